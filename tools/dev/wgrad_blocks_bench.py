"""Dev tool: how the weight-gradient time of one op changes with the size of its grid (cold operands). If a grid of a
quarter of the chip takes much less than 4x the time of the full grid, running several ops side by side in ONE launch
(each with a quarter of the workgroups and a quarter of the split workspace) pays."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from das_amd import ops
shapes = [  # B,H,W,Cin,Cout,k
    (16, 32, 52, 256, 1024, 1), (16, 32, 52, 1024, 256, 1), (16, 32, 52, 256, 256, 3), (16, 128, 208, 256, 256, 1),
    (16, 64, 104, 128, 512, 1), (16, 16, 26, 512, 2048, 1), (16, 16, 26, 512, 512, 3), (16, 128, 208, 64, 64, 3),
    (16, 128, 208, 64, 256, 1), (16, 64, 104, 128, 128, 3),
]
for (B, H, W, Cin, Cout, k) in shapes:
    by = B * H * W * (Cin + Cout) * 2
    nb = max(2, int(700e6 // by) + 1)
    xs = [torch.randn(B, H, W, Cin, device='cuda', dtype=torch.bfloat16) for _ in range(nb)]
    dys = [torch.randn(B, H, W, Cout, device='cuda', dtype=torch.bfloat16) for _ in range(nb)]
    dw = torch.zeros(Cout, k, k, Cin, device='cuda')
    line = f'{H}x{W} Cin={Cin:4d} Cout={Cout:4d} k={k}:'
    for div in (1, 2, 4, 8):
        with ops.tuning(**{'wgrad.pp_blocks': 256 // div, 'wgrad.blocks': 0 if div == 1 else 768 // div}):
            for i in range(nb):
                ops.conv2d_wgrad(xs[i], dys[i], k, k, 1, k // 2, out=dw, accumulate=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 2 * nb
            e0.record()
            for i in range(n):
                ops.conv2d_wgrad(xs[i % nb], dys[i % nb], k, k, 1, k // 2, out=dw, accumulate=True)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / n * 1e3
        line += f'  1/{div}: {us:7.1f} us ({ops.last_kernel()[5:13]})'
    print(line, f'  HBM floor {by / 6.3e6:6.1f} us', flush=True)
