"""Dev tool: 1x1 / 3x3 conv launches on COLD operands (rotating through enough buffer sets to exceed the 256 MiB
infinity cache), with BatchNorm statistics (16 slots) and optionally a residual — what the training step sees,
unlike a loop over one resident buffer set."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from das_amd import ops
shapes = [  # B,H,W,Cin,Cout,k
    (16, 128, 208, 256, 256, 1), (16, 128, 208, 64, 256, 1), (16, 128, 208, 256, 64, 1), (16, 64, 104, 128, 512, 1),
    (16, 64, 104, 512, 128, 1), (16, 32, 52, 256, 1024, 1), (16, 32, 52, 1024, 256, 1), (16, 64, 104, 256, 256, 1),
    (16, 32, 52, 256, 256, 3), (16, 16, 26, 512, 512, 3),
]
torch.manual_seed(0)
for (B, H, W, Cin, Cout, k) in shapes:
    by = B * H * W * (Cin + Cout) * 2
    nb = max(2, int(700e6 // by) + 1)
    xs = [torch.randn(B, H, W, Cin, device='cuda', dtype=torch.bfloat16) for _ in range(nb)]
    ys = [torch.empty(B, H, W, Cout, device='cuda', dtype=torch.bfloat16) for _ in range(nb)]
    rs = [torch.randn(B, H, W, Cout, device='cuda', dtype=torch.bfloat16) for _ in range(min(nb, 4))]
    w = (torch.randn(Cout, k, k, Cin, device='cuda') / (Cin * k * k) ** 0.5).to(torch.bfloat16)
    st = torch.zeros(16 * 2 * Cout, device='cuda', dtype=torch.float32)
    res = []
    for mode in ('plain', 'stats', 'residual'):
        kw = dict(stats=st) if mode == 'stats' else {}
        n = 2 * nb
        for i in range(nb):
            ops.conv2d(xs[i], w, k, k, 1, k // 2, out=ys[i], **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            if mode == 'residual':
                kw = dict(residual=rs[i % len(rs)])
            ops.conv2d(xs[i % nb], w, k, k, 1, k // 2, out=ys[i % nb], **kw)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / n * 1e3)
    print(f'{H}x{W} Cin={Cin:4d} Cout={Cout:4d} k={k}: cold plain {res[0]:7.1f} us  +stats {res[1]:7.1f} us  +residual '
          f'{res[2]:7.1f} us   HBM floor {by / 6.3e6:6.1f} us ({nb} buffer sets)')
