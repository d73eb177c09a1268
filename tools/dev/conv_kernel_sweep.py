"""Dev tool: every conv kernel that can take a shape, on cold operands with the BatchNorm-statistics epilogue (what the
training forward runs): which one should the dispatcher pick?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from das_amd import ops
shapes = [  # B,H,W,Cin,Cout,k
    (16, 32, 52, 256, 1024, 1), (16, 32, 52, 1024, 256, 1), (16, 32, 52, 256, 256, 3), (16, 32, 52, 256, 256, 1),
    (16, 64, 104, 128, 512, 1), (16, 64, 104, 512, 128, 1), (16, 64, 104, 128, 128, 3), (16, 64, 104, 256, 512, 1),
    (16, 16, 26, 512, 2048, 1), (16, 16, 26, 2048, 512, 1), (16, 16, 26, 512, 512, 3), (16, 16, 26, 2048, 256, 1),
    (16, 16, 26, 2048, 2048, 1), (16, 128, 208, 64, 64, 3), (16, 128, 208, 64, 256, 1), (16, 128, 208, 256, 64, 1),
]
BIG = 1 << 30
CFG = {
    'stream': {},
    'glds4pp': {'conv.stream_minrows': 0, 'conv.glds4_minblocks': 1, 'conv.glds4_pp': 1},
    'glds4': {'conv.stream_minrows': 0, 'conv.glds4_minblocks': 1, 'conv.glds4_pp': 0},
    'glds3': {'conv.stream_minrows': 0, 'conv.glds4_minblocks': 0, 'conv.big_minblocks': 1},
    'glds': {'conv.stream_minrows': 0, 'conv.glds4_minblocks': 0, 'conv.big_minblocks': BIG},
}
torch.manual_seed(0)
for (B, H, W, Cin, Cout, k) in shapes:
    by = B * H * W * (Cin + Cout) * 2
    nb = max(2, int(700e6 // by) + 1)
    xs = [torch.randn(B, H, W, Cin, device='cuda', dtype=torch.bfloat16) for _ in range(nb)]
    ys = [torch.empty(B, H, W, Cout, device='cuda', dtype=torch.bfloat16) for _ in range(nb)]
    w = (torch.randn(Cout, k, k, Cin, device='cuda') / (Cin * k * k) ** 0.5).to(torch.bfloat16)
    slots = 16 if B * H * W >= 16384 else 1
    st = torch.zeros(slots * 2 * Cout, device='cuda', dtype=torch.float32)
    line = f'{H}x{W} Cin={Cin:4d} Cout={Cout:4d} k={k}:'
    seen = set()
    for name, cfg in CFG.items():
        with ops.tuning(**cfg):
            for i in range(nb):
                ops.conv2d(xs[i], w, k, k, 1, k // 2, out=ys[i], stats=st)
            got = ops.last_kernel()
            if got in seen:
                continue
            seen.add(got)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 2 * nb
            e0.record()
            for i in range(n):
                ops.conv2d(xs[i % nb], w, k, k, 1, k // 2, out=ys[i % nb], stats=st)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / n * 1e3
        line += f'  {got[5:]:16s} {us:6.1f}'
    fl = 2.0 * B * H * W * Cout * k * k * Cin
    print(line, f'  | hbm {by / 6.3e6:5.1f} mfma {fl / 2.5e9:5.1f} us', flush=True)
