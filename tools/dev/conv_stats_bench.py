"""Dev tool: cost of the BatchNorm statistics epilogue (per-channel sum / sum of squares, f32 atomics per
workgroup) on the large-M training shapes: conv2d with and without `stats`."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from das_amd import ops
shapes = [  # B,H,W,Cin,Cout,k,s
    (16, 128, 208, 256, 256, 1, 1), (16, 128, 208, 64, 256, 1, 1), (16, 128, 208, 256, 64, 1, 1),
    (16, 128, 208, 64, 64, 3, 1), (16, 64, 104, 128, 512, 1, 1), (16, 64, 104, 512, 128, 1, 1),
    (16, 32, 52, 256, 1024, 1, 1), (16, 32, 52, 1024, 256, 1, 1), (16, 32, 52, 256, 256, 3, 1),
    (16, 16, 26, 512, 512, 3, 1), (16, 64, 104, 128, 128, 3, 1),
]
torch.manual_seed(0)
for (B, H, W, Cin, Cout, k, s) in shapes:
    x = torch.randn(B, H, W, Cin, device='cuda', dtype=torch.bfloat16)
    w = (torch.randn(Cout, k, k, Cin, device='cuda') / (Cin * k * k) ** 0.5).to(torch.bfloat16)
    y = ops.conv2d(x, w, k, k, s, k // 2)
    st = torch.zeros(2 * Cout, device='cuda', dtype=torch.float32)
    st16 = torch.zeros(16 * 2 * Cout, device='cuda', dtype=torch.float32)
    res = []
    for stats in (None, st, st16):
        for _ in range(3):
            ops.conv2d(x, w, k, k, s, k // 2, out=y, stats=stats)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            ops.conv2d(x, w, k, k, s, k // 2, out=y, stats=stats)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / n * 1e3)
    by = (x.numel() + y.numel()) * 2
    print(f'{H}x{W} Cin={Cin:4d} Cout={Cout:4d} k={k}: plain {res[0]:7.1f} us  with stats {res[1]:7.1f} us  16 slots {res[2]:7.1f} us   '
          f'HBM floor {by / 6.3e6:6.1f} us')
