"""Dev tool: the step's 3x3 convs (forward form; the data gradients are the same launches with flipped weights) alone:
us per launch, TFLOP/s, the kernel that took them and the MFMA time at 1.3 PF (the tile kernels' K-loop rate)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from das_amd import ops
for a in sys.argv[1:]:
    from das_amd import _lib
    k, v = a.split('=')
    _lib.check(_lib.load().das_tuning_set(k.encode(), int(v)), k)
B = 16
# (launches per step fwd + dgrad, H, W, Cin, Cout); H = 0: the head's ragged levels
SHAPES = [(24, 128, 208, 64, 64), (24, 64, 104, 128, 128), (42, 32, 52, 256, 256), (16, 16, 26, 512, 512), (2, 64, 104, 256, 256),
          (16, 0, 0, 256, 256), (4, 0, 0, 32, 256)]
LEVELS = [(64, 104), (32, 52), (16, 26), (8, 13)]
tot = 0.0
for n, H, W, Cin, Cout in SHAPES:
    if H:
        xs = [torch.randn(B, H, W, Cin, device='cuda', dtype=torch.bfloat16) for _ in range(3)]
        rows = B * H * W
    else:
        xs = [ops.Ragged.from_levels([torch.randn(B, h, w, Cin, device='cuda', dtype=torch.bfloat16) for h, w in LEVELS]) for _ in range(3)]
        rows = xs[0].rows
    w = (torch.randn(Cout, 3, 3, Cin, device='cuda') / (9 * Cin) ** 0.5).to(torch.bfloat16)
    stats = torch.zeros(16 * 2 * Cout, device='cuda')
    y = ops.conv2d(xs[0], w, 3, 3, 1, 1, stats=stats)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(9):
        ops.conv2d(xs[i % 3], w, 3, 3, 1, 1, stats=stats, out=y)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 9 * 1e3
    fl = 2.0 * rows * Cout * 9 * Cin
    tot += n * us
    print(f'{n:3d} x  {H or "rag":>4}x{W or "":<4} {Cin:4d}->{Cout:<4d} {us:7.1f} us  {fl / us / 1e6:6.0f} TF  (1.3 PF: {fl / 1.3e9:5.1f} us)  x{us / (fl / 1.3e9):4.1f}  '
          f'{n * us / 1e3:5.2f} ms/step  {ops.last_kernel()}', flush=True)
print(f'sum {tot / 1e3:.2f} ms per step')
