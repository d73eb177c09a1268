"""Dev tool: one wgrad shape at several batch sizes (time vs pixel rows -> per-step cost and fixed cost)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from das_amd import ops
H, W, Cin, Cout, k = 64, 104, 256, 256, 3
for B in (4, 8, 16, 32):
    x = torch.randn(B, H, W, Cin, device='cuda', dtype=torch.bfloat16)
    dy = torch.randn(B, H, W, Cout, device='cuda', dtype=torch.bfloat16)
    for _ in range(2):
        ops.conv2d_wgrad(x, dy, k, k, 1, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        ops.conv2d_wgrad(x, dy, k, k, 1, 1)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 2.0 * B * H * W * Cout * k * k * Cin
    print(f'B={B:3d}: {ms * 1e3:8.1f} us {fl / ms / 1e9:7.1f} TF')
