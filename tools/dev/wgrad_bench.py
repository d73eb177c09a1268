"""Dev tool: wgrad kernel TFLOP/s on representative training shapes (B=16)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from das_amd import ops
shapes = [  # B,H,W,Cin,Cout,k,s
    (16, 64, 104, 256, 256, 3, 1), (16, 128, 208, 64, 256, 1, 1), (16, 128, 208, 256, 64, 1, 1),
    (16, 128, 208, 64, 64, 3, 1), (16, 64, 104, 128, 128, 3, 1), (16, 32, 52, 1024, 256, 1, 1),
    (16, 32, 52, 256, 1024, 1, 1), (16, 16, 26, 512, 512, 3, 1), (16, 16, 26, 2048, 2048, 1, 1),
    (16, 64, 104, 2304, 256, 1, 1), (16, 32, 52, 256, 256, 3, 1),
]
for (B, H, W, Cin, Cout, k, s) in shapes:
    x = torch.randn(B, H, W, Cin, device='cuda', dtype=torch.bfloat16)
    dy = torch.randn(B, H // s, W // s, Cout, device='cuda', dtype=torch.bfloat16)
    for _ in range(2):
        ops.conv2d_wgrad(x, dy, k, k, s, k // 2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        ops.conv2d_wgrad(x, dy, k, k, s, k // 2)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 2.0 * B * (H // s) * (W // s) * Cout * k * k * Cin
    by = (x.numel() + dy.numel()) * 2
    print(f'{H}x{W} Cin={Cin:4d} Cout={Cout:4d} k={k}: {ms * 1e3:8.1f} us {fl / ms / 1e9:7.1f} TF  min-traffic {by / ms / 1e9:5.2f} TB/s')
