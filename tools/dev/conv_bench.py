"""Dev tool: conv kernel correctness + TFLOP/s on the shapes that dominate the bench."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from das_amd import ops
shapes = [  # B,H,W,Cin,Cout,k,s
    (8, 64, 104, 256, 256, 3, 1), (8, 32, 52, 256, 256, 3, 1), (8, 16, 26, 256, 256, 3, 1),
    (8, 64, 104, 2304, 256, 1, 1), (8, 128, 208, 64, 256, 1, 1), (8, 128, 208, 256, 64, 1, 1),
    (8, 128, 208, 64, 64, 3, 1), (8, 64, 104, 128, 128, 3, 1), (8, 32, 52, 1024, 256, 1, 1),
    (8, 32, 52, 256, 1024, 1, 1), (8, 16, 26, 512, 512, 3, 1), (8, 64, 104, 256, 32, 3, 1),
    (8, 128, 208, 256, 512, 1, 2), (8, 64, 104, 256, 768, 3, 1),
    (8, 128, 208, 256, 256, 1, 1), (8, 64, 104, 128, 512, 1, 1), (8, 64, 104, 512, 128, 1, 1),
    (8, 64, 104, 512, 256, 1, 1), (8, 16, 26, 512, 2048, 1, 1), (8, 16, 26, 2048, 512, 1, 1),
]
if len(sys.argv) > 1:
    shapes = [(int(sys.argv[1]),) + sh[1:] for sh in shapes]
torch.manual_seed(0)
for (B, H, W, Cin, Cout, k, s) in shapes:
    x = torch.randn(B, H, W, Cin, device='cuda', dtype=torch.bfloat16)
    w = (torch.randn(Cout, k, k, Cin, device='cuda') / (Cin * k * k) ** 0.5).to(torch.bfloat16)
    y = ops.conv2d(x, w, k, k, s, k // 2)
    # spot-check against torch on a slice of the batch
    ref = F.conv2d(x[:1].permute(0, 3, 1, 2).float(), w.permute(0, 3, 1, 2).float(), None, s, k // 2)
    err = (y[:1].permute(0, 3, 1, 2).float() - ref).abs().max().item() / ref.abs().max().item()
    for _ in range(3):
        ops.conv2d(x, w, k, k, s, k // 2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        ops.conv2d(x, w, k, k, s, k // 2, out=y)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 2.0 * B * (H // s) * (W // s) * Cout * k * k * Cin
    by = (x.numel() + y.numel()) * 2
    print(f'{H}x{W} Cin={Cin:4d} Cout={Cout:4d} k={k} s={s}: {ms * 1e3:8.1f} us {fl / ms / 1e9:7.1f} TF {by / ms / 1e9:6.2f} TB/s relerr {err:.1e}')
