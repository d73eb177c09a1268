"""Dev tool: the weight-gradient launches of one training bench step (B=16, the 30 most expensive shapes with
their per-step counts, from tools/dev/train_shapes.py), each launched `count` times on fresh random operands.
Run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes) to get the HBM traffic per launch of
conv_wgrad_kernel + wgrad_reduce_kernel for bench.py's roofline.traffic, or bare for a time per step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from das_amd import ops

# (count, H, W, Cin, Cout, k, stride)
SHAPES = [
    (9, 64, 104, 256, 256, 3, 1), (12, 128, 208, 256, 256, 1, 1), (21, 32, 52, 256, 256, 3, 1),
    (12, 128, 208, 64, 64, 3, 1), (27, 32, 52, 256, 1024, 1, 1), (4, 64, 104, 2304, 256, 1, 1),
    (24, 32, 52, 1024, 256, 1, 1), (16, 128, 208, 64, 256, 1, 1), (11, 128, 208, 256, 64, 1, 1),
    (12, 64, 104, 128, 128, 3, 1), (16, 64, 104, 128, 512, 1, 1), (12, 64, 104, 512, 128, 1, 1),
    (8, 64, 104, 512, 256, 1, 1), (8, 16, 26, 512, 512, 3, 1), (12, 16, 26, 512, 2048, 1, 1),
    (4, 64, 104, 256, 32, 3, 1), (4, 128, 208, 256, 128, 1, 1), (8, 16, 26, 2048, 512, 1, 1),
    (3, 64, 104, 512, 512, 1, 1), (3, 32, 52, 1024, 1024, 1, 1), (3, 16, 26, 2048, 2048, 1, 1),
    (1, 512, 832, 8, 64, 7, 2), (4, 128, 208, 128, 128, 3, 2), (4, 64, 104, 256, 256, 3, 2),
    (4, 32, 52, 512, 512, 3, 2), (6, 64, 104, 256, 256, 1, 1), (4, 32, 52, 1024, 512, 1, 1),
    (4, 128, 208, 256, 512, 1, 2), (4, 64, 104, 512, 1024, 1, 2), (4, 32, 52, 1024, 2048, 1, 2),
]
B = 16
tot_ms, tot_fl, n = 0.0, 0.0, 0
alg_bytes = 0.0
for (cnt, H, W, Cin, Cout, k, s) in SHAPES:
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    x = torch.randn(B, H, W, Cin, device='cuda', dtype=torch.bfloat16)
    dy = torch.randn(B, Ho, Wo, Cout, device='cuda', dtype=torch.bfloat16)
    ops.conv2d_wgrad(x, dy, k, k, s, k // 2)   # warm-up (workspace allocation)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(cnt):
        ops.conv2d_wgrad(x, dy, k, k, s, k // 2)
    e1.record()
    torch.cuda.synchronize()
    tot_ms += e0.elapsed_time(e1)
    tot_fl += cnt * 2.0 * B * Ho * Wo * Cout * k * k * Cin
    alg_bytes += cnt * ((x.numel() + dy.numel()) * 2 + Cout * k * k * Cin * 4)
    n += cnt
print(f'{n} launches, {tot_ms:.3f} ms, {tot_fl / tot_ms / 1e9:.1f} TF, algorithmic bytes per launch '
      f'{alg_bytes / n / 1e6:.2f} MB (x + dy read once, dW written once)')
