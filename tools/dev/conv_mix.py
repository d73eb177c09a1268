"""Dev tool: the tile-kernel and stream-kernel launches (forward and data-gradient) of one training bench step (B=16; shapes and
per-step counts from tools/dev/train_shapes.py; the stride-2 data-gradient shapes, 4 launches, are left out), each
launched `count` times. Run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes) for the HBM
traffic per launch behind bench.py's roofline.traffic, or bare for a time per step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from das_amd import ops

# (count, H, W, Cin, Cout, k); H = 0: the head's four ragged FPN levels in one launch
SHAPES = [
    (26, 128, 208, 256, 256, 1), (27, 128, 208, 64, 256, 1), (51, 32, 52, 256, 1024, 1), (28, 64, 104, 128, 512, 1),
    (11, 64, 104, 256, 512, 1), (11, 64, 104, 512, 256, 1), (20, 16, 26, 512, 2048, 1), (6, 64, 104, 512, 512, 1),
    (10, 64, 104, 256, 256, 1), (6, 32, 52, 1024, 1024, 1), (4, 128, 208, 128, 256, 1), (6, 16, 26, 2048, 2048, 1),
    (2, 64, 104, 256, 256, 3), (16, 0, 0, 256, 256, 3), (4, 0, 0, 256, 2304, 1), (4, 0, 0, 2304, 256, 1),
    (4, 0, 0, 32, 256, 3),
    # the 256 x 128 tile (conv_glds3_kernel, plain / ping-pong / split over K) and 128 x 64 tile (conv_glds_kernel) shapes
    (42, 32, 52, 256, 256, 3), (51, 32, 52, 1024, 256, 1), (24, 64, 104, 128, 128, 3), (28, 64, 104, 512, 128, 1),
    (16, 16, 26, 512, 512, 3), (20, 16, 26, 2048, 512, 1), (24, 128, 208, 64, 64, 3),
]
LEVELS = [(64, 104), (32, 52), (16, 26), (8, 13)]
B = 16
tot_ms, tot_fl, n, alg = 0.0, 0.0, 0, 0.0
families = {}      # kernel family (as bench.py tags it) -> [launches, algorithmic bytes] of THIS mix
for (cnt, H, W, Cin, Cout, k) in SHAPES:
    if H:
        x = torch.randn(B, H, W, Cin, device='cuda', dtype=torch.bfloat16)
        rows = B * H * W
    else:
        x = ops.Ragged.from_levels([torch.randn(B, h, w, Cin, device='cuda', dtype=torch.bfloat16) for h, w in LEVELS])
        rows = x.rows
    w = (torch.randn(Cout, k, k, Cin, device='cuda') / (Cin * k * k) ** 0.5).to(torch.bfloat16)
    y = ops.conv2d(x, w, k, k, 1, k // 2)
    fam = ops.last_kernel()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(cnt):
        ops.conv2d(x, w, k, k, 1, k // 2, out=y)
    e1.record()
    torch.cuda.synchronize()
    tot_ms += e0.elapsed_time(e1)
    tot_fl += cnt * 2.0 * rows * Cout * k * k * Cin
    alg += cnt * (rows * (Cin + Cout) * 2 + w.numel() * 2)
    n += cnt
    f = families.setdefault(fam, [0, 0.0])
    f[0] += cnt + 1            # (+ the warm-up launch above: the PMC passes count it too)
    f[1] += (cnt + 1) * (rows * (Cin + Cout) * 2 + w.numel() * 2)
print(f'{n} launches, {tot_ms:.3f} ms, {tot_fl / tot_ms / 1e9:.1f} TF, algorithmic bytes per launch {alg / n / 1e6:.2f} MB '
      f'(x read once, y written once, weights once)')
import json
print('ALGORITHMIC ' + json.dumps({k: dict(launches=v[0], algorithmic_mb_per_launch=round(v[1] / v[0] / 1e6, 2)) for k, v in families.items()}))
