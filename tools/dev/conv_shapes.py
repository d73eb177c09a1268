"""Dev tool: per-shape conv timing of the bench workload (HIP events), sorted by time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from das_amd import ops

dev = torch.device('cuda', 0)
model = bench.build_model(dev)
img = torch.randn(8, 3, bench.H, bench.W, device=dev)
metas = [dict(scale_factor=np.ones(4, dtype=np.float32), filename='')] * 8
for _ in range(3):
    model(img, metas, return_loss=False)
ops.profile_begin()
R = 5
for _ in range(R):
    model(img, metas, return_loss=False)
torch.cuda.synchronize()
PROFILE = ops.profile_end()
agg = {}
for tag, fl, e0, e1, shape in [e[:5] for e in PROFILE]:
    a = agg.setdefault((tag, shape), [0.0, 0.0, 0])
    a[0] += fl; a[1] += e0.elapsed_time(e1) * 1e-3; a[2] += 1
tot = sum(a[1] for a in agg.values())
print(f'total conv ms/step {tot / R * 1e3:.3f}')
for (tag, shape), (fl, sec, n) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    B, H, W, Cin, Cout, k, s, xps = shape
    M = B * (H // s) * (W // s)
    bytes_ = (B * H * W * Cin + M * Cout) * 2
    print(f'{sec / tot * 100:5.1f}% {sec / R * 1e3:7.3f}ms n={n // R:3d} {fl / sec / 1e12:7.1f}TF {bytes_ * (n // R) / (sec / R) / 1e12:6.2f}TB/s '
          f'{tag[11:]:24s} HxW={H}x{W} Cin={Cin} Cout={Cout} k={k} s={s} M={M}')
