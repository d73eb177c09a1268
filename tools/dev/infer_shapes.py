"""Dev tool: per-shape timing of every conv launch of the infer bench step (B=8, 1 stage; HIP events on the launch
stream), sorted by time, with the HBM / MFMA floor of each shape beside it."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from das_amd import ops
from das_amd.datasets import SyntheticPoseDataset, collate

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda', 0)
model = bench.build_model(dev, num_stages=1, train=False)
ds = SyntheticPoseDataset(num_joints=bench.J, img_shape=(bench.H, bench.W), length=B, seed=0)
data = collate([ds[i] for i in range(B)], device=dev)
metas = data['img_metas']
bench.calibrate_scores(model, data['img'], metas)
for _ in range(3):
    model(data['img'], metas, return_loss=False, rescale=True)
torch.cuda.synchronize()
ops.profile_begin()
R = 3
for _ in range(R):
    model(data['img'], metas, return_loss=False, rescale=True)
torch.cuda.synchronize()
PROFILE = ops.profile_end()
agg = {}
for ent in PROFILE:
    tag, fl, e0, e1, shape = ent[:5]
    a = agg.setdefault((tag, shape), [0.0, 0.0, 0])
    a[0] += fl; a[1] += e0.elapsed_time(e1) * 1e-3; a[2] += 1
tot = sum(a[1] for a in agg.values())
print(f'total conv ms/step {tot / R * 1e3:.3f}')
for (tag, shape), (fl, sec, n) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    Bb, H, W, Cin, Cout, k, s, nl = shape
    per, flop = sec / n, fl / n
    rows_out = flop / (2.0 * Cout * k * k * Cin)
    hbm = (rows_out * s * s * Cin + rows_out * Cout + Cout * k * k * Cin) * 2 / 6.3e12
    print(f'{sec / tot * 100:5.1f}% {sec / R * 1e3:7.3f}ms n={n // R:3d} {per * 1e6:7.1f}us (hbm {hbm * 1e6:6.1f} mfma {flop / 2.5e15 * 1e6:6.1f}) '
          f'{fl / sec / 1e12:7.1f}TF {tag:24s} HxW={H}x{W} Cin={Cin} Cout={Cout} k={k} s={s} lv={nl}')
