"""Dev tool: per-shape time and bandwidth of the BatchNorm passes inside the train step (HIP events, side stream off,
hipGraphs off), sorted by time: where does the family lose against the HBM rate?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from das_amd import ops, autograd as ag
from das_amd.datasets import SyntheticPoseDataset, collate
from das_amd.optim import FlatSGD, train_iteration

B = 16
ag.WGRAD_SIDE_STREAM = False
dev = torch.device('cuda', 0)
model = bench.build_model(dev, num_stages=4, train=True)
ds = SyntheticPoseDataset(num_joints=bench.J, img_shape=(bench.H, bench.W), length=B, seed=0)
data = collate([ds[i] for i in range(B)], device=dev)
opt = FlatSGD(model, lr=2e-3, momentum=0.9, weight_decay=1e-4, bias_lr_mult=2.0, bias_decay_mult=0.0, max_grad_norm=35.0)
for _ in range(2):
    train_iteration(model, opt, data, 2e-3)
ops.PROFILE = []
R = 2
for _ in range(R):
    train_iteration(model, opt, data, 2e-3)
torch.cuda.synchronize()
agg = {}
for ent in ops.PROFILE:
    tag, fl, e0, e1, shape = ent[:5]
    if not shape or shape[0] != 'bn':
        continue
    a = agg.setdefault((tag, shape[1:]), [0.0, 0.0, 0])
    a[0] += ent[6]; a[1] += e0.elapsed_time(e1) * 1e-3; a[2] += 1
tot = sum(a[1] for a in agg.values())
print(f'BatchNorm passes: {tot / R * 1e3:.2f} ms/step, {sum(a[0] for a in agg.values()) / tot / 1e12:.2f} TB/s')
for (tag, shape), (by, sec, n) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'{sec / tot * 100:5.1f}% {sec / R * 1e3:6.3f} ms n={n // R:3d} {sec / n * 1e6:7.1f} us  {by / sec / 1e12:5.2f} TB/s  '
          f'floor@6.3 {by / n / 6.3e6:6.1f} us  {tag[:34]:34s} rows={shape[0]} C={shape[1]} {shape[2:]}')
