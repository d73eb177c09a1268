"""Dev tool: per-shape timing of every BatchNorm pass of the training bench step (HIP events on the launch stream):
algorithmic bytes / time per (pass, rows, channels, variant), sorted by time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from das_amd import ops
from das_amd.datasets import SyntheticPoseDataset, collate
from das_amd.optim import FlatSGD, train_iteration
from das_amd import autograd as _ag
_ag.WGRAD_SIDE_STREAM = False
if os.environ.get('DASLIB'):   # dev: a variant build of the library
    from das_amd import _lib as _l
    _l.LIB_PATH = os.path.join(os.path.dirname(_l.LIB_PATH), os.environ['DASLIB'])
dev = torch.device('cuda', 0)
model = bench.build_model(dev, num_stages=4, train=True)
ds = SyntheticPoseDataset(num_joints=bench.J, img_shape=(bench.H, bench.W), length=16, seed=0)
data = collate([ds[i] for i in range(16)], device=dev)
opt = FlatSGD(model, lr=2e-3, momentum=0.9, weight_decay=1e-4, bias_lr_mult=2.0, bias_decay_mult=0.0, max_grad_norm=35.0)
for _ in range(2):
    train_iteration(model, opt, data, 2e-3)
ops.profile_begin()
R = 3
for _ in range(R):
    train_iteration(model, opt, data, 2e-3)
torch.cuda.synchronize()
PROFILE = ops.profile_end()
agg = {}
for ent in PROFILE:
    tag, fl, e0, e1, shape = ent[:5]
    if not (shape and shape[0] == 'bn'):
        continue
    a = agg.setdefault((tag, shape[1:]), [0.0, 0.0, 0])
    a[0] += ent[6]; a[1] += e0.elapsed_time(e1) * 1e-3; a[2] += 1
tot = sum(a[1] for a in agg.values())
print(f'BatchNorm passes: {tot / R * 1e3:.3f} ms/step, {sum(a[0] for a in agg.values()) / tot / 1e12:.2f} TB/s')
print('  share   ms/step   n    us/call   MB/call  TB/s   at 5.5 TB/s this shape would save (ms/step)')
for (tag, shape), (nb, sec, n) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    per = sec / n
    save = (sec - nb / 5.5e12) / R * 1e3
    print(f'{sec / tot * 100:5.1f}% {sec / R * 1e3:7.3f} {n // R:4d} {per * 1e6:8.1f} {nb / n / 1e6:8.1f} {nb / sec / 1e12:6.2f}  {save:6.3f}  {tag[:34]:34s} {shape}')
