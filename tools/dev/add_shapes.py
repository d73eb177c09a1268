"""Dev tool: shapes of the torch add / add_ / copy_ calls of one train step (which gradient sums are left to autograd)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from collections import Counter
import torch
from torch.profiler import ProfilerActivity, profile
import bench
from das_amd.datasets import SyntheticPoseDataset, collate
from das_amd.optim import FlatSGD, train_iteration

B = 16
dev = torch.device('cuda', 0)
model = bench.build_model(dev, num_stages=4, train=True)
ds = SyntheticPoseDataset(num_joints=bench.J, img_shape=(bench.H, bench.W), length=B, seed=0)
data = collate([ds[i] for i in range(B)], device=dev)
opt = FlatSGD(model, lr=2e-3, momentum=0.9, weight_decay=1e-4, bias_lr_mult=2.0, bias_decay_mult=0.0, max_grad_norm=35.0)
for _ in range(2):
    train_iteration(model, opt, data, 2e-3)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    train_iteration(model, opt, data, 2e-3)
    torch.cuda.synchronize()
cnt = Counter()
for ev in prof.events():
    if ev.name in ('aten::add', 'aten::add_', 'aten::copy_', 'aten::cat', 'aten::clone', 'aten::contiguous'):
        shp = tuple(tuple(s) for s in ev.input_shapes[:2] if s)
        n = 1
        for d in (shp[0] if shp else ()):
            n *= d
        if n >= 1 << 20:
            cnt[(ev.name, shp)] += 1
for (name, shp), n in cnt.most_common(40):
    print(f'{n:4d} {name:16s} {shp}')
