"""Dev tool: where do the small torch glue kernels (copies, fills, adds, cats) of one train step come from?
Groups torch-profiler CPU op events by the innermost das_amd source line on their python stack."""
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
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    train_iteration(model, opt, data, 2e-3)
    torch.cuda.synchronize()
WATCH = ('aten::copy_', 'aten::clone', 'aten::contiguous', 'aten::cat', 'aten::fill_', 'aten::zero_', 'aten::add',
         'aten::add_', 'aten::mul', 'aten::to', 'aten::_to_copy', 'aten::zeros', 'aten::index', 'aten::index_put_',
         'aten::empty', 'aten::sum', 'aten::stack')
cnt = Counter()
for ev in prof.events():
    if ev.name not in WATCH:
        continue
    where = 'autograd engine / no das_amd frame'
    for fr in ev.stack:
        if '/das_amd/' in fr:
            where = fr.split('/das_amd/')[-1]
            break
    cnt[(ev.name, where)] += 1
for (name, where), n in cnt.most_common(45):
    print(f'{n:6d}  {name:18s} {where}')
