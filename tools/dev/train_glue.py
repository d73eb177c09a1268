"""Dev tool: torch.profiler over one train bench step: aten copy / add / fill ops by count and input shapes (the glue
between the HIP kernels)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity
from das_amd.datasets import SyntheticPoseDataset, collate
from das_amd.optim import FlatSGD, train_iteration

B = 16
dev = torch.device('cuda', 0)
model = bench.build_model(dev, num_stages=4, train=True)
ds = SyntheticPoseDataset(num_joints=bench.J, img_shape=(bench.H, bench.W), length=B, seed=0)
data = collate([ds[i] for i in range(B)], device=dev)
opt = FlatSGD(model, lr=2e-3, momentum=0.9, weight_decay=1e-4, bias_lr_mult=2.0, bias_decay_mult=0.0, max_grad_norm=35.0)
for _ in range(3):
    train_iteration(model, opt, data, 2e-3)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    train_iteration(model, opt, data, 2e-3)
    torch.cuda.synchronize()
names = ('aten::copy_', 'aten::add_', 'aten::add', 'aten::fill_', 'aten::zero_', 'aten::cat', 'aten::clone', 'aten::contiguous',
         'aten::mul', 'aten::sum', 'aten::zeros', 'aten::empty', 'aten::to', 'aten::_to_copy', 'aten::index_select', 'aten::select_backward',
         'aten::slice_backward', 'aten::zeros_like', 'aten::new_zeros')
by = collections.Counter()
tot = collections.Counter()
for ev in prof.events():
    tot[ev.name] += 1
    if ev.name in names:
        by[(ev.name, str(ev.input_shapes)[:110])] += 1
print('-- op totals'); print(', '.join(f'{k}: {v}' for k, v in tot.most_common(40)))
for (name, shp), n in by.most_common(70):
    print(f'{n:4d}  {name:18s} {shp}')
