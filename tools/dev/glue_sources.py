"""Dev tool: which Python lines of the train step issue the small ATen launches (fill / copy / add / cat ...)?
torch.profiler with stacks over one steady-state step; prints call counts per (op, source line)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import collections
import torch
import bench
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
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    train_iteration(model, opt, data, 2e-3)
    torch.cuda.synchronize()
want = ('aten::fill_', 'aten::zero_', 'aten::copy_', 'aten::add', 'aten::add_', 'aten::cat', 'aten::mul', 'aten::index_select',
        'aten::zeros', 'aten::clone', 'aten::contiguous', 'aten::_to_copy', 'aten::sum', 'aten::div', 'aten::sub')
cnt = collections.Counter()
nost = 0
nostack = collections.Counter()
stacks = {}
for ev in prof.events():
    if ev.name not in want:
        continue
    st = ev.stack or []
    if not st:
        nost += 1
        nostack[(ev.name, str(ev.input_shapes)[:70])] += 1
        continue
    src = next((f for f in st if 'das_amd/' in f), None)
    if src is None:
        src = st[0]
    key = (ev.name, src.split('das_amd/')[-1][:80] + '  ' + str(ev.input_shapes)[:60])
    cnt[key] += 1
    stacks.setdefault(key, st)
print('events without a stack:', nost)
for (name, src), n in cnt.most_common(45):
    print(f'{n:4d}  {name:18s} {src}')
print('--- events without a stack (the autograd engine thread), by op:')
for name, n in nostack.most_common(40):
    print(f'{n:4d}  {name}')
print('--- full stacks of the top entries')
for (name, src), n in cnt.most_common(8):
    print(f'== {n} x {name} @ {src}')
    for f in stacks[(name, src)][:14]:
        print('     ', f[-110:])
