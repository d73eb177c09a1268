"""Dev tool: which ops of ONE train step launch the torch fill kernels (FillFunctor: zeros / zero_ / fill_ — 26 per step at 8.7 us in
profiles/r06_census.txt), with the chain of enclosing ops (autograd nodes included) and the shapes."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402
from das_amd.datasets import SyntheticPoseDataset, collate  # noqa: E402
from das_amd.optim import FlatSGD, train_iteration  # noqa: E402

dev = torch.device('cuda', 0)
model = bench.build_model(dev, num_stages=4, train=True)
ds = SyntheticPoseDataset(num_joints=bench.J, img_shape=(bench.H, bench.W), length=16, seed=0)
data = collate([ds[i] for i in range(16)], device=dev)
opt = FlatSGD(model, lr=2e-3, momentum=0.9, weight_decay=1e-4, bias_lr_mult=2.0, bias_decay_mult=0.0, max_grad_norm=35.0)
for _ in range(4):
    train_iteration(model, opt, data, 2e-3)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    train_iteration(model, opt, data, 2e-3)
    torch.cuda.synchronize()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rows = collections.Counter()
us = collections.Counter()
for ev in prof.events():
    ks = [k for k in (getattr(ev, 'kernels', None) or []) if 'FillFunctor' in k.name]
    if not ks:
        continue
    if ev.cpu_parent is not None and any('FillFunctor' in k.name for k in (getattr(ev.cpu_parent, 'kernels', None) or [])):
        continue      # (the outermost op that owns the kernel)
    chain, p = [], ev
    while p is not None and len(chain) < 6:
        chain.append(p.name)
        p = p.cpu_parent
    stack = getattr(ev, 'stack', None) or []
    frame = next((f.replace(ROOT + '/', '') for f in stack if ('das_amd' in f or 'bench.py' in f) and 'torch/' not in f), '')
    key = (' <- '.join(chain), str(ev.input_shapes)[:60], frame[:80])
    rows[key] += 1
    us[key] += sum(k.duration for k in ks)
for key, n in rows.most_common(40):
    print(f'{n:3d} x {us[key] / n:7.1f} us  {key[0]}  {key[1]}  {key[2]}')
