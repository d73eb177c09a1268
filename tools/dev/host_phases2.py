"""Dev probe: host time per das_amd.ops / torch function inside the head's forward, first step after a synchronisation vs steady state."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from das_amd.datasets import SyntheticPoseDataset, collate
from das_amd.optim import FlatSGD, train_iteration
from das_amd import ops, _lib

dev = torch.device('cuda', 0)
model = bench.build_model(dev, num_stages=4, train=True)
ds = SyntheticPoseDataset(num_joints=bench.J, img_shape=(bench.H, bench.W), length=16, seed=0)
data = collate([ds[i] for i in range(16)], device=dev)
opt = FlatSGD(model, lr=2e-3, momentum=0.9, weight_decay=1e-4, bias_lr_mult=2.0, bias_decay_mult=0.0, max_grad_norm=35.0)
acc, on = {}, [False]


def timed(owner, name, tag):
    orig = getattr(owner, name)

    def f(*a, **k):
        if not on[0]:
            return orig(*a, **k)
        t0 = time.perf_counter()
        try:
            return orig(*a, **k)
        finally:
            acc[tag] = acc.get(tag, 0.0) + (time.perf_counter() - t0) * 1e3
    setattr(owner, name, f)


lib = _lib.load()
for name in list(_lib._SIGS) if hasattr(_lib, '_SIGS') else []:
    pass
for name, fn in list(vars(ops).items()):
    if isinstance(fn, types.FunctionType) and not name.startswith('_'):
        timed(ops, name, 'ops.' + name)
for name in ('cat', 'stack', 'zeros', 'ones', 'empty', 'nonzero_static'):
    timed(torch, name, 'torch.' + name)
fr = model.bbox_head.forward_rows


def fr_on(*a, **k):
    on[0] = True
    try:
        return fr(*a, **k)
    finally:
        on[0] = False


model.bbox_head.forward_rows = fr_on
for _ in range(6):
    train_iteration(model, opt, data, 2e-3)
torch.cuda.synchronize()
rows = []
for i in range(4):
    acc.clear()
    train_iteration(model, opt, data, 2e-3)
    rows.append(dict(acc))
torch.cuda.synchronize()
keys = sorted(set().union(*rows), key=lambda k: -(rows[2].get(k, 0) - rows[0].get(k, 0)))
for k in keys[:12]:
    print('%-34s' % k, ' '.join('%7.2f' % r.get(k, 0.0) for r in rows))
