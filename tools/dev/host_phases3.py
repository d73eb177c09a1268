"""Dev probe: host time per head submodule call (forward hooks), first step after a synchronisation vs steady state."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from das_amd.datasets import SyntheticPoseDataset, collate
from das_amd.optim import FlatSGD, train_iteration

dev = torch.device('cuda', 0)
model = bench.build_model(dev, num_stages=4, train=True)
ds = SyntheticPoseDataset(num_joints=bench.J, img_shape=(bench.H, bench.W), length=16, seed=0)
data = collate([ds[i] for i in range(16)], device=dev)
opt = FlatSGD(model, lr=2e-3, momentum=0.9, weight_decay=1e-4, bias_lr_mult=2.0, bias_decay_mult=0.0, max_grad_norm=35.0)
acc, t_in, order = {}, {}, []
for name, m in model.bbox_head.named_modules():
    if name == '' or any(True for _ in m.children()):
        continue

    def pre(mod, inp, name=name):
        t_in[name] = time.perf_counter()

    def post(mod, inp, out, name=name):
        acc.setdefault(name, []).append((time.perf_counter() - t_in[name]) * 1e3)
        order.append(name)
    m.register_forward_pre_hook(pre)
    m.register_forward_hook(post)
for _ in range(6):
    train_iteration(model, opt, data, 2e-3)
torch.cuda.synchronize()
rows = []
for i in range(4):
    acc.clear(); order.clear()
    train_iteration(model, opt, data, 2e-3)
    rows.append({k: sum(v) for k, v in acc.items()})
    if i == 2:
        seq = list(order)
torch.cuda.synchronize()
keys = sorted(set().union(*rows), key=lambda k: -(rows[2].get(k, 0) - rows[0].get(k, 0)))
for k in keys[:8]:
    print('%-50s' % k, ' '.join('%7.2f' % r.get(k, 0.0) for r in rows))
print('total hooked', ' '.join('%7.2f' % sum(r.values()) for r in rows))
print('call order:', ' '.join(seq[:40]))
