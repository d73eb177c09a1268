"""Dev probe: which tensors does BottleneckChainFn.apply copy? (aten::contiguous events inside apply, with shapes)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from das_amd.datasets import SyntheticPoseDataset, collate
from das_amd.optim import FlatSGD, train_iteration
from torch.profiler import profile, ProfilerActivity

dev = torch.device('cuda', 0)
model = bench.build_model(dev, num_stages=1, train=True)
ds = SyntheticPoseDataset(num_joints=bench.J, img_shape=(bench.H, bench.W), length=2, seed=0)
data = collate([ds[i] for i in range(2)], device=dev)
opt = FlatSGD(model, lr=2e-3, momentum=0.9, weight_decay=1e-4, bias_lr_mult=2.0, bias_decay_mult=0.0, max_grad_norm=35.0)
for _ in range(2):
    train_iteration(model, opt, data, 2e-3)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    x = model.extract_feat(data['img'])
    torch.cuda.synchronize()
for ev in prof.events():
    if ev.name in ('aten::contiguous', 'aten::clone') and ev.stack and any('bottleneck_chain' in f for f in ev.stack[:4]):
        print(ev.name, ev.input_shapes, [f[-60:] for f in ev.stack[:3]])
