"""Dev probe: per-LINE host time inside DASHead.forward_rows in a steady-state step (sys.settrace on that frame only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from das_amd.datasets import SyntheticPoseDataset, collate
from das_amd.optim import FlatSGD, train_iteration
from das_amd import pose_heads

dev = torch.device('cuda', 0)
model = bench.build_model(dev, num_stages=4, train=True)
ds = SyntheticPoseDataset(num_joints=bench.J, img_shape=(bench.H, bench.W), length=16, seed=0)
data = collate([ds[i] for i in range(16)], device=dev)
opt = FlatSGD(model, lr=2e-3, momentum=0.9, weight_decay=1e-4, bias_lr_mult=2.0, bias_decay_mult=0.0, max_grad_norm=35.0)
code = pose_heads.DASHead.forward_rows.__code__ if hasattr(pose_heads, 'DASHead') else None
lines, last = {}, [None, 0.0]


def local(frame, event, arg):
    now = time.perf_counter()
    if last[0] is not None:
        lines[last[0]] = lines.get(last[0], 0.0) + (now - last[1]) * 1e3
    last[0], last[1] = (frame.f_lineno if event != 'return' else None), time.perf_counter()
    return local


def tracer(frame, event, arg):
    if event == 'call' and frame.f_code is code:
        last[0], last[1] = frame.f_lineno, time.perf_counter()
        return local
    return None


for _ in range(8):
    train_iteration(model, opt, data, 2e-3)
sys.settrace(tracer)
train_iteration(model, opt, data, 2e-3)
sys.settrace(None)
torch.cuda.synchronize()
src = open(pose_heads.__file__).read().split('\n')
for ln, ms in sorted(lines.items(), key=lambda kv: -kv[1])[:8]:
    print('%7.2f ms  line %d: %s' % (ms, ln, src[ln - 1].strip()[:120]))
