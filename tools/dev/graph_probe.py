"""Dev probe: capture the trunk graphs at the bench shape after K eager steps (argv: K, width)."""
import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import time
import torch
import bench
from das_amd.datasets import SyntheticPoseDataset, collate
from das_amd.optim import FlatSGD, train_iteration
from das_amd.graphs import enable_trunk_graphs

K = int(sys.argv[1]) if len(sys.argv) > 1 else 1
W = int(sys.argv[2]) if len(sys.argv) > 2 else bench.W
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
dev = torch.device('cuda', 0)
model = bench.build_model(dev, num_stages=4, train=True)
ds = SyntheticPoseDataset(num_joints=bench.J, img_shape=(bench.H, W), length=B, seed=0)
data = collate([ds[i] for i in range(B)], device=dev)
opt = FlatSGD(model, lr=2e-3, momentum=0.9, weight_decay=1e-4, bias_lr_mult=2.0, bias_decay_mult=0.0, max_grad_norm=35.0)
for _ in range(K):
    train_iteration(model, opt, data, 2e-3)
torch.cuda.synchronize()
print('eager steps done', K, W, B, flush=True)
t0 = time.time()
enable_trunk_graphs(model, opt, data['img'])
print(f'captured in {time.time() - t0:.1f} s', flush=True)
for _ in range(3):
    out = train_iteration(model, opt, data, 2e-3)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(6):
    out = train_iteration(model, opt, data, 2e-3)
torch.cuda.synchronize()
print(f'{(time.perf_counter() - t0) / 6 * 1e3:.2f} ms/step with graphs', out['log_vars'], flush=True)
