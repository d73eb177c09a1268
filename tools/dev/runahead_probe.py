"""Dev probe: how far ahead of the GPU can the host queue? After a device synchronisation the host's time per step stays at its
unthrottled ~57 ms until something (the runtime's launch queue, an event wait) holds it to the GPU's pace. Prints host and GPU time
of the steps after the synchronisation; run under different runtime environment settings."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from das_amd.datasets import SyntheticPoseDataset, collate
from das_amd import optim
from das_amd.optim import FlatSGD, train_iteration

if len(sys.argv) > 1:
    optim.MAX_RUN_AHEAD = int(sys.argv[1])
dev = torch.device('cuda', 0)
model = bench.build_model(dev, num_stages=4, train=True)
ds = SyntheticPoseDataset(num_joints=bench.J, img_shape=(bench.H, bench.W), length=16, seed=0)
data = collate([ds[i] for i in range(16)], device=dev)
opt = FlatSGD(model, lr=2e-3, momentum=0.9, weight_decay=1e-4, bias_lr_mult=2.0, bias_decay_mult=0.0, max_grad_norm=35.0)
for _ in range(6):
    train_iteration(model, opt, data, 2e-3)
torch.cuda.synchronize()
host, t = [], time.perf_counter()
a = torch.cuda.Event(enable_timing=True); a.record()
for i in range(10):
    train_iteration(model, opt, data, 2e-3)
    n = time.perf_counter(); host.append((n - t) * 1e3); t = n
b = torch.cuda.Event(enable_timing=True); b.record()
torch.cuda.synchronize()
print('host ms per step after a sync:', ' '.join('%.0f' % h for h in host), '| GPU mean %.1f ms' % (a.elapsed_time(b) / 10),
      '| env', {k: v for k, v in os.environ.items() if k.startswith(('ROC_', 'GPU_', 'HIP_', 'AMD_', 'HSA_')) and k not in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES')}, flush=True)
