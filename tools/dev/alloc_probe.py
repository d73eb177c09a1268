"""Dev probe: does the caching allocator go to the driver (hipMalloc / hipFree) inside steady-state training steps?"""
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
for _ in range(6):
    train_iteration(model, opt, data, 2e-3)
torch.cuda.synchronize()
keys = ('num_device_alloc', 'num_device_free', 'num_alloc_retries', 'reserved_bytes.all.current', 'active_bytes.all.peak', 'num_sync_all_streams')
prev = torch.cuda.memory_stats()
for i in range(10):
    t0 = time.perf_counter()
    train_iteration(model, opt, data, 2e-3)
    dt = (time.perf_counter() - t0) * 1e3
    st = torch.cuda.memory_stats()
    print('step %d host %.1f ms  ' % (i, dt) + '  '.join('%s %+d' % (k.split('.')[0], st.get(k, 0) - prev.get(k, 0)) for k in keys[:3]) +
          '  reserved %.2f GB' % (st['reserved_bytes.all.current'] / 2 ** 30), flush=True)
    prev = st
torch.cuda.synchronize()
