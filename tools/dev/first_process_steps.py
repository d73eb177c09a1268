"""Dev tool: per-step GPU time of the train step from the very first step of a process (run it as the FIRST GPU process of a
fresh box): how many steps does a fresh box / process need before the step time is the steady one?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from das_amd.datasets import SyntheticPoseDataset, collate
from das_amd.optim import FlatSGD, train_iteration

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
if len(sys.argv) > 2 and sys.argv[2] == 'nogc':
    import gc
    gc.collect(); gc.freeze(); gc.disable()
    print('gc frozen + disabled')
dev = torch.device('cuda', 0)
t0 = time.perf_counter()
model = bench.build_model(dev, num_stages=4, train=True)
ds = SyntheticPoseDataset(num_joints=bench.J, img_shape=(bench.H, bench.W), length=16, seed=0)
data = collate([ds[i] for i in range(16)], device=dev)
opt = FlatSGD(model, lr=2e-3, momentum=0.9, weight_decay=1e-4, bias_lr_mult=2.0, bias_decay_mult=0.0, max_grad_norm=35.0)
torch.cuda.synchronize()
print(f'setup {time.perf_counter() - t0:.1f} s')
evs = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
host = []
evs[0].record()
for i in range(N):
    h0 = time.perf_counter()
    train_iteration(model, opt, data, 2e-3)
    host.append((time.perf_counter() - h0) * 1e3)
    evs[i + 1].record()
torch.cuda.synchronize()
ts = [evs[i].elapsed_time(evs[i + 1]) for i in range(N)]
print('gpu ms per step :', ' '.join(f'{t:.1f}' for t in ts))
print('host ms per step:', ' '.join(f'{t:.1f}' for t in host))
print('reserved GB', torch.cuda.memory_reserved() / 2 ** 30)
