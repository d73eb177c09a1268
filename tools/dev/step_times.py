"""Dev tool: per-step wall time and host (launch-queueing) time of the train bench's step in a fresh process; with
`graphs` as second argument the trunk is captured as hipGraphs after step 3. usage: step_times.py [steps] [graphs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from das_amd.datasets import SyntheticPoseDataset, collate
from das_amd.optim import FlatSGD, train_iteration

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B = 16
dev = torch.device('cuda', 0)
t0 = time.perf_counter()
model = bench.build_model(dev, num_stages=4, train=True)
ds = SyntheticPoseDataset(num_joints=bench.J, img_shape=(bench.H, bench.W), length=B, seed=0)
data = collate([ds[i] for i in range(B)], device=dev)
opt = FlatSGD(model, lr=2e-3, momentum=0.9, weight_decay=1e-4, bias_lr_mult=2.0, bias_decay_mult=0.0, max_grad_norm=35.0)
torch.cuda.synchronize()
print(f'setup {time.perf_counter() - t0:.2f} s')
ts, hs = [], []
for i in range(n):
    if i == 3 and len(sys.argv) > 2 and sys.argv[2] == 'graphs':
        from das_amd.graphs import enable_trunk_graphs
        enable_trunk_graphs(model, opt, data['img'])
        print('trunk captured as hipGraphs before step 3')
    torch.cuda.synchronize()
    t = time.perf_counter()
    train_iteration(model, opt, data, 2e-3)
    th = time.perf_counter()              # everything of the step is queued: the host's part ends here
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t) * 1e3)
    hs.append((th - t) * 1e3)
    if i in (2, 5):
        from das_amd import ops as _o
        print('after step', i, 'schedules built', _o.last_wgrad_plan()['schedules_built'])
print('wall ms', ' '.join(f'{v:.1f}' for v in ts))
print('host ms', ' '.join(f'{v:.1f}' for v in hs))
from das_amd import ops
print('schedules built', ops.last_wgrad_plan()['schedules_built'])
