"""Dev tool: how long does the main stream idle at the end of backward, waiting for the weight-gradient side streams?
An event pair around the joins of autograd.finish_backward (recorded on the main stream right before the first wait_stream and
right after the last one): with the host running ahead, the pair's elapsed time is the main stream's idle time at the join.
Also: the step with 1 / 2 / 3 side streams and with the last N batches kept on the main stream (none built: just the measure)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from das_amd import autograd as ag
from das_amd.datasets import SyntheticPoseDataset, collate
from das_amd.optim import FlatSGD, train_iteration

dev = torch.device('cuda', 0)
model = bench.build_model(dev, num_stages=4, train=True)
ds = SyntheticPoseDataset(num_joints=bench.J, img_shape=(bench.H, bench.W), length=16, seed=0)
data = collate([ds[i] for i in range(16)], device=dev)
opt = FlatSGD(model, lr=2e-3, momentum=0.9, weight_decay=1e-4, bias_lr_mult=2.0, bias_decay_mult=0.0, max_grad_norm=35.0)
pairs = []
orig = ag.finish_backward


def probed():
    ag.flush_wgrads()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    orig()
    b.record()
    pairs.append((a, b))


ag.finish_backward = probed
import das_amd.optim as _o
for _ in range(4):
    train_iteration(model, opt, data, 2e-3)
torch.cuda.synchronize()
pairs.clear()
N = 12
evs = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
evs[0].record()
for i in range(N):
    train_iteration(model, opt, data, 2e-3)
    evs[i + 1].record()
torch.cuda.synchronize()
steps = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(N))
waits = sorted(a.elapsed_time(b) for a, b in pairs)
print(f'step median {steps[N // 2]:.2f} ms; joins per step {len(pairs) / N:.1f}; main-stream wait at the join: median {waits[len(waits) // 2]:.3f} ms, '
      f'min {waits[0]:.3f}, max {waits[-1]:.3f}')
