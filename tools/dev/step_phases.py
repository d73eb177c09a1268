"""Dev tool: GPU time of the phases of a train bench step (events on the main stream): forward up to the loss function,
the loss function (target assignment + four losses), parse/sum, backward, optimizer."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import time
import torch
import bench
from das_amd import losses, pose_heads
from das_amd.datasets import SyntheticPoseDataset, collate
from das_amd.optim import FlatSGD

B = 16
dev = torch.device('cuda', 0)
model = bench.build_model(dev, num_stages=4, train=True)
ds = SyntheticPoseDataset(num_joints=bench.J, img_shape=(bench.H, bench.W), length=B, seed=0)
data = collate([ds[i] for i in range(B)], device=dev)
opt = FlatSGD(model, lr=2e-3, momentum=0.9, weight_decay=1e-4, bias_lr_mult=2.0, bias_decay_mult=0.0, max_grad_norm=35.0)
ev = {}


cpu = {}


def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    ev.setdefault(name, []).append(e)
    cpu.setdefault(name, []).append(time.perf_counter())


orig = losses.das_head_loss_rows


def timed_loss(*a, **k):
    mark('loss_in')
    r = orig(*a, **k)
    mark('loss_out')
    return r


losses.das_head_loss_rows = timed_loss
N = 8
for it in range(3 + N):
    if it == 3:
        ev.clear(); cpu.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    mark('start')
    opt.zero_grad()
    out = model.train_step(data, None)
    mark('fwd_done')
    out['loss'].backward()
    mark('bwd_done')
    opt.all_reduce_grads()
    opt.step(2e-3)
    mark('end')
torch.cuda.synchronize()
print(f'wall ms/step {(time.perf_counter() - t0) / N * 1e3:.2f}')
names = ['start', 'loss_in', 'loss_out', 'fwd_done', 'bwd_done', 'end']
for a, b in zip(names, names[1:]):
    ms = sum(x.elapsed_time(y) for x, y in zip(ev[a], ev[b])) / N
    cms = sum(y - x for x, y in zip(cpu[a], cpu[b])) / N * 1e3
    print(f'{a:9s} -> {b:9s} GPU timeline {ms:8.3f} ms   CPU issue time {cms:8.3f} ms')
