#!/usr/bin/env python3
"""What does the per-step host synchronisation of the target assignment cost? Median step time with the real
prepare_targets against a run that reuses the first step's prep (same synthetic batch every step: identical result)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import bench
from das_amd.datasets import SyntheticPoseDataset, collate
from das_amd.optim import FlatSGD, train_iteration

dev = torch.device('cuda', 0)
model = bench.build_model(dev, num_stages=4, train=True)
ds = SyntheticPoseDataset(num_joints=bench.J, img_shape=(bench.H, bench.W), length=16, seed=0)
data = collate([ds[i] for i in range(16)], device=dev)
opt = FlatSGD(model, lr=2e-3, momentum=0.9, weight_decay=1e-4, bias_lr_mult=2.0, bias_decay_mult=0.0, max_grad_norm=35.0)
for _ in range(4):
    train_iteration(model, opt, data, 2e-3)
head = model.bbox_head
real = head.prepare_targets
cache = {}


def cached(*a, **k):
    if 'p' not in cache:
        cache['p'] = real(*a, **k)
    return cache['p']


def run(n=16):
    for _ in range(2):
        train_iteration(model, opt, data, 2e-3)
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    evs[0].record()
    for i in range(n):
        train_iteration(model, opt, data, 2e-3)
        evs[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(n))
    return ts[n // 2]


for rep in range(3):
    head.prepare_targets = real
    a = run()
    head.prepare_targets = cached
    b = run()
    print('real prepare_targets %.2f ms   cached prep %.2f ms   (%+.2f)' % (a, b, b - a), flush=True)
