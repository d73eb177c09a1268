"""Dev tool: the launch-count changes of round 5 (odd channel counts padded inside the flat storage, zeroed GroupNorm
workspaces from pools, one finalize launch per fused BatchNorm consumer) ON vs OFF in ONE process: each configuration gets
its own model + optimizer (the storage layout differs), the runs are interleaved R times, median step per configuration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from das_amd import autograd as ag, nn as dnn, optim
from das_amd.datasets import SyntheticPoseDataset, collate
from das_amd.optim import FlatSGD, train_iteration

R, N = int(os.environ.get('R', 4)), int(os.environ.get('N', 10))
dev = torch.device('cuda', 0)
ds = SyntheticPoseDataset(num_joints=bench.J, img_shape=(bench.H, bench.W), length=16, seed=0)
data = collate([ds[i] for i in range(16)], device=dev)


def setup(on):
    optim.PAD_ODD_CHANNELS = dnn.ZEROED_GN_WS = ag.FINALIZE_MANY = on


cfgs = {}
for on in (True, False):
    setup(on)
    model = bench.build_model(dev, num_stages=4, train=True)
    opt = FlatSGD(model, lr=2e-3, momentum=0.9, weight_decay=1e-4, bias_lr_mult=2.0, bias_decay_mult=0.0, max_grad_norm=35.0)
    for _ in range(4):
        train_iteration(model, opt, data, 2e-3)
    cfgs[on] = (model, opt)
res = {True: [], False: []}
for rep in range(R):
    for on in (True, False):
        setup(on)
        model, opt = cfgs[on]
        for _ in range(2):
            train_iteration(model, opt, data, 2e-3)
        torch.cuda.synchronize()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
        evs[0].record()
        for i in range(N):
            train_iteration(model, opt, data, 2e-3)
            evs[i + 1].record()
        torch.cuda.synchronize()
        ts = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(N))
        res[on].append(ts[N // 2])
for on in (True, False):
    v = sorted(res[on])
    print(f'{"on " if on else "off"}  median {v[len(v) // 2]:7.2f} ms   range {v[0]:.2f} .. {v[-1]:.2f}')
