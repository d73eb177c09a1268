#!/usr/bin/env python3
"""Twenty optimisation steps of the benchmark model (bf16, B = 16, the same synthetic batch) with the fused train-mode forms
of rounds 4 and 5 ON against all of them OFF (the reference's order of operations, kernel by kernel): the loss curves must track
each other — a fusion that corrupted a gradient or a running statistic would show within a few steps."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import bench
from das_amd import autograd as ag, nn as dnn
from das_amd.datasets import SyntheticPoseDataset, collate
from das_amd.optim import FlatSGD, train_iteration

dev = torch.device('cuda', 0)
ds = SyntheticPoseDataset(num_joints=bench.J, img_shape=(bench.H, bench.W), length=16, seed=0)
data = collate([ds[i] for i in range(16)], device=dev)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def run(on):
    from das_amd import losses, optim
    dnn.UPCONV_AT_LOW_RES = dnn.UPMERGE_FUSED = dnn.DEFERRED_SKIPS = on
    ag.DUAL_APPLY = ag.MASK_BITS = ag.GN_REMASK = ag.RES_BITS = on
    # (round 5: chained consumers, the fused DCNv2 forward, padded flat storage, zero pools, grouped finalize launches, fused targets)
    dnn.CHAIN_CONSUMERS = ag.DCN_FUSED = optim.PAD_ODD_CHANNELS = dnn.ZEROED_GN_WS = ag.FINALIZE_MANY = losses.FUSED_TARGETS = on
    torch.manual_seed(0)
    model = bench.build_model(dev, num_stages=4, train=True)
    opt = FlatSGD(model, lr=2e-3, momentum=0.9, weight_decay=1e-4, bias_lr_mult=2.0, bias_decay_mult=0.0, max_grad_norm=35.0)
    out = []
    for _ in range(N):
        out.append(float(train_iteration(model, opt, data, 2e-3)['log_vars']['loss']))
    rv = torch.cat([b.flatten().float() for n, b in model.named_buffers() if n.endswith('running_var')])
    return out, rv


a, rva = run(True)
b, rvb = run(False)
c, rvc = run(False)
print('step   fused      unfused    unfused(again)')
for i, (x, y, z) in enumerate(zip(a, b, c)):
    print('%3d  %10.3f %10.3f %10.3f' % (i, x, y, z))
rel = lambda u, v: float((u - v).abs().max() / v.abs().max())
print('running_var after %d steps: fused vs unfused %.2e; unfused vs unfused %.2e' % (N, rel(rva, rvb), rel(rvc, rvb)))
