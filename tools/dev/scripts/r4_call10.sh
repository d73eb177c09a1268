#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r4j
mkdir -p $O
for ra in 3 4 2b 0; do
timeout 300 python3 - > $O/bench_ra$ra.txt 2>&1 <<P
import sys, time, torch
sys.argv=['bench.py']
import bench
from das_amd import optim
ra='$ra'
optim.MAX_RUN_AHEAD = int(ra.rstrip('b'))
if ra.endswith('b'):
    _E = torch.cuda.Event
    torch.cuda.Event = lambda *a, **k: _E(blocking=True) if not a and not k else _E(*a, **k)
from das_amd.datasets import SyntheticPoseDataset, collate
from das_amd.optim import FlatSGD, train_iteration
dev=torch.device('cuda',0)
model=bench.build_model(dev,num_stages=4,train=True)
ds=SyntheticPoseDataset(num_joints=bench.J,img_shape=(bench.H,bench.W),length=16,seed=0)
data=collate([ds[i] for i in range(16)],device=dev)
opt=FlatSGD(model,lr=2e-3,momentum=0.9,weight_decay=1e-4,bias_lr_mult=2.0,bias_decay_mult=0.0,max_grad_norm=35.0)
for _ in range(3): train_iteration(model,opt,data,2e-3)
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(12): train_iteration(model,opt,data,2e-3)
torch.cuda.synchronize(); print('run-ahead', ra, 'ms/step', (time.perf_counter()-t0)*1000/12)
P
cat $O/bench_ra$ra.txt | tail -1
done
timeout 600 python3 -m pytest tests/test_model_gpu.py -x -q -m gpu -k unconsumed 2>&1 | grep -E "^E|passed|failed" | head
