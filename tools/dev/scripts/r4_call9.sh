#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r4i
mkdir -p $O
timeout 300 python3 tools/dev/step_times.py 12 > $O/step_times.txt 2>&1
tail -4 $O/step_times.txt
for ra in 2 0; do
timeout 300 python3 - > $O/bench_ra$ra.txt 2>&1 <<P
import sys, time, torch
sys.argv=['bench.py']
import bench
from das_amd import optim
optim.MAX_RUN_AHEAD = $ra
from das_amd.datasets import SyntheticPoseDataset, collate
from das_amd.optim import FlatSGD, train_iteration
dev=torch.device('cuda',0)
model=bench.build_model(dev,num_stages=4,train=True)
ds=SyntheticPoseDataset(num_joints=bench.J,img_shape=(bench.H,bench.W),length=16,seed=0)
data=collate([ds[i] for i in range(16)],device=dev)
opt=FlatSGD(model,lr=2e-3,momentum=0.9,weight_decay=1e-4,bias_lr_mult=2.0,bias_decay_mult=0.0,max_grad_norm=35.0)
for _ in range(3): train_iteration(model,opt,data,2e-3)
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(10): train_iteration(model,opt,data,2e-3)
torch.cuda.synchronize(); print('run-ahead', $ra, 'ms/step', (time.perf_counter()-t0)*100)
P
cat $O/bench_ra$ra.txt | tail -1
done
timeout 600 python3 -m pytest tests/test_model_gpu.py tests/test_flat_paths_gpu.py -x -q -m gpu 2>&1 | tail -3
