#!/bin/bash
# Dev: train bench under a few wgrad tuning overrides.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
{
  echo "== default"
  timeout 600 python bench.py --steps 10 --warmup 3 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['family_ms_per_step'], d['roofline']['achieved'], d['last_losses'])"
  echo "== 1024 blocks"
  DAS_DEV_WGRAD_BLOCKS=1024 timeout 600 python bench.py --steps 10 --warmup 3 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['family_ms_per_step'], d['roofline']['achieved'])"
  echo "== pp K>=1024"
  DAS_DEV_WGRAD_PP=1024 timeout 600 python bench.py --steps 10 --warmup 3 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['family_ms_per_step'], d['roofline']['achieved'])"
} > gpurun_out/train_ab.log 2>&1
grep -v amdgpu.ids gpurun_out/train_ab.log | tail -20
