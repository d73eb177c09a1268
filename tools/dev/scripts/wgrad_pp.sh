#!/bin/bash
# Dev: A/B the ping-pong weight-gradient kernel (DAS_DEV_WGRAD_PP = minimum K) against the default one.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
{
  echo "== parity (default)"
  timeout 600 python -m pytest tests/test_hip_backward.py tests/test_hip_backward_head.py tests/test_flat_paths_gpu.py -x -q -m gpu 2>&1 | tail -5
  echo "== parity (pp, K>=256)"
  DAS_DEV_WGRAD_PP=256 timeout 600 python -m pytest tests/test_hip_backward.py tests/test_hip_backward_head.py tests/test_flat_paths_gpu.py -x -q -m gpu 2>&1 | tail -5
  echo "== base"
  timeout 300 python tools/dev/wgrad_bench.py
  echo "== base, 1024 blocks"
  DAS_DEV_WGRAD_BLOCKS=1024 timeout 300 python tools/dev/wgrad_bench.py
  echo "== base, 512 blocks"
  DAS_DEV_WGRAD_BLOCKS=512 timeout 300 python tools/dev/wgrad_bench.py
  echo "== pp"
  DAS_DEV_WGRAD_PP=256 timeout 300 python tools/dev/wgrad_bench.py
  echo "== base by batch"
  timeout 300 python tools/dev/wgrad_abl.py
  echo "== pp by batch"
  DAS_DEV_WGRAD_PP=256 timeout 300 python tools/dev/wgrad_abl.py
} > gpurun_out/wgrad_pp.log 2>&1
grep -v amdgpu.ids gpurun_out/wgrad_pp.log | tail -80
