#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 1500 python3 -m pytest tests/test_upconv_gpu.py tests/test_hip_kernels.py tests/test_hip_backward_elem.py tests/test_train_gpu.py tests/test_bn_fused_gpu.py tests/test_hip_backward.py tests/test_model_gpu.py -q -m gpu 2>&1 | tail -15 > gpurun_out/r4/pytest_fold.txt
cat gpurun_out/r4/pytest_fold.txt
timeout 300 python3 tools/dev/upstats_bench.py 2>&1 | grep -v amdgpu | tee gpurun_out/r4/upstats_bench.txt
timeout 600 python3 tools/dev/tune_step.py -r 3 UPCONV=0 > gpurun_out/r4/tune_upconv.txt 2>&1
cat gpurun_out/r4/tune_upconv.txt
