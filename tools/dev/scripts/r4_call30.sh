#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 900 python3 -m pytest tests/test_upconv_gpu.py tests/test_hip_kernels.py tests/test_model_gpu.py tests/test_train_gpu.py tests/test_hip_backward_head.py -q -m gpu -x 2>&1 | tail -3
timeout 600 python3 tools/dev/bn_shapes.py > gpurun_out/r4/bn_shapes2.txt 2>&1
head -3 gpurun_out/r4/bn_shapes2.txt; grep -E ", 0\)$|'upstats'" gpurun_out/r4/bn_shapes2.txt | head -30
timeout 600 python3 tools/dev/tune_step.py -n 12 -r 3 > gpurun_out/r4/tune_now.txt 2>&1; cat gpurun_out/r4/tune_now.txt
