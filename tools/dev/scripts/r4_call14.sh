#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 900 python3 -m pytest tests/test_upconv_gpu.py tests/test_train_gpu.py tests/test_model_gpu.py tests/test_bn_fused_gpu.py -q -m gpu -s 2>&1 | tail -40 > gpurun_out/r4/pytest_upconv.txt
cat gpurun_out/r4/pytest_upconv.txt
timeout 600 python3 tools/dev/tune_step.py -r 3 UPCONV=0 > gpurun_out/r4/tune_upconv.txt 2>&1
cat gpurun_out/r4/tune_upconv.txt
