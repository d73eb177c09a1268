#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 900 python3 -m pytest tests/test_hip_backward_elem.py tests/test_hip_kernels.py tests/test_hip_backward_head.py tests/test_model_gpu.py tests/test_train_step_gpu.py tests/test_topologies_gpu.py -q -m gpu -x 2>&1 | tail -5
timeout 900 python3 tools/dev/tune_step.py -r 5 GNMASK=0 > gpurun_out/r4/tune_gn.txt 2>&1
cat gpurun_out/r4/tune_gn.txt
