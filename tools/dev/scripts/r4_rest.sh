#!/bin/bash
# the tests after the first failure of the -x run
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 3000 python3 -m pytest tests/test_full_width_gpu.py tests/test_graphs_gpu.py tests/test_hip_backward.py tests/test_hip_backward_head.py tests/test_hip_kernels.py tests/test_loss_gpu.py tests/test_model_gpu.py tests/test_pipeline_gpu.py tests/test_syncbn_gpu.py tests/test_topologies_gpu.py tests/test_train_gpu.py tests/test_train_step_gpu.py tests/test_flow_gpu.py -q -m gpu --durations=8 > gpurun_out/r4/pytest_rest.txt 2>&1
tail -25 gpurun_out/r4/pytest_rest.txt
