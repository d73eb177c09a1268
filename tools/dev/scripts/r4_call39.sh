#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 900 python3 -m pytest tests/test_bn_fused_gpu.py -q -m gpu -x 2>&1 | tail -8
timeout 900 python3 -m pytest tests/test_train_gpu.py tests/test_model_gpu.py tests/test_topologies_gpu.py tests/test_graphs_gpu.py tests/test_train_step_gpu.py tests/test_ddp_gpu.py tests/test_conv_tiles_gpu.py -q -m gpu -x 2>&1 | tail -4
timeout 900 python3 tools/dev/tune_step.py -r 5 RESBITS=0 > gpurun_out/r4/tune_resbits.txt 2>&1
cat gpurun_out/r4/tune_resbits.txt
