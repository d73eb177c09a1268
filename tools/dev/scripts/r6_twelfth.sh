#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6m
timeout 900 python -m pytest tests/test_bn_fused_gpu.py tests/test_conv_tiles_gpu.py tests/test_hip_backward.py -q -p no:cacheprovider > gpurun_out/r6m/tests.log 2>&1; echo "tests rc=$?" > gpurun_out/r6m/rc.txt
timeout 900 python tools/dev/tune_step.py -n 10 -r 5 conv.epi_prefetch=0 > gpurun_out/r6m/tune.txt 2>&1; echo "tune rc=$?" >> gpurun_out/r6m/rc.txt
cat gpurun_out/r6m/rc.txt; tail -4 gpurun_out/r6m/tests.log | cut -c1-250; tail -3 gpurun_out/r6m/tune.txt
