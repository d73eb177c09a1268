#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6n
timeout 900 python -m pytest tests/test_bn_fused_gpu.py tests/test_conv_tiles_gpu.py -q -p no:cacheprovider > gpurun_out/r6n/tests.log 2>&1; echo "tests rc=$?" > gpurun_out/r6n/rc.txt
timeout 900 python tools/dev/tune_step.py -n 10 -r 5 conv.kstream=0 conv.kstream=1 conv.kstream=7 conv.kstream=15 > gpurun_out/r6n/tune.txt 2>&1; echo "tune rc=$?" >> gpurun_out/r6n/rc.txt
cat gpurun_out/r6n/rc.txt; tail -3 gpurun_out/r6n/tests.log | cut -c1-200; tail -6 gpurun_out/r6n/tune.txt
