#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6k
timeout 900 python -m pytest tests/test_conv_tiles_gpu.py tests/test_bn_fused_gpu.py -q -k "kstream" > gpurun_out/r6k/tests.log 2>&1; echo "kstream tests rc=$?" > gpurun_out/r6k/rc.txt
timeout 600 python tools/dev/kstream_ab.py > gpurun_out/r6k/kstream_ab.md 2>&1; echo "ab rc=$?" >> gpurun_out/r6k/rc.txt
timeout 900 python tools/dev/tune_step.py -n 10 -r 5 conv.kstream=1 > gpurun_out/r6k/tune.txt 2>&1; echo "tune rc=$?" >> gpurun_out/r6k/rc.txt
cat gpurun_out/r6k/rc.txt; tail -15 gpurun_out/r6k/tests.log | cut -c1-250; cat gpurun_out/r6k/kstream_ab.md; tail -3 gpurun_out/r6k/tune.txt
