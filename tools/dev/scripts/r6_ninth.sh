#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6i
timeout 900 python -m pytest tests/test_conv_tiles_gpu.py -q -x -k "splitk" > gpurun_out/r6i/splitk_tests.log 2>&1; echo "splitk tests rc=$?" > gpurun_out/r6i/rc.txt
timeout 900 python tools/dev/tune_step.py -n 10 -r 5 conv.splitk_inkernel=1 > gpurun_out/r6i/tune.txt 2>&1; echo "tune rc=$?" >> gpurun_out/r6i/rc.txt
timeout 600 python -m pytest tests/test_cli_gpu.py -q -k two_ranks > gpurun_out/r6i/two_rank.log 2>&1; echo "tworank rc=$?" >> gpurun_out/r6i/rc.txt
cat gpurun_out/r6i/rc.txt; tail -4 gpurun_out/r6i/splitk_tests.log | cut -c1-300; tail -3 gpurun_out/r6i/tune.txt; tail -2 gpurun_out/r6i/two_rank.log
