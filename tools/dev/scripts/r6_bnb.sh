#!/bin/bash
# Round 6: conv_epilogue's BNB template flag — tests, then the step on the new and the old library, alternating processes
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6s
mkdir -p $O; rm -rf $O/*
trap 'echo "exit $?" >> $O/rc.txt' EXIT
timeout 1500 python3 -m pytest tests/test_conv_tiles_gpu.py tests/test_bn_fused_gpu.py tests/test_hip_backward.py tests/test_hip_kernels.py -q -m gpu -x 2>&1 | tail -6 > $O/tests.log
echo "tests rc=$?" >> $O/rc.txt
tail -3 $O/tests.log
for i in 1 2 3; do for l in libdas_hip_old.so libdas_hip.so; do echo "== $l" >> $O/ab.txt; DASLIB=$l timeout 400 python3 tools/dev/tune_step.py -n 10 -r 3 2>&1 | grep "defaults" >> $O/ab.txt; done; done
cat $O/ab.txt
