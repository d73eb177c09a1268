#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6g
timeout 900 python tools/dev/tune_step.py -n 10 -r 5 conv.stream_nt=1 conv.stream_nt=4 conv.stream_nt=5 bn.nt_fwd=7,bn.nt_bwd=7 conv.stream_nt=5,bn.nt_fwd=7,bn.nt_bwd=7 conv.stream_nt=1,bn.nt_fwd=7,bn.nt_bwd=7 > gpurun_out/r6g/tune.txt 2>&1; echo "tune rc=$?" > gpurun_out/r6g/rc.txt
timeout 600 python -m pytest tests/test_conv_tiles_gpu.py tests/test_bn_fused_gpu.py -q -x -k "stream" > gpurun_out/r6g/stream_tests.log 2>&1; echo "stream tests rc=$?" >> gpurun_out/r6g/rc.txt
cat gpurun_out/r6g/rc.txt; tail -8 gpurun_out/r6g/tune.txt; tail -3 gpurun_out/r6g/stream_tests.log
