#!/bin/bash
# (record of a negative result, profiles/r05_negative_results.txt: the DAS_STREAM_PIPE6 code path this compared was removed again)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5
mkdir -p $O
timeout 900 python -m pytest tests/test_bn_fused_gpu.py tests/test_conv_tiles_gpu.py -q -x 2>&1 | tail -5 | tee $O/pipe_tests.txt
DASLIB=libdas_hip_nopipe.so timeout 600 python tools/dev/train_shapes.py > $O/ts_nopipe.txt 2>&1
timeout 600 python tools/dev/train_shapes.py > $O/ts_pipe.txt 2>&1
DASLIB=libdas_hip_nopipe.so timeout 600 python tools/dev/train_shapes.py > $O/ts_nopipe2.txt 2>&1
timeout 600 python tools/dev/train_shapes.py > $O/ts_pipe2.txt 2>&1
for f in nopipe pipe nopipe2 pipe2; do echo "== $f"; sed -n 2,4p $O/ts_$f.txt; grep "x1_stream" $O/ts_$f.txt | grep "mode=rb\|mode=bm\|mode=rbm" | cut -c1-175; done
