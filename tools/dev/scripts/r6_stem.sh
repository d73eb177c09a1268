#!/bin/bash
# Round 6: the stem kernel — tests, step A/B, inference A/B
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6p
mkdir -p $O; rm -rf $O/*
trap 'echo "exit $?" >> $O/rc.txt' EXIT
timeout 900 python3 -m pytest tests/test_conv_tiles_gpu.py -q -m gpu -k "stem7x7" -x 2>&1 | tail -15 > $O/tests.log
echo "tests rc=$?" >> $O/rc.txt
tail -5 $O/tests.log
timeout 900 python3 tools/dev/tune_step.py -n 10 -r 5 conv.stem7x7=0 > $O/tune.txt 2>&1
echo "tune rc=$?" >> $O/rc.txt
tail -3 $O/tune.txt
timeout 600 python3 tools/dev/tune_infer.py -n 20 -r 5 conv.stem7x7=0 > $O/tune_infer.txt 2>&1
tail -3 $O/tune_infer.txt
timeout 300 python3 tools/dev/stem_ab.py > $O/stem_ab.md 2>&1
tail -9 $O/stem_ab.md
