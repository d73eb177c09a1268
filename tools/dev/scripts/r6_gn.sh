#!/bin/bash
# Round 6: GroupNorm sums in the conv epilogue — head / full-width tests, step A/B, inference A/B
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6q
mkdir -p $O; rm -rf $O/*
trap 'echo "exit $?" >> $O/rc.txt' EXIT
timeout 1500 python3 -m pytest tests -q -m gpu -x -k "head or full_width or groupnorm or gn or graph or detector or smoke or model or tiles" 2>&1 | tail -15 > $O/tests.log
echo "tests rc=$?" >> $O/rc.txt
tail -4 $O/tests.log
timeout 900 python3 tools/dev/tune_step.py -n 10 -r 5 GNF=0 > $O/tune.txt 2>&1
tail -3 $O/tune.txt
timeout 600 python3 tools/dev/tune_infer.py -n 20 -r 5 GNF=0 > $O/tune_infer.txt 2>&1
tail -3 $O/tune_infer.txt
