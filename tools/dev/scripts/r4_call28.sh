#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 900 python3 tools/dev/tune_step.py -r 5 DUAL=0 DEFER=0 > gpurun_out/r4/tune_dual.txt 2>&1
cat gpurun_out/r4/tune_dual.txt
