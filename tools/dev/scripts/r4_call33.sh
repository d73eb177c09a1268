#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 900 python3 tools/dev/tune_step.py -r 5 BITS=0 > gpurun_out/r4/tune_bits.txt 2>&1
cat gpurun_out/r4/tune_bits.txt
