#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 900 python3 tools/dev/tune_step.py -r 3 conv.big_minblocks=210 conv.big_minblocks=420 conv.splitk_target=420 conv.splitk_target=420,conv.splitk_minsteps=8 > gpurun_out/r4/tune_bigmin.txt 2>&1
cat gpurun_out/r4/tune_bigmin.txt
