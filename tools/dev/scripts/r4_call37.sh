#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
WGB=1 timeout 600 python3 tools/dev/train_shapes.py > gpurun_out/r4/train_shapes_wg1.txt 2>&1
grep -E "wgrad" gpurun_out/r4/train_shapes_wg1.txt | head -60
