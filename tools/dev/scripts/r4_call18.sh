#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 600 python3 tools/dev/train_shapes.py > gpurun_out/r4/train_shapes.txt 2>&1
head -100 gpurun_out/r4/train_shapes.txt
