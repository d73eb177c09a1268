#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 600 python3 tools/dev/bn_shapes.py > gpurun_out/r4/bn_shapes.txt 2>&1
head -70 gpurun_out/r4/bn_shapes.txt
