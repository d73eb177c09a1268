#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 900 python3 tools/dev/loss_track.py 20 2>&1 | grep -v amdgpu | tee gpurun_out/r4/loss_track.txt
