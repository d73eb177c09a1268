#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 3000 python3 -m pytest tests -q -m gpu -x --durations=15 > gpurun_out/r4/pytest_full.txt 2>&1
tail -40 gpurun_out/r4/pytest_full.txt
