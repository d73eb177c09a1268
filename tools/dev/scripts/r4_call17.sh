#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 600 python3 tools/dev/glue_sources.py > gpurun_out/r4/glue.txt 2>&1
tail -80 gpurun_out/r4/glue.txt
