#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r4e
mkdir -p $O
timeout 300 python3 tools/dev/pt3_single.py > $O/pt3_single.txt 2>&1
cat $O/pt3_single.txt
