#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r4c
mkdir -p $O
timeout 300 python3 tools/dev/pt3_stamps.py > $O/pt3_stamps.txt 2>&1
cat $O/pt3_stamps.txt
timeout 300 python3 tools/dev/pt3_stamps.py conv.pt3_mintiles=0 > $O/pt3_stamps_old.txt 2>&1
