#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r4f
mkdir -p $O
timeout 600 python3 tools/dev/cu_pressure.py own 16 32 64 > $O/cu_pressure_own.txt 2>&1
cat $O/cu_pressure_own.txt
timeout 600 python3 tools/dev/cu_pressure.py share 16 32 64 > $O/cu_pressure_share.txt 2>&1
cat $O/cu_pressure_share.txt
