#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
bash tools/dev/scripts/r4_tune.sh
timeout 600 python3 tools/dev/cu_pressure.py own 16 32 64 > gpurun_out/r4/cu_pressure_own.txt 2>&1
cat gpurun_out/r4/cu_pressure_own.txt
timeout 600 python3 tools/dev/cu_pressure.py share 16 32 64 > gpurun_out/r4/cu_pressure_share.txt 2>&1
cat gpurun_out/r4/cu_pressure_share.txt
