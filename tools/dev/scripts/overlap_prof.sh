#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out/prof
rm -rf gpurun_out/prof/ov_$tag
timeout 900 rocprofv3 --kernel-trace -d gpurun_out/prof/ov_$tag -o tr -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline $* > gpurun_out/prof/ov_${tag}.log 2>&1
db=$(find gpurun_out/prof/ov_$tag -name "*.db" | head -1)
python3 tools/dev/rocprof_overlap.py "$db" > gpurun_out/prof/ov_${tag}.txt 2>&1
rm -rf gpurun_out/prof/ov_$tag
cat gpurun_out/prof/ov_${tag}.txt | head -30
