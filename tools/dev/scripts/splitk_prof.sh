#!/bin/bash
# Dev: kernel-level times of the split-K bench (B = $1)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/prof; rm -rf gpurun_out/prof/sk
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/sk -o sk -- python3 tools/dev/splitk_bench.py $1 > gpurun_out/prof/sk.log 2>&1
db=$(find gpurun_out/prof/sk -name "*.db" | head -1)
python3 tools/dev/rocprof_summary.py "$db" gpurun_out/prof/sk_stats.md "splitk_bench $1"
rm -rf gpurun_out/prof/sk
grep -E "splitk|glds" gpurun_out/prof/sk_stats.md | cut -c1-150
