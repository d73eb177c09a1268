#!/bin/bash
# Dev: rocprofv3 kernel trace of the infer bench (summary -> gpurun_out/prof/infer_stats.md).
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
rm -rf gpurun_out/prof/inf
CMD="python3 bench.py --workload infer --steps 10 --warmup 3 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/inf -o inf -- $CMD > gpurun_out/prof/infer_prof.log 2>&1
db=$(find gpurun_out/prof/inf -name "*.db" | head -1)
python3 tools/dev/rocprof_summary.py "$db" gpurun_out/prof/infer_stats.md "rocprofv3 --kernel-trace --stats -- $CMD"
python3 tools/dev/rocprof_gaps.py "$db" > gpurun_out/prof/infer_gaps.txt 2>&1
rm -rf gpurun_out/prof/inf
head -32 gpurun_out/prof/infer_stats.md | cut -c1-150
tail -4 gpurun_out/prof/infer_gaps.txt
