#!/bin/bash
# Dev: rocprofv3 kernel trace of the train bench (summary -> gpurun_out/prof/train_stats.md, idle gaps).
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
rm -rf gpurun_out/prof/tr
CMD="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline"
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/tr -o tr -- $CMD > gpurun_out/prof/train_prof.log 2>&1
echo "rc=$?"
tail -1 gpurun_out/prof/train_prof.log | cut -c1-200
db=$(find gpurun_out/prof/tr -name "*.db" | head -1)
python3 tools/dev/rocprof_summary.py "$db" gpurun_out/prof/train_stats.md "rocprofv3 --kernel-trace --stats -- $CMD"
python3 tools/dev/rocprof_gaps.py "$db" > gpurun_out/prof/train_gaps.txt 2>&1
rm -rf gpurun_out/prof/tr
head -45 gpurun_out/prof/train_stats.md | cut -c1-170
tail -5 gpurun_out/prof/train_gaps.txt
