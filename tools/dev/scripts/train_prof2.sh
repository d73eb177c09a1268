#!/bin/bash
# Dev: rocprofv3 kernel trace of the train bench: $1 = tag, rest = extra bench flags. Summary -> gpurun_out/prof/<tag>_stats.md
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out/prof
rm -rf gpurun_out/prof/tr_$tag
CMD="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline $*"
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/tr_$tag -o tr -- $CMD > gpurun_out/prof/${tag}_prof.log 2>&1
echo "rc=$?"
tail -1 gpurun_out/prof/${tag}_prof.log | cut -c1-160
db=$(find gpurun_out/prof/tr_$tag -name "*.db" | head -1)
python3 tools/dev/rocprof_summary.py "$db" gpurun_out/prof/${tag}_stats.md "rocprofv3 --kernel-trace --stats -- $CMD"
python3 tools/dev/rocprof_gaps.py "$db" > gpurun_out/prof/${tag}_gaps.txt 2>&1
rm -rf gpurun_out/prof/tr_$tag
