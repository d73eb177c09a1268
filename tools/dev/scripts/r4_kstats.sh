#!/bin/bash
# usage: r4_kstats.sh <grep pattern>  -> median step time + per-kernel stats lines of the train bench
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 300 python3 tools/dev/tune_step.py -n 12 -r 3 2>&1 | grep -v amdgpu
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-also > /tmp/kt.log 2>&1
python3 tools/dev/rocprof_summary.py $(find /tmp/kt -name "*.db" | head -1) gpurun_out/r4/kstats_after.md x > /dev/null; grep -E "$1" gpurun_out/r4/kstats_after.md | cut -c1-150
