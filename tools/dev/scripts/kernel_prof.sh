#!/bin/bash
# Dev: kernel-level times (rocprofv3 --kernel-trace --stats) of one dev tool run. usage: kernel_prof.sh <grep pattern> <script> [args]
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/prof; rm -rf gpurun_out/prof/kp
pat=$1; shift
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/kp -o kp -- python3 "$@" > gpurun_out/prof/kp.log 2>&1
db=$(find gpurun_out/prof/kp -name "*.db" | head -1)
python3 tools/dev/rocprof_summary.py "$db" gpurun_out/prof/kp_stats.md "$*" > /dev/null
rm -rf gpurun_out/prof/kp
grep -E "$pat" gpurun_out/prof/kp_stats.md | cut -c1-160
