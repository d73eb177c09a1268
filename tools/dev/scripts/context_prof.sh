#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
rm -rf gpurun_out/prof/ctx
timeout 900 rocprofv3 --kernel-trace -d gpurun_out/prof/ctx -o tr -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-wgrad-stream $* > gpurun_out/prof/ctx.log 2>&1
db=$(find gpurun_out/prof/ctx -name "*.db" | head -1)
python3 tools/dev/rocprof_context.py "$db" > gpurun_out/prof/ctx.txt 2>&1
rm -rf gpurun_out/prof/ctx
head -70 gpurun_out/prof/ctx.txt
