#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
rm -rf /tmp/ctxi
timeout 900 rocprofv3 --kernel-trace -d /tmp/ctxi -o tr -- python3 bench.py --workload infer --steps 10 --warmup 2 --no-cpu-baseline > /tmp/ctxi.log 2>&1
db=$(find /tmp/ctxi -name "*.db" | head -1)
python3 tools/dev/rocprof_context.py "$db" 2>&1 | head -40
