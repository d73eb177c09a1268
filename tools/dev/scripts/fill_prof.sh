#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
rm -rf /tmp/fillp
timeout 900 rocprofv3 --kernel-trace -d /tmp/fillp -o tr -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-wgrad-stream $* > /tmp/fillp.log 2>&1
db=$(find /tmp/fillp -name "*.db" | head -1)
python3 tools/dev/rocprof_fill.py "$db" 9
