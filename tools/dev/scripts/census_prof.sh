#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
rm -rf /tmp/cen
timeout 900 rocprofv3 --kernel-trace -d /tmp/cen -o tr -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-wgrad-stream > /tmp/cen.log 2>&1
python3 tools/dev/rocprof_step_census.py $(find /tmp/cen -name "*.db" | head -1)
