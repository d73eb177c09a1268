#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python3 tools/dev/dcn_bwd_probe.py > /tmp/kt.log 2>&1
python3 tools/dev/rocprof_summary.py $(find /tmp/kt -name "*.db" | head -1) /tmp/kt.md x > /dev/null; grep -E "deform|calls" /tmp/kt.md | cut -c1-150
grep -v amdgpu /tmp/kt.log | tail -6
