#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 600 python3 tools/dev/dcn_bwd_probe.py 2>&1 | grep -v amdgpu
