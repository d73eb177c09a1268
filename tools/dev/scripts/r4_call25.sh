#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 300 python3 -m pytest tests/test_hip_backward_elem.py -q -m gpu -x 2>&1 | tail -2
timeout 300 python3 tools/dev/upT_bench.py 2>&1 | grep -v amdgpu
