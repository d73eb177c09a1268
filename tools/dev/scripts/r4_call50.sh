#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 900 python3 tools/dev/tune_step.py -r 5 WGRAD_BATCH=16 WGRAD_BATCH=8 WGRAD_BATCH=4 2>&1 | grep -v amdgpu
