#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_ddp_gpu.py -q -m gpu -x 2>&1 | tail -15
