#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_full_width_gpu.py -q -m gpu 2>&1 | tail -5
