#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x --ignore=tests/test_full_width_gpu.py 2>&1 | tail -15 > $O/suite_odd.txt
tail -15 $O/suite_odd.txt
R=5 N=8 timeout 900 python tools/dev/ab_launches.py 2>&1 | tail -3 | tee $O/ab_launches.txt
