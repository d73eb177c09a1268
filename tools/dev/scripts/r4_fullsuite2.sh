#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 3200 python3 -m pytest tests -q -m gpu --durations=10 > gpurun_out/r4/pytest_full.txt 2>&1
tail -25 gpurun_out/r4/pytest_full.txt
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
