#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | cut -c1-200
timeout 600 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-also 2>/dev/null | tail -1 | cut -c1-160
