#!/bin/bash
# A/B of bench.py flag sets on ONE box: each argument is one flag string; two rounds, img/s per run
cd "$GRAFT_REPO_ROOT"
for r in 1 2; do
  for f in "$@"; do
    v=$(python bench.py --no-cpu-baseline $f 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "[$f] $v"
  done
done
