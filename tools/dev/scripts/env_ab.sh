#!/bin/bash
# A/B of environment settings on ONE box: each argument is "VAR=value ..." (or "" for defaults); two rounds
cd "$GRAFT_REPO_ROOT"
for r in 1 2; do
  for e in "$@"; do
    v=$(env $e python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "[$e] $v"
  done
done
