#!/bin/bash
# rocprofv3 kernel sums of the train bench under two environment settings (one box)
# usage: env_ab_prof.sh "<VAR=v for A>" "<VAR=v for B>" <grep pattern>
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for tag in A B; do
  if [ $tag = A ]; then E="$1"; else E="$2"; fi
  rm -rf /tmp/eabp_$tag
  env $E timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/eabp_$tag -o tr -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-wgrad-stream > /tmp/eabp_$tag.log 2>&1
  db=$(find /tmp/eabp_$tag -name "*.db" | head -1)
  python3 tools/dev/rocprof_summary.py "$db" /tmp/eabp_$tag.md "$E" > /dev/null
  echo "== $tag ($E)"
  grep -E "total kernel|$3" /tmp/eabp_$tag.md | cut -c1-150
done
