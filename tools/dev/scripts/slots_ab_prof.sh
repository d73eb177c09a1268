#!/bin/bash
# rocprofv3 kernel sums of the train bench with das_amd.nn.<$SLOTVAR, default _MID_SLOTS> = $1 and = $2 (one box); $3 = grep pattern
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
cat > /tmp/run_slots.py <<'PY'
import sys
v = int(sys.argv[1])
name = sys.argv[2] if len(sys.argv) > 2 else '_MID_SLOTS'
sys.argv = ['bench.py', '--steps', '5', '--warmup', '2', '--no-cpu-baseline', '--no-wgrad-stream']
sys.path.insert(0, '.')
import das_amd.nn as n
setattr(n, name, v)
import bench
bench.main()
PY
for v in $1 $2; do
  rm -rf /tmp/slp_$v
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/slp_$v -o tr -- python3 /tmp/run_slots.py $v $SLOTVAR > /tmp/slp_$v.log 2>&1
  db=$(find /tmp/slp_$v -name "*.db" | head -1)
  python3 tools/dev/rocprof_summary.py "$db" /tmp/slp_$v.md "slots $v" > /dev/null
  echo "== mid slots $v"
  grep -E "total kernel|$3" /tmp/slp_$v.md | cut -c1-150
done
