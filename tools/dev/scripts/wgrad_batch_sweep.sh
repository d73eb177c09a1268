#!/bin/bash
# A/B: weight gradients per deferred launch (bench.py --wgrad-batch), train workload
for b in "$@"; do
  python bench.py --no-cpu-baseline --wgrad-batch $b 2>&1 | tail -1 > /tmp/line.json
  python - "$b" <<'PY'
import sys, json
d = json.loads(open('/tmp/line.json').read())
r = d['roofline']
print('wgrad-batch', sys.argv[1], 'img/s', d['value'], 'ms', d['ms_per_step'], 'wgrad ms', r['family_ms_per_step'], 'TF', r['achieved'], 'mem', d.get('peak_mem_gb'))
PY
done
