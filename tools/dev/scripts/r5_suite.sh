#!/bin/bash
# Dev: the whole -m gpu suite (optionally without the full-width file: $1 = "fast") + the default bench line.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5
mkdir -p $O
if [ "$1" = "fast" ]; then IGN="--ignore=tests/test_full_width_gpu.py"; else IGN=""; fi
timeout 2400 python -m pytest tests -m gpu -q -x $IGN 2>&1 | tail -15 > $O/suite.txt
tail -15 $O/suite.txt
timeout 900 python3 bench.py --no-also 2>$O/bench.err | tail -1 > $O/bench_line.json
python3 - <<'PY'
import json
l = json.load(open('gpurun_out/r5/bench_line.json'))
pr = l['priced_step']
print(l['value'], 'img/s', l['ms_per_step'], 'ms; pass', pr['step_ms_this_pass'], 'families', pr['families_ms_sum'], 'unreliable', pr['unreliable'], 'headline', l['roofline']['kernel'], l['roofline']['frac'])
print({k: v for k, v in pr['families_ms'].items()})
print(l['cpu_baseline']['value'], l['cpu_baseline']['protocol'])
PY
