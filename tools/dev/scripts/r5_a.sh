#!/bin/bash
# Dev (round 5): tests of the library's launch timing and the weight-gradient wave arrangements, the train line with and without
# the arrangements, and the round-4 tree under host starvation (what moved round 4's Python-recorded event pairs).
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5
mkdir -p $O
timeout 900 python -m pytest tests/test_prof_gpu.py tests/test_conv_tiles_gpu.py -q -x -k "prof or wgrad" 2>&1 | tail -40 > $O/tests_b.txt
tail -30 $O/tests_b.txt
timeout 600 python3 bench.py --no-cpu-baseline --no-also 2>$O/b_on.err | tail -1 > $O/b_shapes_on.json
timeout 600 python3 bench.py --no-cpu-baseline --no-also --tune wgrad.shapes=0 2>$O/b_off.err | tail -1 > $O/b_shapes_off.json
python3 - <<'PY'
import json
for f in ('on', 'off'):
    try:
        l = json.load(open(f'gpurun_out/r5/b_shapes_{f}.json'))
        fm = l['priced_step']['families_ms']
        print(f, l['ms_per_step'], 'wgrad', fm.get('conv_wgrad_kernel<bf16>'), 'pp', fm.get('conv_wgrad_pp_kernel'), 'pass', l['priced_step']['step_ms_this_pass'])
    except Exception as e:
        print(f, 'failed', e)
PY
if [ -d r4tree ]; then
  cd r4tree
  pids=""
  trap '[ -n "$pids" ] && kill $pids 2>/dev/null' EXIT   # (an outer timeout must not leave the 32 busy loops running on a shared box)
  for i in $(seq 1 32); do python3 -c "while True: pass" & pids="$pids $!"; done
  timeout 900 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also 2>../$O/r4_hog.err | tail -1 > ../$O/r4_hog_line.json
  kill $pids
  timeout 900 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also 2>../$O/r4_plain.err | tail -1 > ../$O/r4_plain_line.json
  cd ..
  python3 - <<'PY'
import json
for f in ('hog', 'plain'):
    try:
        l = json.load(open(f'gpurun_out/r5/r4_{f}_line.json'))
        pr = l['priced_step']
        print('r4 tree', f, l['ms_per_step'], 'pass', pr['step_ms_this_pass'], 'families', pr['families_ms_sum'], 'headline', l['roofline']['kernel'], l['roofline']['frac'],
              {k: v for k, v in list(pr['families_ms'].items())[:6]})
    except Exception as e:
        print(f, 'failed', e)
PY
fi
