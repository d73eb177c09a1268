#!/bin/bash
# Dev (round 5): is the default line's step time off after the suite, or because of the new paths? bench twice + interleaved A/B.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5
mkdir -p $O
timeout 600 python -m pytest tests/test_train_gpu.py -q -x -k chaining 2>&1 | grep -E "assert|Error|passed|failed" | head -8
for i in 1 2; do
  timeout 600 python3 bench.py --no-also --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; l=json.loads(sys.stdin.read()); print('bench', l['ms_per_step'], 'pass', l['priced_step']['step_ms_this_pass'], 'fam', l['priced_step']['families_ms_sum'])"
done
timeout 900 python3 tools/dev/tune_step.py -r 3 CHAIN=0 wgrad.shapes=0 > $O/tune_e.txt 2>&1
tail -5 $O/tune_e.txt
