#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5
mkdir -p $O
timeout 900 python3 bench.py --steps 20 --warmup 5 2>$O/bench_chk.err | tail -1 > $O/bench_chk.json
python3 - <<'PY'
import json
l = json.load(open('gpurun_out/r5/bench_chk.json'))
print(l['value'], l['ms_per_step'], l['timed_steps_ms'], l['priced_step']['families_ms_sum'], l['priced_step']['unreliable'], l['cpu_baseline']['value'], l['cpu_baseline']['protocol'])
PY
timeout 900 python3 bench.py --gpus 2 --share-gpu --norm SyncBN --no-cpu-baseline --no-also --steps 6 --warmup 2 2>$O/share2.err | tail -1 | cut -c1-300
tail -3 $O/share2.err
