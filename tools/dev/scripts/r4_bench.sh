#!/bin/bash
# usage: r4_bench.sh <tag> [bench args]
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
T=$1; shift
mkdir -p gpurun_out/r4
timeout 600 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/r4/bench_$T.json 2> gpurun_out/r4/bench_$T.err
tail -3 gpurun_out/r4/bench_$T.err
python3 - <<P
import json
d=json.load(open('gpurun_out/r4/bench_$T.json'))
print('VALUE', d['value'], d['ms_per_step'])
print(json.dumps(d.get('priced_step'), indent=1))
print(json.dumps(d.get('also'), indent=1))
P
