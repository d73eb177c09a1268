#!/bin/bash
# odd-channel layers on the direct paths + zeroed GroupNorm workspaces: fast suite, bench line, launch census
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x --ignore=tests/test_full_width_gpu.py 2>&1 | tail -15 > $O/suite_odd.txt
tail -15 $O/suite_odd.txt
timeout 900 python3 bench.py --no-also --no-cpu-baseline 2>$O/bench_odd.err | tail -1 > $O/bench_line_odd.json
cut -c1-200 $O/bench_line_odd.json; echo
rm -rf $O/tr
timeout 900 rocprofv3 --kernel-trace --stats -d $O/tr -o tr -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-also > $O/odd_prof.log 2>&1
db=$(find $O/tr -name "*.db" | head -1)
python3 tools/dev/rocprof_step_census.py "$db" > $O/census_odd.txt 2>&1
rm -rf $O/tr
head -30 $O/census_odd.txt
