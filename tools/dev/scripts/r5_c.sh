#!/bin/bash
# Dev (round 5): the sharper full-width parity tests (numbers printed), where the step's small ATen launches come from, launch census.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5
mkdir -p $O
timeout 1700 python -m pytest tests/test_full_width_gpu.py -q -x -s 2>&1 | grep -v "^$" | tail -40 > $O/tests_fullwidth.txt
tail -40 $O/tests_fullwidth.txt
timeout 600 python3 tools/dev/glue_sources.py > $O/glue.txt 2>&1
head -90 $O/glue.txt
rm -rf $O/tr
timeout 900 rocprofv3 --kernel-trace --stats -d $O/tr -o tr -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-also > $O/census_prof.log 2>&1
db=$(find $O/tr -name "*.db" | head -1)
python3 tools/dev/rocprof_step_census.py "$db" > $O/census.txt 2>&1
rm -rf $O/tr
head -60 $O/census.txt
