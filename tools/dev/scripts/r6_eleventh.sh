#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6l
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r6l/gputests.log 2>&1; echo "gputests rc=$?" > gpurun_out/r6l/rc.txt
timeout 900 python tools/dev/tune_step.py -n 10 -r 5 conv.kstream=0 conv.kstream=1 conv.kstream=7 > gpurun_out/r6l/tune.txt 2>&1; echo "tune rc=$?" >> gpurun_out/r6l/rc.txt
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r6l/bench.json 2> gpurun_out/r6l/bench.err; echo "bench rc=$?" >> gpurun_out/r6l/rc.txt
cat gpurun_out/r6l/rc.txt; tail -4 gpurun_out/r6l/gputests.log | cut -c1-250; tail -5 gpurun_out/r6l/tune.txt; head -c 300 gpurun_out/r6l/bench.json
