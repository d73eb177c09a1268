#!/bin/bash
# round 6, first GPU call: full GPU suite, the BN statistics tests with printed maxima, bench baseline, vendor yardstick, arena demo
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6a
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r6a/gputests.log 2>&1; echo "gputests rc=$?" > gpurun_out/r6a/rc.txt
timeout 600 python -m pytest tests/test_full_width_gpu.py -q -s -k "statistics or benchmarked_b16" > gpurun_out/r6a/bn_stats.log 2>&1; echo "bnstats rc=$?" >> gpurun_out/r6a/rc.txt
timeout 300 python tools/dev/arena_revert_demo.py > gpurun_out/r6a/arena_demo.txt 2>&1; echo "arena rc=$?" >> gpurun_out/r6a/rc.txt
timeout 600 python tools/dev/gemm_yardstick.py > gpurun_out/r6a/yardstick.md 2> gpurun_out/r6a/yardstick.err; echo "yard rc=$?" >> gpurun_out/r6a/rc.txt
timeout 900 python bench.py > gpurun_out/r6a/bench.json 2> gpurun_out/r6a/bench.err; echo "bench rc=$?" >> gpurun_out/r6a/rc.txt
cat gpurun_out/r6a/rc.txt; tail -3 gpurun_out/r6a/gputests.log
