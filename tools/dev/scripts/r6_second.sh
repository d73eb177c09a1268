#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6b
timeout 600 python -m pytest tests/test_hip_kernels.py -q -s -k "decode" > gpurun_out/r6b/decode.log 2>&1; echo "decode rc=$?" > gpurun_out/r6b/rc.txt
timeout 900 python -m pytest tests/test_full_width_gpu.py -q -s -k "statistics or benchmarked_b16" > gpurun_out/r6b/bn_stats.log 2>&1; echo "bnstats rc=$?" >> gpurun_out/r6b/rc.txt
timeout 300 python tools/dev/arena_revert_demo.py > gpurun_out/r6b/arena_demo.txt 2>&1; echo "arena rc=$?" >> gpurun_out/r6b/rc.txt
cat gpurun_out/r6b/rc.txt; grep -E "passed|failed" gpurun_out/r6b/*.log gpurun_out/r6b/arena_demo.txt
