#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6f
timeout 900 python tools/dev/tune_step.py -n 10 -r 5 wgrad.pp_mink=64 wgrad.pp_mink=128 conv.glds3_pp_mink=512 conv.balance_rows=2 wgrad.pp_mink=128,conv.glds3_pp_mink=512 > gpurun_out/r6f/tune.txt 2>&1; echo "tune rc=$?" > gpurun_out/r6f/rc.txt
timeout 600 python -m pytest tests/test_full_width_gpu.py -q -s -k "inference_graph" > gpurun_out/r6f/infer_graph_test.log 2>&1; echo "igtest rc=$?" >> gpurun_out/r6f/rc.txt
cat gpurun_out/r6f/rc.txt; tail -8 gpurun_out/r6f/tune.txt; tail -3 gpurun_out/r6f/infer_graph_test.log
