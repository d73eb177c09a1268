#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6e
timeout 900 python -m pytest tests/test_conv_tiles_gpu.py -q -x -k "mf32" > gpurun_out/r6e/mf32_tests.log 2>&1; echo "mf32 tests rc=$?" > gpurun_out/r6e/rc.txt
timeout 600 python tools/dev/mfma32_ab.py > gpurun_out/r6e/mfma32_ab.md 2>&1; echo "ab rc=$?" >> gpurun_out/r6e/rc.txt
timeout 900 python tools/dev/tune_step.py -n 10 -r 5 conv.glds4_mfma32=1 > gpurun_out/r6e/tune_mf32.txt 2>&1; echo "tune rc=$?" >> gpurun_out/r6e/rc.txt
timeout 600 python -m pytest tests/test_full_width_gpu.py -q -s -k "inference_graph" > gpurun_out/r6e/infer_graph_test.log 2>&1; echo "igtest rc=$?" >> gpurun_out/r6e/rc.txt
cat gpurun_out/r6e/rc.txt; tail -3 gpurun_out/r6e/mf32_tests.log; cat gpurun_out/r6e/mfma32_ab.md; tail -3 gpurun_out/r6e/tune_mf32.txt; tail -3 gpurun_out/r6e/infer_graph_test.log
