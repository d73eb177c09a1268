#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 900 python3 -m pytest tests/test_upconv_gpu.py tests/test_train_step_gpu.py tests/test_graphs_gpu.py -q -m gpu -x 2>&1 | tail -3
timeout 600 python3 tools/dev/glue_sources.py > gpurun_out/r4/glue.txt 2>&1
grep -n "events without a stack" -A60 gpurun_out/r4/glue.txt | head -90
