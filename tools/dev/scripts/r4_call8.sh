#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r4h
mkdir -p $O
timeout 1500 python3 -m pytest tests/test_hip_backward_head.py tests/test_train_gpu.py tests/test_flat_paths_gpu.py tests/test_train_step_gpu.py tests/test_loss_gpu.py tests/test_topologies_gpu.py tests/test_model_gpu.py -x -q -m gpu > $O/pytest_a.txt 2>&1
tail -8 $O/pytest_a.txt
timeout 300 python3 tools/dev/glue_sources.py > $O/glue_sources.txt 2>&1
head -40 $O/glue_sources.txt
bash tools/dev/scripts/r4_bench.sh b
