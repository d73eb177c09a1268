#!/bin/bash
# Dev (round 5): head consumer chaining — tests, A/B, launch census.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5
mkdir -p $O
timeout 1700 python -m pytest tests/test_train_gpu.py tests/test_model_gpu.py tests/test_topologies_gpu.py tests/test_loss_gpu.py tests/test_hip_backward_head.py -q -x 2>&1 | tail -12 > $O/tests_d.txt
tail -12 $O/tests_d.txt
timeout 600 python3 tools/dev/tune_step.py -r 4 CHAIN=0 CHAIN=1 > $O/tune_chain.txt 2>&1
tail -6 $O/tune_chain.txt
