#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_syncbn_gpu.py tests/test_ddp_gpu.py tests/test_bn_fused_gpu.py tests/test_train_gpu.py tests/test_upconv_gpu.py -q -m gpu -x 2>&1 | tail -8
