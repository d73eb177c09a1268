#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_ddp_gpu.py tests/test_syncbn_gpu.py tests/test_upconv_gpu.py tests/test_bn_fused_gpu.py -q -m gpu -x 2>&1 | tail -15
timeout 600 python3 bench.py --gpus 2 --share-gpu --norm SyncBN --no-cpu-baseline --no-also --steps 4 --warmup 2 2>/dev/null | tail -1 | cut -c1-200
