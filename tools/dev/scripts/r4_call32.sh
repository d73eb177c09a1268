#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 600 python3 -m pytest tests/test_bn_fused_gpu.py -q -m gpu -x 2>&1 | tail -3
timeout 600 python3 tools/dev/train_shapes.py > gpurun_out/r4/train_shapes2.txt 2>&1
grep -E "mode=r?b[my]" gpurun_out/r4/train_shapes2.txt | head -30
timeout 600 python3 tools/dev/bn_shapes.py > gpurun_out/r4/bn_shapes3.txt 2>&1
grep -E "bits|upmerge" gpurun_out/r4/bn_shapes3.txt | head
