#!/bin/bash
# Dev (round 5): conv-tile / BatchNorm-fusion / weight-gradient / prof tests after the removal of conv_pt3 and the new
# weight-gradient wave arrangements; interleaved A/B of wgrad.shapes.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5
mkdir -p $O
timeout 1500 python -m pytest tests/test_prof_gpu.py tests/test_conv_tiles_gpu.py tests/test_bn_fused_gpu.py -q -x 2>&1 | tail -15 > $O/tests_c.txt
tail -15 $O/tests_c.txt
timeout 600 python3 tools/dev/tune_step.py -r 4 wgrad.shapes=0 wgrad.shapes=1 > $O/tune_wgrad_shapes.txt 2>&1
tail -12 $O/tune_wgrad_shapes.txt
