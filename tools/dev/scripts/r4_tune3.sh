#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 2000 python3 tools/dev/tune_step.py -n 12 -r 5 conv.stream_percu=3 conv.stream_percu=4 conv.stream_minrows=4096 conv.glds4_minblocks=208 conv.glds4_minblocks=256 \
  bn.stream_minbytes=50000000 bn.stream_minbytes=250000000 bn.reduce_blocks=512 conv.pt3_mintiles=257 > gpurun_out/r4/tune_step3.txt 2>&1
cat gpurun_out/r4/tune_step3.txt
