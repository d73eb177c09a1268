#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 1500 python3 tools/dev/tune_step.py conv.stream_percu=1 conv.stream_percu=3 conv.stream_minrows=4096 conv.stream_minrows=65536 \
  conv.splitk_target=128 conv.splitk_target=512 conv.splitk_minsteps=8 conv.splitk_minsteps=18 conv.big_minblocks=64 conv.big_minblocks=160 \
  conv.glds4_minblocks=96 conv.glds4_minblocks=208 conv.glds3_pp_mink=512 conv.glds3_pp_mink=2304 conv.tail_split=0 conv.glds4_mf=8 \
  bn.vpt=4 bn.vpt=16 bn.stream_minbytes=50000000 bn.stream_minbytes=250000000 bn.reduce_blocks=512 wgrad.bkm=64 wgrad.pp_mink=512 \
  conv.c64_mintiles=0 conv.pt3_mintiles=257 WGRAD_BATCH=16 WGRAD_BATCH=64 SLOTS=16/4 SLOTS=8/8 SLOTS=4/2 > gpurun_out/r4/tune_step.txt 2>&1
cat gpurun_out/r4/tune_step.txt
