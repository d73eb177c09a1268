#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 1500 python3 tools/dev/tune_step.py -n 8 -r 4 conv.stream_percu=3 conv.stream_minrows=4096 conv.glds4_minblocks=208 bn.stream_minbytes=50000000 bn.stream_minbytes=250000000 \
  bn.reduce_blocks=512 bn.vpt=16 SLOTS=16/4 conv.pt3_mintiles=257 conv.glds3_pp_mink=512 \
  conv.stream_percu=3,bn.reduce_blocks=512,bn.stream_minbytes=250000000,conv.pt3_mintiles=257,SLOTS=16/4 > gpurun_out/r4/tune_step2.txt 2>&1
cat gpurun_out/r4/tune_step2.txt
timeout 300 python3 tools/dev/cu_pressure.py own 16 > gpurun_out/r4/cu_pressure_own16.txt 2>&1
cat gpurun_out/r4/cu_pressure_own16.txt
