#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 1500 python3 tools/dev/tune_step.py -r 3 bn.upmerge_blocks=256 bn.upmerge_blocks=1024 bn.vpt=4 bn.vpt=16 bn.stream_minbytes=50000000 bn.stream_minbytes=250000000 SLOTS=16/4 SLOTS=4/2 bn.reduce_blocks=512 elem.upstats_ppb=64 elem.upstats_ppb=256 WGRAD_BATCH=16 WGRAD_BATCH=64 conv.stream_percu=3 > gpurun_out/r4/tune_end.txt 2>&1
cat gpurun_out/r4/tune_end.txt
