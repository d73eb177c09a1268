#!/bin/bash
# final_r5.sh + pmc_r5.sh on one tree, plus: the DCNv2 data-gradient GEMM (256 -> 2304) on the tile kernels instead of the stream kernel
cd "$GRAFT_REPO_ROOT"
bash tools/dev/scripts/final_r5.sh > gpurun_out/final5.log 2>&1
bash tools/dev/scripts/pmc_r5.sh > gpurun_out/pmc5.log 2>&1
mkdir -p gpurun_out/r5
TUNE=conv.stream_minrows=200000 timeout 600 python tools/dev/train_shapes.py > gpurun_out/r5/ts_nostream_mid.txt 2>&1
grep "Cout=2304" gpurun_out/r5/ts_nostream_mid.txt | cut -c1-175
cut -c1-220 gpurun_out/final5/train_bench_line.json; echo; head -3 gpurun_out/final5/census.txt; tail -4 gpurun_out/final5/step_times.txt
