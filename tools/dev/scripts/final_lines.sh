#!/bin/bash
# Dev: the three bench lines (default flags) + the rocprofv3 kernel summaries of the train and infer commands.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/final; rm -f gpurun_out/final/*
timeout 900 python bench.py 2>/dev/null | tail -1 > gpurun_out/final/train_bench_line.json
timeout 600 python bench.py --workload infer 2>/dev/null | tail -1 > gpurun_out/final/infer_bench_line.json
timeout 600 python bench.py --workload decode 2>/dev/null | tail -1 > gpurun_out/final/decode_bench_line.json
bash tools/dev/scripts/train_prof2.sh final > gpurun_out/final/train_prof.out 2>&1
bash tools/dev/scripts/infer_prof.sh > gpurun_out/final/infer_prof.out 2>&1
cp gpurun_out/prof/final_stats.md gpurun_out/final/train_stats.md
cp gpurun_out/prof/final_gaps.txt gpurun_out/final/train_gaps.txt
cp gpurun_out/prof/infer_stats.md gpurun_out/prof/infer_gaps.txt gpurun_out/final/
for f in train infer decode; do cut -c1-330 gpurun_out/final/${f}_bench_line.json; echo; done
