#!/bin/bash
# Dev: everything profiles/r03_* is built from, on ONE tree: the bench lines (default flags; train also with the trunk
# replayed as hipGraphs), the full-protocol CPU baseline, rocprofv3 kernel summaries of the train command in both stream layouts
# (as benchmarked: launch by launch + side stream; and with the weight gradients on the main stream, the layout
# bench.py's per-family measurement passes use), the infer command, the tile kernels' phase stamps, the loader bench.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/final3
mkdir -p $O; rm -rf $O/*
timeout 900 python3 bench.py 2>/dev/null | tail -1 > $O/train_bench_line.json
timeout 900 python3 bench.py --graphs --no-cpu-baseline 2>/dev/null | tail -1 > $O/train_graphs_bench_line.json
timeout 600 python3 bench.py --workload infer 2>/dev/null | tail -1 > $O/infer_bench_line.json
timeout 600 python3 bench.py --workload decode 2>/dev/null | tail -1 > $O/decode_bench_line.json
timeout 900 python3 bench.py --cpu-baseline-only --cpu-baseline-full 2>/dev/null | tail -1 > $O/cpu_baseline_full.json
prof() {  # name, bench args
  rm -rf $O/tr
  timeout 900 rocprofv3 --kernel-trace --stats -d $O/tr -o tr -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline $2 > $O/$1_prof.log 2>&1
  db=$(find $O/tr -name "*.db" | head -1)
  python3 tools/dev/rocprof_summary.py "$db" $O/$1_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline $2" > /dev/null
  python3 tools/dev/rocprof_gaps.py "$db" > $O/$1_idle_gaps.txt 2>&1
  rm -rf $O/tr
}
prof train ""
prof train_eager_serial "--no-wgrad-stream"
prof infer "--workload infer"
timeout 300 python3 tools/dev/conv_stamps.py > $O/conv_phase_stamps.txt 2>&1
timeout 900 python3 tools/dev/loader_bench.py > $O/loader_bench.txt 2>&1
timeout 300 python3 tools/dev/step_times.py 30 graphs > $O/step_times.txt 2>&1
for f in train train_graphs infer decode; do cut -c1-300 $O/${f}_bench_line.json; echo; done
cut -c1-400 $O/cpu_baseline_full.json
