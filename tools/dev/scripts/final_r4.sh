#!/bin/bash
# Dev: everything profiles/r04_* is built from, on ONE tree: bench lines (default flags; --graphs; --norm SyncBN; infer; decode;
# two ranks sharing the GPU with SyncBN — plumbing, not a measurement), the full-protocol CPU baseline, rocprofv3 kernel
# summaries of the train command in both stream layouts and of the infer command, launch census, host / wall step times,
# the persistent-tile kernel's phase stamps and A/B table, the one-tile kernels' stamps.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/final4
mkdir -p $O; rm -rf $O/*
timeout 900 python3 bench.py 2>/dev/null | tail -1 > $O/train_bench_line.json
timeout 900 python3 bench.py --graphs --no-cpu-baseline --no-also 2>/dev/null | tail -1 > $O/train_graphs_bench_line.json
timeout 900 python3 bench.py --norm SyncBN --no-cpu-baseline --no-also 2>/dev/null | tail -1 > $O/train_syncbn_bench_line.json
timeout 900 python3 bench.py --gpus 2 --share-gpu --norm SyncBN --no-cpu-baseline --no-also --steps 6 --warmup 2 2>$O/share2.err | tail -1 > $O/train_syncbn_2ranks_one_gpu_line.json
timeout 600 python3 bench.py --workload infer 2>/dev/null | tail -1 > $O/infer_bench_line.json
timeout 600 python3 bench.py --workload decode 2>/dev/null | tail -1 > $O/decode_bench_line.json
timeout 900 python3 bench.py --cpu-baseline-only --cpu-baseline-full 2>/dev/null | tail -1 > $O/cpu_baseline_full.json
prof() {  # name, bench args
  rm -rf $O/tr
  timeout 900 rocprofv3 --kernel-trace --stats -d $O/tr -o tr -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-also $2 > $O/$1_prof.log 2>&1
  db=$(find $O/tr -name "*.db" | head -1)
  python3 tools/dev/rocprof_summary.py "$db" $O/$1_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-also $2" > /dev/null
  python3 tools/dev/rocprof_gaps.py "$db" > $O/$1_idle_gaps.txt 2>&1
  [ "$1" = "train" ] && python3 tools/dev/rocprof_step_census.py "$db" > $O/census.txt 2>&1
  rm -rf $O/tr
}
prof train ""
prof train_eager_serial "--no-wgrad-stream"
prof infer "--workload infer"
timeout 300 python3 tools/dev/step_times.py 16 > $O/step_times.txt 2>&1
timeout 300 python3 tools/dev/conv_stamps.py > $O/conv_phase_stamps.txt 2>&1
timeout 300 python3 tools/dev/pt3_stamps.py > $O/pt3_stamps.txt 2>&1
timeout 300 python3 tools/dev/pt3_bench.py > $O/pt3_bench.txt 2>&1
timeout 300 python3 tools/dev/pt3_single.py > $O/pt3_single.txt 2>&1
for f in train train_graphs train_syncbn train_syncbn_2ranks_one_gpu infer decode; do cut -c1-260 $O/${f}_bench_line.json; echo; done
cut -c1-300 $O/cpu_baseline_full.json; tail -3 $O/share2.err
