#!/bin/bash
# Dev: everything profiles/r06_* is built from, on ONE tree: the driver's command (plain, and under rocprofv3 in the serial stream
# layout: pricing repro block), bench lines (graphs, SyncBN, infer, decode, two ranks sharing the GPU with SyncBN — plumbing),
# the full-protocol CPU baseline, kernel summaries, launch census, host / wall step times.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/final6
mkdir -p $O; rm -rf $O/*
timeout 900 python3 bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/train_bench_line.json
timeout 900 python3 bench.py --graphs --no-cpu-baseline --no-also 2>/dev/null | tail -1 > $O/train_graphs_bench_line.json
timeout 900 python3 bench.py --norm SyncBN --no-cpu-baseline --no-also 2>/dev/null | tail -1 > $O/train_syncbn_bench_line.json
timeout 900 python3 bench.py --gpus 2 --share-gpu --norm SyncBN --no-cpu-baseline --no-also --steps 6 --warmup 2 2>$O/share2.err | tail -1 > $O/train_syncbn_2ranks_one_gpu_plumbing_line.json
timeout 600 python3 bench.py --workload infer 2>/dev/null | tail -1 > $O/infer_bench_line.json
timeout 600 python3 bench.py --workload infer --no-infer-graph --no-cpu-baseline 2>/dev/null | tail -1 > $O/infer_eager_bench_line.json
timeout 900 python3 bench.py --gpus 2 --share-gpu --grad-comm bf16 --no-cpu-baseline --no-also --steps 6 --warmup 2 2>$O/share2_bf16.err | tail -1 > $O/train_2ranks_one_gpu_bf16_buckets_plumbing_line.json
timeout 600 python3 -m pytest tests/test_cli_gpu.py -q -k two_ranks > $O/two_rank_test.log 2>&1
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
prof infer "--workload infer"
# the driver's command under the profiler, weight gradients on the main stream (the layout of the per-launch passes): the
# bench line it prints + the kernel summary of the same process -> pricing repro block of this box
rm -rf $O/tr
timeout 900 rocprofv3 --kernel-trace --stats -d $O/tr -o tr -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-wgrad-stream 2>$O/serial_prof.err | tail -1 > $O/serial_prof_line.json
db=$(find $O/tr -name "*.db" | head -1)
python3 tools/dev/rocprof_summary.py "$db" $O/train_eager_serial_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-wgrad-stream" > /dev/null
python3 tools/dev/pricing_repro.py $O/serial_prof_line.json "$db" "(under rocprofv3, weight gradients on the main stream)" > $O/pricing_repro_block.md
python3 tools/dev/pricing_repro.py $O/train_bench_line.json "$db" "(the plain driver command; kernel times from the rocprofv3 run)" >> $O/pricing_repro_block.md
rm -rf $O/tr
timeout 300 python3 tools/dev/step_times.py 16 > $O/step_times.txt 2>&1
for f in train train_graphs train_syncbn train_syncbn_2ranks_one_gpu_plumbing train_2ranks_one_gpu_bf16_buckets_plumbing infer infer_eager decode; do cut -c1-260 $O/${f}_bench_line.json; echo; done
tail -2 $O/two_rank_test.log; cut -c1-300 $O/cpu_baseline_full.json; tail -3 $O/share2.err; cat $O/pricing_repro_block.md
