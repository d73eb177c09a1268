#!/bin/bash
# Dev (VERDICT r4 #1d): the driver's command on this box, plain and under rocprofv3 --kernel-trace, and the largest families'
# event-pair time next to their kernel time -> one block of profiles/r05_pricing_repro.md per box. Optional $1 = "hog":
# a third run with busy host processes beside the trainer (does a starved launch thread move the priced families?).
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5/repro_$(hostname)_$$
mkdir -p $O
timeout 900 python3 bench.py --steps 20 --warmup 5 2>$O/plain.err | tail -1 > $O/plain_line.json
rm -rf $O/tr
timeout 900 rocprofv3 --kernel-trace --stats -d $O/tr -o tr -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-wgrad-stream 2>$O/prof.err | tail -1 > $O/prof_line.json
db=$(find $O/tr -name "*.db" | head -1)
python3 tools/dev/pricing_repro.py $O/prof_line.json "$db" "(under rocprofv3)" > $O/repro.md
python3 tools/dev/pricing_repro.py $O/plain_line.json "$db" "(plain run; kernel times from the rocprofv3 run)" >> $O/repro.md
if [ "$1" = "hog" ]; then
  pids=""
  trap '[ -n "$pids" ] && kill $pids 2>/dev/null' EXIT   # (an outer timeout must not leave the 32 busy loops running on a shared box)
  for i in $(seq 1 32); do python3 -c "while True: pass" & pids="$pids $!"; done
  timeout 900 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also 2>$O/hog.err | tail -1 > $O/hog_line.json
  kill $pids
  python3 tools/dev/pricing_repro.py $O/hog_line.json "$db" "(32 busy host processes beside the trainer; kernel times from the rocprofv3 run)" >> $O/repro.md
fi
python3 tools/dev/rocprof_summary.py "$db" $O/kernel_stats.md "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-wgrad-stream" > /dev/null
rm -rf $O/tr
cat $O/repro.md
cut -c1-400 $O/plain_line.json
