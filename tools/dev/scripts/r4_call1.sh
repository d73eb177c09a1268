#!/bin/bash
# round 4, first data-gathering call: baseline line, per-shape / per-mode tables, glue attribution, census
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r4a
mkdir -p $O
timeout 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
timeout 300 python3 tools/dev/train_shapes.py > $O/train_shapes.txt 2>&1
timeout 300 python3 tools/dev/bn_shapes.py > $O/bn_shapes.txt 2>&1
timeout 300 python3 tools/dev/glue_sources.py > $O/glue_sources.txt 2>&1
timeout 200 python3 tools/dev/stream_bench.py > $O/stream_bench.txt 2>&1
rm -rf /tmp/cen
timeout 600 rocprofv3 --kernel-trace -d /tmp/cen -o tr -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > /tmp/cen.log 2>&1
python3 tools/dev/rocprof_step_census.py $(find /tmp/cen -name "*.db" | head -1) > $O/census.txt 2>&1
python3 tools/dev/rocprof_seq.py $(find /tmp/cen -name "*.db" | head -1) copyBuffer 2400 > $O/seq_copy.txt 2>&1
python3 tools/dev/rocprof_seq.py $(find /tmp/cen -name "*.db" | head -1) fillBuffer 2400 > $O/seq_fill.txt 2>&1
ls -la $O
