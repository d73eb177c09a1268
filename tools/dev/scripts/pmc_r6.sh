#!/bin/bash
# Round 6 PMC passes over the REAL train step (each pass its own run, --kernel-trace only, as the pool requires)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/pmc6
mkdir -p $O; rm -rf $O/*
cp profiles/traffic.json $O/traffic.json
MF="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE"
run() {  # tag counters
  timeout 900 rocprofv3 --pmc $2 --kernel-trace -d $O/$1 -o $1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also > $O/$1.log 2>&1
  echo "$1 rc=$?"
}
run f FETCH_SIZE
run w WRITE_SIZE
run m "$MF"
cd tools/dev
DAS_ROUND=r06 python3 pmc_step_tables.py $(find ../../$O/f -name "*.db" | head -1) $(find ../../$O/w -name "*.db" | head -1) $(find ../../$O/m -name "*.db" | head -1) ../../$O 2>&1 | tail -3
cd ../..
rm -rf $O/f $O/w $O/m
ls -la $O
head -30 $O/r06_stream_modes.md
