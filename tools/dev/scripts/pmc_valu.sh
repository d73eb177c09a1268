#!/bin/bash
# Dev: vector-ALU instruction counts of every kernel of the train bench step (one PMC pass, kernel trace only).
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/pmc; rm -rf gpurun_out/pmc/valu
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace -d gpurun_out/pmc/valu -o valu -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline $* > gpurun_out/pmc/valu.log 2>&1; echo "rc=$?"
db=$(find gpurun_out/pmc/valu -name "*.db" | head -1)
python3 tools/dev/pmc_valu_summary.py "$db" gpurun_out/pmc/valu.md | cut -c1-190
rm -rf gpurun_out/pmc/valu
