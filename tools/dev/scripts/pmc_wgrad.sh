#!/bin/bash
# Dev: HBM traffic counters (separate --pmc passes, no trace domains besides kernel-trace) for the weight-gradient
# launches of one train step (tools/dev/wgrad_mix.py) and for the forward conv kernels of the infer workload.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
rm -rf gpurun_out/pmc/*
timeout 200 python3 tools/dev/wgrad_mix.py > gpurun_out/pmc/wgrad_mix_bare.log 2>&1
timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc/wf -o wf -- python3 tools/dev/wgrad_mix.py > gpurun_out/pmc/wf.log 2>&1
echo "fetch pass rc=$?"
timeout 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc/ww -o ww -- python3 tools/dev/wgrad_mix.py > gpurun_out/pmc/ww.log 2>&1
echo "write pass rc=$?"
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc/inf -o inf -- python3 bench.py --workload infer --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc/inf.log 2>&1
echo "infer fetch pass rc=$?"
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc/infw -o infw -- python3 bench.py --workload infer --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc/infw.log 2>&1
echo "infer write pass rc=$?"
for t in wf ww inf infw; do
  db=$(find gpurun_out/pmc/$t -name "*.db" | head -1)
  [ -n "$db" ] && python3 tools/dev/pmc_summary.py "$db" gpurun_out/pmc/$t.md "$t" gpurun_out/pmc/$t.json | tail -1
  # the raw databases are large: keep only the summaries
  rm -rf gpurun_out/pmc/$t
done
grep -v amdgpu.ids gpurun_out/pmc/wgrad_mix_bare.log | tail -2
grep -i wgrad gpurun_out/pmc/wf.md gpurun_out/pmc/ww.md | head
