#!/bin/bash
# Dev: PMC passes of round 2 (each pass = its own run with --kernel-trace only, as the pool requires):
#   HBM traffic (FETCH_SIZE / WRITE_SIZE) of the weight gradients of one train step (tools/dev/wgrad_mix.py),
#   matrix-core counters of the weight gradients, of the conv_glds4 mix (tools/dev/conv_mix.py) and of the infer workload.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
rm -rf gpurun_out/pmc/*
MF="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE"
timeout 200 python3 tools/dev/wgrad_mix.py > gpurun_out/pmc/wgrad_mix_bare.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc/wf -o wf -- python3 tools/dev/wgrad_mix.py > gpurun_out/pmc/wf.log 2>&1; echo "wf rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc/ww -o ww -- python3 tools/dev/wgrad_mix.py > gpurun_out/pmc/ww.log 2>&1; echo "ww rc=$?"
timeout 300 rocprofv3 --pmc $MF --kernel-trace -d gpurun_out/pmc/wm -o wm -- python3 tools/dev/wgrad_mix.py > gpurun_out/pmc/wm.log 2>&1; echo "wm rc=$?"
timeout 300 rocprofv3 --pmc $MF --kernel-trace -d gpurun_out/pmc/cm -o cm -- python3 tools/dev/conv_mix.py > gpurun_out/pmc/cm.log 2>&1; echo "cm rc=$?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc/cf -o cf -- python3 tools/dev/conv_mix.py > gpurun_out/pmc/cf.log 2>&1; echo "cf rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc/cw -o cw -- python3 tools/dev/conv_mix.py > gpurun_out/pmc/cw.log 2>&1; echo "cw rc=$?"
timeout 400 rocprofv3 --pmc $MF --kernel-trace -d gpurun_out/pmc/im -o im -- python3 bench.py --workload infer --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc/im.log 2>&1; echo "im rc=$?"
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc/inf -o inf -- python3 bench.py --workload infer --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc/inf.log 2>&1; echo "inf rc=$?"
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc/infw -o infw -- python3 bench.py --workload infer --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc/infw.log 2>&1; echo "infw rc=$?"
for t in wf ww cf cw inf infw; do
  db=$(find gpurun_out/pmc/$t -name "*.db" | head -1)
  [ -n "$db" ] && python3 tools/dev/pmc_summary.py "$db" gpurun_out/pmc/$t.md "$t" gpurun_out/pmc/$t.json | tail -1
  rm -rf gpurun_out/pmc/$t
done
for t in wm cm im; do
  db=$(find gpurun_out/pmc/$t -name "*.db" | head -1)
  [ -n "$db" ] && python3 tools/dev/pmc_mfma_summary.py "$db" gpurun_out/pmc/$t.md "$t" | tail -1
  rm -rf gpurun_out/pmc/$t
done
grep -v amdgpu.ids gpurun_out/pmc/wgrad_mix_bare.log | tail -1
head -14 gpurun_out/pmc/wm.md | cut -c1-220
