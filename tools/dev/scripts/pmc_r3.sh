#!/bin/bash
# Dev: PMC passes of round 3 (each pass = its own run with --kernel-trace only, as the pool requires):
#   matrix-core counters of EVERY tile kernel on the train shapes (tools/dev/conv_mix.py) and of the weight gradients
#   (tools/dev/wgrad_mix.py), HBM traffic (FETCH_SIZE / WRITE_SIZE) of the conv mix, the weight gradients and the
#   BatchNorm passes (tools/dev/bn_mix.py).
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc3
rm -rf gpurun_out/pmc3/*
MF="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE"
for mix in conv_mix wgrad_mix bn_mix; do
  timeout 200 python3 tools/dev/$mix.py > gpurun_out/pmc3/${mix}_bare.log 2>&1
done
run() {  # tag counters mix
  timeout 300 rocprofv3 --pmc $2 --kernel-trace -d gpurun_out/pmc3/$1 -o $1 -- python3 tools/dev/$3.py > gpurun_out/pmc3/$1.log 2>&1
  echo "$1 rc=$?"
}
run cm "$MF" conv_mix
run wm "$MF" wgrad_mix
run cf FETCH_SIZE conv_mix
run cw WRITE_SIZE conv_mix
run wf FETCH_SIZE wgrad_mix
run ww WRITE_SIZE wgrad_mix
run bf FETCH_SIZE bn_mix
run bw WRITE_SIZE bn_mix
for t in cf cw wf ww bf bw; do
  db=$(find gpurun_out/pmc3/$t -name "*.db" | head -1)
  [ -n "$db" ] && python3 tools/dev/pmc_summary.py "$db" gpurun_out/pmc3/$t.md "$t" gpurun_out/pmc3/$t.json | tail -1
  rm -rf gpurun_out/pmc3/$t
done
for t in wm cm; do
  db=$(find gpurun_out/pmc3/$t -name "*.db" | head -1)
  [ -n "$db" ] && python3 tools/dev/pmc_mfma_summary.py "$db" gpurun_out/pmc3/$t.md "$t" | tail -1
  rm -rf gpurun_out/pmc3/$t
done
grep -v amdgpu.ids gpurun_out/pmc3/*_bare.log | tail -16
head -20 gpurun_out/pmc3/cm.md | cut -c1-220
