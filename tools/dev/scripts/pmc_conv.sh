#!/bin/bash
# Dev: HBM traffic counters (separate --pmc passes) for the conv_glds4_kernel launches of one train step.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
timeout 200 python3 tools/dev/conv_mix.py > gpurun_out/pmc/conv_mix_bare.log 2>&1
timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc/cf -o cf -- python3 tools/dev/conv_mix.py > gpurun_out/pmc/cf.log 2>&1
echo "fetch pass rc=$?"
timeout 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc/cw -o cw -- python3 tools/dev/conv_mix.py > gpurun_out/pmc/cw.log 2>&1
echo "write pass rc=$?"
for t in cf cw; do
  db=$(find gpurun_out/pmc/$t -name "*.db" | head -1)
  [ -n "$db" ] && python3 tools/dev/pmc_summary.py "$db" gpurun_out/pmc/$t.md "$t" gpurun_out/pmc/$t.json | tail -1
  rm -rf gpurun_out/pmc/$t
done
grep -v amdgpu.ids gpurun_out/pmc/conv_mix_bare.log | tail -2
grep -i "glds" gpurun_out/pmc/cf.md gpurun_out/pmc/cw.md | head
