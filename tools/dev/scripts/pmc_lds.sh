#!/bin/bash
# LDS counter pass (bank conflicts, duty) over the weight-gradient replay, the conv mix and the infer workload
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS GRBM_GUI_ACTIVE"
timeout 300 rocprofv3 --pmc $C --kernel-trace -d /tmp/lw -o lw -- python3 tools/dev/wgrad_mix.py > /tmp/lw.log 2>&1; echo "lw rc=$?"
timeout 300 rocprofv3 --pmc $C --kernel-trace -d /tmp/lc -o lc -- python3 tools/dev/conv_mix.py > /tmp/lc.log 2>&1; echo "lc rc=$?"
for t in lw lc; do
  db=$(find /tmp/$t -name "*.db" | head -1)
  [ -n "$db" ] && python3 tools/dev/pmc_lds_summary.py "$db" gpurun_out/pmc/$t.md "$t" | tail -1
done
head -14 gpurun_out/pmc/lw.md | cut -c1-200
head -16 gpurun_out/pmc/lc.md | cut -c1-200
