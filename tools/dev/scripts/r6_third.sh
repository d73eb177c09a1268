#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6c
timeout 600 python tools/dev/arena_revert_demo.py > gpurun_out/r6c/arena_demo.txt 2>&1; echo "arena rc=$?" > gpurun_out/r6c/rc.txt
timeout 900 python tools/dev/tune_step.py -n 10 -r 3 bn.nt_fwd=1 bn.nt_fwd=2 bn.nt_fwd=4 bn.nt_fwd=7 bn.nt_bwd=1 bn.nt_bwd=2 bn.nt_bwd=4 bn.nt_bwd=7 bn.nt_fwd=7,bn.nt_bwd=7 bn.nt_fwd=4,bn.nt_bwd=6 > gpurun_out/r6c/tune_nt.txt 2>&1; echo "tune rc=$?" >> gpurun_out/r6c/rc.txt
timeout 600 python tools/dev/wgrad_ops.py > gpurun_out/r6c/wgrad_ops.txt 2>&1; echo "wgops rc=$?" >> gpurun_out/r6c/rc.txt
cat gpurun_out/r6c/rc.txt; tail -4 gpurun_out/r6c/arena_demo.txt; cat gpurun_out/r6c/tune_nt.txt | tail -14
