#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 900 python3 -m pytest tests/test_flat_paths_gpu.py tests/test_hip_backward_elem.py tests/test_train_step_gpu.py tests/test_train_gpu.py tests/test_conv_tiles_gpu.py -q -m gpu -x -k "pack or pool or train or bf16_exact or optimizer" 2>&1 | tail -4
timeout 300 python3 tools/dev/tune_step.py -n 12 -r 3 > gpurun_out/r4/tune_after_pack.txt 2>&1; cat gpurun_out/r4/tune_after_pack.txt | grep -v amdgpu
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-also > /tmp/kt.log 2>&1
python3 tools/dev/rocprof_summary.py $(find /tmp/kt -name "*.db" | head -1) gpurun_out/r4/kstats_after.md x > /dev/null; grep -E "pack_conv|maxpool|sgd_kernel" gpurun_out/r4/kstats_after.md | cut -c1-140
