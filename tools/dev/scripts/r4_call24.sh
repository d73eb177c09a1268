#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 300 python3 -m pytest tests/test_upconv_gpu.py tests/test_hip_backward_elem.py -q -m gpu -x 2>&1 | tail -3
timeout 300 python3 tools/dev/upmerge_bench.py 2>&1 | grep -v amdgpu | tee gpurun_out/r4/upmerge_bench.txt
