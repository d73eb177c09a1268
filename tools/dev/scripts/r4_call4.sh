#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r4d
mkdir -p $O
timeout 600 python3 -m pytest tests/test_conv_tiles_gpu.py tests/test_bn_fused_gpu.py -x -q -m gpu -k "pt3 or epilogues_forced or ragged_levels_forced or dgrad_forced or real_sizes or dgrad_epilogue" > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
timeout 300 python3 tools/dev/pt3_stamps.py > $O/pt3_stamps.txt 2>&1
cat $O/pt3_stamps.txt
timeout 300 python3 tools/dev/pt3_bench.py > $O/pt3_bench.txt 2>&1
cat $O/pt3_bench.txt
