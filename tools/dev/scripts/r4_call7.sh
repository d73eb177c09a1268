#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r4g
mkdir -p $O
timeout 900 python3 -m pytest tests/test_ddp_gpu.py tests/test_graphs_gpu.py tests/test_syncbn_gpu.py -x -q -m gpu > $O/pytest_ddp.txt 2>&1
tail -15 $O/pytest_ddp.txt
timeout 600 python3 -m pytest tests/test_conv_tiles_gpu.py -x -q -m gpu -k "cu_reserve or wgrad" > $O/pytest_res.txt 2>&1
tail -8 $O/pytest_res.txt
timeout 600 python3 -m pytest tests/test_cli_gpu.py -x -q -m gpu > $O/pytest_cli.txt 2>&1
tail -8 $O/pytest_cli.txt
