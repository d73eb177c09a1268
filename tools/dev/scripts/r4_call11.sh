#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r4k
mkdir -p $O
timeout 300 python3 tools/dev/glue_sources.py > $O/glue_sources.txt 2>&1
head -50 $O/glue_sources.txt | cut -c1-200
timeout 600 python3 -m pytest tests/test_model_gpu.py -x -q -m gpu -k unconsumed 2>&1 | grep -E "^E|passed|failed" | head -5
