#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 900 python3 bench.py --gpus 2 --share-gpu --no-cpu-baseline --no-also --steps 4 --warmup 2 2>/tmp/err.txt | tail -1 | cut -c1-400
tail -3 /tmp/err.txt | cut -c1-300
