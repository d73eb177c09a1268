#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for n in 6656 13312 26624; do timeout 300 python3 tools/dev/dcn_chunk_probe.py $n 2>&1 | grep -v amdgpu; done
