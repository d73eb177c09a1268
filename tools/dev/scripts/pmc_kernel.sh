#!/bin/bash
# Dev: a few SQ / TCC counters for the kernels of one dev tool run. usage: pmc_kernel.sh <tag> <counters...> -- <python script> [args]
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
tag=$1; shift
ctr=""
while [ "$1" != "--" ]; do ctr="$ctr $1"; shift; done
shift
mkdir -p gpurun_out/pmc; rm -rf gpurun_out/pmc/$tag
timeout 300 rocprofv3 --pmc $ctr --kernel-trace -d gpurun_out/pmc/$tag -o $tag -- python3 "$@" > gpurun_out/pmc/$tag.log 2>&1; echo "rc=$?"
db=$(find gpurun_out/pmc/$tag -name "*.db" | head -1)
python3 tools/dev/pmc_summary.py "$db" gpurun_out/pmc/$tag.md "$tag" | tail -1
rm -rf gpurun_out/pmc/$tag
grep -E "deform|kernel \|" gpurun_out/pmc/$tag.md | head -30 | cut -c1-200
