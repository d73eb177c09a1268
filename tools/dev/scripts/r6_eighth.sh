#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6h
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r6h/gputests.log 2>&1; echo "gputests rc=$?" > gpurun_out/r6h/rc.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6h/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r6h/rc.txt
cat gpurun_out/r6h/rc.txt; tail -5 gpurun_out/r6h/gputests.log | cut -c1-300; tail -2 gpurun_out/r6h/smoke.log | cut -c1-300
