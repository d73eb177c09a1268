#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/gputests
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/gputests/gputests.log 2>&1; echo "gputests rc=$?" > gpurun_out/gputests/rc.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/gputests/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/gputests/rc.txt
cat gpurun_out/gputests/rc.txt; tail -5 gpurun_out/gputests/gputests.log | cut -c1-300; tail -2 gpurun_out/gputests/smoke.log | cut -c1-300
