#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6d
timeout 600 python -m pytest tests/test_full_width_gpu.py -q -s -k "inference_graph" > gpurun_out/r6d/infer_graph_test.log 2>&1; echo "igtest rc=$?" > gpurun_out/r6d/rc.txt
timeout 600 python bench.py --workload infer --no-cpu-baseline > gpurun_out/r6d/infer.json 2> gpurun_out/r6d/infer.err; echo "infer rc=$?" >> gpurun_out/r6d/rc.txt
timeout 600 python bench.py --workload infer --no-cpu-baseline --no-infer-graph > gpurun_out/r6d/infer_eager.json 2> gpurun_out/r6d/infer_eager.err; echo "infer eager rc=$?" >> gpurun_out/r6d/rc.txt
timeout 600 python tools/dev/copy_sources.py > gpurun_out/r6d/copy_sources.txt 2>&1; echo "copysrc rc=$?" >> gpurun_out/r6d/rc.txt
cat gpurun_out/r6d/rc.txt; tail -3 gpurun_out/r6d/infer_graph_test.log; head -c 600 gpurun_out/r6d/infer.json; echo; head -c 400 gpurun_out/r6d/infer_eager.json; echo; tail -70 gpurun_out/r6d/copy_sources.txt
