#!/bin/bash
# the full GPU suite three times in a row (no -x): any test that fails in one run and passes in another is a flaky tolerance
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6j
for i in 1 2 3; do
  timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r6j/run$i.log 2>&1; echo "run $i rc=$?" >> gpurun_out/r6j/rc.txt
  tail -1 gpurun_out/r6j/run$i.log; grep "^FAILED" gpurun_out/r6j/run$i.log
done
cat gpurun_out/r6j/rc.txt
