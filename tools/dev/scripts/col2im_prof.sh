#!/bin/bash
# Dev: kernel times of the DCNv2 backward sampling microbenchmark at a few offset scales.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
for s in 0 0.5 3; do
  rm -rf gpurun_out/prof/c2i
  timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/c2i -o c2i -- python3 tools/dev/col2im_bench.py $s > gpurun_out/prof/c2i.log 2>&1
  db=$(find gpurun_out/prof/c2i -name "*.db" | head -1)
  echo "== offset std $s"
  python3 - "$db" <<'EOF'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
for n, cnt, avg in c.execute("select name, count(*), avg(end-start) from kernels where name like '%col2im%' group by name"):
    print(f'  {avg / 1e3:9.1f} us x{cnt}  {n[:70]}')
EOF
done
rm -rf gpurun_out/prof/c2i
