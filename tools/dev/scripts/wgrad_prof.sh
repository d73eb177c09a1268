#!/bin/bash
# Dev: per-kernel times of the wgrad microbenchmark by batch (main kernel vs the split reduction).
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
rm -rf gpurun_out/prof/wg*
rocprofv3 --kernel-trace --stats -d gpurun_out/prof/wg -o wg -- python3 tools/dev/wgrad_abl.py > gpurun_out/wgrad_prof.log 2>&1
db=$(find gpurun_out/prof/wg -name "*.db" | head -1)
python3 - "$db" <<'EOF'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='view' or type='table'")]
rows = c.execute('select name, start, end from kernels order by start').fetchall()
# last 10 launches of each batch size come in order: print per-kernel averages by groups of consecutive launches
import re, collections
seq = [(re.sub(r'\(anonymous namespace\)::', '', n)[:60], (e - s) / 1e3) for n, s, e in rows if 'wgrad' in n]
groups = collections.OrderedDict()
i = 0
for n, us in seq:
    groups.setdefault(n, []).append(us)
for n, v in groups.items():
    # 4 batch sizes x 12 launches
    k = len(v) // 4
    print(n, [round(sum(v[j * k:(j + 1) * k]) / k, 1) for j in range(4)])
EOF
