#!/bin/bash
# Dev: GPU time of the data path alone (kernel + copy trace of tools/dev/loader_bench.py --no-train)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r3
mkdir -p $O; rm -rf $O/ltr
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $O/ltr -o ltr -- python3 tools/dev/loader_bench.py --no-train --batches 4 > $O/loader_prof.log 2>&1
db=$(find $O/ltr -name "*.db" | head -1)
python3 tools/dev/rocprof_summary.py "$db" $O/loader_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python3 tools/dev/loader_bench.py --no-train --batches 4" > /dev/null
python3 - "$db" <<'PY' > $O/loader_copies.txt 2>&1
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if 'copy' in t.lower() or 'memory' in t.lower()])
for t in tabs:
    if 'memory_cop' in t.lower() and 'rocpd' not in t.lower():
        cols = [r[1] for r in c.execute(f'pragma table_info({t})')]
        print(t, cols)
        try:
            for r in c.execute(f'select name, count(*), sum(end-start)/1e6, sum(size)/1e6 from {t} group by name'):
                print(r)
        except Exception as e:
            print('ERR', e)
PY
rm -rf $O/ltr
grep -v amdgpu $O/loader_prof.log | tail -8
head -40 $O/loader_kernel_stats.md | cut -c1-200
cat $O/loader_copies.txt | cut -c1-300
