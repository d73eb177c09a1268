"""Kernels of ONE steady-state train step (between two optimizer steps) of a rocprofv3 --kernel-trace run, by launch
count: which small launches are left? usage: python tools/dev/rocprof_step_census.py results.db"""
import re
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
rows = c.execute('select name, start, end from kernels order by start').fetchall()
sgd = [i for i, r in enumerate(rows) if 'sgd_kernel' in r[0]]
# the optimizer launches twice per step (two groups): take the window between the last launches of two steps
ends = [sgd[i] for i in range(len(sgd)) if i + 1 == len(sgd) or sgd[i + 1] - sgd[i] > 50]
a, b = ends[-3], ends[-2]
agg = defaultdict(lambda: [0, 0.0])
for n, s, e in rows[a + 1:b + 1]:
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'at::native::', '', n)[:90]
    agg[n][0] += 1
    agg[n][1] += (e - s) / 1e3
tot = sum(v[0] for v in agg.values())
print(f'{tot} launches, {sum(v[1] for v in agg.values()) / 1e3:.2f} ms of kernel time in the step')
small = {k: v for k, v in agg.items() if v[1] / v[0] < 12.0}
print(f'launches under 12 us on average: {sum(v[0] for v in small.values())}, {sum(v[1] for v in small.values()) / 1e3:.2f} ms')
for k, v in sorted(small.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f'{v[0]:5d} x {v[1] / v[0]:6.1f} us  {k}')
