"""For the small glue kernels (copies, fills, elementwise) of a rocprofv3 --kernel-trace run: which kernels surround them?
usage: python tools/dev/rocprof_context.py results.db"""
import re
import sqlite3
import sys
from collections import Counter

c = sqlite3.connect(sys.argv[1])
rows = c.execute('select name, start, end from kernels order by start').fetchall()


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'at::native::', '', n)
    return re.sub(r'\(.*', '', n)[:48]


names = [short(r[0]) for r in rows]
glue = ('__amd_rocclr_copyBuffer', '__amd_rocclr_fillBuffer', 'vectorized_elementwise', 'elementwise_kernel', 'unrolled_elementwise',
        'CatArray', 'index_elementwise', 'reduce_kernel')
cnt, dur = Counter(), Counter()
for i, n in enumerate(names):
    if not any(g in n for g in glue):
        continue
    j = i - 1
    while j >= 0 and any(g in names[j] for g in glue):
        j -= 1
    k = i + 1
    while k < len(names) and any(g in names[k] for g in glue):
        k += 1
    key = (n[:34], names[j] if j >= 0 else '-', names[k] if k < len(names) else '-')
    cnt[key] += 1
    dur[key] += rows[i][2] - rows[i][1]
for key, t in dur.most_common(60):
    print(f'{t / 1e6:8.3f} ms  n={cnt[key]:5d}  {key[0]:34s} after [{key[1]}] before [{key[2]}]')
