"""Dev tool: the kernel sequence of the last N dispatches of a rocprofv3 --kernel-trace result (rocpd sqlite .db), with
the context of every `pattern` kernel (what runs before / after the copies and fills).
usage: python tools/dev/rocprof_seq.py <results.db> <pattern> [last_n]"""
import re
import sqlite3
import sys
from collections import Counter


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'void |at::native::', '', n)
    return n[:60]


db, pat = sys.argv[1], sys.argv[2]
last = int(sys.argv[3]) if len(sys.argv) > 3 else 400
c = sqlite3.connect(db)
rows = c.execute('select name, start, end from kernels order by start').fetchall()[-last:]
ctx = Counter()
for i, (n, s, e) in enumerate(rows):
    if pat in n:
        prev = next((short(rows[j][0]) for j in range(i - 1, -1, -1) if pat not in rows[j][0]), '-')
        nxt = next((short(rows[j][0]) for j in range(i + 1, len(rows)) if pat not in rows[j][0]), '-')
        ctx[(prev, nxt)] += 1
for (p, n), k in ctx.most_common(60):
    print(f'{k:4d}  after [{p}]  before [{n}]')
