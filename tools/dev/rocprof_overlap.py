"""How much of the kernel time of a rocprofv3 --kernel-trace run overlapped (several streams): sum of durations vs the
union of the busy intervals, and per kernel the share of its time during which another kernel was running.
usage: python tools/dev/rocprof_overlap.py results.db"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute('select name, start, end from kernels order by start').fetchall()
tot = sum(e - s for _, s, e in rows)
busy, cur_s, cur_e = 0, None, None
for _, s, e in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = rows[-1][2] - rows[0][1]
print(f'kernels {len(rows)}  sum of durations {tot / 1e6:.1f} ms  busy (union) {busy / 1e6:.1f} ms  span {span / 1e6:.1f} ms')
# overlap share per kernel name (sweep)
ev = []
for i, (n, s, e) in enumerate(rows):
    ev.append((s, 1, i))
    ev.append((e, -1, i))
ev.sort()
active, last = set(), None
ov = [0] * len(rows)
for t, d, i in ev:
    if last is not None and len(active) > 1:
        for j in active:
            ov[j] += t - last
    last = t
    if d == 1:
        active.add(i)
    else:
        active.discard(i)
agg = {}
for (n, s, e), o in zip(rows, ov):
    n = re.sub(r'\(.*', '', n)[:60]
    a = agg.setdefault(n, [0, 0, 0])
    a[0] += e - s
    a[1] += o
    a[2] += 1
for n, (d, o, k) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f'{d / 1e6:9.2f} ms  overlapped {100.0 * o / d:5.1f}%  n={k:5d}  {n}')
