"""How full are the launches of a rocprofv3 --kernel-trace run? Per kernel name: time, and the CU-time left idle by
launches with fewer workgroups than CUs (duration x (1 - workgroups / 256); persistent / multi-round launches count as
full). usage: python tools/dev/rocprof_fill.py results.db [steps]"""
import re
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
gx = 'grid_x' if 'grid_x' in cols else 'grid_size_x' if 'grid_size_x' in cols else None
wx = 'workgroup_x' if 'workgroup_x' in cols else 'workgroup_size_x' if 'workgroup_size_x' in cols else None
if gx is None:
    print('columns:', cols)
    sys.exit(0)
rows = c.execute(f'select name, start, end, {gx}, grid_y, grid_z, {wx}, workgroup_y, workgroup_z from kernels').fetchall() \
    if 'grid_y' in cols else c.execute(f'select name, start, end, {gx}, 1, 1, {wx}, 1, 1 from kernels').fetchall()
agg = defaultdict(lambda: [0, 0.0, 0.0])
for n, s, e, g0, g1, g2, w0, w1, w2 in rows:
    wgs = (g0 * g1 * g2) / max(1, w0 * w1 * w2)
    d = (e - s) / 1e3
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'\(.*', '', n)[:60]
    a = agg[n]
    a[0] += 1
    a[1] += d
    a[2] += d * max(0.0, 1.0 - wgs / 256.0)
tot = sum(a[1] for a in agg.values())
idle = sum(a[2] for a in agg.values())
print(f'kernel time {tot / 1e3 / steps:.2f} ms per step, under-fill idle {idle / 1e3 / steps:.2f} ms per step (CU-time / 256)')
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][2])[:25]:
    print(f'{a[2] / 1e3 / steps:7.3f} ms idle of {a[1] / 1e3 / steps:7.3f} ms  n={a[0] / steps:6.1f}  {n}')
