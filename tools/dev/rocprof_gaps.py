"""Idle-gap analysis of a rocprofv3 --kernel-trace result (rocpd sqlite .db): where does the GPU wait for
the host? Prints total busy / idle time over the traced window and the largest gaps with the kernels on
either side, plus idle time grouped by the kernel that FOLLOWS the gap.
usage: python tools/dev/rocprof_gaps.py <results.db> [skip_first_n_kernels]"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = n.replace('unsigned short', 'bf16')
    return n[:70]


def main(db, skip=0):
    c = sqlite3.connect(db)
    rows = c.execute('select name, start, end from kernels order by start').fetchall()[int(skip):]
    # steady state only: the window between the first and the last optimizer step (sgd_kernel launches of > 100 us) if
    # the run has several, so that model construction and the first (allocating) steps stay out of the numbers
    sgd = [i for i, (n, s, e) in enumerate(rows) if 'sgd_kernel' in n and e - s > 100000]
    if len(sgd) >= 4:
        rows = rows[sgd[1] + 1:sgd[-1] + 1]
        print(f'steady-state window: {len(sgd) - 2} optimizer steps')
    busy = sum(e - s for _, s, e in rows)
    span = rows[-1][2] - rows[0][1]
    gaps = []
    last_end = rows[0][2]
    for i in range(1, len(rows)):
        n, s, e = rows[i]
        if s > last_end:
            gaps.append((s - last_end, i))
        last_end = max(last_end, e)
    idle = sum(g for g, _ in gaps)
    print(f'{len(rows)} kernels, span {span / 1e6:.2f} ms, busy(sum) {busy / 1e6:.2f} ms, idle {idle / 1e6:.2f} ms '
          f'({idle / span * 100:.1f}%), gaps > 20us: {sum(1 for g, _ in gaps if g > 20000)}')
    by_next = defaultdict(lambda: [0, 0])
    for g, i in gaps:
        k = by_next[short(rows[i][0])]
        k[0] += g; k[1] += 1
    hist = defaultdict(lambda: [0, 0])
    for g, _ in gaps:
        b = 2 if g < 2000 else 5 if g < 5000 else 10 if g < 10000 else 20 if g < 20000 else 100 if g < 100000 else 1000
        hist[b][0] += g; hist[b][1] += 1
    print('-- gap histogram (bucket upper bound us: count, total ms): ' +
          ', '.join(f'<{b}: {hist[b][1]}, {hist[b][0] / 1e6:.2f}' for b in sorted(hist)))
    print('-- idle time by the kernel that follows the gap')
    for n, (t, cnt) in sorted(by_next.items(), key=lambda kv: -kv[1][0])[:25]:
        print(f'  {t / 1e6:8.3f} ms  {cnt:6d} gaps  avg {t / cnt / 1e3:7.1f} us  {n}')
    print('-- largest gaps')
    for g, i in sorted(gaps, reverse=True)[:25]:
        print(f'  {g / 1e3:9.1f} us  after [{short(rows[i - 1][0])}]  before [{short(rows[i][0])}]')


if __name__ == '__main__':
    main(*sys.argv[1:3])
