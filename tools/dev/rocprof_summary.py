"""Summarise a rocprofv3 --kernel-trace --stats result (rocpd sqlite .db) as a markdown table.
usage: python tools/dev/rocprof_summary.py gpurun_out/prof/x_results.db profiles/name.md "command line"
"""
import re
import sqlite3
import sys


def main(db, out, cmd=''):
    c = sqlite3.connect(db)
    rows = c.execute('select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) '
                     'from kernels group by name order by 3 desc').fetchall()
    tot = sum(r[2] for r in rows)
    with open(out, 'w') as f:
        f.write(f'# rocprofv3 --kernel-trace --stats summary\n\ncommand: `{cmd}`\n\n')
        f.write(f'total kernel time: {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches\n\n')
        f.write('| % | calls | total ms | avg us | min us | max us | kernel |\n|---|---|---|---|---|---|---|\n')
        for n, cnt, s, a, mn, mx in rows:
            n = re.sub(r'\(anonymous namespace\)::', '', n)
            n = n.replace('unsigned short', 'bf16')
            f.write(f'| {s / tot * 100:.2f} | {cnt} | {s / 1e6:.3f} | {a / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | `{n[:110]}` |\n')
    print('wrote', out)


if __name__ == '__main__':
    main(*sys.argv[1:4])
