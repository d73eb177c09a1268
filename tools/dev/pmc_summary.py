"""Summarise rocprofv3 --pmc results (rocpd sqlite): per-kernel mean of each counter per dispatch.
usage: python tools/dev/pmc_summary.py results.db out.md "command"
"""
import re
import sqlite3
import sys


def main(db, out, cmd=''):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    view = 'counters_collection' if 'counters_collection' in tabs else None
    if view is None:
        cand = [t for t in tabs if 'counter' in t.lower()]
        raise SystemExit(f'no counters_collection view; counter-like tables: {cand}')
    cols = [r[1] for r in c.execute(f'pragma table_info({view})')]
    kcol = 'kernel_name' if 'kernel_name' in cols else 'name'
    rows = c.execute(f'select {kcol}, counter_name, count(*), avg(value), sum(value) from {view} group by 1, 2 '
                     f'order by 5 desc').fetchall()
    with open(out, 'w') as f:
        f.write(f'# rocprofv3 --pmc summary\n\ncommand: `{cmd}`\n\n')
        f.write('Units: FETCH_SIZE / WRITE_SIZE are in KiB as reported; on gfx950 FETCH_SIZE under-reports wide coalesced '
                'reads by 2x (MI355X_MICROARCH.md, HBM section) - the "corrected" column doubles it.\n\n')
        f.write('| kernel | counter | dispatches | mean per dispatch | corrected MB per dispatch |\n|---|---|---|---|---|\n')
        for k, cn, n, avg, tot in rows[:60]:
            k = re.sub(r'\(anonymous namespace\)::', '', k).replace('unsigned short', 'bf16')[:90]
            corr = avg * 1024 / 1e6 * (2 if cn == 'FETCH_SIZE' else 1)
            f.write(f'| `{k}` | {cn} | {n} | {avg:.1f} | {corr:.2f} |\n')
    print('wrote', out)


if __name__ == '__main__':
    main(*sys.argv[1:4])
