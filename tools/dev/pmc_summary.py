"""Summarise rocprofv3 --pmc results (rocpd sqlite): per-kernel mean of each counter per dispatch.
usage: python tools/dev/pmc_summary.py results.db out.md "command" [out.json]
The optional JSON maps the kernel name (template arguments kept, parameter list dropped) to the corrected HBM bytes
per dispatch; bench.py reads it to fill `roofline.traffic` for the dominant kernel."""
import json
import re
import sqlite3
import sys


def norm(k):
    k = re.sub(r'\(anonymous namespace\)::', '', k).replace('unsigned short', 'bf16')
    k = re.sub(r'^void ', '', k)
    depth, out = 0, []
    for ch in k:            # cut at the parameter list: the first '(' outside template brackets
        if ch == '<':
            depth += 1
        elif ch == '>':
            depth -= 1
        elif ch == '(' and depth == 0:
            break
        out.append(ch)
    return ''.join(out)


def main(db, out, cmd='', out_json=None):
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
    table = {}
    with open(out, 'w') as f:
        f.write(f'# rocprofv3 --pmc summary\n\ncommand: `{cmd}`\n\n')
        f.write('Units: FETCH_SIZE / WRITE_SIZE are in KiB as reported; on gfx950 FETCH_SIZE under-reports wide coalesced '
                'reads by 2x (MI355X_MICROARCH.md, HBM section) - the "corrected" column doubles it. WRITE_SIZE is '
                'uncalibrated there and taken as reported.\n\n')
        f.write('| kernel | counter | dispatches | mean per dispatch | corrected MB per dispatch |\n|---|---|---|---|---|\n')
        for k, cn, n, avg, tot in rows:
            corr = avg * 1024 / 1e6 * (2 if cn == 'FETCH_SIZE' else 1)
            e = table.setdefault(norm(k), dict(dispatches=n))
            e['fetch_mb' if cn == 'FETCH_SIZE' else cn.lower() + '_mb'] = round(corr, 3)
        for k, cn, n, avg, tot in rows[:80]:
            corr = avg * 1024 / 1e6 * (2 if cn == 'FETCH_SIZE' else 1)
            f.write(f'| `{norm(k)[:90]}` | {cn} | {n} | {avg:.1f} | {corr:.2f} |\n')
    print('wrote', out)
    if out_json:
        with open(out_json, 'w') as f:
            json.dump(dict(command=cmd, note='corrected MB per dispatch: FETCH_SIZE KiB x 2 (gfx950), WRITE_SIZE KiB as reported',
                           kernels=table), f, indent=1, sort_keys=True)
        print('wrote', out_json)


if __name__ == '__main__':
    main(*sys.argv[1:5])
