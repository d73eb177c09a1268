"""Summarise a rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS GRBM_GUI_ACTIVE pass:
per kernel the share of the LDS-active cycles lost to bank conflicts and the LDS duty (active cycles per CU-cycle).
usage: python tools/dev/pmc_lds_summary.py results.db out.md [command]"""
import sqlite3
import sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
from pmc_summary import norm  # noqa: E402


def main(db, out, cmd=''):
    c = sqlite3.connect(db)
    rows = c.execute('select kernel_name, counter_name, count(*), avg(value) from counters_collection group by 1, 2').fetchall()
    dur = {}
    for k, n, a in c.execute('select name, count(*), avg(end-start) from kernels group by 1'):
        dur[norm(k)] = (n, a)
    tab = {}
    for k, cn, n, avg in rows:
        tab.setdefault(norm(k), {})[cn] = avg
    order = sorted(tab, key=lambda k: -(dur.get(k, (0, 0))[0] * dur.get(k, (0, 0))[1]))
    with open(out, 'w') as f:
        f.write(f'# rocprofv3 --pmc LDS counters\n\ncommand: `{cmd}`\n\nmean per dispatch; conflict share = SQ_LDS_BANK_CONFLICT / '
                'SQ_LDS_IDX_ACTIVE; LDS duty = SQ_LDS_IDX_ACTIVE / (GRBM_GUI_ACTIVE [summed over 8 XCDs] x 32 CUs)\n\n'
                '| kernel | dispatches | avg us | IDX_ACTIVE | BANK_CONFLICT | ADDR_CONFLICT | INSTS_LDS | conflict share | LDS duty |\n'
                '|---|---|---|---|---|---|---|---|---|\n')
        for k in order[:30]:
            n, ns = dur.get(k, (0, 0.0))
            t = tab[k]
            act, bc, gui = t.get('SQ_LDS_IDX_ACTIVE', 0.0), t.get('SQ_LDS_BANK_CONFLICT', 0.0), t.get('GRBM_GUI_ACTIVE', 0.0)
            if act <= 0:
                continue
            f.write(f'| `{k[:60]}` | {n} | {ns / 1e3:.1f} | {act:.3g} | {bc:.3g} | {t.get("SQ_LDS_ADDR_CONFLICT", 0.0):.3g} | '
                    f'{t.get("SQ_INSTS_LDS", 0.0):.3g} | {bc / act:.3f} | {act / (gui * 32) if gui else 0:.3f} |\n')
    print('wrote', out)


if __name__ == '__main__':
    main(*sys.argv[1:4])
