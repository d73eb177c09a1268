"""Summarise a rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES pass: per kernel the mean duration next to a lower bound of its
vector-ALU time = SQ_INSTS_VALU (wave instructions) x 4 cycles / (1024 SIMDs x 2.4 GHz) — a kernel whose bound is close
to its duration is VALU bound, not memory bound.
usage: python tools/dev/pmc_valu_summary.py results.db out.md"""
import sqlite3
import sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
from pmc_summary import norm  # noqa: E402


def main(db, out):
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
        f.write('| kernel | dispatches | avg us | total ms | VALU wave-instr | per wave | VALU bound us | bound / time |\n|---|---|---|---|---|---|---|---|\n')
        for k in order[:45]:
            n, ns = dur.get(k, (0, 0.0))
            valu, waves = tab[k].get('SQ_INSTS_VALU', 0.0), tab[k].get('SQ_WAVES', 0.0)
            bound = valu * 4 / (1024 * 2.4e9) * 1e6
            f.write(f'| `{k[:60]}` | {n} | {ns / 1e3:.1f} | {n * ns / 1e6:.2f} | {valu:.3g} | {valu / waves if waves else 0:.0f} | {bound:.1f} | '
                    f'{bound / (ns / 1e3) if ns else 0:.2f} |\n')
    print(open(out).read())


if __name__ == '__main__':
    main(*sys.argv[1:3])
