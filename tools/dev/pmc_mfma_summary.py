"""Summarise a rocprofv3 --pmc pass with the matrix-core counters: per kernel the mean per dispatch of
SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES, SQ_INSTS_VALU_MFMA_MOPS_BF16, GRBM_GUI_ACTIVE and
  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128)   (GRBM_GUI_ACTIVE is summed over the 8 XCDs, checked
              against the kernel time x clock; x 128 = 1024 SIMDs / 8: the fraction of the matrix pipes' cycles spent
              executing MFMA — it agrees with MFMA TF / 2500)
  mfma TF   = SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512 FLOP / kernel time (from the kernel trace of the same pass)
usage: python tools/dev/pmc_mfma_summary.py results.db out.md "command"."""
import re
import sqlite3
import sys

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
from pmc_summary import norm  # noqa: E402


def main(db, out, cmd=''):
    c = sqlite3.connect(db)
    rows = c.execute('select kernel_name, counter_name, count(*), avg(value) from counters_collection group by 1, 2').fetchall()
    dur = {norm(k): (n, a) for k, n, a in c.execute('select name, count(*), avg(end-start) from kernels group by 1')}
    tab = {}
    for k, cn, n, avg in rows:
        tab.setdefault(norm(k), {})[cn] = (n, avg)
    order = sorted(tab, key=lambda k: -(dur.get(k, (0, 0))[0] * dur.get(k, (0, 0))[1]))
    with open(out, 'w') as f:
        f.write(f'# rocprofv3 --pmc matrix-core counters\n\ncommand: `{cmd}`\n\n')
        f.write('mean per dispatch; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE [summed over 8 XCDs] x 128); '
                'MFMA TF = SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512 / kernel time of this (profiled, slower-clocked) pass\n\n')
        f.write('| kernel | dispatches | time us | MFMA_BUSY_CYCLES | SQ_BUSY_CYCLES | SQ_WAVE_CYCLES | MFMA_MOPS_BF16 | GUI_ACTIVE | mfma_busy | MFMA TF |\n')
        f.write('|---|---|---|---|---|---|---|---|---|---|\n')
        for k in order[:40]:
            t = tab[k]
            g = lambda n: t.get(n, (0, 0.0))[1]
            n, us = dur.get(k, (0, 0.0))
            mb, ga, mops = g('SQ_VALU_MFMA_BUSY_CYCLES'), g('GRBM_GUI_ACTIVE'), g('SQ_INSTS_VALU_MFMA_MOPS_BF16')
            if mb == 0 and mops == 0:
                continue
            busy = mb / (ga * 128) if ga else 0.0
            tf = mops * 512 / (us * 1e-9) / 1e12 if us else 0.0
            f.write(f'| `{k[:70]}` | {n} | {us / 1e3:.1f} | {mb:.3g} | {g("SQ_BUSY_CYCLES"):.3g} | {g("SQ_WAVE_CYCLES"):.3g} | '
                    f'{mops:.3g} | {ga:.3g} | {busy:.3f} | {tf:.0f} |\n')
    print('wrote', out)


if __name__ == '__main__':
    main(*sys.argv[1:4])
