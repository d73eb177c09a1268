"""Rounds 4-5: HBM traffic and matrix-core counters of the REAL train step's launches (no proxy mix).
Inputs: the rocpd databases of three `rocprofv3 --pmc <set> --kernel-trace -- python3 bench.py --steps 2 --warmup 1
--no-cpu-baseline --no-also` passes (FETCH_SIZE; WRITE_SIZE; the MFMA set) — tools/dev/scripts/pmc_r5.sh.
Outputs under <out>: {RND}_pmc_step_traffic.md (every kernel variant: dispatches, mean us, fetch / write MB per dispatch, TB/s),
{RND}_stream_modes.md (conv1x1_stream_kernel<KB, WN, MODE> only), {RND}_pmc_mfma_step.md, and the `train` entries of traffic.json
(bytes per launch of every bench.py family, from the step's own launches).
usage: pmc_step_tables.py fetch.db write.db mfma.db <out dir>"""
import json
import os
import re
import sqlite3
import sys

from pmc_summary import norm

RND = os.environ.get('DAS_ROUND', 'r06')      # file-name prefix of the tables; ROUND_NO in their titles
ROUND_NO = RND.lstrip('r0') or '0'


def per_kernel(db, counter=None):
    c = sqlite3.connect(db)
    out = {}
    if counter:
        cols = [r[1] for r in c.execute('pragma table_info(counters_collection)')]
        kcol = 'kernel_name' if 'kernel_name' in cols else 'name'
        for k, cn, n, avg in c.execute(f'select {kcol}, counter_name, count(*), avg(value) from counters_collection group by 1, 2'):
            out.setdefault(norm(k), {})[cn] = (n, avg)
    dur = {}
    for k, n, us in c.execute('select name, count(*), avg(end - start) / 1000.0 from kernels group by 1'):
        dur[norm(k)] = (n, us)
    return out, dur


def targs(k):
    """template arguments of a normalised kernel name: 'conv_glds3_kernel<bf16, bf16, true, false>' -> ['bf16', 'bf16', 'true', 'false']"""
    i = k.find('<')
    return [a.strip() for a in k[i + 1:k.rfind('>')].split(',')] if i >= 0 else []


def _glds3(pp):     # conv_glds3_kernel<T, OT, PP, BITS, BNB, PLAIN>
    return lambda k: k.startswith('conv_glds3_kernel<') and targs(k)[2] == ('true' if pp else 'false')


def _glds4(pp, bmt):     # conv_glds4_kernel<T, OT, PP, BMT, MF, BNB, PLAIN>
    return lambda k: k.startswith('conv_glds4_kernel<') and targs(k)[2] == ('true' if pp else 'false') and targs(k)[3] == str(bmt)


FAMILIES = (   # bench.py family tag -> matcher on the normalised kernel name
    ('conv_glds4_kernel<pp,288>', _glds4(True, 288)),
    ('conv_glds4_kernel<pp>', _glds4(True, 256)),
    ('conv_glds3_kernel<pp>', _glds3(True)),
    ('conv_glds3_kernel', _glds3(False)),
    ('conv_glds_kernel', lambda k: k.startswith('conv_glds_kernel<')),
    ('conv_reg_kernel', lambda k: k.startswith('conv_reg_kernel<')),
    ('conv1x1_stream_kernel', lambda k: k.startswith('conv1x1_stream_kernel<')),
    ('conv1x1_kstream_kernel', lambda k: k.startswith('conv1x1_kstream_kernel<')),
    ('conv3x3_c64_kernel', lambda k: k.startswith('conv3x3_c64_kernel<')),
    ('conv_stem7x7_kernel', lambda k: k.startswith('conv_stem7x7_kernel<')),
    ('conv_wgrad_pp_kernel', lambda k: k == 'conv_wgrad_pp_kernel' or 'AccMap256' in k),
    ('conv_wgrad_kernel<bf16>', lambda k: k.startswith('conv_wgrad_kernel<') or 'AccMap128' in k or k.startswith('conv_wgrad_c64_kernel')
     or k.startswith('wgrad_c64_reduce_kernel')),
    ('bn_apply_kernel + bn_bwd_apply_dz_kernel + bn_bwd_reduce_kernel + bn_bwd_apply_kernel', lambda k: k.startswith('bn_') or k.startswith('fold_slots_kernel')),
)


def main(fdb, wdb, mdb, out):
    fe, dur = per_kernel(fdb, 'FETCH_SIZE')
    wr, _ = per_kernel(wdb, 'WRITE_SIZE')
    CMD = 'rocprofv3 --pmc {} --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also'
    rows = []
    for k, d in fe.items():
        n, f = d.get('FETCH_SIZE', (0, 0.0))
        w = wr.get(k, {}).get('WRITE_SIZE', (0, 0.0))[1]
        fmb, wmb = f * 1024 / 1e6 * 2, w * 1024 / 1e6          # FETCH_SIZE x 2 on gfx950 (MI355X_MICROARCH.md, HBM section)
        us = dur.get(k, (0, 0.0))[1]
        rows.append((k, n, us, fmb, wmb))
    rows.sort(key=lambda r: -r[1] * r[2])
    with open(f'{out}/{RND}_pmc_step_traffic.md', 'w') as f:
        f.write('# HBM traffic of every kernel of the REAL train step (round ' + ROUND_NO + ')\n\n')
        f.write(f'commands: `{CMD.format("FETCH_SIZE")}` and the same with `WRITE_SIZE` (separate passes, kernel trace only: '
                'tools/dev/scripts/pmc_r*.sh). The launches are the step\'s own (B = 16, 4-stage MSPN-50 + FPN + head; warm-up, timed and '
                'per-family measurement passes of bench.py: every step issues the same launches), not a proxy mix. FETCH_SIZE is doubled '
                '(gfx950 reports half of a wide coalesced read, MI355X_MICROARCH.md, HBM section); MB = 1e6 bytes; us = mean kernel '
                'duration in the FETCH pass (kernels run one at a time under --pmc).\n\n')
        f.write('| kernel | dispatches | mean us | fetch MB | write MB | (fetch + write) / time, TB/s |\n|---|---|---|---|---|---|\n')
        for k, n, us, fmb, wmb in rows[:70]:
            f.write(f'| `{k[:100]}` | {n} | {us:.1f} | {fmb:.1f} | {wmb:.1f} | {(fmb + wmb) / max(us, 1e-9):.2f} |\n')
    with open(f'{out}/{RND}_stream_modes.md', 'w') as f:
        f.write('# conv1x1_stream_kernel<KB, WN, MODE> on the step\'s real operands (round ' + ROUND_NO + ')\n\n')
        f.write('K = 32 KB input channels, WN waves across 32-channel groups; MODE 0 forward + BatchNorm statistics, 2 data gradient + '
                'second gradient, 3 data gradient + fused BatchNorm-backward sums with the mask from y (+ second gradient), 4 the same with '
                f'the mask recomputed from raw. Same passes as {RND}_pmc_step_traffic.md.\n\n')
        f.write('| variant | dispatches | mean us | fetch MB | write MB | TB/s |\n|---|---|---|---|---|---|\n')
        tot = [0, 0.0, 0.0]
        for k, n, us, fmb, wmb in sorted((r for r in rows if r[0].startswith('conv1x1_stream_kernel<')), key=lambda r: r[0]):
            f.write(f'| `{k}` | {n} | {us:.1f} | {fmb:.1f} | {wmb:.1f} | {(fmb + wmb) / max(us, 1e-9):.2f} |\n')
            tot[0] += n; tot[1] += n * us; tot[2] += n * (fmb + wmb)
        f.write(f'\nfamily: {tot[0]} dispatches, {tot[2] / max(tot[1], 1e-9):.2f} TB/s of measured HBM traffic\n')
    # matrix-core counters
    mf, mdur = per_kernel(mdb, 'mfma')
    with open(f'{out}/{RND}_pmc_mfma_step.md', 'w') as f:
        f.write('# Matrix-core counters of the REAL train step\'s tile / weight-gradient kernels (round ' + ROUND_NO + ')\n\n')
        f.write(f'command: `{CMD.format("SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE")}`. '
                'mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128): GUI_ACTIVE is summed over the 8 XCDs (32 CUs x 4 SIMDs each); '
                'TF from MOPS_BF16 x 512 FLOP / mean duration.\n\n')
        f.write('| kernel | dispatches | mean us | mfma_busy | TF (counter) |\n|---|---|---|---|---|\n')
        mrows = []
        for k, d in mf.items():
            if 'SQ_VALU_MFMA_BUSY_CYCLES' not in d or 'GRBM_GUI_ACTIVE' not in d:
                continue
            n, busy = d['SQ_VALU_MFMA_BUSY_CYCLES']
            gui = d['GRBM_GUI_ACTIVE'][1]
            mops = d.get('SQ_INSTS_VALU_MFMA_MOPS_BF16', (0, 0.0))[1]
            us = mdur.get(k, (0, 0.0))[1]
            if busy <= 0 or us <= 0:
                continue
            mrows.append((k, n, us, busy / (gui * 128.0), mops * 512 / us / 1e6))
        for k, n, us, b, tf in sorted(mrows, key=lambda r: -r[1] * r[2])[:40]:
            f.write(f'| `{k[:100]}` | {n} | {us:.1f} | {b:.3f} | {tf:.0f} |\n')
    # traffic.json: the step's own launches per bench.py family
    path = f'{out}/traffic.json'
    doc = json.load(open(path))
    src = ('rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only; FETCH_SIZE x 2 as the guide prescribes for '
           'gfx950) over the REAL train step: `bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also` (round ' + ROUND_NO + ', tools/dev/scripts/pmc_' + RND.replace('0', '') + '.sh); '
           'per launch of the family, reduce / fold passes counted as launches as bench.py counts them')
    for tag, match in FAMILIES:
        ks = [r for r in rows if match(r[0])]
        n = sum(r[1] for r in ks)
        if not n:
            continue
        fm = sum(r[1] * r[3] for r in ks) / n
        wm = sum(r[1] * r[4] for r in ks) / n
        doc['families'].setdefault(tag, {})['train'] = dict(dispatches=n, fetch_mb=round(fm, 2), write_mb=round(wm, 2),
                                                            hbm_mb_per_launch=round(fm + wm, 2), source=src)
    doc['note'] = ('HBM bytes per launch behind bench.py roofline*.traffic. train: measured on the step\'s own launches (round ' + ROUND_NO + '); '
                   'compare with the line\'s algorithmic_mb_per_launch directly. infer: round-2 passes.')
    json.dump(doc, open(path, 'w'), indent=1, sort_keys=True)
    print('wrote tables and', path)


if __name__ == '__main__':
    main(*sys.argv[1:5])
