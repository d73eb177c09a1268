"""Round 3: update the `train` entries of profiles/traffic.json (HBM bytes per launch behind bench.py's roofline*.traffic)
from the PMC summaries tools/dev/scripts/pmc_r3.sh leaves in gpurun_out/pmc3/ (FETCH_SIZE x 2 + WRITE_SIZE, separate
passes, --kernel-trace only), and write the tracked tables profiles/r03_pmc_*.md.
usage: python tools/dev/make_traffic_r03.py gpurun_out/pmc3 profiles"""
import json
import shutil
import sys

PASSES = 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only; FETCH_SIZE x 2 as the guide prescribes for gfx950)'


def load(d, name):
    return json.load(open(f'{d}/{name}.json'))['kernels']


def fam(fe, wr, match):
    ks = [k for k in fe if match(k)]
    n = sum(fe[k]['dispatches'] for k in ks)
    if not n:
        return None
    f = sum(fe[k]['dispatches'] * fe[k].get('fetch_mb', 0) for k in ks) / n
    w = sum(wr[k]['dispatches'] * wr[k].get('write_size_mb', 0) for k in ks if k in wr) / n
    return dict(fetch_mb=round(f, 2), write_mb=round(w, 2), hbm_mb_per_launch=round(f + w, 2), dispatches=n)


def main(d, out):
    cf, cw, wf, ww, bf, bw = (load(d, n) for n in ('cf', 'cw', 'wf', 'ww', 'bf', 'bw'))
    path = f'{out}/traffic.json'
    doc = json.load(open(path))
    fams = doc['families']
    src_c = PASSES + ' over tools/dev/conv_mix.py: the tile- and stream-kernel shapes (forward + data gradient) of the train step with their per-step counts, B=16'
    src_w = PASSES + ' over tools/dev/wgrad_mix.py: the 279 weight-gradient ops of one train step (B=16) in the 13 batched launches backward issues; kernel + its reduce pass, per kernel launch'
    src_b = PASSES + ' over tools/dev/bn_mix.py: the BatchNorm passes of one train step (B=16) with their per-step counts'
    tags = (('conv_glds4_kernel<pp,288>', cf, cw, src_c, lambda k: k.startswith('conv_glds4_kernel<') and k.endswith('true, 288>')),
            ('conv_glds4_kernel<pp>', cf, cw, src_c, lambda k: k.startswith('conv_glds4_kernel<') and k.endswith('true, 256>')),
            ('conv_glds3_kernel<pp>', cf, cw, src_c, lambda k: k.startswith('conv_glds3_kernel<') and k.endswith('true>')),
            ('conv_glds3_kernel', cf, cw, src_c, lambda k: k.startswith('conv_glds3_kernel<') and k.endswith('false>')),
            ('conv_glds_kernel', cf, cw, src_c, lambda k: k.startswith('conv_glds_kernel<')),
            ('conv1x1_stream_kernel', cf, cw, src_c, lambda k: k.startswith('conv1x1_stream_kernel<')),
            ('conv3x3_c64_kernel', cf, cw, src_c, lambda k: k.startswith('conv3x3_c64_kernel<')),
            ('conv_wgrad_pp_kernel', wf, ww, src_w, lambda k: k == 'conv_wgrad_pp_kernel' or 'AccMap256' in k),
            ('conv_wgrad_kernel<bf16>', wf, ww, src_w, lambda k: k.startswith('conv_wgrad_kernel<') or 'AccMap128' in k or
             k.startswith('conv_wgrad_c64_kernel') or k.startswith('wgrad_c64_reduce_kernel')),
            ('bn_apply_kernel + bn_bwd_apply_dz_kernel + bn_bwd_reduce_kernel + bn_bwd_apply_kernel', bf, bw, src_b,
             lambda k: k.startswith('bn_') or k.startswith('fold_slots_kernel')))
    for tag, fe, wr, src, match in tags:
        e = fam(fe, wr, match)
        if e:
            e['source'] = src
            fams.setdefault(tag, {})['train'] = e
    # algorithmic bytes of exactly the launches the counters saw (tools/dev/conv_mix.py prints them per family)
    for line in open(f'{d}/conv_mix_bare.log'):
        if line.startswith('ALGORITHMIC '):
            json.dump(dict(note='algorithmic bytes per launch (x once, y once, weights once) of the launches of tools/dev/conv_mix.py, '
                                'per kernel family as ops.last_kernel() names it; the PMC passes behind profiles/traffic.json '
                                'counted the same launches (warm-up launch of every shape included)',
                           families=json.loads(line[len('ALGORITHMIC '):])), open(f'{out}/r03_conv_mix_algorithmic.json', 'w'))
    try:
        alg = json.load(open(f'{out}/r03_conv_mix_algorithmic.json'))['families']
    except OSError:
        alg = {}
    if 'conv_glds3_kernel<pp>' in alg and 'conv_glds3_kernel<splitk>' in alg:   # (one kernel template to the counters)
        a, b = alg['conv_glds3_kernel<pp>'], alg.pop('conv_glds3_kernel<splitk>')
        n = a['launches'] + b['launches']
        alg['conv_glds3_kernel<pp>'] = dict(launches=n, algorithmic_mb_per_launch=(
            a['launches'] * a['algorithmic_mb_per_launch'] + b['launches'] * b['algorithmic_mb_per_launch']) / n)
    for tag, a in alg.items():
        e = fams.get(tag, {}).get('train')
        if e and e['dispatches'] == a['launches']:
            e['algorithmic_mb_per_launch_same_launches'] = round(a['algorithmic_mb_per_launch'], 2)
            e['hbm_over_algorithmic'] = round(e['hbm_mb_per_launch'] / a['algorithmic_mb_per_launch'], 3)
    json.dump(doc, open(path, 'w'), indent=1, sort_keys=True)
    # tracked tables
    for name, dst, cmd in (('cm', 'r03_pmc_mfma_conv_mix', 'tools/dev/conv_mix.py'), ('wm', 'r03_pmc_mfma_wgrad_mix', 'tools/dev/wgrad_mix.py'),
                           ('cf', 'r03_pmc_fetch_conv_mix', 'tools/dev/conv_mix.py'), ('cw', 'r03_pmc_write_conv_mix', 'tools/dev/conv_mix.py'),
                           ('wf', 'r03_pmc_fetch_wgrad_mix', 'tools/dev/wgrad_mix.py'), ('ww', 'r03_pmc_write_wgrad_mix', 'tools/dev/wgrad_mix.py')):
        txt = open(f'{d}/{name}.md').read().replace(f'command: `{name}`', f'command: `rocprofv3 --pmc <counters below> --kernel-trace -- python3 {cmd}` '
                                                    f'(tools/dev/scripts/pmc_r3.sh)')
        open(f'{out}/{dst}.md', 'w').write(txt)
    # BatchNorm table: fetch + write per kernel against the algorithmic bytes of the mix
    alg = None
    for line in open(f'{d}/bn_mix_bare.log'):
        if line.startswith('BatchNorm mix'):
            alg = float(line.split(' ms, ')[1].split(' GB')[0])
    rows, tf, tw = [], 0.0, 0.0
    for k in sorted(bf):
        if not k.startswith('bn_'):
            continue
        n, f, w = bf[k]['dispatches'], bf[k].get('fetch_mb', 0.0), bw.get(k, {}).get('write_size_mb', 0.0)
        rows.append((k, n, f, w))
        tf += n * f
        tw += n * w
    with open(f'{out}/r03_pmc_bn.md', 'w') as f:
        f.write('# HBM traffic of the BatchNorm passes of one train step (rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE)\n\n')
        f.write('command: `rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 tools/dev/bn_mix.py` and the same with WRITE_SIZE '
                '(separate passes, tools/dev/scripts/pmc_r3.sh). bn_mix.py launches the step\'s BatchNorm passes (B=16, 4-stage '
                'MSPN-50 + FPN) with their per-step counts on rotating buffers. FETCH_SIZE is doubled (gfx950 reports half of a '
                'wide coalesced read, MI355X_MICROARCH.md, HBM section); MB = 1e6 bytes.\n\n')
        f.write('| kernel | dispatches | fetch MB / dispatch | write MB / dispatch | total GB |\n|---|---|---|---|---|\n')
        for k, n, fe, w in rows:
            f.write(f'| `{k}` | {n} | {fe:.1f} | {w:.1f} | {n * (fe + w) / 1e3:.2f} |\n')
        f.write(f'\nsum: fetch {tf / 1e3:.1f} GB + write {tw / 1e3:.1f} GB = **{(tf + tw) / 1e3:.1f} GB**; algorithmic bytes of the same '
                f'launches (every operand of every pass once): **{alg:.1f} GB** -> measured / algorithmic = {(tf + tw) / 1e3 / alg:.2f}\n\n')
        f.write('bare timing of the same mix (no profiler):\n\n```\n' + open(f'{d}/bn_mix_bare.log').read().split('\n', 1)[1] + '```\n')
    print('updated', path)


if __name__ == '__main__':
    main(*sys.argv[1:3])
