"""Build profiles/traffic.json (HBM bytes per launch of the conv kernel families, for bench.py's roofline.traffic)
from the per-pass PMC summaries that tools/dev/scripts/pmc_wgrad.sh and pmc_conv.sh leave in gpurun_out/pmc/:
  wf / ww   : FETCH_SIZE / WRITE_SIZE of tools/dev/wgrad_mix.py (the weight-gradient launches of one train step)
  cf / cw   : FETCH_SIZE / WRITE_SIZE of tools/dev/conv_mix.py (the conv_glds4_kernel launches of one train step)
  inf / infw: FETCH_SIZE / WRITE_SIZE of `bench.py --workload infer` (the forward conv kernels)
usage: python tools/dev/make_traffic_json.py gpurun_out/pmc profiles/traffic.json"""
import json
import sys

PASSES = 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only)'


def load(d, name):
    return json.load(open(f'{d}/{name}.json'))['kernels']


def family(fetch, write, match):
    ks = [k for k in fetch if match(k)]
    n = sum(fetch[k]['dispatches'] for k in ks)
    if not n:
        return None
    f = sum(fetch[k]['dispatches'] * fetch[k].get('fetch_mb', 0) for k in ks) / n
    w = sum(write[k]['dispatches'] * write[k].get('write_size_mb', 0) for k in ks if k in write) / n
    return dict(fetch_mb=round(f, 2), write_mb=round(w, 2), hbm_mb_per_launch=round(f + w, 2), dispatches=n)


WGRAD_OPS = 279 * 2     # tools/dev/wgrad_mix.py: 279 ops per replay of the step's batches, warm-up + timed replay


def main(d, out):
    wf, ww, cf, cw, inf, infw = (load(d, n) for n in ('wf', 'ww', 'cf', 'cw', 'inf', 'infw'))
    fam = {}
    # weight gradient: grouped launches (several ops per dispatch) of conv_wgrad_pp_kernel (256 x 256 ping-pong) and
    # conv_wgrad_kernel (128 x 128) + one wgrad_reduce_kernel per group: all their bytes / the number of ops replayed
    f = sum(v['dispatches'] * v.get('fetch_mb', 0) for k, v in wf.items() if 'wgrad' in k) / WGRAD_OPS
    w = sum(v['dispatches'] * v.get('write_size_mb', 0) for k, v in ww.items() if 'wgrad' in k) / WGRAD_OPS
    fam['conv_wgrad_kernel<bf16>'] = dict(train=dict(
        fetch_mb=round(f, 2), write_mb=round(w, 2), hbm_mb_per_launch=round(f + w, 2), ops=WGRAD_OPS,
        source=PASSES + ' over tools/dev/wgrad_mix.py: the 279 weight-gradient ops of one train step (B=16) replayed in '
                        'the 13 launches (<= 32 deferred ops each) backward issues them in; conv_wgrad_pp_kernel + conv_wgrad_kernel + '
                        'wgrad_reduce_kernel bytes per op (algorithmic 123.66 MB per op)'))
    src_mix = PASSES + (' over tools/dev/conv_mix.py: the conv shapes (forward + data-gradient) of the train step with '
                        'their per-step counts, B=16')
    src_inf = PASSES + ' over `bench.py --workload infer --steps 3 --warmup 1` (B=8, forward launches)'
    # bench.py tags a launch with das_last_kernel(): map each tag to the rocprof kernel names of that family
    tags = (('conv_glds4_kernel<pp,288>', lambda k: k.startswith('conv_glds4_kernel<') and k.endswith('true, 288>')),
            ('conv_glds4_kernel<pp>', lambda k: k.startswith('conv_glds4_kernel<') and k.endswith('true, 256>')),
            ('conv_glds4_kernel', lambda k: k.startswith('conv_glds4_kernel<') and k.endswith('false, 256>')),
            ('conv_glds3_kernel', lambda k: k.startswith('conv_glds3_kernel<')),
            ('conv_glds_kernel', lambda k: k.startswith('conv_glds_kernel<')),
            ('conv1x1_stream_kernel', lambda k: k.startswith('conv1x1_stream_kernel<')),
            ('conv_reg_kernel', lambda k: k.startswith('conv_reg_kernel<')))
    # (the <splitk> tags — main kernel over blockIdx.y + splitk_finish_kernel — share the unsplit kernels' names in the
    # trace, so they get no entry of their own: roofline.traffic stays null when one of them is the dominant family)
    for tag, match in tags:
        for wl, (fe, wr, src) in (('train', (cf, cw, src_mix)), ('infer', (inf, infw, src_inf))):
            e = family(fe, wr, match)
            if e:
                e['source'] = src
                fam.setdefault(tag, {})[wl] = e
    json.dump(dict(note='HBM MB per launch: FETCH_SIZE KiB x 2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE KiB, '
                        'means over the dispatches of the named run',
                   families=fam), open(out, 'w'), indent=1, sort_keys=True)
    for k, v in fam.items():
        print(k, {w: e['hbm_mb_per_launch'] for w, e in v.items()})


if __name__ == '__main__':
    main(*sys.argv[1:3])
