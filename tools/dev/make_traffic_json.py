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


def main(d, out):
    wf, ww, cf, cw, inf, infw = (load(d, n) for n in ('wf', 'ww', 'cf', 'cw', 'inf', 'infw'))
    fam = {}
    # weight gradient = main kernel (128 x 128 or ping-pong 256 x 256) + split reduction per op: all bytes of the
    # weight-gradient kernels divided by the number of ops (= dispatches of the reduce kernels)
    nops = sum(v['dispatches'] for k, v in wf.items() if 'wgrad_reduce' in k)
    f = sum(v['dispatches'] * v.get('fetch_mb', 0) for k, v in wf.items() if 'wgrad' in k) / nops
    w = sum(v['dispatches'] * v.get('write_size_mb', 0) for k, v in ww.items() if 'wgrad' in k) / nops
    fam['conv_wgrad_kernel<bf16>'] = dict(train=dict(
        fetch_mb=round(f, 2), write_mb=round(w, 2), hbm_mb_per_launch=round(f + w, 2),
        source=PASSES + ' over tools/dev/wgrad_mix.py: the 30 most expensive weight-gradient shapes of the train '
                        'step with their per-step counts, B=16; conv_wgrad_kernel + wgrad_reduce_kernel per op'))
    e = family(cf, cw, lambda k: k.startswith('conv_glds4_kernel<bf16, bf16'))
    e['source'] = PASSES + (' over tools/dev/conv_mix.py: the conv_glds4_kernel shapes (forward + data-gradient) of '
                            'the train step with their per-step counts, B=16')
    fam['conv_glds4_kernel<bf16, bf16>'] = dict(train=e)
    src_inf = PASSES + ' over `bench.py --workload infer --steps 3 --warmup 1` (B=8, forward launches)'
    for prefix, tag in (('conv_glds4_kernel<bf16, bf16', 'conv_glds4_kernel<bf16, bf16>'),
                        ('conv_glds3_kernel<bf16, bf16', 'conv_glds3_kernel<bf16, bf16>'),
                        ('conv_glds_kernel<bf16, bf16, 128, 128>', 'conv_glds_kernel<bf16, bf16, 128, 128>'),
                        ('conv_glds_kernel<bf16, bf16, 64, 128>', 'conv_glds_kernel<bf16, bf16, 64, 128>')):
        e = family(inf, infw, lambda k, p=prefix: k.startswith(p))
        if e:
            e['source'] = src_inf
            fam.setdefault(tag, {})['infer'] = e
    json.dump(dict(note='HBM MB per launch: FETCH_SIZE KiB x 2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE KiB, '
                        'means over the dispatches of the named run',
                   families=fam), open(out, 'w'), indent=1, sort_keys=True)
    for k, v in fam.items():
        print(k, {w: e['hbm_mb_per_launch'] for w, e in v.items()})


if __name__ == '__main__':
    main(*sys.argv[1:3])
