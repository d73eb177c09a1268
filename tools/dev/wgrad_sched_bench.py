"""Dev tool: the weight gradients of one training step (tools/dev/wgrad_batches.json) re-chunked into launches of
batch=N ops, timed launch by launch (HIP events, operands rotate through a pool so that they are cold), against the
floor of each launch: max(FLOP / 2.5 PF x 1 / 0.32 [the kernels' measured main-loop rate, 810 TF], bytes / 6.2 TB/s)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from das_amd import ops, _lib

batches = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'wgrad_batches.json')))
nb, only = 32, None
for a in sys.argv[1:]:
    k, v = a.split('=')
    if k == 'batch':
        nb = int(v)
    elif k == 'cls':
        only = int(v)
    else:
        _lib.check(_lib.load().das_tuning_set(k.encode(), int(v)), k)
flat = [o for b in batches for o in b]


def cls_of(o):
    return 0 if (o['k'] * o['k'] * o['Cin'] >= 256 and o['Cout'] >= 256) else 1


if only is not None:
    flat = [o for o in flat if cls_of(o) == only]
batches = [flat[i:i + nb] for i in range(0, len(flat), nb)]
pool = {}


def operands(o, slot):
    key = (json.dumps(o, sort_keys=True), slot)
    if key not in pool:
        k, s, p = o['k'], o['s'], o['p']
        if 'ragged' in o:
            x = ops.Ragged.from_levels([torch.randn(o['B'], h, w, o['Cin'], device='cuda', dtype=torch.bfloat16) for h, w in o['ragged']])
            dy = ops.Ragged.from_levels([torch.randn(o['B'], h, w, o['Cout'], device='cuda', dtype=torch.bfloat16) for h, w in o['ragged']])
            rows = x.rows
        else:
            Ho, Wo = (o['H'] + 2 * p - k) // s + 1, (o['W'] + 2 * p - k) // s + 1
            x = torch.randn(o['B'], o['H'], o['W'], o['Cin'], device='cuda', dtype=torch.bfloat16)
            dy = torch.randn(o['B'], Ho, Wo, o['Cout'], device='cuda', dtype=torch.bfloat16)
            rows = o['B'] * Ho * Wo
        out = torch.zeros(o['Cout'], k, k, o['Cin'], device='cuda')
        pool[key] = (x, dy, out, rows)
    return pool[key]


def run(timed):
    evs = []
    for b in batches:
        items, fl, by = [], [0.0, 0.0], 0.0
        for i, o in enumerate(b):
            x, dy, out, rows = operands(o, i)
            items.append((x, dy, o['k'], o['k'], o['s'], o['p'], out))
            fl[cls_of(o)] += 2.0 * rows * o['Cout'] * o['k'] * o['k'] * o['Cin']
            by += (ops._data(x).numel() + ops._data(dy).numel()) * 2 + out.numel() * 4
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.conv2d_wgrad_batch(items)
        e1.record()
        evs.append((e0, e1, fl, by, len(b)))
    torch.cuda.synchronize()
    return evs


run(False)
run(False)
evs = run(True)
tot = tfl = tfloor = 0.0
for i, (e0, e1, fl, by, n) in enumerate(evs):
    us = e0.elapsed_time(e1) * 1e3
    floor = max(sum(fl) / 810e6, by / 6.2e6)
    tot += us; tfl += sum(fl); tfloor += floor
    print(f'launch {i:3d} n={n:2d} pp {fl[0] / 1e9:8.1f} GF plain {fl[1] / 1e9:8.1f} GF {by / 1e6:8.1f} MB  {us:8.1f} us  {sum(fl) / us / 1e6:6.1f} TF  floor {floor:7.1f} us x{us / floor:4.2f}')
print(f'total {tot / 1e3:.3f} ms, {tfl / tot / 1e6:.1f} TF, floor {tfloor / 1e3:.3f} ms')
