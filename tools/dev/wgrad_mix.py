"""Dev tool: the weight gradients of one training bench step (B=16), replayed batch by batch as backward hands them to
das_conv2d_wgrad_batch (tools/dev/wgrad_batches.json, recorded by tools/dev/dump_wgrad_batches.py: 13 batches of <= 32 deferred ops, 279 ops;
the DCNv2 GEMM weight gradients go out alone) on random operands. Run under `rocprofv3 --pmc ...` (one counter set per
pass) for the HBM traffic / MFMA counters of conv_wgrad_kernel + conv_wgrad_pp_kernel + wgrad_reduce_kernel per op
behind bench.py's roofline.traffic, or bare for a time per step."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from das_amd import ops

batches = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'wgrad_batches.json')))
args = [a for a in sys.argv[1:]]
for a in args:
    if a.startswith('batch='):       # re-chunk the op sequence (A/B of the batch size)
        nb = int(a[6:])
        flat = [o for b in batches for o in b]
        batches = [flat[i:i + nb] for i in range(0, len(flat), nb)]
    else:                            # key=value -> das_tuning_set
        from das_amd import _lib
        k, v = a.split('=')
        _lib.check(_lib.load().das_tuning_set(k.encode(), int(v)), k)
pool = {}


def operands(o, slot):
    """x, dy for one op: one pair of buffers per (shape, position in the batch), reused across batches"""
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


def run():
    fl, by, n = 0.0, 0.0, 0
    for b in batches:
        items = []
        for i, o in enumerate(b):
            x, dy, out, rows = operands(o, i)
            items.append((x, dy, o['k'], o['k'], o['s'], o['p'], out))
            fl += 2.0 * rows * o['Cout'] * o['k'] * o['k'] * o['Cin']
            by += (ops._data(x).numel() + ops._data(dy).numel()) * 2 + out.numel() * 4
            n += 1
        ops.conv2d_wgrad_batch(items)
    return fl, by, n


run()                                        # warm-up: operand pool, workspace
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
fl, by, n = run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
print(f'{n} ops in {len(batches)} batches, {ms:.3f} ms, {fl / ms / 1e9:.1f} TF, algorithmic bytes per op {by / n / 1e6:.2f} MB '
      f'(x + dy read once, dW written once); ops_per_run {n} runs 2')
