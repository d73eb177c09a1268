"""Dev tool: every distinct weight-gradient shape of the train step (tools/dev/wgrad_batches.json) timed alone (cold-ish:
the operand pool is larger than the caches), with its count per step, TFLOP/s and HBM floor."""
import json, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from das_amd import ops

for a in sys.argv[1:]:                   # key=value -> das_tuning_set
    from das_amd import _lib
    k_, v_ = a.split('=')
    _lib.check(_lib.load().das_tuning_set(k_.encode(), int(v_)), k_)
batches = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'wgrad_batches.json')))
cnt = collections.Counter(json.dumps(o, sort_keys=True) for b in batches for o in b)
rows = []
for key, n in cnt.items():
    o = json.loads(key)
    k, s, p = o['k'], o['s'], o['p']
    if 'ragged' in o:
        x = ops.Ragged.from_levels([torch.randn(o['B'], h, w, o['Cin'], device='cuda', dtype=torch.bfloat16) for h, w in o['ragged']])
        dy = ops.Ragged.from_levels([torch.randn(o['B'], h, w, o['Cout'], device='cuda', dtype=torch.bfloat16) for h, w in o['ragged']])
        M = x.rows
        desc = f"ragged {o['Cin']}->{o['Cout']} k{k}"
    else:
        Ho, Wo = (o['H'] + 2 * p - k) // s + 1, (o['W'] + 2 * p - k) // s + 1
        x = torch.randn(o['B'], o['H'], o['W'], o['Cin'], device='cuda', dtype=torch.bfloat16)
        dy = torch.randn(o['B'], Ho, Wo, o['Cout'], device='cuda', dtype=torch.bfloat16)
        M = o['B'] * Ho * Wo
        desc = f"{o['H']}x{o['W']} {o['Cin']}->{o['Cout']} k{k} s{s}"
    out = torch.zeros(o['Cout'], k, k, o['Cin'], device='cuda')
    item = [(x, dy, k, k, s, p, out)]
    for _ in range(2):
        ops.conv2d_wgrad_batch(item)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.conv2d_wgrad_batch(item)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    fl = 2.0 * M * o['Cout'] * k * k * o['Cin']
    by = (ops._data(x).numel() + ops._data(dy).numel()) * 2 + out.numel() * 4
    rows.append((n * us, n, us, fl / us / 1e6, by / 6.2e6, desc, ops.last_kernel()))
    del x, dy, out
tot = sum(r[0] for r in rows)
print(f'sum over the step {tot / 1e3:.2f} ms (ops alone, warm)')
for t, n, us, tf, floor, desc, kern in sorted(rows, reverse=True):
    print(f'{t / tot * 100:5.1f}% n={n:3d} {us:7.1f} us {tf:6.0f} TF  hbm floor {floor:6.1f} us  x{us / floor:4.1f}  {desc:34s} {kern}')
