"""Dev tool: host time of one das_conv2d_wgrad_batch call (descriptor building in Python + the C launcher), the GPU
running behind asynchronously."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from das_amd import ops
batches = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'wgrad_batches.json')))
b = max(batches, key=len)[:32]
items = []
for o in b:
    k, s, p = o['k'], o['s'], o['p']
    if 'ragged' in o:
        x = ops.Ragged.from_levels([torch.randn(2, h, w, o['Cin'], device='cuda', dtype=torch.bfloat16) for h, w in o['ragged']])
        dy = ops.Ragged.from_levels([torch.randn(2, h, w, o['Cout'], device='cuda', dtype=torch.bfloat16) for h, w in o['ragged']])
    else:
        Ho, Wo = (o['H'] + 2 * p - k) // s + 1, (o['W'] + 2 * p - k) // s + 1
        x = torch.randn(2, o['H'], o['W'], o['Cin'], device='cuda', dtype=torch.bfloat16)
        dy = torch.randn(2, Ho, Wo, o['Cout'], device='cuda', dtype=torch.bfloat16)
    items.append((x, dy, k, k, s, p, torch.zeros(o['Cout'], k, k, o['Cin'], device='cuda')))
for _ in range(3):
    ops.conv2d_wgrad_batch(items)
torch.cuda.synchronize()
n = 50
t0 = time.perf_counter()
for _ in range(n):
    ops.conv2d_wgrad_batch(items)
host = (time.perf_counter() - t0) / n
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / n
print(f'{len(items)} ops per call: host {host * 1e6:.0f} us per call ({host * 1e6 / len(items):.1f} us per op), with GPU {tot * 1e6:.0f} us')
