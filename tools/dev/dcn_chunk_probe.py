#!/usr/bin/env python3
"""DCNv2 backward (dcol GEMM + col2im gather + offset-gradient kernel) on the step's head geometry: one launch each over all
rows against the same work image-group by image-group through a small ring buffer for dcol (the 652 MB dcol tensor then never
leaves the 256 MiB cache: written, read twice, overwritten)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from das_amd import ops

dev = 'cuda'
B, C, O = 16, 256, 256
sizes = [(64, 104), (32, 52), (16, 26), (8, 13)]
x = ops.Ragged.from_levels([torch.randn(B, h, w, C, device=dev).bfloat16() for h, w in sizes])
rows = x.rows
om = x.new(32, torch.float32)
om.data.zero_()
dy = torch.randn(rows, O, device=dev).bfloat16()
wt = (torch.randn(9 * C, 1, 1, O, device=dev) / 48).bfloat16()
junk = torch.empty(600 << 20, dtype=torch.uint8, device=dev)
TARGET = int(sys.argv[1]) if len(sys.argv) > 1 else 6656


def t(fn, n=6):
    out = []
    for _ in range(n):
        for _ in range(25):       # ~4 ms of queued fills: the host queues fn's launches while the GPU is still busy with them
            junk.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        out.append(a.elapsed_time(b) * 1e3)
    return sorted(out)[n // 2]


def whole():
    dcol = ops.conv2d(ops.Ragged(dy, B, sizes), wt, 1, 1)
    return ops.deform_im2col3x3_backward(x, om, dcol)


chunks = []
r = 0
for (h, w) in sizes:
    per = max(1, min(B, TARGET // (h * w)))
    for b0 in range(0, B, per):
        nb = min(per, B - b0)
        chunks.append((r + b0 * h * w, nb, h, w))
    r += B * h * w
ring = [torch.empty(max(nb * h * w for _, nb, h, w in chunks), 9 * C, device=dev, dtype=torch.bfloat16) for _ in range(2)]


def chunked():
    for i, (r0, nb, h, w) in enumerate(chunks):
        n = nb * h * w
        dc = ring[i & 1][:n]
        ops.conv2d(dy[r0:r0 + n].view(1, 1, n, O), wt, 1, 1, out=dc.view(1, 1, n, 9 * C))
        ops.deform_im2col3x3_backward(x.data[r0:r0 + n].view(nb, h, w, C), om.data[r0:r0 + n].view(nb, h, w, 32), dc.view(nb, h, w, 9 * C))


print('%d chunks (target %d rows)' % (len(chunks), TARGET))
for rep in range(2):
    print('whole %.1f us   chunked %.1f us' % (t(whole), t(chunked)), flush=True)
