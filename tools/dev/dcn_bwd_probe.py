#!/usr/bin/env python3
"""das_deform_im2col3x3_backward on the step's head geometry (B = 16, four levels, 256 channels): offsets zero (the
initial state), random offsets of +-0.6 / +-1.5 px, and all masks shut (no hits: the candidate search alone)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from das_amd import ops

dev = 'cuda'
B, C = 16, 256
sizes = [(64, 104), (32, 52), (16, 26), (8, 13)]
x = ops.Ragged.from_levels([torch.randn(B, h, w, C, device=dev).bfloat16() for h, w in sizes])
rows = x.rows
dcol = x.new(9 * C, torch.bfloat16)
dcol.data.copy_(torch.randn(rows, 9 * C, device=dev).bfloat16())
junk = torch.empty(512 << 20, dtype=torch.uint8, device=dev)


def t(fn, n=8):
    out = []
    for _ in range(n):
        junk.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        out.append(a.elapsed_time(b) * 1e3)
    return sorted(out)[n // 2]


def om_of(scale, logit):
    om = x.new(32, torch.float32)
    om.data.zero_()
    om.data[:, :18] = (torch.rand(rows, 18, device=dev) * 2 - 1) * scale
    om.data[:, 18:27] = logit
    return om


for name, scale, logit in (('offsets 0', 0.0, 0.0), ('offsets +-0.6', 0.6, 0.0), ('offsets +-1.5', 1.5, 0.0),
                           ('offsets +-3', 3.0, 0.0), ('masks shut (no hits)', 0.6, -200.0)):
    om = om_of(scale, logit)
    print('%-24s %8.1f us' % (name, t(lambda: ops.deform_im2col3x3_backward(x, om, dcol))), flush=True)
