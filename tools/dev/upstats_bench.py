#!/usr/bin/env python3
"""bilinear_ac_stats_kernel on the step's three up_conv shapes, pixels per workgroup swept (elem.upstats_ppb), against the
plain upsampling kernel. usage: upstats_bench.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from das_amd import _lib, ops
from das_amd.nn import bn_stats_buffer_rows

lib = _lib.load()
dev = 'cuda'
junk = torch.empty(512 << 20, dtype=torch.uint8, device=dev)


def t(fn, n=20):
    out = []
    for _ in range(n):
        junk.zero_()                       # cold operands
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        out.append(a.elapsed_time(b) * 1e3)
    return sorted(out)[n // 2]


for (B, H, W, C) in [(16, 64, 104, 256), (16, 32, 52, 256), (16, 16, 26, 256)]:
    x = torch.randn(B, H, W, C, device=dev).bfloat16()
    Ho, Wo = 2 * H, 2 * W
    rows = B * Ho * Wo
    st = bn_stats_buffer_rows(rows, C, x.device)
    line = ['%dx%d -> %dx%d slots %d: plain %.1f us |' % (H, W, Ho, Wo, st.numel() // (2 * C), t(lambda: ops.upsample_bilinear_ac(x, Ho, Wo)))]
    for ppb in (0, 0, 32, 64, 128, 256, 512, 1024):
        _lib.check(lib.das_tuning_set(b'elem.upstats_ppb', ppb), 'set')
        line.append('%d: %.1f' % (ppb, t(lambda: ops.upsample_bilinear_ac(x, Ho, Wo, stats=st))))
    _lib.check(lib.das_tuning_set(b'elem.upstats_ppb', 0), 'set')
    line.append('| plain again %.1f' % t(lambda: ops.upsample_bilinear_ac(x, Ho, Wo)))
    print(' '.join(line), flush=True)
