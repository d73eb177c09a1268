#!/usr/bin/env python3
"""bilinear_ac_bwd5_kernel: pixels per workgroup swept (elem.upstats_ppb)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from das_amd import _lib, ops
lib = _lib.load()
dev = 'cuda'
junk = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
def t(fn, n=12, cold=True):
    out = []
    for _ in range(n):
        if cold:
            junk.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        out.append(a.elapsed_time(b) * 1e3)
    return sorted(out)[n // 2]
for (B, H, W, C) in [(16, 64, 104, 256), (16, 32, 52, 256)]:
    dy = torch.randn(B, 2 * H, 2 * W, C, device=dev).bfloat16()
    line = ['%dx%d:' % (2 * H, 2 * W)]
    for ppb in (0, 16, 32, 64, 0):
        _lib.check(lib.das_tuning_set(b'elem.upstats_ppb', ppb), 'set')
        line.append('%d/%dKB: %.1f (warm %.1f)' % (ppb & 0xffff, ppb >> 16, t(lambda: ops.upsample_bilinear_ac_backward(dy, H, W)), t(lambda: ops.upsample_bilinear_ac_backward(dy, H, W), cold=False)))
    _lib.check(lib.das_tuning_set(b'elem.upstats_ppb', 0), 'set')
    print(' '.join(line), flush=True)
