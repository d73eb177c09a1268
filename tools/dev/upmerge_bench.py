#!/usr/bin/env python3
"""The kernels of the fused upsample-unit merge (csrc/upmerge.hip) on the step's three shapes, cold operands.
usage: upmerge_bench.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from das_amd import _lib, ops
from das_amd.nn import bn_stats_buffer_rows

lib = _lib.load()
dev = 'cuda'
junk = torch.empty(512 << 20, dtype=torch.uint8, device=dev)


def t(fn, n=12):
    out = []
    for _ in range(n):
        junk.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        out.append(a.elapsed_time(b) * 1e3)
    return sorted(out)[n // 2]


def setk(k, v):
    _lib.check(lib.das_tuning_set(k.encode(), v), k)


for (B, H, W, C) in [(16, 64, 104, 256), (16, 32, 52, 256), (16, 16, 26, 256)]:
    Ho, Wo = 2 * H, 2 * W
    z = torch.randn(B, H, W, C, device=dev).bfloat16()
    raw1 = torch.randn(B, Ho, Wo, C, device=dev).bfloat16()
    dy = torch.randn_like(raw1)
    f = lambda: torch.rand(C, device=dev) + 0.5
    bn1, bn2 = (f() - 1, f(), f(), f() - 1), (f() - 1, f(), f(), f() - 1)
    out = ops.upmerge_forward(raw1, z, bn1, bn2)
    mb = raw1.numel() * 2 / 1e6
    line = ['%dx%d (%.0f MB): fwd %.1f us' % (Ho, Wo, mb, t(lambda: ops.upmerge_forward(raw1, z, bn1, bn2)))]
    st = bn_stats_buffer_rows(B * Ho * Wo, C, z.device)
    line.append('stats-only %.1f' % t(lambda: ops.upsample_bilinear_ac(z, Ho, Wo, stats=st, stats_only=True)))
    line.append('stats-lowres %.1f' % t(lambda: ops.upsample_stats_lowres(z, Ho, Wo, st)))
    for blocks in (256, 512, 1024):
        setk('bn.upmerge_blocks', blocks)
        line.append('reduce[%d] %.1f' % (blocks, t(lambda: ops.upmerge_backward_reduce(dy, out, raw1, z, bn1[0], bn1[1], bn2[0], bn2[1]))))
    setk('bn.upmerge_blocks', 512)
    dzm, sums = ops.upmerge_backward_reduce(dy, out, raw1, z, bn1[0], bn1[1], bn2[0], bn2[1])
    P = ops.upsample_bilinear_ac_backward(dzm, H, W)
    line.append('| up^T %.1f' % t(lambda: ops.upsample_bilinear_ac_backward(dzm, H, W)))
    line.append('lowres %.1f' % t(lambda: ops.upmerge_backward_lowres(P, z, Ho, Wo, sums, bn2[2], bn2[0], bn2[1], B * Ho * Wo)))
    line.append('bn1 apply %.1f' % t(lambda: ops.bn_backward_apply(dzm, raw1, bn1[0], bn1[1], bn1[2], sums[:2 * C])))
    print(' '.join(line), flush=True)
    # the plain reduce pass of the same size, block caps
    for blocks in (256,):
        setk('bn.reduce_blocks', blocks)
        print('    bn_bwd_reduce+apply (mask from y) blocks %d: %.1f us' % (blocks, t(lambda: ops.bn_train_backward(
            dy, out, raw1, bn1[0], bn1[1], bn1[2], True, True, beta=bn1[3]))), flush=True)
    setk('bn.reduce_blocks', 256)
