"""Dev tool ("untested hypothesis, cheap to test"): do the K = 64 variants of conv1x1_stream_kernel want
more bytes in flight? A/B of the product library (4 LDS stages of 8 KiB, 3 tiles ahead, two workgroups per CU = 48 KiB in
flight per CU) against a dev build with 8 stages / 7 ahead (`make -C das_amd/csrc deep`), cold operands, the step's K = 64
shapes and modes. usage: stream_depth_ab.py [deep]   (run twice: each process loads one library)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from das_amd import _lib
deep = len(sys.argv) > 1 and sys.argv[1] == 'deep'
if deep:
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libdas_hip_deep.so')
from das_amd import ops
BF = torch.bfloat16
B = 16
for (H, W, Cin, Cout) in [(128, 208, 64, 256), (128, 208, 64, 64), (64, 104, 64, 256)]:
    nb = 5
    xs = [torch.randn(B, H, W, Cin, device='cuda', dtype=BF) for _ in range(nb)]
    ys = [torch.empty(B, H, W, Cout, device='cuda', dtype=BF) for _ in range(nb)]
    rs = [torch.randn(B, H, W, Cout, device='cuda', dtype=BF) for _ in range(3)]
    w = (torch.randn(Cout, 1, 1, Cin, device='cuda') / Cin ** 0.5).to(BF)
    st = torch.zeros(8 * 2 * Cout, device='cuda')
    mean, invstd = torch.zeros(Cout, device='cuda'), torch.ones(Cout, device='cuda')
    gamma, beta = torch.ones(Cout, device='cuda'), torch.zeros(Cout, device='cuda')
    row = f'{"deep  " if deep else "4 / 3 "} {H}x{W} {Cin}->{Cout}: '
    for mode, nten in (('stats', Cin + Cout), ('res', Cin + 2 * Cout), ('res_bnb_y', Cin + 4 * Cout)):
        def call(i):
            kw = dict(stats=st) if mode == 'stats' else dict(residual=rs[i % 3]) if mode == 'res' else \
                dict(stats=st, residual=rs[i % 3], bn_bwd=ops.BnBwd(rs[(i + 1) % 3], rs[(i + 2) % 3], mean, invstd, gamma, beta, True))
            ops.conv2d(xs[i % nb], w, 1, 1, 1, 0, out=ys[i % nb], **kw)
        for i in range(nb):
            call(i)
        n = 6 * nb
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            call(i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        row += f'{mode} {us:6.1f} us {B * H * W * nten * 2 / us / 1e6:5.2f} TB/s ({ops.last_kernel()[:14]})   '
    print(row, flush=True)
