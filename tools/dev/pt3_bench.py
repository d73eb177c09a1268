"""Dev tool: the persistent 256 x 128 tile grid (conv_pt3_kernel) against the one-tile-per-workgroup kernels it replaces,
on the train step's multi-round shapes and epilogue modes, COLD operands (buffer sets rotated past the 256 MiB cache).
usage: pt3_bench.py [key=value ...]   (das_tuning_set before the run)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from das_amd import ops, _lib
for a in sys.argv[1:]:
    k, v = a.split('=')
    _lib.check(_lib.load().das_tuning_set(k.encode(), int(v)), k)
BF = torch.bfloat16
B = 16
SHAPES = [  # H, W, Cin, Cout, k
    (64, 104, 128, 128, 3), (64, 104, 512, 128, 1), (64, 104, 256, 128, 1), (128, 208, 128, 128, 1),
    (64, 104, 1024, 128, 1), (128, 208, 256, 128, 3),
]
torch.manual_seed(0)
for (H, W, Cin, Cout, k) in SHAPES:
    by = B * H * W * (Cin + Cout) * 2
    nb = max(2, int(600e6 // by) + 1)
    xs = [torch.randn(B, H, W, Cin, device='cuda', dtype=BF) for _ in range(nb)]
    ys = [torch.empty(B, H, W, Cout, device='cuda', dtype=BF) for _ in range(nb)]
    rs = [torch.randn(B, H, W, Cout, device='cuda', dtype=BF) for _ in range(min(nb, 3))]
    raws = [torch.randn(B, H, W, Cout, device='cuda', dtype=BF) for _ in range(min(nb, 3))]
    w = (torch.randn(Cout, k, k, Cin, device='cuda') / (Cin * k * k) ** 0.5).to(BF)
    st = torch.zeros(8 * 2 * Cout, device='cuda')
    mean, invstd = torch.zeros(Cout, device='cuda'), torch.ones(Cout, device='cuda')
    gamma, beta = torch.ones(Cout, device='cuda'), torch.zeros(Cout, device='cuda')
    row = f'{H}x{W} {Cin}->{Cout} k{k}: '
    fl = 2.0 * B * H * W * Cin * Cout * k * k
    for mode in ('plain', 'stats', 'res', 'bnb_x', 'res_bnb_y'):
        def call(i):
            kw = {}
            if mode == 'stats':
                kw = dict(stats=st)
            elif mode == 'res':
                kw = dict(residual=rs[i % len(rs)])
            elif mode == 'bnb_x':
                kw = dict(stats=st, bn_bwd=ops.BnBwd(raws[i % len(raws)], None, mean, invstd, gamma, beta, True))
            elif mode == 'res_bnb_y':
                kw = dict(stats=st, residual=rs[i % len(rs)],
                          bn_bwd=ops.BnBwd(raws[i % len(raws)], rs[(i + 1) % len(rs)], mean, invstd, gamma, beta, True))
            ops.conv2d(xs[i % nb], w, k, k, 1, k // 2, out=ys[i % nb], **kw)
        res = []
        for mint in (0, 257):
            with ops.tuning(**{'conv.pt3_mintiles': mint}):
                for i in range(nb):
                    call(i)
                n = 3 * nb
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(n):
                    call(i)
                e1.record()
                torch.cuda.synchronize()
                res.append((e0.elapsed_time(e1) / n * 1e3, ops.last_kernel().replace('conv_', '').replace('_kernel', '')))
        row += f'{mode} {res[0][0]:6.1f} ({res[0][1]}) -> {res[1][0]:6.1f} us ({res[1][1]}, {fl / res[1][0] / 1e6:5.0f} TF)  '
    print(row, flush=True)
    del xs, ys, rs, raws
