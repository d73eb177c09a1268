"""Dev tool: conv1x1_stream_kernel on the train step's shapes and modes against the HBM floor of each call (bytes of
every operand the mode touches / 6.2 TB/s), operands rotated through a pool larger than the caches."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from das_amd import ops
B = 16
SHAPES = [(128, 208, 256, 256), (128, 208, 64, 256), (128, 208, 256, 64), (64, 104, 128, 512), (64, 104, 512, 128),
          (32, 52, 256, 1024), (64, 104, 256, 512)]
POOL = 4


def bench(fn, n=POOL * 3):
    for i in range(POOL):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i % POOL)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (H, W, Cin, Cout) in SHAPES:
    M = B * H * W
    xs = [torch.randn(B, H, W, Cin, device='cuda', dtype=torch.bfloat16) for _ in range(POOL)]
    ys = [torch.empty(B, H, W, Cout, device='cuda', dtype=torch.bfloat16) for _ in range(POOL)]
    rs = [torch.randn(B, H, W, Cout, device='cuda', dtype=torch.bfloat16) for _ in range(POOL)]
    raws = [torch.randn(B, H, W, Cout, device='cuda', dtype=torch.bfloat16) for _ in range(POOL)]
    w = (torch.randn(Cout, 1, 1, Cin, device='cuda') / Cin ** 0.5).to(torch.bfloat16)
    stats = torch.zeros(16 * 2 * Cout, device='cuda')
    mean, invstd = torch.zeros(Cout, device='cuda'), torch.ones(Cout, device='cuda')
    gamma, beta = torch.ones(Cout, device='cuda'), torch.zeros(Cout, device='cuda')
    row = f'{H}x{W} {Cin}->{Cout}: '
    modes = [('fwd+stats', lambda i: ops.conv2d(xs[i], w, 1, 1, 1, 0, stats=stats, out=ys[i]), Cin + Cout),
             ('dgrad+res', lambda i: ops.conv2d(xs[i], w, 1, 1, 1, 0, residual=rs[i], out=ys[i]), Cin + 2 * Cout),
             ('dgrad+bnb(raw)', lambda i: ops.conv2d(xs[i], w, 1, 1, 1, 0, out=ys[i], stats=stats,
                                                     bn_bwd=ops.BnBwd(raws[i], None, mean, invstd, gamma, beta, True)), Cin + 2 * Cout),
             ('dgrad+res+bnb(raw,y)', lambda i: ops.conv2d(xs[i], w, 1, 1, 1, 0, residual=rs[i], out=ys[i], stats=stats,
                                                           bn_bwd=ops.BnBwd(raws[i], rs[(i + 1) % POOL], mean, invstd, gamma, beta, True)), Cin + 4 * Cout)]
    for name, fn, ch in modes:
        us = bench(fn)
        floor = M * ch * 2 / 6.2e6
        row += f'{name} {us:6.1f} us x{us / floor:4.2f} ({ops.last_kernel().replace("conv", "").replace("_kernel", "")[:12]})  '
    print(row)
    del xs, ys, rs, raws
