"""Dev tool: every BatchNorm pass alone at the step's large shapes (rotating buffers larger than the caches): us per
call and TB/s of algorithmic bytes, next to the 5.9 TB/s a plain 2R1W / 3R1W pass reaches (tools/dev/probe/triad_probe.hip)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from das_amd import ops
for a in sys.argv[1:]:
    from das_amd import _lib
    k, v = a.split('=')
    _lib.check(_lib.load().das_tuning_set(k.encode(), int(v)), k)
BF = torch.bfloat16
for rows, C in ((425984, 256), (106496, 512), (26624, 1024), (425984, 64)):
    shape = (16, rows // 16, 1, C)
    NB = 4
    xs = [torch.randn(shape, device='cuda').to(BF) for _ in range(NB)]
    rs = [torch.randn(shape, device='cuda').to(BF) for _ in range(NB)]
    ys = [torch.randn(shape, device='cuda').to(BF) for _ in range(NB)]
    g, b = torch.ones(C, device='cuda'), torch.zeros(C, device='cuda')
    rm, rv = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
    st = torch.rand(16 * 2 * C, device='cuda') * rows / 16
    mean, invstd = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
    sums = torch.rand(16 * 2 * C, device='cuda')
    nb = rows * C * 2
    cases = [
        ('apply', 2, lambda i: ops.bn_train_apply(xs[i % NB], st, g, b, rm, rv, relu=True)),
        ('apply + residual', 3, lambda i: ops.bn_train_apply(xs[i % NB], st, g, b, rm, rv, residual=rs[i % NB], relu=True)),
        ('apply_dz', 3, lambda i: ops.bn_backward_apply(xs[i % NB], rs[i % NB], mean, invstd, g, sums)),
        ('backward, recomputed mask (reduce + apply)', 5, lambda i: ops.bn_train_backward(xs[i % NB], None, rs[i % NB], mean, invstd, g, True, False, beta=b)),
        ('backward, y mask + d residual (reduce + apply)', 8, lambda i: ops.bn_train_backward(xs[i % NB], ys[i % NB], rs[i % NB], mean, invstd, g, True, True, beta=b)),
    ]
    for name, passes, fn in cases:
        for i in range(3):
            fn(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 12
        for i in range(n):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        print(f'rows={rows:7d} C={C:5d}  {name:48s} {us:8.1f} us  {passes * nb / us / 1e6:5.2f} TB/s', flush=True)
    del xs, rs, ys
