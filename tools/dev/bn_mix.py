"""Dev tool: the BatchNorm passes of one training bench step (B=16; shapes and per-step counts from tools/dev/bn_shapes.py),
each launched `count` times on rotating buffers. Run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate
passes, --kernel-trace only) for the HBM traffic per launch next to the algorithmic bytes, or bare for the times."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from das_amd import ops

BF = torch.bfloat16
for a in sys.argv[1:]:                   # key=value -> das_tuning_set (A/B runs, e.g. bn.stream_minbytes=0)
    from das_amd import _lib
    k, v = a.split('=')
    _lib.check(_lib.load().das_tuning_set(k.encode(), int(v)), k)
# (rows, C): apply no-residual, apply + residual, apply_dz, classic backward (recomputed mask), classic backward (y mask + d residual)
SHAPES = [
    ((425984, 256), 14, 16, 8, 13, 7), ((425984, 64), 27, 0, 24, 3, 0), ((106496, 512), 10, 16, 12, 10, 4),
    ((106496, 128), 28, 0, 28, 0, 0), ((106496, 256), 10, 4, 4, 6, 4), ((26624, 1024), 10, 24, 20, 10, 4),
    ((26624, 256), 50, 4, 44, 6, 4), ((6656, 2048), 10, 12, 8, 10, 4), ((6656, 512), 20, 0, 20, 0, 0),
]
NBUF = 3
tot_ms, tot_b = 0.0, 0.0
for (rows, C), n_a, n_ar, n_dz, n_c2, n_c3 in SHAPES:
    H = 16
    shape = (H, rows // H, 1, C)
    xs = [torch.randn(shape, device='cuda').to(BF) for _ in range(NBUF)]
    rs = [torch.randn(shape, device='cuda').to(BF) for _ in range(NBUF)]
    g, b = torch.ones(C, device='cuda'), torch.zeros(C, device='cuda')
    rm, rv = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
    st = torch.rand(8 * 2 * C, device='cuda') * rows / 8
    mean, invstd = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
    sums = torch.rand(8 * 2 * C, device='cuda')
    nb = rows * C * 2
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(n_a):
        ops.bn_train_apply(xs[i % NBUF], st, g, b, rm, rv, relu=True)
    for i in range(n_ar):
        ops.bn_train_apply(xs[i % NBUF], st, g, b, rm, rv, residual=rs[i % NBUF], relu=True)
    for i in range(n_dz):
        ops.bn_backward_apply(xs[i % NBUF], rs[i % NBUF], mean, invstd, g, sums)
    for i in range(n_c2):
        ops.bn_train_backward(xs[i % NBUF], None, rs[i % NBUF], mean, invstd, g, True, False, beta=b)
    for i in range(n_c3):
        ops.bn_train_backward(xs[i % NBUF], rs[(i + 1) % NBUF], rs[i % NBUF], mean, invstd, g, True, True, beta=b)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    by = nb * (2 * n_a + 3 * n_ar + 3 * n_dz + 5 * n_c2 + 8 * n_c3)
    tot_ms += ms
    tot_b += by
    print(f'rows={rows} C={C}: {ms:7.3f} ms  {by / ms / 1e9:6.2f} TB/s of algorithmic bytes')
print(f'BatchNorm mix of one step: {tot_ms:.3f} ms, {tot_b / 1e9:.1f} GB algorithmic, {tot_b / tot_ms / 1e9:.2f} TB/s')
