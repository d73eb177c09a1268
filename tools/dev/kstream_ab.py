"""Dev tool: the step's long-K 1x1 convs on the tile kernels vs conv1x1_kstream_kernel (tuning key conv.kstream), forward with
statistics and the data gradient with the fused BatchNorm backward (mode 4), interleaved rounds, median (min)."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from das_amd import _lib, ops  # noqa: E402

lib = _lib.load()
B = 16
SHAPES = [(B, 64, 104, 512, 128), (B, 64, 104, 512, 256), (B, 64, 104, 512, 512), (B, 32, 52, 1024, 256), (B, 32, 52, 1024, 512), (B, 32, 52, 1024, 1024), (B, 128, 208, 512, 128)]
ROUNDS, INNER = 7, 5
torch.manual_seed(0)
print('| shape | mode | tile kernel | us (min) | kstream us (min) | tile / kstream | HBM floor us (6.2 TB/s) |')
print('|---|---|---|---|---|---|---|')
for (b, H, W, Cin, Cout) in SHAPES:
    rows = b * H * W
    x = torch.randn(b, H, W, Cin, device='cuda', dtype=torch.bfloat16)
    w = (torch.randn(Cout, 1, 1, Cin, device='cuda') / Cin ** 0.5).to(torch.bfloat16)
    raw = torch.randn(b, H, W, Cout, device='cuda', dtype=torch.bfloat16)
    mean, invstd = torch.zeros(Cout, device='cuda'), torch.ones(Cout, device='cuda')
    gamma, beta = torch.ones(Cout, device='cuda'), torch.zeros(Cout, device='cuda')
    bnb = ops.BnBwd(raw, None, mean, invstd, gamma, beta, True)
    y = torch.empty(b, H, W, Cout, device='cuda', dtype=torch.bfloat16)
    for mode in ('fwd+stats', 'dgrad+bnb'):
        def run():
            st = torch.zeros(2 * Cout, device='cuda')
            if mode == 'fwd+stats':
                ops.conv2d(x, w, 1, 1, 1, 0, stats=st, out=y)
            else:
                ops.conv2d(x, w, 1, 1, 1, 0, stats=st, bn_bwd=bnb, out=y)
        names, ts = {}, {0: [], 1: []}
        for arm in (0, 1):
            lib.das_tuning_set(b'conv.kstream', 31 * arm)
            for _ in range(3):
                run()
            names[arm] = lib.das_last_kernel().decode()
        for _ in range(ROUNDS):
            for arm in (0, 1):
                lib.das_tuning_set(b'conv.kstream', 31 * arm)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(INNER):
                    run()
                e1.record()
                torch.cuda.synchronize()
                ts[arm].append(e0.elapsed_time(e1) / INNER * 1e3)
        lib.das_tuning_reset()
        by = rows * (Cin + Cout * (2 if mode != 'fwd+stats' else 1)) * 2
        m0, m1 = statistics.median(ts[0]), statistics.median(ts[1])
        print(f'| {H}x{W} {Cin}->{Cout} | {mode} | `{names[0]}` | {m0:.1f} ({min(ts[0]):.1f}) | {m1:.1f} ({min(ts[1]):.1f}) [`{names[1]}`] | '
              f'{m0 / m1:.2f} | {by / 6.2e6:.1f} |', flush=True)
print('(each timed call includes a 2C-float torch.zeros fill: the same ~3 us in both arms)')
