"""Dev tool: what does a launch of the step's most frequent kernels cost on a problem that is all fixed cost?
(back-to-back launches, HIP events; the bytes column says how little there is to move)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from das_amd import ops

BF = torch.bfloat16


def bench(name, fn, mb, n=60):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f'{name:58s} {e0.elapsed_time(e1) / n * 1e3:7.1f} us  {mb:7.1f} MB')


for (B, H, W, C) in [(16, 16, 26, 512), (16, 16, 26, 2048), (16, 32, 52, 256), (16, 32, 52, 1024), (16, 64, 104, 128)]:
    rows = B * H * W
    x = torch.randn(B, H, W, C, device='cuda').to(BF)
    res = torch.randn(B, H, W, C, device='cuda').to(BF)
    g, b = torch.ones(C, device='cuda'), torch.zeros(C, device='cuda')
    rm, rv = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
    mb = rows * C * 2 / 1e6
    for slots in (1, 16):
        st = torch.rand(slots * 2 * C, device='cuda') * rows / slots
        bench(f'bn_apply {H}x{W}x{C} slots={slots}', lambda: ops.bn_train_apply(x, st, g, b, rm, rv, relu=True), 2 * mb)
    st = torch.rand(16 * 2 * C, device='cuda') * rows / 16
    bench(f'bn_apply {H}x{W}x{C} +residual', lambda: ops.bn_train_apply(x, st, g, b, rm, rv, residual=res, relu=True), 3 * mb)
    mean, invstd = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
    for slots in (1, 16):
        sums = torch.rand(slots * 2 * C, device='cuda')
        bench(f'bn_backward_apply(dz) {H}x{W}x{C} slots={slots}', lambda: ops.bn_backward_apply(x, res, mean, invstd, g, sums), 3 * mb)
    bench(f'bn_train_backward classic relu {H}x{W}x{C}', lambda: ops.bn_train_backward(x, res, res, mean, invstd, g, True, False, beta=b), 5 * mb)

print('--- convs (cold operands rotate through 6 buffers)')
for (B, H, W, Cin, Cout, k) in [(16, 16, 26, 512, 2048, 1), (16, 16, 26, 2048, 512, 1), (16, 16, 26, 512, 512, 3), (16, 32, 52, 1024, 256, 1),
                                (16, 32, 52, 256, 256, 3), (16, 32, 52, 512, 1024, 1)]:
    xs = [torch.randn(B, H, W, Cin, device='cuda').to(BF) for _ in range(6)]
    ys = [torch.empty(B, H, W, Cout, device='cuda', dtype=BF) for _ in range(6)]
    w = (torch.randn(Cout, k, k, Cin, device='cuda') / (Cin * k * k) ** 0.5).to(BF)
    rows = B * H * W
    mb = rows * (Cin + Cout) * 2 / 1e6
    i = [0]
    for slots in (0, 1, 4, 16):
        st = torch.zeros(max(slots, 1) * 2 * Cout, device='cuda') if slots else None

        def fn():
            i[0] += 1
            ops.conv2d(xs[i[0] % 6], w, k, k, 1, k // 2, out=ys[i[0] % 6], stats=st)
        fn()
        bench(f'conv {H}x{W} {Cin}->{Cout} k{k} stats slots={slots} ({ops.last_kernel()[5:]})', fn, mb)
