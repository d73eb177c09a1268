"""Dev tool: DCNv2 forward of a head layer — das_dcn3x3_fused (with / without the col side output) against
das_deform_im2col3x3 + the 1x1 GEMM over col, on the head's four ragged levels at B = 16 and B = 8 (cold operands: three
buffer sets in rotation)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from das_amd import ops

LEVELS = [(64, 104), (32, 52), (16, 26), (8, 13)]
for B in (16, 8):
    C = O = 256
    xs = [ops.Ragged.from_levels([torch.randn(B, h, w, C, device='cuda').to(torch.bfloat16) for h, w in LEVELS]) for _ in range(3)]
    oms = []
    for _ in range(3):
        lv = []
        for h, w in LEVELS:
            t = torch.zeros(B, h, w, 32, device='cuda')
            t[..., :18] = torch.randn(B, h, w, 18, device='cuda') * 1.5
            t[..., 18:27] = torch.randn(B, h, w, 9, device='cuda')
            lv.append(t)
        oms.append(ops.Ragged.from_levels(lv))
    w = (torch.randn(O, 1, 1, 9 * C, device='cuda') / (9 * C) ** 0.5).to(torch.bfloat16)
    bias = torch.randn(O, device='cuda')

    def timeit(fn, n=9):
        fn(0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            fn(i % 3)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    t_two = timeit(lambda i: ops.conv2d(ops.deform_im2col3x3(xs[i], oms[i]), w, 1, 1, shift=bias))
    t_im = timeit(lambda i: ops.deform_im2col3x3(xs[i], oms[i]))
    t_fc = timeit(lambda i: ops.dcn3x3_fused(xs[i], oms[i], w, bias, want_col=True))
    t_f = timeit(lambda i: ops.dcn3x3_fused(xs[i], oms[i], w, bias))
    rows = xs[0].rows
    fl = 2.0 * rows * O * 9 * C
    print(f'B={B} rows={rows}: im2col {t_im:.1f} us, im2col + GEMM {t_two:.1f} us ({ops.last_kernel()}), fused + col {t_fc:.1f} us, '
          f'fused {t_f:.1f} us ({fl / t_f / 1e6:.0f} TF)', flush=True)
