"""Dev tool: conv_pt3_kernel against conv_glds3_kernel<pp> on SINGLE-round launches (one tile per workgroup either way):
any difference is inside the kernel (K loop, prologue, epilogue), not in the schedule."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from das_amd import ops
BF = torch.bfloat16
B = 16
for (H, W, Cin, Cout, k) in [(32, 52, 256, 256, 3), (32, 52, 1024, 256, 1), (32, 52, 512, 128, 3)]:
    nb = 6
    xs = [torch.randn(B, H, W, Cin, device='cuda', dtype=BF) for _ in range(nb)]
    ys = [torch.empty(B, H, W, Cout, device='cuda', dtype=BF) for _ in range(nb)]
    w = (torch.randn(Cout, k, k, Cin, device='cuda') / (Cin * k * k) ** 0.5).to(BF)
    row = f'{H}x{W} {Cin}->{Cout} k{k}: '
    for name, tune in (('glds3<pp>', {'conv.pt3_mintiles': 0, 'conv.glds3_pp_mink': 0, 'conv.glds4_minblocks': 0, 'conv.splitk_target': 0}),
                       ('pt3', {'conv.pt3_mintiles': 1, 'conv.glds4_minblocks': 0, 'conv.splitk_target': 0})):
        with ops.tuning(**tune):
            for i in range(nb):
                ops.conv2d(xs[i], w, k, k, 1, k // 2, out=ys[i])
            n = 5 * nb
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(n):
                ops.conv2d(xs[i % nb], w, k, k, 1, k // 2, out=ys[i % nb])
            e1.record()
            torch.cuda.synchronize()
            row += f'{name} {e0.elapsed_time(e1) / n * 1e3:6.1f} us ({ops.last_kernel()})   '
    print(row, flush=True)
