"""Dev tool: conv_glds3_kernel plain vs ping-pong on the train step's 256 x 128-tile shapes (B = 16), cold-ish rotation."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from das_amd import ops
B = 16
SHAPES = [(32, 52, 256, 256, 3), (64, 104, 128, 128, 3), (64, 104, 512, 128, 1), (32, 52, 1024, 256, 1), (128, 208, 128, 128, 3),
          (16, 26, 512, 512, 3), (16, 26, 2048, 512, 1), (32, 52, 1024, 512, 1), (128, 208, 64, 128, 3)]
for (H, W, Cin, Cout, k) in SHAPES:
    xs = [torch.randn(B, H, W, Cin, device='cuda', dtype=torch.bfloat16) for _ in range(4)]
    w = (torch.randn(Cout, k, k, Cin, device='cuda') / (Cin * k * k) ** 0.5).to(torch.bfloat16)
    stats = torch.zeros(16 * 2 * Cout, device='cuda')
    row = f'{H}x{W} {Cin}->{Cout} k{k}: '
    for name, v in (('plain', -1), ('pp', 0)):
        with ops.tuning(**{'conv.glds3_pp_mink': v, 'conv.glds4_minblocks': 0, 'conv.stream_minrows': 0}):
            for x in xs:
                ops.conv2d(x, w, k, k, 1, k // 2, stats=stats)
            kern = ops.last_kernel()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                for x in xs:
                    ops.conv2d(x, w, k, k, 1, k // 2, stats=stats)
            e1.record()
            torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        tf = 2.0 * B * H * W * Cout * k * k * Cin / us / 1e6
        row += f'{name} {us:6.1f} us {tf:5.0f} TF ({kern.replace("conv_", "").replace("_kernel", "")})  '
    print(row)
