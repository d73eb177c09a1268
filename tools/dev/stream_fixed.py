"""Dev tool: the persistent 1x1 kernel on small problems — what does a launch cost before the first byte counts?
Run under rocprofv3 --kernel-trace --stats for pure kernel durations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from das_amd import ops
B = 16
for (H, W, Cin, Cout) in [(32, 52, 64, 64), (32, 52, 64, 256), (32, 52, 256, 256), (32, 52, 256, 1024), (64, 104, 64, 64),
                          (64, 104, 256, 256), (128, 208, 64, 64)]:
    xs = [torch.randn(B, H, W, Cin, device='cuda', dtype=torch.bfloat16) for _ in range(4)]
    ys = [torch.empty(B, H, W, Cout, device='cuda', dtype=torch.bfloat16) for _ in range(4)]
    w = (torch.randn(Cout, 1, 1, Cin, device='cuda') / Cin ** 0.5).to(torch.bfloat16)
    for mode in ('plain', 'stats1', 'stats16', 'stats64'):
        stats = torch.zeros(int(mode[5:]) * 2 * Cout, device='cuda') if mode != 'plain' else None

        def fn(i):
            ops.conv2d(xs[i % 4], w, 1, 1, 1, 0, out=ys[i % 4], stats=stats)
        for i in range(4):
            fn(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(40):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        mb = B * H * W * (Cin + Cout) * 2 / 1e6
        print(f'{H}x{W} {Cin}->{Cout} {mode:7s} {e0.elapsed_time(e1) / 40 * 1e3:6.1f} us  {mb:6.1f} MB  {ops.last_kernel()}')
