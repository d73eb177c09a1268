"""Dev tool: the small-M, long-K conv layers (16x26 / 32x52 stages) with and without split-K, and against the 256-row
tile kernels the default dispatch would pick; B = argv[1] (8 = infer bench, 16 = train bench)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from das_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
SHAPES = [  # H, W, Cin, Cout, k, stride
    (16, 26, 512, 512, 3, 1), (16, 26, 2048, 512, 1, 1), (32, 52, 512, 512, 3, 2), (16, 26, 256, 256, 3, 1),
    (16, 26, 2048, 256, 1, 1), (16, 26, 512, 2048, 1, 1), (32, 52, 256, 256, 3, 1), (32, 52, 1024, 256, 1, 1),
    (32, 52, 256, 1024, 1, 1), (32, 52, 1024, 512, 1, 1), (64, 104, 128, 128, 3, 1), (64, 104, 512, 128, 1, 1),
    (32, 52, 1024, 1024, 1, 1), (16, 26, 2048, 2048, 1, 1), (8, 13, 256, 256, 3, 1),
]
CONFIGS = [('r1', {'conv.splitk_target': 0})] + [
    (f'm{m}t{t}s{ms}', {'conv.splitk_kernels': m, 'conv.splitk_target': t, 'conv.splitk_minsteps': ms})
    for (m, t, ms) in ((1, 256, 8), (2, 256, 8), (4, 256, 8), (7, 256, 8), (7, 256, 4), (7, 256, 12), (7, 208, 8), (6, 256, 6))]
print(f'B={B}  us per launch (kernel picked)')
for (H, W, Cin, Cout, k, s) in SHAPES:
    x = torch.randn(B, H, W, Cin, device='cuda', dtype=torch.bfloat16)
    w = (torch.randn(Cout, k, k, Cin, device='cuda') / (Cin * k * k) ** 0.5).to(torch.bfloat16)
    stats = torch.zeros(2 * Cout, device='cuda')
    row = f'{H}x{W} {Cin}->{Cout} k{k} s{s}: '
    for name, kv in CONFIGS:
        with ops.tuning(**kv):
            for _ in range(3):
                ops.conv2d(x, w, k, k, s, k // 2, stats=stats)
            kern = ops.last_kernel()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.conv2d(x, w, k, k, s, k // 2, stats=stats)
            e1.record()
            torch.cuda.synchronize()
        short = kern.replace('conv_', '').replace('_kernel', '')
        row += f'{name} {e0.elapsed_time(e1) / 20 * 1e3:6.1f} ({short})  '
    print(row)
