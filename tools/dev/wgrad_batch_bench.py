"""Dev tool: n weight gradients as n launches vs ONE batched launch (das_conv2d_wgrad_batch), cold operands."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from das_amd import ops
groups = {
    '4 x (32x52 256->1024 1x1)': [(16, 32, 52, 256, 1024, 1)] * 4,
    '4 x (32x52 256->256 3x3)': [(16, 32, 52, 256, 256, 3)] * 4,
    'layer3 block: 1024->256, 256->256 3x3, 256->1024': [(16, 32, 52, 1024, 256, 1), (16, 32, 52, 256, 256, 3), (16, 32, 52, 256, 1024, 1)] * 2,
    'layer1 block: 256->64, 64->64 3x3, 64->256': [(16, 128, 208, 256, 64, 1), (16, 128, 208, 64, 64, 3), (16, 128, 208, 64, 256, 1)] * 2,
    'layer2 block: 512->128, 128->128 3x3, 128->512': [(16, 64, 104, 512, 128, 1), (16, 64, 104, 128, 128, 3), (16, 64, 104, 128, 512, 1)] * 2,
    'layer4 block: 2048->512, 512->512 3x3, 512->2048': [(16, 16, 26, 2048, 512, 1), (16, 16, 26, 512, 512, 3), (16, 16, 26, 512, 2048, 1)] * 2,
}
for name, shapes in groups.items():
    by = sum(B * H * W * (Cin + Cout) * 2 for (B, H, W, Cin, Cout, k) in shapes)
    nb = max(2, int(700e6 // by) + 1)
    sets = []
    for _ in range(nb):
        items = []
        for (B, H, W, Cin, Cout, k) in shapes:
            items.append((torch.randn(B, H, W, Cin, device='cuda', dtype=torch.bfloat16),
                          torch.randn(B, H, W, Cout, device='cuda', dtype=torch.bfloat16), k, k, 1, k // 2,
                          torch.zeros(Cout, k, k, Cin, device='cuda')))
        sets.append(items)

    def run(batched):
        for it in sets:
            go(it, batched)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 2 * nb
        e0.record()
        for i in range(n):
            go(sets[i % nb], batched)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    def go(items, batched):
        if batched:
            ops.conv2d_wgrad_batch(items)
        else:
            for (x, dy, k, _, s, p, out) in items:
                ops.conv2d_wgrad(x, dy, k, k, s, p, out=out, accumulate=True)
    a, b = run(False), run(True)
    print(f'{name}: separate {a:7.1f} us   batched {b:7.1f} us   ({a / b:.2f}x)', flush=True)
