"""Dev tool: BatchNorm forward-apply / backward kernels at one of the bench's typical shapes.
usage: python tools/dev/bn_bench.py <shape index> [relu_mask_mode: none|y|recompute]
Prints the wall time per call (HIP events) and the HBM rate of the whole call; run it under
`rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from das_amd import ops

SHAPES = [(16, 128, 208, 64), (16, 128, 208, 256), (16, 64, 104, 128), (16, 64, 104, 512), (16, 32, 52, 256),
          (16, 32, 52, 1024), (16, 16, 26, 512), (16, 16, 26, 2048)]
B, H, W, C = SHAPES[int(sys.argv[1])]
mode = sys.argv[2] if len(sys.argv) > 2 else 'recompute'
dev = 'cuda'
dt = torch.bfloat16
raw = torch.randn(B, H, W, C, device=dev).to(dt)
dy = torch.randn(B, H, W, C, device=dev).to(dt)
res = torch.randn(B, H, W, C, device=dev).to(dt)
gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
stats = torch.stack([raw.float().sum((0, 1, 2)), raw.float().square().sum((0, 1, 2))]).reshape(-1).contiguous()
nbytes = raw.numel() * 2


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


with_res = mode == 'y'
y, mean, invstd = ops.bn_train_apply(raw, stats, gamma, beta, None, None, 0.1, 1e-5, residual=res if with_res else None,
                                     relu=mode != 'none')
t = timeit(lambda: ops.bn_train_apply(raw, stats, gamma, beta, None, None, 0.1, 1e-5,
                                      residual=res if with_res else None, relu=mode != 'none'))
acc = (2 + with_res)
print(f'shape {B}x{H}x{W}x{C} mode {mode}: fwd apply {t:7.1f} us  {acc * nbytes / t / 1e6:5.2f} TB/s ({acc} tensor passes)')
yy = y if mode == 'y' else None
t = timeit(lambda: ops.bn_train_backward(dy, yy, raw, mean, invstd, gamma, mode != 'none', with_res, beta=beta))
acc = {'none': 2 + 2 + 1, 'y': 3 + 3 + 2, 'recompute': 2 + 2 + 1}[mode]
print(f'    backward (reduce + apply) {t:7.1f} us  {acc * nbytes / t / 1e6:5.2f} TB/s ({acc} tensor passes)')
