"""Dev tool: conv_glds4_kernel<pp> on v_mfma_f32_16x16x32_bf16 vs v_mfma_f32_32x32x16_bf16 (tuning key conv.glds4_mfma32), the
step's 256 x 256-tile shapes, interleaved rounds in one process, median (min) per arm; random normal operands."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from das_amd import _lib, ops  # noqa: E402

lib = _lib.load()
B = 16
# B, H, W, Cin, Cout, k  (das_head.py:112-161 towers as one plane of 8840 rows per image; mspn_mmpose.py:81-157 1x1 convs)
SHAPES = [(B, 64, 104, 256, 256, 3), (B, 64, 104, 512, 256, 1), (B, 32, 52, 1024, 1024, 1), (B, 16, 26, 2048, 2048, 1),
          (B, 32, 52, 1024, 512, 1), (B, 16, 26, 512, 2048, 1), (B, 64, 104, 512, 512, 1), (B, 32, 52, 512, 1024, 1)]
ROUNDS, INNER = 7, 5
torch.manual_seed(0)
print('| shape | kernel 16x16x32 | us (min) | TF | kernel 32x32x16 | us (min) | TF | 32 / 16 |')
print('|---|---|---|---|---|---|---|---|')
for (b, H, W, Cin, Cout, k) in SHAPES:
    x = torch.randn(b, H, W, Cin, device='cuda', dtype=torch.bfloat16)
    w = (torch.randn(Cout, k, k, Cin, device='cuda') / (Cin * k * k) ** 0.5).to(torch.bfloat16)
    y = ops.conv2d(x, w, k, k, 1, k // 2)
    names, ts = {}, {0: [], 1: []}
    for arm in (0, 1):
        lib.das_tuning_set(b'conv.glds4_mfma32', arm)
        lib.das_tuning_set(b'conv.glds4_mf', 8)
        for _ in range(3):
            ops.conv2d(x, w, k, k, 1, k // 2, out=y)
        names[arm] = lib.das_last_kernel().decode()
    for _ in range(ROUNDS):
        for arm in (0, 1):
            lib.das_tuning_set(b'conv.glds4_mfma32', arm)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(INNER):
                ops.conv2d(x, w, k, k, 1, k // 2, out=y)
            e1.record()
            torch.cuda.synchronize()
            ts[arm].append(e0.elapsed_time(e1) / INNER * 1e3)
    lib.das_tuning_reset()
    fl = 2.0 * b * H * W * Cout * k * k * Cin
    m0, m1 = statistics.median(ts[0]), statistics.median(ts[1])
    print(f'| {H}x{W} {Cin}->{Cout} k{k} | `{names[0]}` | {m0:.1f} ({min(ts[0]):.1f}) | {fl / m0 / 1e6:.0f} | `{names[1]}` | {m1:.1f} ({min(ts[1]):.1f}) | '
          f'{fl / m1 / 1e6:.0f} | {m0 / m1:.3f} |', flush=True)
