"""Dev tool: phases of conv_pt3_kernel from in-kernel wall-clock stamps (stamped dev build: make -C das_amd/csrc stamps).
Per workgroup: entry, prologue issued, first step landed, end of the first tile's K loop, end of its epilogue, the same
for the second tile, kernel end."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import numpy as np
import torch
from das_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libdas_hip_stamps.so')
from das_amd import ops
lib = _lib.load()
for a in sys.argv[1:]:
    k, v = a.split('=')
    _lib.check(lib.das_tuning_set(k.encode(), int(v)), k)
lib.das_dev_set_stamps.restype, lib.das_dev_set_stamps.argtypes = C.c_int, [C.c_void_p]
NWG = 8192
stamps = torch.zeros(NWG * 8, dtype=torch.int64, device='cuda')
assert lib.das_dev_set_stamps(stamps.data_ptr()) == 0
shapes = [(16, 64, 104, 128, 128, 3), (16, 64, 104, 512, 128, 1), (16, 128, 208, 256, 128, 3)]
torch.manual_seed(0)
for (B, H, W, Cin, Cout, k) in shapes:
    by = B * H * W * (Cin + Cout) * 2
    nb = max(2, int(700e6 // by) + 1)
    xs = [torch.randn(B, H, W, Cin, device='cuda', dtype=torch.bfloat16) for _ in range(nb)]
    ys = [torch.empty(B, H, W, Cout, device='cuda', dtype=torch.bfloat16) for _ in range(nb)]
    w = (torch.randn(Cout, k, k, Cin, device='cuda') / (Cin * k * k) ** 0.5).to(torch.bfloat16)
    for i in range(nb):
        ops.conv2d(xs[i], w, k, k, 1, k // 2, out=ys[i])
    stamps.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.conv2d(xs[0], w, k, k, 1, k // 2, out=ys[0])
    e1.record()
    torch.cuda.synchronize()
    t = stamps.cpu().numpy().reshape(NWG, 8)
    t = t[t[:, 0] != 0].astype(np.float64) * 0.01
    t0 = t[:, 0].min()
    d = lambda a: f'{np.median(a):6.1f} (max {a.max():6.1f})'
    two = t[t[:, 5] > 0]
    print(f'{H}x{W} {Cin}->{Cout} k{k}: {ops.last_kernel()} wgs={len(t)} ({len(two)} with two tiles) event {e0.elapsed_time(e1) * 1e3:6.1f} us '
          f'span {t[:, 7].max() - t0:6.1f}\n   skew {d(t[:, 0] - t0)} setup+issue {d(t[:, 1] - t[:, 0])} first landed {d(t[:, 2] - t[:, 1])} '
          f'K1 {d(t[:, 3] - t[:, 2])} epi1 {d(t[:, 4] - t[:, 3])}')
    if len(two):
        print(f'   K2 {d(two[:, 5] - two[:, 4])} epi2 {d(two[:, 6] - two[:, 5])} tail {d(two[:, 7] - two[:, 6])}')
