"""Dev tool: where does a tile-kernel launch spend its time? Loads the stamped dev build of the library
(`make -C das_amd/csrc stamps`), runs the step's representative conv shapes on cold operands and splits every launch
into dispatch skew / prologue / first tile landed / K loop / epilogue from the per-workgroup wall-clock stamps
(100 MHz) the tile kernels record. Usage: python tools/dev/conv_stamps.py [shape-set]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import numpy as np
import torch
from das_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libdas_hip_stamps.so')
from das_amd import ops

lib = _lib.load()
lib.das_dev_set_stamps.restype, lib.das_dev_set_stamps.argtypes = C.c_int, [C.c_void_p]
NWG = 8192
stamps = torch.zeros(NWG * 8, dtype=torch.int64, device='cuda')
assert lib.das_dev_set_stamps(stamps.data_ptr()) == 0

shapes = [  # B,H,W,Cin,Cout,k
    (16, 32, 52, 256, 256, 3), (16, 32, 52, 1024, 256, 1), (16, 64, 104, 128, 128, 3), (16, 64, 104, 512, 128, 1),
    (16, 16, 26, 2048, 2048, 1), (16, 16, 26, 512, 2048, 1), (16, 64, 104, 512, 256, 1), (16, 32, 52, 1024, 1024, 1),
    (8, 64, 104, 256, 256, 3), (16, 64, 104, 256, 256, 3),
]
torch.manual_seed(0)
for (B, H, W, Cin, Cout, k) in shapes:
    by = B * H * W * (Cin + Cout) * 2
    nb = max(2, int(700e6 // by) + 1)
    xs = [torch.randn(B, H, W, Cin, device='cuda', dtype=torch.bfloat16) for _ in range(nb)]
    ys = [torch.empty(B, H, W, Cout, device='cuda', dtype=torch.bfloat16) for _ in range(nb)]
    w = (torch.randn(Cout, k, k, Cin, device='cuda') / (Cin * k * k) ** 0.5).to(torch.bfloat16)
    st = torch.zeros(16 * 2 * Cout, device='cuda', dtype=torch.float32)
    for i in range(nb):
        ops.conv2d(xs[i], w, k, k, 1, k // 2, out=ys[i], stats=st)
    rec = []
    for i in range(nb):
        stamps.zero_()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.conv2d(xs[i], w, k, k, 1, k // 2, out=ys[i], stats=st)
        e1.record()
        torch.cuda.synchronize()
        t = stamps.cpu().numpy().reshape(NWG, 8)
        t = t[t[:, 0] != 0]
        rec.append((e0.elapsed_time(e1) * 1e3, t.copy()))
    kern = ops.last_kernel()
    ev = np.median([r[0] for r in rec])
    t = rec[-1][1].astype(np.float64) * 0.01  # us
    if len(t) == 0:
        print(f'{H}x{W} {Cin}->{Cout} k{k}: {kern} (no stamps) event {ev:.1f} us')
        continue
    t0 = t[:, 0].min()
    span = t[:, 4].max() - t0
    d = lambda a: f'{np.median(a):6.1f} (max {a.max():6.1f})'
    has2 = (t[:, 2] > 0).all()
    mid = t[:, 2] if has2 else t[:, 1]
    print(f'{H}x{W} {Cin}->{Cout} k{k} B={B}: {kern} wgs={len(t)} event {ev:6.1f} us  span {span:6.1f} us\n'
          f'    start skew {d(t[:, 0] - t0)}  setup {d(t[:, 1] - t[:, 0])}  first tile {d(mid - t[:, 1])}  '
          f'K loop {d(t[:, 3] - mid)}  epilogue {d(t[:, 4] - t[:, 3])}  end spread {d(t[:, 4].max() - t[:, 4])}')
    if (t[:, 5] > 0).all():
        s7 = np.where(t[:, 7] > 0, t[:, 7], t[:, 6])
        print(f'    epilogue: acc->LDS+sync {d(t[:, 5] - t[:, 3])}  rows->global {d(t[:, 6] - t[:, 5])}  sync {d(s7 - t[:, 6])}  '
              f'stats {d(t[:, 4] - s7)}')
