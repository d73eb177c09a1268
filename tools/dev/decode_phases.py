"""Dev tool: decode_kernel time at B = 8 under settings that switch phases off (no candidates / one kept pose /
no suppression), kernel only (outputs preallocated, HIP events around the C call)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from das_amd import _lib, ops

lib = _lib.load()
dev = 'cuda'
for name, (J, HW) in {'infer J=15 512x832': (15, [(64, 104), (32, 52), (16, 26), (8, 13)]),
                      'mupots J=21 768x1024': (21, [(96, 128), (48, 64), (24, 32), (12, 16)])}.items():
    B = 8
    g = torch.Generator().manual_seed(0)
    cls = [torch.randn(B, h, w, 1, generator=g) for h, w in HW]
    ctr = [torch.randn(B, h, w, 1, generator=g) + 0.5 for h, w in HW]
    lo, hi = -12.0, 4.0
    for _ in range(30):
        mid = 0.5 * (lo + hi)
        n = float(sum(((torch.sigmoid(c + mid) * torch.sigmoid(t)) > 0.07).sum() for c, t in zip(cls, ctr))) / B
        lo, hi = (mid, hi) if n < 150 else (lo, mid)
    cls = [c + lo for c in cls]
    pose = []
    for h, w in HW:
        p = torch.randn(B, h, w, 3 + 6 * J, generator=g)
        p[..., 3:3 + 3 * J] *= 30.0
        pose.append(p)
    cls, ctr, pose = [t.to(dev) for t in cls], [t.to(dev) for t in ctr], [t.to(dev) for t in pose]
    sf = torch.ones(B, 2, device=dev)
    for tag, kw in (('normal', {}), ('no candidates', dict(score_thr=0.9999)), ('nms_post=1', dict(nms_post=1)),
                    ('no suppression', dict(nms_thr=2.0))):
        a = dict(nms_pre=1000, nms_post=100, score_thr=0.07, nms_thr=0.9)
        a.update(kw)
        out = ops.decode(cls, ctr, pose, [8, 16, 32, 64], sf, J, a['nms_pre'], a['nms_post'], a['score_thr'], a['nms_thr'])
        torch.cuda.synchronize()
        cand = None
        ts = []
        for _ in range(30):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            # (ops.decode allocates; time the whole call but subtract nothing: compare variants instead)
            e0.record()
            out = ops.decode(cls, ctr, pose, [8, 16, 32, 64], sf, J, a['nms_pre'], a['nms_post'], a['score_thr'], a['nms_thr'])
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        print(f'{name:22s} {tag:16s} {ts[len(ts) // 2]:7.1f} us   kept/img {out["count"].float().mean().item():.1f}', flush=True)
