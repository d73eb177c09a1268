"""Dev tool: fused RealNVP log-density kernels, forward / backward time at a few row counts."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from das_amd.pose_heads import RealNVP, RealNVP2D
from das_amd.train_ops import realnvp_log_prob

for dim, cls in ((3, RealNVP), (2, RealNVP2D)):
    flow = cls().cuda()
    for N in ([int(a) for a in sys.argv[1:]] or (256, 4096, 65536, 262144)):
        x = torch.randn(N, dim, device='cuda', requires_grad=True)
        g = torch.randn(N, device='cuda')
        for _ in range(2):
            realnvp_log_prob(flow, x).backward(g)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        n = 5
        tf = tb = 0.0
        for _ in range(n):
            e[0].record()
            lp = realnvp_log_prob(flow, x)
            e[1].record()
            lp.backward(g)
            e[2].record()
            torch.cuda.synchronize()
            tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
        print(f'D={dim} N={N:7d}: forward {tf / n * 1e3:8.1f} us  backward {tb / n * 1e3:8.1f} us')
