import sys, os
sys.path.insert(0, os.getcwd())
import torch
from das_amd import ops
B, C = 16, 256
sizes = [(64, 104), (32, 52), (16, 26), (8, 13)]
R = ops.Ragged.from_levels
mk = lambda: R([torch.randn(B, h, w, C, device='cuda').to(torch.bfloat16) for h, w in sizes])
xs, dys = [mk() for _ in range(3)], [mk() for _ in range(3)]
g, b = torch.ones(C, device='cuda'), torch.zeros(C, device='cuda')
ys, sts = [], []
for x in xs:
    y, st = ops.groupnorm(x, g, b, 32, 1e-5, relu=True, out=x.new(C), return_stats=True)
    ys.append(y); sts.append(st)
for i in range(3):
    ops.groupnorm_backward(dys[i], ys[i], xs[i], sts[i], g, 32, 1e-5, True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(12):
    ops.groupnorm_backward(dys[i % 3], ys[i % 3], xs[i % 3], sts[i % 3], g, 32, 1e-5, True)
e1.record(); torch.cuda.synchronize()
print(f'groupnorm_backward (reduce + apply): {e0.elapsed_time(e1) / 12 * 1e3:.1f} us per call')
