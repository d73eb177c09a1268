"""Dev tool: one weight gradient, N back-to-back launches (for rocprofv3 --kernel-trace --stats A/B of library builds).
usage: wgrad_one.py B H W Cin Cout k stride pad [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from das_amd import ops
B, H, W, Cin, Cout, k, s, p = [int(v) for v in sys.argv[1:9]]
reps = int(sys.argv[9]) if len(sys.argv) > 9 else 20
Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
xs = [torch.randn(B, H, W, Cin, device='cuda', dtype=torch.bfloat16) for _ in range(4)]
dys = [torch.randn(B, Ho, Wo, Cout, device='cuda', dtype=torch.bfloat16) for _ in range(4)]
out = torch.zeros(Cout, k, k, Cin, device='cuda')
for i in range(reps):
    ops.conv2d_wgrad(xs[i % 4], dys[i % 4], k, k, s, p, out=out, accumulate=True)
torch.cuda.synchronize()
print(ops.last_kernel())
