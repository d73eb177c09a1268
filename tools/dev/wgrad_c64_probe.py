"""Dev probe: conv_wgrad_c64_kernel + its reduce pass alone (B=16, 128x208), for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from das_amd import ops
B, H, W = int(sys.argv[1]) if len(sys.argv) > 1 else 16, 128, 208
xs = [torch.randn(B, H, W, 64, device='cuda', dtype=torch.bfloat16) for _ in range(3)]
dys = [torch.randn(B, H, W, 64, device='cuda', dtype=torch.bfloat16) for _ in range(3)]
out = torch.zeros(64, 3, 3, 64, device='cuda')
for i in range(20):
    ops.conv2d_wgrad(xs[i % 3], dys[i % 3], 3, 3, 1, 1, out=out)
torch.cuda.synchronize()
print(ops.last_kernel())
