"""Dev tool: DCNv2 backward sampling (col2im) at the head's ragged geometry, B=16, C=256."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from das_amd import ops
B, C = 16, 256
sizes = [(64, 104), (32, 52), (16, 26), (8, 13)]
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
R = ops.Ragged.from_levels
x = R([torch.randn(B, h, w, C, device='cuda').to(torch.bfloat16) for h, w in sizes])
om = R([torch.cat([torch.randn(B, h, w, 18, device='cuda') * scale, torch.randn(B, h, w, 9, device='cuda'),
                   torch.zeros(B, h, w, 5, device='cuda')], -1) for h, w in sizes])
dc = R([torch.randn(B, h, w, 9 * C, device='cuda').to(torch.bfloat16) for h, w in sizes])
for _ in range(2):
    ops.deform_im2col3x3_backward(x, om, dc)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    dx, dom = ops.deform_im2col3x3_backward(x, om, dc)
e1.record()
torch.cuda.synchronize()
print(f'offset std {scale}: {e0.elapsed_time(e1) / 5 * 1e3:.1f} us per call (incl. zero fills)  checksum {float(dx.data.float().abs().sum()):.4e} {float(dom.data.abs().sum()):.4e}')
