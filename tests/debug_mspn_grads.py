import sys, os
sys.path[:0] = ['/root/repo', '/root/repo/tests', '/root/repo/tests/golden']
import numpy as np, torch, cases, das_amd
from oracle import backbone as ob
z = np.load('/root/repo/tests/golden/mspn_s2_train.npz')
shapes = [[int(i) for i in row if i >= 0] for row in z['sd_shapes']]
sd = cases.sd_from_manifest(z['sd_keys'], shapes, z['sd_dtypes'], 1)
x = cases.randn(7, 2, 3, 64, 96)
gs = [cases.randn(80+i, 2, 16, 16>>i, 24>>i) for i in range(4)]
def hip():
    m = das_amd.MSPN2(unit_channels=16, num_stages=2, num_blocks=[1,1,1,1], compute_dtype='f32'); m.load_state_dict(sd); m.cuda().train()
    outs = m(x.cuda()); sum((o.float()*g.cuda()).sum() for o, g in zip(outs, gs)).backward()
    return {k: p.grad.double().cpu() for k, p in m.named_parameters() if p.grad is not None}
def orc(dt):
    osd = {k: (v.to(dt).clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k else (v.to(dt) if v.is_floating_point() else v.clone())) for k, v in sd.items()}
    oo = ob.mspn2_forward(osd, x.to(dt), 2, (1,1,1,1), train=True); sum((o*g.to(dt)).sum() for o, g in zip(oo, gs)).backward()
    return {k: v.grad.double() for k, v in osd.items() if v.is_floating_point() and v.requires_grad and v.grad is not None}
h1, h2, o32, o64 = hip(), hip(), orc(torch.float32), orc(torch.float64)
def worst(a, b):
    return max(((a[k]-b[k]).abs().max().item()/max(b[k].abs().max().item(),1e-12), k) for k in b if k in a)
print('hip run1 vs run2 ', worst(h1, h2))
print('hip vs oracle f64', worst(h1, o64))
print('hip vs oracle f32', worst(h1, o32))
print('f32 vs f64 oracle', worst(o32, o64))
