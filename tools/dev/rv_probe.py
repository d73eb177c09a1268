"""Dev probe: which switch moves the running variances? N steps with everything on except the switches named in argv[2:],
compared with everything on; and everything on twice (the run-to-run floor)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import bench
from das_amd import autograd as ag, nn as dnn, losses, optim
from das_amd.datasets import SyntheticPoseDataset, collate
from das_amd.optim import FlatSGD, train_iteration

dev = torch.device('cuda', 0)
ds = SyntheticPoseDataset(num_joints=bench.J, img_shape=(bench.H, bench.W), length=16, seed=0)
data = collate([ds[i] for i in range(16)], device=dev)
N = int(sys.argv[1])
SW = {'FINALIZE_MANY': (ag, 'FINALIZE_MANY'), 'PAD': (optim, 'PAD_ODD_CHANNELS'), 'GNWS': (dnn, 'ZEROED_GN_WS'),
      'TARGETS': (losses, 'FUSED_TARGETS'), 'CHAIN': (dnn, 'CHAIN_CONSUMERS'), 'DCNF': (ag, 'DCN_FUSED'),
      'UPMERGE': (dnn, 'UPMERGE_FUSED'), 'DEFER': (dnn, 'DEFERRED_SKIPS'), 'DUAL': (ag, 'DUAL_APPLY')}


def run(off):
    for k, (m, a) in SW.items():
        setattr(m, a, k not in off)
    torch.manual_seed(0)
    model = bench.build_model(dev, num_stages=4, train=True)
    opt = FlatSGD(model, lr=2e-3, momentum=0.9, weight_decay=1e-4, bias_lr_mult=2.0, bias_decay_mult=0.0, max_grad_norm=35.0)
    out = [float(train_iteration(model, opt, data, 2e-3)['log_vars']['loss']) for _ in range(N)]
    names = [n for n, b in model.named_buffers() if n.endswith('running_var')]
    rv = [b.flatten().float().clone() for n, b in model.named_buffers() if n.endswith('running_var')]
    pars = {n: p.detach().flatten().float().clone() for n, p in model.named_parameters()}
    return out, names, rv, pars


base = run(())
for off in [()] + [(k,) for k in sys.argv[2:]]:
    out, names, rv, pars = run(off)
    worst = max(((float((u - v).abs().max() / v.abs().max()), n) for n, u, v in zip(names, rv, base[2])), key=lambda t: t[0])
    print('off=%-16s loss[0] %.3f loss[-1] %.3f  worst running_var difference %.3e at %s' % (','.join(off) or '-', out[0], out[-1], worst[0], worst[1]), flush=True)
    # parameters: how far two runs drift apart relative to how far the parameter MOVED from its initial value (torch.manual_seed(0) init)
    pw = sorted(((float((pars[n] - base[3][n]).abs().max() / max(float(base[3][n].abs().max()), 1e-12)), n) for n in pars), reverse=True)[:3]
    print('      parameters furthest apart (relative to their largest element): ' + '; '.join('%.2e %s' % t for t in pw), flush=True)
