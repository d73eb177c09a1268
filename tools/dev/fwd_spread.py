"""Dev probe: the spread of the FIRST forward's losses over repeated evaluations of one model on one batch (train mode, no
optimizer step: the only differences between evaluations are the orders of float atomics), per switch configuration."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import bench
from das_amd import autograd as ag, nn as dnn, losses, optim
from das_amd.datasets import SyntheticPoseDataset, collate

dev = torch.device('cuda', 0)
ds = SyntheticPoseDataset(num_joints=bench.J, img_shape=(bench.H, bench.W), length=16, seed=0)
data = collate([ds[i] for i in range(16)], device=dev)
SW = {'FINALIZE_MANY': (ag, 'FINALIZE_MANY'), 'GNWS': (dnn, 'ZEROED_GN_WS'), 'TARGETS': (losses, 'FUSED_TARGETS'),
      'CHAIN': (dnn, 'CHAIN_CONSUMERS'), 'DCNF': (ag, 'DCN_FUSED'), 'UPMERGE': (dnn, 'UPMERGE_FUSED'),
      'DEFER': (dnn, 'DEFERRED_SKIPS'), 'DUAL': (ag, 'DUAL_APPLY'), 'BITS': (ag, 'MASK_BITS')}
dtype = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
for off in [()] + [(k,) for k in sys.argv[2:]]:
    for k, (m, a) in SW.items():
        setattr(m, a, k not in off)
    torch.manual_seed(0)
    model = bench.build_model(dev, num_stages=4, train=True, dtype=dtype)
    vals = []
    for _ in range(6):
        out = model.train_step(data, None)
        vals.append({k: float(v) for k, v in out['log_vars'].items()})
        del out
    keys = list(vals[0])
    print('off=%-14s' % (','.join(off) or '-'), '  '.join('%s %.4f..%.4f' % (k[5:9] or 'sum', min(v[k] for v in vals), max(v[k] for v in vals)) for k in keys), flush=True)
