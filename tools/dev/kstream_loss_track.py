"""Dev: twenty optimisation steps of the benchmark model (bf16, B = 16, one synthetic batch) with conv1x1_kstream_kernel on
(conv.kstream = 35, the default: both BatchNorm-backward modes; 3: the recomputed-mask mode only) and off (0), same initial weights: the loss curves must track each other within the run-to-run
spread of two identical runs (float atomics) — a wrong partial sum or statistic in the new kernel would show within a few steps."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch  # noqa: E402

import bench  # noqa: E402
from das_amd import _lib  # noqa: E402
from das_amd.datasets import SyntheticPoseDataset, collate  # noqa: E402
from das_amd.optim import FlatSGD, train_iteration  # noqa: E402

lib = _lib.load()
dev = torch.device('cuda', 0)
ds = SyntheticPoseDataset(num_joints=bench.J, img_shape=(bench.H, bench.W), length=16, seed=0)
data = collate([ds[i] for i in range(16)], device=dev)
N = 20


def run(mask):
    lib.das_tuning_reset()
    lib.das_tuning_set(b'conv.kstream', mask)
    torch.manual_seed(0)
    model = bench.build_model(dev, num_stages=4, train=True)
    opt = FlatSGD(model, lr=2e-3, momentum=0.9, weight_decay=1e-4, bias_lr_mult=2.0, bias_decay_mult=0.0, max_grad_norm=35.0)
    out = [float(train_iteration(model, opt, data, 2e-3)['log_vars']['loss']) for _ in range(N)]
    rv = torch.cat([b.flatten().float() for n, b in model.named_buffers() if n.endswith('running_var')])
    return out, rv


d, rvd = run(35)
a, rva = run(3)
b, rvb = run(0)
c, rvc = run(0)
print('step   kstream=35 (default)   kstream=3   kstream=0   kstream=0 (again)')
for i, (w, x, y, z) in enumerate(zip(d, a, b, c)):
    print('%3d  %16.3f %15.3f %10.3f %10.3f' % (i, w, x, y, z))
rel = lambda p, q: float(((p - q).abs() / (q.abs() + 1e-3)).max())
print('running_var after %d steps, max relative difference: 35 vs off %.2e; 3 vs off %.2e; off vs off %.2e'
      % (N, rel(rvd, rvb), rel(rva, rvb), rel(rvb, rvc)))
