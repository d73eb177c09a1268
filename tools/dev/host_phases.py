"""Dev probe: host time of the phases of a training step, for the first step after a device synchronisation (nothing can hold the
host back) and for the steady state — the phase whose time grows is where the host waits for the GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from das_amd.datasets import SyntheticPoseDataset, collate
from das_amd.optim import FlatSGD, train_iteration
from das_amd import optim, losses, train_ops as T

dev = torch.device('cuda', 0)
model = bench.build_model(dev, num_stages=4, train=True)
ds = SyntheticPoseDataset(num_joints=bench.J, img_shape=(bench.H, bench.W), length=16, seed=0)
data = collate([ds[i] for i in range(16)], device=dev)
opt = FlatSGD(model, lr=2e-3, momentum=0.9, weight_decay=1e-4, bias_lr_mult=2.0, bias_decay_mult=0.0, max_grad_norm=35.0)
acc = {}


def timed(owner, name, tag=None):
    orig = getattr(owner, name)

    def f(*a, **k):
        t0 = time.perf_counter()
        try:
            return orig(*a, **k)
        finally:
            acc[tag or name] = acc.get(tag or name, 0.0) + (time.perf_counter() - t0) * 1e3
    setattr(owner, name, f)


timed(model, 'extract_feat')
timed(model.bbox_head, 'prepare_targets')
timed(model.bbox_head, 'forward_train', 'head.forward_train')
timed(model.bbox_head, 'forward_rows', 'head.forward_rows')
timed(losses, 'das_head_loss_rows')
timed(T, 'rle_pose_loss_sums')
timed(torch.Tensor, 'backward')
timed(opt, 'all_reduce_grads')
timed(opt, 'step', 'opt.step')
timed(opt, 'zero_grad')
for _ in range(6):
    train_iteration(model, opt, data, 2e-3)
torch.cuda.synchronize()
rows = []
for i in range(8):
    acc.clear()
    t0 = time.perf_counter()
    train_iteration(model, opt, data, 2e-3)
    acc['TOTAL'] = (time.perf_counter() - t0) * 1e3
    rows.append(dict(acc))
torch.cuda.synchronize()
keys = ['TOTAL', 'zero_grad', 'extract_feat', 'prepare_targets', 'head.forward_train', 'head.forward_rows', 'das_head_loss_rows',
        'rle_pose_loss_sums', 'backward', 'all_reduce_grads', 'opt.step']
print('%-22s' % 'phase (host ms)', ' '.join('%7s' % ('step%d' % i) for i in range(8)))
for k in keys:
    print('%-22s' % k, ' '.join('%7.1f' % r.get(k, 0.0) for r in rows))
