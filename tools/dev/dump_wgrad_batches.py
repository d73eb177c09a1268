"""Dev tool: record the weight-gradient batches of one training bench step (what backward hands to
das_conv2d_wgrad_batch, in order) -> tools/dev/wgrad_batches.json, replayed by tools/dev/wgrad_mix.py."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from das_amd import ops
from das_amd.datasets import SyntheticPoseDataset, collate
from das_amd.optim import FlatSGD, train_iteration

B = 16
dev = torch.device('cuda', 0)
model = bench.build_model(dev, num_stages=4, train=True)
ds = SyntheticPoseDataset(num_joints=bench.J, img_shape=(bench.H, bench.W), length=B, seed=0)
data = collate([ds[i] for i in range(B)], device=dev)
opt = FlatSGD(model, lr=2e-3, momentum=0.9, weight_decay=1e-4, bias_lr_mult=2.0, bias_decay_mult=0.0, max_grad_norm=35.0)
for _ in range(2):
    train_iteration(model, opt, data, 2e-3)
log = []
orig = ops.conv2d_wgrad_batch


def rec(items):
    batch = []
    for (x, dy, KH, KW, s, p, out) in items:
        if isinstance(x, ops.Ragged):
            batch.append(dict(ragged=[list(t) for t in x.sizes], B=x.B, Cin=x.C, Cout=dy.C, k=KH, s=s, p=p))
        else:
            batch.append(dict(B=x.shape[0], H=x.shape[1], W=x.shape[2], Cin=x.shape[3], Cout=dy.shape[3], k=KH, s=s, p=p))
    log.append(batch)
    return orig(items)


ops.conv2d_wgrad_batch = rec
train_iteration(model, opt, data, 2e-3)
torch.cuda.synchronize()
out = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/wgrad_batches.json'
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(log, open(out, 'w'))
print(len(log), 'batches,', sum(len(b) for b in log), 'ops ->', out)
