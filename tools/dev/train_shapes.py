"""Dev tool: per-shape timing of every conv-family launch (forward, data-grad, weight-grad) of the
training bench step (HIP events on the launch stream), sorted by time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from das_amd import ops
from das_amd.datasets import SyntheticPoseDataset, collate
from das_amd.optim import FlatSGD, train_iteration

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
if os.environ.get('DASLIB'):   # dev: a variant build of the library (make nobits / deep / stamps)
    from das_amd import _lib as _l
    _l.LIB_PATH = os.path.join(os.path.dirname(_l.LIB_PATH), os.environ['DASLIB'])
from das_amd import autograd as _ag
_ag.WGRAD_SIDE_STREAM = False
if os.environ.get('WGB'):
    _ag.WGRAD_BATCH = int(os.environ['WGB'])   # (dev: unbatched weight gradients -> per-layer times)   # kernels one at a time: per-shape times are not stretched by overlapped weight gradients
if os.environ.get('TUNE'):     # dev: dispatch keys for the whole run, TUNE=conv.balance_rows=0,conv.tail_split=0
    from das_amd import _lib as _l2
    for kv in os.environ['TUNE'].split(','):
        k, v = kv.split('=')
        _l2.check(_l2.load().das_tuning_set(k.encode(), int(v)), k)
dev = torch.device('cuda', 0)
model = bench.build_model(dev, num_stages=4, train=True)
ds = SyntheticPoseDataset(num_joints=bench.J, img_shape=(bench.H, bench.W), length=B, seed=0)
data = collate([ds[i] for i in range(B)], device=dev)
opt = FlatSGD(model, lr=2e-3, momentum=0.9, weight_decay=1e-4, bias_lr_mult=2.0, bias_decay_mult=0.0, max_grad_norm=35.0)
for _ in range(2):
    train_iteration(model, opt, data, 2e-3)
ops.profile_begin()
R = 2
for _ in range(R):
    train_iteration(model, opt, data, 2e-3)
torch.cuda.synchronize()
PROFILE = ops.profile_end()
agg = {}
for ent in PROFILE:
    tag, fl, e0, e1, shape = ent[:5]
    if shape and shape[0] == 'bn':
        continue
    if shape[0] == 'batch':
        shape = (0, 0, 0, 0, 0, 0, 0, shape[1])
    a = agg.setdefault((tag, shape), [0.0, 0.0, 0])
    a[0] += fl; a[1] += e0.elapsed_time(e1) * 1e-3; a[2] += 1
tot = sum(a[1] for a in agg.values())
print(f'total conv-family ms/step {tot / R * 1e3:.3f}')
fam = {}
for (tag, shape), (fl, sec, n) in agg.items():
    f = fam.setdefault(tag, [0.0, 0.0])
    f[0] += fl; f[1] += sec
for tag, (fl, sec) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print(f'  {sec / R * 1e3:8.3f} ms {fl / sec / 1e12:7.1f} TF  {tag}')
for (tag, shape), (fl, sec, n) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:140]:
    Bb, H, W, Cin, Cout, k, s, nl = shape[:8]
    mode = shape[8] if len(shape) > 8 else ''
    # floors per launch: operands once through HBM at 6.3 TB/s; MFMA at 2.5 PF
    per = sec / n
    flop = fl / n
    rows_out = flop / (2.0 * max(Cout, 1) * k * k * max(Cin, 1)) if Cin else 0
    hbm = (rows_out * (Cin if k == 1 and s == 1 else min(s * s, k * k) * Cin / (s * s) if False else Cin) + rows_out * Cout) * 2 / 6.3e12 if Cin else 0
    mf = flop / 2.5e15
    print(f'{sec / tot * 100:5.1f}% {sec / R * 1e3:7.3f}ms n={n // R:3d} {per * 1e6:7.1f}us (hbm {hbm * 1e6:6.1f} mfma {mf * 1e6:6.1f}) '
          f'{fl / sec / 1e12:7.1f}TF {tag[5:30]:22s} HxW={H}x{W} Cin={Cin} Cout={Cout} k={k} s={s} lv={nl} mode={mode}')
