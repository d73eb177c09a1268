"""Dev tool: which Python call sites issue the small device copies / fills of one train step (the kernel trace shows ~230
`__amd_rocclr_copyBuffer` launches per step)? torch.profiler with stacks over ONE step; aten::copy_ / fill_ / zero_ / clone
grouped by the innermost das_amd / bench frame."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402
from das_amd.datasets import SyntheticPoseDataset, collate  # noqa: E402
from das_amd.optim import FlatSGD, train_iteration  # noqa: E402

dev = torch.device('cuda', 0)
INFER = '--infer' in sys.argv      # the B = 8 1-stage inference step (forward + decode) instead of the train step
if INFER:
    model = bench.build_model(dev, num_stages=1, train=False)
    ds = SyntheticPoseDataset(num_joints=bench.J, img_shape=(bench.H, bench.W), length=8, seed=0)
    data = collate([ds[i] for i in range(8)], device=dev)
    bench.calibrate_scores(model, data['img'], data['img_metas'])

    def one_step():
        return model(data['img'], data['img_metas'], return_loss=False, rescale=True)
else:
    model = bench.build_model(dev, num_stages=4, train=True)
    ds = SyntheticPoseDataset(num_joints=bench.J, img_shape=(bench.H, bench.W), length=16, seed=0)
    data = collate([ds[i] for i in range(16)], device=dev)
    opt = FlatSGD(model, lr=2e-3, momentum=0.9, weight_decay=1e-4, bias_lr_mult=2.0, bias_decay_mult=0.0, max_grad_norm=35.0)

    def one_step():
        return train_iteration(model, opt, data, 2e-3)
for _ in range(4):
    one_step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    one_step()
    torch.cuda.synchronize()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sites = collections.Counter()
names = ('aten::copy_', 'aten::fill_', 'aten::zero_', 'aten::clone', 'aten::contiguous', 'aten::_to_copy', 'aten::cat', 'aten::stack',
         'aten::add_', 'aten::add', 'aten::mul', 'aten::sum', 'aten::empty_like')
for ev in prof.events():
    if ev.name in names and ev.cpu_parent is not None and ev.cpu_parent.name.startswith('aten::'):
        continue      # (count the outermost aten op only)
    if ev.name in names:
        stack = getattr(ev, 'stack', None) or []
        frame = next((f for f in stack if ('das_amd' in f or 'bench.py' in f) and 'torch/' not in f), stack[0] if stack else '?')
        sites[(ev.name, frame.replace(ROOT + '/', ''))] += 1
kern = collections.Counter()
for ev in prof.events():
    if ev.device_type is not None and 'cuda' in str(ev.device_type).lower():
        kern[ev.name[:60]] += 1
print('device-side events of the step (name: count), small ATen / runtime ones:')
for k, v in kern.most_common():
    if 'rocclr' in k or 'Memcpy' in k or 'Memset' in k or 'at::native' in k:
        print(f'  {v:5d}  {k}')
print('aten ops by innermost das_amd frame:')
for (n, f), v in sites.most_common(60):
    print(f'  {v:5d}  {n:18s} {f}')


# ---- the same question answered on the Python side (the profiler's stacks are empty on this build): a dispatch mode that
# records the innermost das_amd / bench frame of every ATen call that launches a copy / fill / cat / elementwise kernel
import traceback  # noqa: E402

from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

WATCH = ('copy_', '_to_copy', 'clone', 'cat', 'stack', 'fill_', 'zero_', 'add', 'mul', 'sub', 'div', 'sigmoid', 'contiguous',
         'ones', 'zeros', 'full', 'empty_like', 'index', 'slice_scatter', 'select_scatter')


class Sites(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.sites = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split('.')[0]
        out = func(*args, **(kwargs or {}))
        if name in WATCH and any(torch.is_tensor(a) and a.is_cuda for a in list(args) + ([out] if torch.is_tensor(out) else [])):
            fr = [f for f in traceback.extract_stack() if ('das_amd' in f.filename or 'bench.py' in f.filename)
                  and 'copy_sources' not in f.filename]
            where = f'{os.path.relpath(fr[-1].filename, ROOT)}:{fr[-1].lineno} {fr[-1].name}' if fr else '?'
            self.sites[(name, where)] += 1
        return out


with Sites() as rec:
    one_step()
    torch.cuda.synchronize()
print('ATen calls on device tensors in one step, by innermost das_amd / bench frame (dispatch mode):')
for (n, w), v in rec.sites.most_common(50):
    print(f'  {v:5d}  {n:14s} {w}')
