"""Dev probe: where does the host BLOCK inside a training step? Wraps the calls that can wait for the GPU (Tensor.tolist /
item / cpu / nonzero, Event.synchronize, Stream.synchronize, cuda.synchronize) with a timer and the calling line; prints, for
the last steps, every such call that took longer than 0.2 ms."""
import os, sys, time, traceback
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import bench
from das_amd.datasets import SyntheticPoseDataset, collate
from das_amd.optim import FlatSGD, train_iteration

log = []


def wrap(owner, name):
    orig = getattr(owner, name)

    def f(*a, **k):
        t0 = time.perf_counter()
        out = orig(*a, **k)
        dt = (time.perf_counter() - t0) * 1e3
        if dt > 0.2:
            fr = [x for x in traceback.extract_stack()[:-1] if 'das_amd' in x.filename or 'bench.py' in x.filename]
            where = '%s:%d' % (os.path.basename(fr[-1].filename), fr[-1].lineno) if fr else '?'
            log.append((name, dt, where))
        return out
    setattr(owner, name, f)


for n in ('tolist', 'item', 'cpu', 'nonzero', '__float__', '__int__', '__bool__'):
    wrap(torch.Tensor, n)
wrap(torch.cuda.Event, 'synchronize')
wrap(torch.cuda.Stream, 'synchronize')
wrap(torch.cuda, 'synchronize')
dev = torch.device('cuda', 0)
model = bench.build_model(dev, num_stages=4, train=True)
ds = SyntheticPoseDataset(num_joints=bench.J, img_shape=(bench.H, bench.W), length=16, seed=0)
data = collate([ds[i] for i in range(16)], device=dev)
opt = FlatSGD(model, lr=2e-3, momentum=0.9, weight_decay=1e-4, bias_lr_mult=2.0, bias_decay_mult=0.0, max_grad_norm=35.0)
for _ in range(5):
    train_iteration(model, opt, data, 2e-3)
torch.cuda.synchronize()
for i in range(8):
    log.clear()
    t0 = time.perf_counter()
    train_iteration(model, opt, data, 2e-3)
    dt = (time.perf_counter() - t0) * 1e3
    print('step %d: host %.1f ms; blocking calls: %s' % (i, dt, '; '.join('%s %.1f ms @ %s' % x for x in log) or 'none'), flush=True)
torch.cuda.synchronize()
