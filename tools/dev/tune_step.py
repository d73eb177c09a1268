"""Dev tool: train-step time under dispatch-threshold variations, ONE process (same box, same clocks, same model):
for each `key=value[,key=value...]` argument the tuning is set, 2 warm-up + N timed steps run, the tuning is reset.
The baseline (defaults) runs first, in the middle and last: the spread of those three is the noise floor.
usage: tune_step.py [-n steps] cfg1 cfg2 ...      special keys: WGRAD_BATCH=<n> (das_amd.autograd), SLOTS=<full>,<mid> and UPCONV=<0|1> (das_amd.nn)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
if os.environ.get('DASLIB'):   # dev: a variant build of the library (csrc/Makefile: make variant VAR=...)
    from das_amd import _lib as _l0
    _l0.LIB_PATH = os.path.join(os.path.dirname(_l0.LIB_PATH), os.environ['DASLIB'])
import bench
from das_amd import _lib, autograd as ag, nn as dnn
from das_amd.datasets import SyntheticPoseDataset, collate
from das_amd.optim import FlatSGD, train_iteration

args = sys.argv[1:]
N = 10
if args and args[0] == '-n':
    N = int(args[1]); args = args[2:]
lib = _lib.load()
dev = torch.device('cuda', 0)
model = bench.build_model(dev, num_stages=4, train=True)
ds = SyntheticPoseDataset(num_joints=bench.J, img_shape=(bench.H, bench.W), length=16, seed=0)
data = collate([ds[i] for i in range(16)], device=dev)
opt = FlatSGD(model, lr=2e-3, momentum=0.9, weight_decay=1e-4, bias_lr_mult=2.0, bias_decay_mult=0.0, max_grad_norm=35.0)
for _ in range(4):
    train_iteration(model, opt, data, 2e-3)


_UP_EV = data['gt_poses_3d'][0]._das_uploaded
_hiccup = [0.0]
_bwd_res = [0]
_tb = torch.Tensor.backward


def _backward(self, *a, **k):
    if _bwd_res[0]:
        lib.das_tuning_set(b'comm.reserved_cus', _bwd_res[0])
    try:
        return _tb(self, *a, **k)
    finally:
        if _bwd_res[0]:
            lib.das_tuning_set(b'comm.reserved_cus', 0)


torch.Tensor.backward = _backward
_head_ft = model.bbox_head.forward_train


def _slow_head(*a, **k):
    if _hiccup[0] > 0:
        time.sleep(_hiccup[0])
    return _head_ft(*a, **k)


model.bbox_head.forward_train = _slow_head


def run(cfg):
    _lib.check(lib.das_tuning_reset(), 'reset')
    wb, fs, ms_ = ag.WGRAD_BATCH, dnn._FULL_SLOTS, dnn._MID_SLOTS
    for kv in [c for c in cfg.split(',') if c]:
        k, v = kv.split('=')
        if k == 'WGRAD_BATCH':
            ag.WGRAD_BATCH = int(v)
        elif k == 'UPCONV':
            dnn.UPCONV_AT_LOW_RES = bool(int(v))
        elif k == 'RESBITS':
            ag.RES_BITS = bool(int(v))
        elif k == 'GNMASK':
            ag.GN_REMASK = bool(int(v))
        elif k == 'BITS':
            ag.MASK_BITS = bool(int(v))
        elif k == 'DCNF':
            ag.DCN_FUSED = bool(int(v))
        elif k == 'UPLOADEV':      # 0: the ground truth without its upload events (the detector's side stream then waits for the step before)
            for key in ('gt_poses_3d', 'centers2d', 'depths'):
                for t in data[key]:
                    if int(v):
                        t._das_uploaded = _UP_EV
                    elif hasattr(t, '_das_uploaded'):
                        del t._das_uploaded
        elif k == 'SLEEP':         # a host hiccup of v ms in every step, right before the head is queued
            _hiccup[0] = int(v) * 1e-3
        elif k == 'BWDRES':        # comm.reserved_cus = v during backward only (a CU budget beside the weight gradients' side stream)
            _bwd_res[0] = int(v)
        elif k == 'SIDEPP':
            ag.SIDE_PP_SHARE = int(v)
        elif k == 'CHAIN':
            dnn.CHAIN_CONSUMERS = bool(int(v))
        elif k == 'DUAL':
            ag.DUAL_APPLY = bool(int(v))
        elif k == 'DEFER':
            dnn.DEFERRED_SKIPS = bool(int(v))
        elif k == 'UPMERGE':
            dnn.UPMERGE_FUSED = bool(int(v))
        elif k == 'SLOTS':
            dnn._FULL_SLOTS, dnn._MID_SLOTS = int(v.split('/')[0]), int(v.split('/')[1])
        else:
            _lib.check(lib.das_tuning_set(k.encode(), int(v)), k)
    for _ in range(2):
        train_iteration(model, opt, data, 2e-3)
    torch.cuda.synchronize()
    # per-step GPU time (events on the stream), MEDIAN over the steps: robust against the occasional slow step of a noisy box
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
    evs[0].record()
    for i in range(N):
        train_iteration(model, opt, data, 2e-3)
        evs[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(N))
    ms = ts[N // 2]
    ag.WGRAD_BATCH, dnn._FULL_SLOTS, dnn._MID_SLOTS = wb, fs, ms_
    ag.DCN_FUSED = True
    ag.SIDE_PP_SHARE = 2
    _hiccup[0] = 0.0
    _bwd_res[0] = 0
    for key in ('gt_poses_3d', 'centers2d', 'depths'):
        for t in data[key]:
            t._das_uploaded = _UP_EV
    dnn.CHAIN_CONSUMERS = dnn.UPCONV_AT_LOW_RES = dnn.UPMERGE_FUSED = dnn.DEFERRED_SKIPS = ag.DUAL_APPLY = ag.MASK_BITS = ag.GN_REMASK = ag.RES_BITS = True
    _lib.check(lib.das_tuning_reset(), 'reset')
    return ms


R = 1
if args and args[0] == '-r':
    R = int(args[1]); args = args[2:]
if R > 1:       # interleaved repetitions: defaults, cfg1, cfg2, ..., R times; mean and range per configuration
    res = {c: [] for c in [''] + args}
    for rep in range(R):
        for c in [''] + args:
            res[c].append(run(c))
    b = sum(res['']) / R
    for c in [''] + args:
        v = res[c]
        med = sorted(v)[R // 2]
        print(f'{(c or "defaults"):60s} median {med:7.2f} ms  ({med - sorted(res[""])[R // 2]:+.2f})  range {min(v):.2f} .. {max(v):.2f}', flush=True)
    sys.exit(0)
base = [run('')]
print(f'{"defaults":60s} {base[0]:7.2f} ms', flush=True)
half = len(args) // 2
for i, cfg in enumerate(args):
    if i == half:
        base.append(run(''))
        print(f'{"defaults (again)":60s} {base[-1]:7.2f} ms', flush=True)
    ms = run(cfg)
    print(f'{cfg:60s} {ms:7.2f} ms  ({ms - sum(base) / len(base):+.2f})', flush=True)
base.append(run(''))
print(f'{"defaults (last)":60s} {base[-1]:7.2f} ms   noise floor: {max(base) - min(base):.2f} ms')
