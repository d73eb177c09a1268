"""Dev tool (VERDICT r3 #3a): what do the persistent one-workgroup-per-CU kernels do when a collective's kernels hold
CUs beside them? One GPU cannot run RCCL with more than one rank, so `das_dev_occupy_cus` stands in: N workgroups that
hold their CU slot for the length of a train step, launched on a side stream right before the step (the worst case — a
real all-reduce only runs during backward). For each N: the step time and the persistent families' times with every
persistent grid sized for the whole chip (comm.reserved_cus = 0) and for N fewer CUs (comm.reserved_cus = N).
mode `own`: 160 KiB of LDS per occupier (it owns its CU); mode `share`: 4 KiB / 256 threads (what an RCCL channel's
workgroup takes; it shares the CU when registers / wave slots allow).
usage: cu_pressure.py [own|share] [N ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import torch
import bench
from das_amd import _lib, ops, autograd as ag
from das_amd.datasets import SyntheticPoseDataset, collate
from das_amd.optim import FlatSGD, train_iteration

mode = sys.argv[1] if len(sys.argv) > 1 else 'own'
Ns = [int(a) for a in sys.argv[2:]] or [16, 32, 64]
lib = _lib.load()
B = 16
dev = torch.device('cuda', 0)
model = bench.build_model(dev, num_stages=4, train=True)
ds = SyntheticPoseDataset(num_joints=bench.J, img_shape=(bench.H, bench.W), length=B, seed=0)
data = collate([ds[i] for i in range(B)], device=dev)
opt = FlatSGD(model, lr=2e-3, momentum=0.9, weight_decay=1e-4, bias_lr_mult=2.0, bias_decay_mult=0.0, max_grad_norm=35.0)
side = torch.cuda.Stream()
FAMS = ('conv_wgrad_pp_kernel', 'conv_wgrad_kernel<bf16>', 'conv1x1_stream_kernel', 'conv3x3_c64_kernel', 'conv_glds4_kernel<pp>',
        'conv_glds3_kernel<pp>')


def occupy(n, usec):
    if n <= 0:
        return
    lds, thr = (160 * 1024 - 512, 256) if mode == 'own' else (4096, 256)
    _lib.check(lib.das_dev_occupy_cus(n, thr, lds, usec, C.c_void_p(side.cuda_stream)), 'das_dev_occupy_cus')


def run(n, reserve, steps=6):
    _lib.check(lib.das_tuning_set(b'comm.reserved_cus', reserve), 'tuning')
    for _ in range(2):
        train_iteration(model, opt, data, 2e-3)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for _ in range(steps):
        occupy(n, 130000)
        e0.record()
        train_iteration(model, opt, data, 2e-3)
        e1.record()
        torch.cuda.synchronize()      # (the occupiers outlive the step: drained before the next one starts)
        tot += e0.elapsed_time(e1)
    ms = tot / steps
    # family times under the same pressure (weight gradients on the main stream, events around every launch)
    side_was, ag.WGRAD_SIDE_STREAM = ag.WGRAD_SIDE_STREAM, False
    train_iteration(model, opt, data, 2e-3)
    torch.cuda.synchronize()
    ops.profile_begin()
    occupy(n, 160000)
    train_iteration(model, opt, data, 2e-3)
    torch.cuda.synchronize()
    fam = {}
    for ent in ops.profile_end():
        fam[ent[0]] = fam.get(ent[0], 0.0) + ent[2].elapsed_time(ent[3])
    ag.WGRAD_SIDE_STREAM = side_was
    return ms, fam


print(f'mode {mode}: occupiers hold {"a whole CU each (160 KiB LDS)" if mode == "own" else "4 KiB LDS / 256 threads each"}')
base, fam0 = run(0, 0)
print(f'N=0   reserve=0  : step {base:7.2f} ms   ' + '  '.join(f'{k.replace("conv_", "").replace("_kernel", "")} {fam0.get(k, 0):5.2f}' for k in FAMS))
for n in Ns:
    for reserve in ((0, n) if n != 16 else (0, 16, 32)):
        ms, fam = run(n, reserve)
        print(f'N={n:<3d} reserve={reserve:<3d}: step {ms:7.2f} ms   ' +
              '  '.join(f'{k.replace("conv_", "").replace("_kernel", "")} {fam.get(k, 0):5.2f}' for k in FAMS), flush=True)
_lib.check(lib.das_tuning_set(b'comm.reserved_cus', 0), 'tuning')
