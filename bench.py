"""bench.py — DAS hot-path throughput on MI355X.

Workloads (BASELINE.json `configs`):
  --workload train (default)  configs[2]/[3]: MSPN-50 4-stage + FPN + DASHead (J=15), bf16 activations with
                              f32 master weights, batch 16 x 3x512x832 synthetic frames + GT per GPU, one FULL
                              train step = forward + 4 losses + backward + gradient all-reduce (N>1) + clip + SGD.
  --workload infer            configs[1]: MSPN-50 1-stage, batch 8, forward + decode.
  --workload decode           configs[4]: exp_mupots geometry (1024x768, J=21), decode + OKS-NMS only, 512 images/step.
Inputs are resident in HBM when the timed region starts. N>1: one process per GPU (torchrun), independent
per-rank batches (weak scaling); training exchanges gradients with RCCL all-reduce, inference has no collective.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline     — the kernel family carrying the most algorithmic FLOPs, FLOPs / HIP-event time measured on the launch
                 stream against the dense MFMA peak (+ PMC traffic per launch from profiles/traffic.json)
  roofline_hbm — the largest HBM-bound family beside it (launch mix below the ridge), algorithmic bytes / time against
                 the HBM peak
  cpu_baseline — the CPU oracle (oracle/, "port") timed on this host on a bounded sample (N=1 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

H, W, J = 512, 832, 15
PEAK_BF16_TFLOPS = 2500.0   # dense MFMA bf16, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0       # HBM3E, same guide
PEAK_F32_TFLOPS = 157.3
# conv MACs x 2 per image at 512x832, J=15
FWD_GFLOP = {1: 227.0, 2: 326.9, 3: 426.8, 4: 526.7}
TRAIN_GFLOP = {k: 3 * v for k, v in FWD_GFLOP.items()}


def model_cfg(num_stages=1, dtype='bf16', norm='BN'):
    return dict(
        type='DAS', pretrained=None,
        backbone=dict(type='MSPN2', unit_channels=256, num_stages=num_stages, num_units=4, num_blocks=[3, 4, 6, 3],
                      norm_cfg=dict(type=norm), compute_dtype=dtype),
        neck=dict(type='FPN', in_channels=[256] * 4, out_channels=256, start_level=1, add_extra_convs='on_output',
                  num_outs=4, relu_before_extra_convs=True, norm_cfg=dict(type=norm)),
        bbox_head=dict(type='DASHead', num_classes=1, in_channels=256, feat_channels=256, stacked_convs=2,
                       strides=[8, 16, 32, 64], regress_ranges=((-1, 80), (80, 160), (160, 320), (320, 1e8)),
                       center_sample_radius=1.5, num_joints=J, depth_factor=20, z_norm=50, root_idx=2,
                       cls_branch=(256,), reg_branch=((256,),) * 4, centerness_on_reg=True, conv_bias=True,
                       dcn_on_last_conv=True,
                       recursive_update=dict(prev_loss=True, num_heads=4, in_channels=256, feat_channels=256,
                                             num_layers=1, dim=3, num_joints=J)),
        train_cfg=dict(code_weight=[1.0, 1.0, 1] + [2] * J * 6),
        test_cfg=dict(nms_across_levels=False, nms_pre=1000, nms_post=100, nms_thr=0.9, score_thr=0.07))


def build_model(dev, seed=0, dtype='bf16', num_stages=1, train=False, norm='BN'):
    import das_amd
    torch.manual_seed(seed)
    model = das_amd.build_model(model_cfg(num_stages, dtype, norm))
    model.init_weights()
    # random-init heads predict ~zero offsets; give the sampling / regression convs some spread so that
    # the deformable and resampling kernels see non-trivial coordinates, as a trained net would
    with torch.no_grad():
        for n, p in model.bbox_head.named_parameters():
            if 'conv_offset.weight' in n or 'sampling_offset.weight' in n or 'conv_poses.0.weight' in n:
                p.normal_(0, 0.02)
    return model.to(dev).train(train)


def calibrate_scores(model, img, metas, target=150):
    """Shift conv_cls.bias so that ~`target` locations per image pass score_thr."""
    with torch.no_grad():
        cls, pose, ctr = model.bbox_head(model.extract_feat(img))
        c = torch.cat([t.float().reshape(t.shape[0], -1) for t in cls], 1)
        k = torch.cat([t.float().reshape(t.shape[0], -1) for t in ctr], 1)
        lo, hi = -20.0, 20.0
        for _ in range(40):
            mid = 0.5 * (lo + hi)
            n = ((torch.sigmoid(c + mid) * torch.sigmoid(k)) > 0.07).float().sum(1).mean().item()
            lo, hi = (mid, hi) if n < target else (lo, mid)
        model.bbox_head.conv_cls.bias.add_(0.5 * (lo + hi))


def host_cpu_info():
    """(threads to use, description): physical cores this process may run on — logical CPUs in the affinity mask,
    capped by the cgroup CPU quota, divided by the SMT width — and the CPU model name. Read from /proc and /sys: this
    runs after the process has initialised the GPU, where starting a child process (lscpu) is not something to risk."""
    logical = len(os.sched_getaffinity(0))
    quota = None
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            q, per = f.read().split()
        if q != 'max':
            quota = max(1, int(float(q) / float(per)))
    except (OSError, ValueError):
        pass
    model, tpc = 'unknown CPU', 1
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    model = line.split(':', 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        cpu0 = min(os.sched_getaffinity(0))
        with open(f'/sys/devices/system/cpu/cpu{cpu0}/topology/thread_siblings_list') as f:
            sib = f.read().strip()
        n = 0
        for part in sib.split(','):       # "0,128" or "0-1"
            lo, _, hi = part.partition('-')
            n += int(hi or lo) - int(lo) + 1
        tpc = max(1, n)
    except (OSError, ValueError):
        pass
    cores = max(1, logical // tpc)
    if quota is not None:
        cores = min(cores, quota)
    desc = f'{model}, {logical} logical CPUs in the affinity mask, SMT x{tpc}' + \
        (f', cgroup quota {quota} CPUs' if quota is not None else ', no cgroup quota')
    return cores, desc


def cpu_baseline(workload, budget_s=30.0, full=False, threads=None):
    """The CPU oracle (a port of the reference's algorithm, oracle/) timed on this host.
    BASELINE.md section 4 protocol (tools/analysis_tools/benchmark.py:63-90 of the reference): all physical cores,
    5 warm-up iterations, mean wall-clock over >= 20 iterations — run with `full=True` (bench.py --cpu-baseline-full,
    minutes). The default bench line keeps to a bounded sample: 1 warm-up and as many iterations as fit in
    `budget_s` seconds (at least 2), and says so in `sample`.
      train: one optimisation step at B=2 — 4-stage forward, 4 losses, backward, clip_grad_norm_(35), SGD(momentum)
      infer: 1-stage forward + decode at B=2 (BASELINE configs[0])."""
    from oracle import backbone as ob, decode as od, head as oh, loss as ol
    import das_amd
    from das_amd.datasets import SyntheticPoseDataset
    torch.manual_seed(0)
    cores, desc = host_cpu_info()
    if threads:
        cores = threads
    torch.set_num_threads(cores)
    try:
        torch.set_num_interop_threads(1)
    except RuntimeError:
        pass   # (already started)
    B = 2
    stages = 4 if workload == 'train' else 1
    model = das_amd.build_model(model_cfg(stages, 'f32'))
    model.init_weights()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    sd['bbox_head.conv_cls.bias'] += 3.0
    params = []
    if workload == 'train':
        for k, v in sd.items():
            if v.is_floating_point() and 'running' not in k and not k.endswith('.mask'):
                v.requires_grad_(True)
                params.append(v)
    bsd = {k[9:]: v for k, v in sd.items() if k.startswith('backbone.')}
    nsd = {k[5:]: v for k, v in sd.items() if k.startswith('neck.')}
    hsd = {k[10:]: v for k, v in sd.items() if k.startswith('bbox_head.')}
    hcfg = dict(num_joints=J, root_idx=2, depth_factor=20, z_norm=50, strides=[8, 16, 32, 64], stacked_convs=2,
                num_heads=4, num_layers=1, regress_ranges=((-1, 80), (80, 160), (160, 320), (320, 1e8)),
                code_weight=[1.0, 1.0, 1] + [2] * J * 6, prev_loss=True)
    tcfg = model_cfg()['test_cfg']
    ds = SyntheticPoseDataset(num_joints=J, img_shape=(H, W), length=4, seed=0)
    ss = [ds[i] for i in range(B)]
    img = torch.stack([s['img'] for s in ss])
    metas = [dict(scale_factor=np.ones(4, dtype=np.float32), filename='')] * B
    gts = {k: [s[k] for s in ss] for k in ('gt_labels_3d', 'gt_poses_3d', 'centers2d', 'depths')}
    sgd = torch.optim.SGD(params, lr=2e-3, momentum=0.9, weight_decay=1e-4) if params else None

    if workload != 'train':  # same candidate load as the GPU run: ~150 locations above score_thr per image
        with torch.no_grad():
            feats = ob.fpn_forward(nsd, ob.mspn2_forward(bsd, img, 1, (3, 4, 6, 3)))
            c, p, k = oh.head_forward(hsd, feats, hcfg, '', False)
            cc = torch.cat([t.reshape(B, -1) for t in c], 1)
            kk = torch.cat([t.reshape(B, -1) for t in k], 1)
            lo, hi = -20.0, 20.0
            for _ in range(40):
                mid = 0.5 * (lo + hi)
                n = ((torch.sigmoid(cc + mid) * torch.sigmoid(kk)) > 0.07).float().sum().item() / B
                lo, hi = (mid, hi) if n < 150 else (lo, mid)
            hsd['conv_cls.bias'] += 0.5 * (lo + hi)

    def step():
        if workload == 'train':
            sgd.zero_grad(set_to_none=True)
            feats = ob.fpn_forward(nsd, ob.mspn2_forward(bsd, img, stages, (3, 4, 6, 3), train=True), train=True)
            outs = oh.head_forward(hsd, feats, hcfg, '', True)
            sum(ol.head_loss(hsd, '', *outs, gts, hcfg).values()).backward()
            torch.nn.utils.clip_grad_norm_([p for p in params if p.grad is not None], 35.0)
            sgd.step()
            return None
        with torch.no_grad():
            feats = ob.fpn_forward(nsd, ob.mspn2_forward(bsd, img, 1, (3, 4, 6, 3)))
            c, p, k = oh.head_forward(hsd, feats, hcfg, '', False)
            return od.get_poses(c, p, k, metas, J, hcfg['strides'], tcfg)
    # Protocol: BASELINE.md section 4 (5 warm-ups, mean of 20 iterations) whenever it fits `full_fit_s` seconds on this
    # host — judged from the first (cold) iteration —, else a bounded sample that says so in `protocol`.
    full_fit_s = 120.0
    t0 = time.perf_counter()
    step()
    first = time.perf_counter() - t0          # (cold: allocator, thread pool, page faults — typically twice a steady one)
    done = 1
    if not full:
        t0 = time.perf_counter()
        step()
        second = time.perf_counter() - t0
        done = 2
        if first + second + 23 * second <= full_fit_s:
            full = True
    warmups = 5 if full else done
    for _ in range(warmups - done):
        step()
    warm = first
    t0, n = time.perf_counter(), 0
    while True:
        step()
        n += 1
        el = time.perf_counter() - t0
        if (full and n >= 20) or (not full and n >= 2 and (el + warm > budget_s or n >= 20)):
            break
    dt = time.perf_counter() - t0
    what = ('4-stage forward + 4 losses + backward + clip_grad_norm_(35) + SGD(momentum 0.9, wd 1e-4) step'
            if workload == 'train' else '1-stage forward + decode')
    proto = ('BASELINE.md section 4 protocol: 5 warm-ups, mean of 20 iterations' if full else
             f'bounded sample: {warmups} warm-up, mean of {n} iterations (time cap {budget_s:.0f} s; the full 5 + 20 '
             f'protocol did not fit {full_fit_s:.0f} s on this host — first iteration {first:.1f} s, second {second:.1f} s —, `bench.py '
             f'--cpu-baseline-full` forces it)')
    return dict(value=round(B * n / dt, 4), unit='img/s', cores=cores, kind='port', protocol='full' if full else 'bounded',
                warmups=warmups, iterations=n,
                sample=f'{n} x (batch {B} x 3 x {H} x {W}) {what}; CPU oracle fp32, torch {torch.__version__}, '
                       f'{cores} threads (intra-op), 1 inter-op; host: {desc}; {proto}')


def _tbytes(t):
    """bytes of a tensor / Ragged / list of them (0 for anything else)"""
    if t is None:
        return 0
    if hasattr(t, 'sizes') and hasattr(t, 'data'):
        t = t.data
    if isinstance(t, torch.Tensor):
        return t.numel() * t.element_size()
    if isinstance(t, (list, tuple)):
        return sum(_tbytes(u) for u in t)
    return 0


# The families beside the convolutions and the BatchNorm passes (which das_amd.ops times itself): ops.<name> ->
# (family, algorithmic bytes of a call from its arguments and result — every operand of every pass once).
def _io(a, k, out):
    return sum(_tbytes(t) for t in a) + sum(_tbytes(t) for t in k.values()) + _tbytes(out)


OTHER_FAMILIES = {
    # DCNv2 sampling (the GEMM half runs on the conv kernels): x, offsets / masks -> col; backward: dcol, x, offsets twice
    # (the input-gradient gather and the offset-gradient kernel each read them) -> dx, d offsets
    'deform_im2col3x3': ('dcnv2 sampling (im2col, col2im gather, offset gradient)', _io),
    'deform_im2col3x3_backward': ('dcnv2 sampling (im2col, col2im gather, offset gradient)',
                                  lambda a, k, out: 2 * _io(a, k, None) - _tbytes(a[2]) + _tbytes(out)),
    # GroupNorm(32) + ReLU: statistics pass + apply pass (2R + 1W); backward reduce + apply (6R + 1W)
    'groupnorm': ('groupnorm (stats + apply, backward reduce + apply)', lambda a, k, out: 3 * _tbytes(a[0])),
    'groupnorm_backward': ('groupnorm (stats + apply, backward reduce + apply)', lambda a, k, out: 7 * _tbytes(a[2])),
    'maxpool3x3s2': ('elementwise (pool, upsampling, skip adds, layout, bias sums)', _io),
    'maxpool3x3s2_backward': ('elementwise (pool, upsampling, skip adds, layout, bias sums)', _io),
    'upsample_bilinear_ac': ('elementwise (pool, upsampling, skip adds, layout, bias sums)', _io),
    'upsample_bilinear_ac_backward': ('elementwise (pool, upsampling, skip adds, layout, bias sums)', _io),
    'add_upsample_nearest': ('elementwise (pool, upsampling, skip adds, layout, bias sums)', _io),
    'upsample_nearest_backward': ('elementwise (pool, upsampling, skip adds, layout, bias sums)', _io),
    'add3': ('elementwise (pool, upsampling, skip adds, layout, bias sums)', _io),
    'colsum': ('elementwise (pool, upsampling, skip adds, layout, bias sums)', _io),
    'pack_image': ('elementwise (pool, upsampling, skip adds, layout, bias sums)', _io),
    'to_nchw_f32': ('elementwise (pool, upsampling, skip adds, layout, bias sums)', _io),
    'offset_sample': ('head elementwise (offset_sample, blend, assemble / finalize)', _io),
    'offset_sample_backward': ('head elementwise (offset_sample, blend, assemble / finalize)', _io),
    'sigmoid_blend': ('head elementwise (offset_sample, blend, assemble / finalize)', _io),
    'sigmoid_blend_backward': ('head elementwise (offset_sample, blend, assemble / finalize)', _io),
    'head_assemble': ('head elementwise (offset_sample, blend, assemble / finalize)', _io),
    'head_assemble_backward': ('head elementwise (offset_sample, blend, assemble / finalize)', _io),
    'head_finalize': ('head elementwise (offset_sample, blend, assemble / finalize)', _io),
    'pack_conv_weights': ('optimizer (weight packing, clip norm, SGD)', lambda a, k, out: _tbytes(a[0]) + _tbytes(a[1]) + _tbytes(a[2])),
}
# C entry points without a tensor-level wrapper in ops (das_amd.train_ops calls them): timed at the ctypes boundary.
# bytes: (index of the element-count argument, bytes per element) or None (latency-bound kernels over ~10^4 rows).
OTHER_C = {
    'das_grad_sumsq': ('optimizer (weight packing, clip norm, SGD)', (1, 4)),
    'das_sgd_momentum_step': ('optimizer (weight packing, clip norm, SGD)', (3, 20)),      # p, g, m read; p, m written
    'das_assign_targets': ('losses (targets, focal / L1 / BCE, RealNVP, RLE)', None),
    'das_positive_rows': ('losses (targets, focal / L1 / BCE, RealNVP, RLE)', None),
    'das_sigmoid_focal_loss': ('losses (targets, focal / L1 / BCE, RealNVP, RLE)', None),
    'das_smooth_l1_loss': ('losses (targets, focal / L1 / BCE, RealNVP, RLE)', None),
    'das_bce_logits_loss': ('losses (targets, focal / L1 / BCE, RealNVP, RLE)', None),
    'das_realnvp_log_prob_multi': ('losses (targets, focal / L1 / BCE, RealNVP, RLE)', None),
    'das_realnvp_log_prob_multi_backward': ('losses (targets, focal / L1 / BCE, RealNVP, RLE)', None),
    'das_rle_prepare': ('losses (targets, focal / L1 / BCE, RealNVP, RLE)', None),
    'das_rle_loss': ('losses (targets, focal / L1 / BCE, RealNVP, RLE)', None),
    'das_rle_backward': ('losses (targets, focal / L1 / BCE, RealNVP, RLE)', None),
}


def instrument_other_families(ops):
    """Every call of the op wrappers / C entry points above becomes a PROFILE entry while ops.PROFILE is a list: the span of
    the library's own event pairs (das_prof_*, recorded inside the entry point around its launches) the call produced;
    entries use the layout of das_amd.ops._timed. Idempotent."""
    if getattr(ops, '_bench_instrumented', False):
        return
    ops._bench_instrumented = True
    from das_amd import _lib

    def timed(fn, family, nbytes):
        def call(*a, **k):
            if ops.PROFILE is None:
                return fn(*a, **k)
            i0 = ops._prof_mark()
            out = fn(*a, **k)
            sp = ops._Span(i0, ops._prof_mark())
            ops.PROFILE.append((family, 0.0, sp, sp, ('other',), 1, float(nbytes(a, k, out)), 1))
            return out
        return call
    for name, (family, nb) in OTHER_FAMILIES.items():
        setattr(ops, name, timed(getattr(ops, name), family, nb))
    lib = _lib.load()
    for name, (family, spec) in OTHER_C.items():
        nb = (lambda a, k, out, spec=spec: (int(getattr(a[spec[0]], 'value', a[spec[0]])) * spec[1]) if spec else 0.0)
        setattr(lib, name, timed(getattr(lib, name), family, nb))


PRICED_REPS = 5            # repetitions of the per-launch pass; per launch index the MEDIAN is priced (the sum of per-launch
                           # minima understates every single step: it made the self-check easier and fraction_of_roof kinder)
PRICED_FIT = 1.02          # families_ms_sum must fit into the passes' MEDIAN step time within this factor
PRICED_PASS_OVER_STEP = 1.15   # ... and the pass's step time must stay within this factor of the timed region's
PRICED_RETRIES = 3


def _priced_pass(ops, run_step, reps):
    """`reps` single steps, each between das_prof_begin / das_prof_end. Returns (wall ms per step, per step the list of
    (tag, flops, ms, ops, algorithmic bytes, launches))."""
    walls, passes = [], []
    for _ in range(reps):
        torch.cuda.synchronize()
        w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ops.profile_begin()
        w0.record()
        run_step()
        w1.record()
        torch.cuda.synchronize()
        ents = ops.profile_end()
        walls.append(w0.elapsed_time(w1))
        passes.append([(e[0], e[1], e[2].elapsed_time(), e[5] if len(e) > 5 else 1, e[6] if len(e) > 6 else 0.0,
                        e[7] if len(e) > 7 else 1) for e in ents])
    return walls, passes


def _fold_passes(passes):
    """Per family [flops, seconds (MEDIAN per launch index over the passes), ops, bytes, launches, seconds (minimum)] of ONE
    step. The launch sequence of a step is deterministic; should two passes disagree on it, the families' sums are
    compared instead (aligned = False)."""
    import statistics
    aligned = all(len(p) == len(passes[0]) and all(x[0] == y[0] for x, y in zip(p, passes[0])) for p in passes[1:])
    fam = {}
    if aligned:
        for i, ent in enumerate(passes[0]):
            ts = [p[i][2] for p in passes]
            f = fam.setdefault(ent[0], [0.0, 0.0, 0, 0.0, 0, 0.0])
            f[0] += ent[1]
            f[1] += statistics.median(ts) * 1e-3
            f[2] += ent[3]
            f[3] += ent[4]
            f[4] += ent[5]
            f[5] += min(ts) * 1e-3
    else:
        per = []
        for p in passes:
            d = {}
            for ent in p:
                f = d.setdefault(ent[0], [0.0, 0.0, 0, 0.0, 0, 0.0])
                f[0] += ent[1]; f[1] += ent[2] * 1e-3; f[2] += ent[3]; f[3] += ent[4]; f[4] += ent[5]
            per.append(d)
        for k in per[0]:
            have = [d[k] for d in per if k in d]
            mid = sorted(have, key=lambda f: f[1])[len(have) // 2]
            fam[k] = mid[:5] + [min(f[1] for f in have)]
    return fam, aligned


def roofline_from_profile(ops, run_step, dtype, ms_per_step=None, reps=PRICED_REPS):
    """Event pairs recorded INSIDE the library around every launch of the conv families (forward, data gradient, the two
    weight-gradient kernel classes), of the BatchNorm passes and of every other HIP op of the step (OTHER_FAMILIES /
    OTHER_C) — das_prof_*, on the stream each launch goes to, nothing of the interpreter between an event and its launch.
    `reps` single-step passes; a launch is priced at its MEDIAN over the passes (minimum reported beside it). The pass
    checks itself: the families must fit into the passes' median step time (x PRICED_FIT) and that step time must stay within
    PRICED_PASS_OVER_STEP of the timed region's; otherwise the pass is repeated (PRICED_RETRIES times) and, failing
    that, `priced_step.unreliable` is set and the headline comes from roofline_hbm / roofline_mfma.
    Returns (roofline, roofline_mfma, roofline_hbm, roofline_bn, priced):
      roofline       the family with the LARGEST TIME in the step (selection rule stated in the object; BatchNorm counts
                     per pass type there, roofline_bn has the passes together), priced against the roof its own launch
                     mix sits under: FLOP per algorithmic byte above the ridge (peak FLOP/s / peak B/s = 312 for bf16)
                     -> dense MFMA peak, below -> HBM peak;
      roofline_mfma  the family that carries the most algorithmic FLOPs among those above the ridge;
      roofline_hbm   the conv family with the largest time among those below the ridge;
      roofline_bn    the BatchNorm passes together (apply, backward apply, backward reduce + apply), HBM-bound;
      priced         every millisecond of the measured step: ms per family (conv families, BatchNorm, DCNv2 sampling,
                     GroupNorm, elementwise, head elementwise, losses, optimizer) + `torch glue + launch gaps` = the
                     step's wall time in the same pass minus the families' sum."""
    # Kernel quality is measured with the kernels running one at a time: the weight gradients' side stream (which
    # overlaps them with the main stream in the timed region) is switched off for these passes, otherwise a launch's
    # event-to-event time would include whatever ran beside it.
    import statistics
    from das_amd import autograd as ag
    instrument_other_families(ops)
    side_was, ag.WGRAD_SIDE_STREAM = ag.WGRAD_SIDE_STREAM, False
    run_step()          # (untimed: the first step of this stream layout allocates its workspaces)
    attempts, problems = 0, []
    try:
        while True:
            attempts += 1
            walls, passes = _priced_pass(ops, run_step, reps)
            if not passes or not passes[0]:
                return None, None, None, None, None
            fam, aligned = _fold_passes(passes)
            wall_ms, wall_min = statistics.median(walls), min(walls)
            fam_sum = sum(f[1] for f in fam.values()) * 1e3
            bad = []
            if fam_sum > PRICED_FIT * wall_ms:
                bad.append(f'families {fam_sum:.2f} ms > {PRICED_FIT} x pass step {wall_ms:.2f} ms')
            if ms_per_step is not None and wall_ms > PRICED_PASS_OVER_STEP * ms_per_step:
                bad.append(f'pass step {wall_ms:.2f} ms > {PRICED_PASS_OVER_STEP} x timed step {ms_per_step:.2f} ms')
            # several ranks: every pass is a sequence of collectives (the steps' gradient all-reduces), so the ranks must agree
            # on whether another one runs — any rank's failed self-check repeats the pass on all of them
            again = bool(bad) and attempts <= PRICED_RETRIES
            if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
                flag = torch.tensor([1.0 if again else 0.0], device='cuda')
                torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
                again = bool(flag.item() > 0) and attempts <= PRICED_RETRIES
            if not again:
                break
            problems.append('; '.join(bad) if bad else 'repeated with the other ranks')
    finally:
        ag.WGRAD_SIDE_STREAM = side_was
    unreliable = bool(bad)
    if bad:
        problems.append('; '.join(bad))
    peak = PEAK_BF16_TFLOPS if dtype == 'bf16' else PEAK_F32_TFLOPS
    ridge = peak * 1e12 / (PEAK_HBM_GBS * 1e9)    # FLOP per byte above which a launch mix is matrix-core bound

    def entry(tag, v, rule):
        fl, sec, nops, by, nl = v[:5]
        mfma = by > 0 and fl / by >= ridge
        ach = fl / sec / 1e12 if mfma else by / sec / 1e9
        pk = peak if mfma else PEAK_HBM_GBS
        return dict(bound='mfma' if mfma else 'hbm', kernel=tag, achieved=round(ach, 2 if mfma else 1), peak=pk,
                    unit='TFLOP/s' if mfma else 'GB/s', frac=round(ach / pk, 4), traffic=None, selection=rule,
                    launches_per_step=nl, ops_per_step=nops, avg_launch_us=round(sec / max(nl, 1) * 1e6, 2),
                    family_ms_per_step=round(sec * 1e3, 3), family_ms_per_step_min=round(v[5] * 1e3, 3),
                    algorithmic_mb_per_launch=round(by / max(nl, 1) / 1e6, 2),
                    flop_per_byte=round(fl / max(by, 1.0), 1), ridge_flop_per_byte=round(ridge, 1),
                    tflops=round(fl / sec / 1e12, 2), gbs=round(by / sec / 1e9, 1))
    other_names = {f for f, _ in OTHER_FAMILIES.values()} | {f for f, _ in OTHER_C.values()}
    conv = {k: v for k, v in fam.items() if v[0] > 0}
    bn = {k: v for k, v in fam.items() if v[0] == 0 and k not in other_names}
    ranked = {k: v for k, v in fam.items() if k not in other_names}
    above = {k: x for k, x in conv.items() if x[3] > 0 and x[0] / x[3] >= ridge}
    below = {k: x for k, x in conv.items() if x[3] > 0 and x[0] / x[3] < ridge}
    roof_mfma = roof_hbm = roof_bn = None
    if above:
        t, x = max(above.items(), key=lambda kv: kv[1][0])
        roof_mfma = entry(t, x, 'most algorithmic FLOPs among the families above the ridge')
    if below:
        t, x = max(below.items(), key=lambda kv: kv[1][1])
        roof_hbm = entry(t, x, 'largest time among the conv families below the ridge')
    if bn:
        tot = [sum(x[i] for x in bn.values()) for i in range(6)]
        roof_bn = entry(' + '.join(sorted(bn)), tot, 'all BatchNorm passes of the step (forward apply, backward apply, '
                        'backward reduce + apply; the fused forms count under the pass they replace: upsample-unit merge, '
                        'cross-stage skip add, projection-shortcut dual apply); algorithmic bytes = every operand of every '
                        'pass once')
    if unreliable and (roof_hbm or roof_mfma):
        # a pass that does not fit its own step must not pick the headline: fall back to the two per-roof entries
        cands = [r for r in (roof_hbm, roof_mfma) if r is not None]
        roof = dict(max(cands, key=lambda r: r['family_ms_per_step']))
        roof['selection'] = ('priced_step.unreliable: the per-launch pass failed its self-check, so the headline is the '
                             'larger of roofline_hbm / roofline_mfma instead of the family with the largest time')
    else:
        tag, v = max(ranked.items(), key=lambda kv: kv[1][1])
        roof = entry(tag, v, 'largest time per step among the conv families and the BatchNorm passes; BatchNorm is ranked '
                     'per pass type here (forward apply / backward apply / backward reduce + apply) — together the passes '
                     'are roofline_bn, which may exceed this family')
    roof['all_families'] = {k: dict(tflops=round(x[0] / x[1] / 1e12, 2), gbs=round(x[3] / x[1] / 1e9, 1),
                                    flop_per_byte=round(x[0] / max(x[3], 1.0), 1), ms_per_step=round(x[1] * 1e3, 3),
                                    ms_per_step_min=round(x[5] * 1e3, 3), launches=x[4], ops=x[2])
                            for k, x in fam.items()}
    # every millisecond of the step: families + the rest (ATen glue launches, launch gaps, event overhead of this pass)
    fam_ms = {k: x[1] * 1e3 for k, x in fam.items()}
    at_roof = 0.0
    for k, x in fam.items():
        at_roof += max(x[0] / (peak * 1e12), x[3] / (PEAK_HBM_GBS * 1e9)) * 1e3
    priced = dict(step_ms_this_pass=round(wall_ms, 3), step_ms_this_pass_min=round(wall_min, 3),
                  reps=reps, attempts=attempts, launch_sequences_aligned=aligned, unreliable=unreliable,
                  checks=dict(families_fit_step=f'families_ms_sum <= {PRICED_FIT} x step_ms_this_pass',
                              pass_vs_timed_step=f'step_ms_this_pass <= {PRICED_PASS_OVER_STEP} x ms_per_step',
                              timed_ms_per_step=None if ms_per_step is None else round(ms_per_step, 3),
                              failures=problems),
                  note='weight gradients on the main stream; one HIP event pair per C entry point, recorded inside the '
                       'library right around its launches (das_prof_*); each launch priced at its MEDIAN over the '
                       'repetitions, step_ms_this_pass = the median pass (families_ms_min: the minima instead); every family at max(FLOPs / MFMA peak, '
                       'algorithmic bytes / HBM peak) gives families_ms_at_roof',
                  families_ms={k: round(v, 3) for k, v in sorted(fam_ms.items(), key=lambda kv: -kv[1])},
                  families_ms_min={k: round(x[5] * 1e3, 3) for k, x in sorted(fam.items(), key=lambda kv: -kv[1][1])},
                  families_ms_sum=round(sum(fam_ms.values()), 3), families_ms_at_roof=round(at_roof, 3),
                  fraction_of_roof=round(at_roof / max(sum(fam_ms.values()), 1e-9), 4),
                  torch_glue_and_gaps_ms=round(wall_ms - sum(fam_ms.values()), 3))
    return roof, roof_mfma, roof_hbm, roof_bn, priced


def attach_traffic(roof, workload, batch):
    """roofline.traffic: HBM bytes per launch of the dominant kernel family from the PMC passes (FETCH_SIZE x 2 +
    WRITE_SIZE; collected in separate rocprofv3 --pmc runs, tools/dev/scripts/pmc_*.sh -> profiles/traffic.json).
    Only a measurement taken on this workload's shapes and batch size is used; anything else stays null."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'traffic.json')
    default_batch = {'train': 16, 'infer': 8}[workload]
    try:
        with open(path) as f:
            entry = json.load(f)['families'][roof['kernel']][workload]
    except (OSError, KeyError, ValueError):
        return
    if batch != default_batch:
        return
    roof['traffic'] = round(entry['hbm_mb_per_launch'] * 1e6)
    roof['traffic_unit'] = 'bytes per launch (HBM, PMC)'
    roof['traffic_source'] = entry['source']
    if 'algorithmic_mb_per_launch_same_launches' in entry:
        # the counters ran over the family's shapes with plain operands (x, w -> y); the step's own launches also read
        # residuals / BatchNorm-backward operands, so `traffic` is to be compared with THIS figure, not with the line's
        # algorithmic_mb_per_launch
        roof['traffic_algorithmic'] = round(entry['algorithmic_mb_per_launch_same_launches'] * 1e6)
        roof['traffic_over_algorithmic'] = entry['hbm_over_algorithmic']
    elif roof.get('algorithmic_mb_per_launch'):
        # (round 4: the counters ran over the step's OWN launches, so the line's algorithmic bytes are the comparable figure)
        roof['traffic_over_algorithmic'] = round(roof['traffic'] / (roof['algorithmic_mb_per_launch'] * 1e6), 3)


def decode_workload(args, rank, world, dev, quiet=False, batch=None, steps=None, warmup=None, cpu_leg=True):
    """BASELINE configs[4]: exp_mupots geometry (1024x768 input, J=21, strides 8..64 -> 16 320 locations per
    image), decode only: score / threshold / per-level top-k / OKS-NMS of `batch` images per step (one workgroup
    per image). Synthetic eval-mode head outputs, calibrated to ~150 candidates per image above score_thr."""
    import torch.distributed as dist
    from das_amd import ops
    Jm, HW = 21, [(96, 128), (48, 64), (24, 32), (12, 16)]
    B = batch or args.batch or 512
    steps = steps or args.steps or 20
    warmup = warmup if warmup is not None else (args.warmup if args.warmup is not None else 3)
    g = torch.Generator(device='cpu').manual_seed(100 + rank)
    cls, ctr, pose = [], [], []
    for h, w in HW:
        cls.append(torch.randn(B, h, w, 1, generator=g))
        ctr.append(torch.randn(B, h, w, 1, generator=g) + 0.5)
        p = torch.randn(B, h, w, 3 + 6 * Jm, generator=g)   # (depth, offsets, uvd, sigma) as the eval head emits
        p[..., 3:3 + 3 * Jm] *= 30.0
        p[..., 2] = p[..., 2].abs() * 0.2 + 0.2
        pose.append(p)
    # shift the class logits so that ~150 locations per image pass score_thr = 0.07
    sc = torch.cat([(torch.sigmoid(c) * torch.sigmoid(t)).reshape(B, -1) for c, t in zip(cls, ctr)], 1)
    lo, hi = -12.0, 4.0
    for _ in range(30):
        mid = 0.5 * (lo + hi)
        n = float(sum(((torch.sigmoid(c + mid) * torch.sigmoid(t)) > 0.07).sum() for c, t in zip(cls, ctr))) / B
        lo, hi = (mid, hi) if n < 150 else (lo, mid)
    del sc
    cls = [(c + lo).to(dev).contiguous() for c in cls]
    ctr = [t.to(dev).contiguous() for t in ctr]
    pose = [p.to(dev).contiguous() for p in pose]
    sf = torch.ones(B, 2, dtype=torch.float32, device=dev)
    strides = [8, 16, 32, 64]

    def step():
        return ops.decode(cls, ctr, pose, strides, sf, Jm, 1000, 100, 0.07, 0.9)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(warmup):
        out = step()
    sync_all()
    evs = []
    t0 = time.perf_counter()
    for _ in range(steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = step()
        e1.record()
        evs.append((e0, e1))
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    if rank != 0:
        return
    kern_s = sum(a.elapsed_time(b) for a, b in evs) * 1e-3 / steps     # events bracket the launch on its stream
    nloc = sum(h * w for h, w in HW)
    # Bytes the path REQUIRES per image: the class and centerness logit of every location (the pose maps are only
    # gathered at the candidates that pass the threshold: 3 + 6 J floats each), and the kept poses written. SURVEY 8(d)'s
    # figure (every pose value of every location, 3.36 MB at J = 15) is what the reference's dense tensor code touches,
    # not what the decode needs: priced on it this latency-bound kernel looked like 79 % of the HBM peak.
    cand = int(out['ncand'].float().mean().item()) if 'ncand' in out else 150
    poses = int(out['count'].sum().item())
    bytes_img = nloc * 2 * 4 + cand * (3 + 6 * Jm) * 4 + (poses / B) * (3 * Jm + 3 + 1) * 4
    # the kernel is one workgroup per image (a sort / NMS latency chain): per-image time at a batch that fills the chip
    # (B) and at the serving batch of configs[1] (8 images: 8 of 256 CUs busy)
    small = [t[:8] for t in cls], [t[:8] for t in ctr], [t[:8] for t in pose]
    for _ in range(3):
        ops.decode(small[0], small[1], small[2], strides, sf[:8], Jm, 1000, 100, 0.07, 0.9)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.decode(small[0], small[1], small[2], strides, sf[:8], Jm, 1000, 100, 0.07, 0.9)
    e1.record()
    torch.cuda.synchronize()
    us_b8 = e0.elapsed_time(e1) / 20 * 1e3
    line = {
        'metric': 'imgs/sec decode', 'value': round(B * world * steps / dt, 1), 'unit': 'img/s', 'n_gpus': world,
        'steps': steps, 'warmup': warmup, 'ms_per_step': round(dt / steps * 1e3, 3), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f'BASELINE configs[4]: exp_mupots geometry, 1024x768 input, J={Jm}, {nloc} locations/img, '
                               f'decode only (threshold, top-k 1000 per level, OKS-NMS 0.9, keep 100), batch {B} per GPU',
                   'per_gpu_batch': B, 'global_batch': B * world, 'parallelism': f'replicas{world}'},
        'poses_per_sec': round(poses * world * steps / dt, 1), 'poses_per_step_rank0': poses,
        'roofline': {'bound': 'hbm', 'kernel': 'decode_kernel', 'achieved': round(bytes_img * B / kern_s / 1e9, 1),
                     'peak': 8000.0, 'unit': 'GB/s', 'frac': round(bytes_img * B / kern_s / 8e12, 4), 'traffic': None,
                     'avg_launch_us': round(kern_s * 1e6, 1), 'required_bytes_per_img': int(bytes_img),
                     'us_per_img': round(kern_s * 1e6 / B, 3), 'launch_us_at_batch_8': round(us_b8, 1),
                     'us_per_img_at_batch_8': round(us_b8 / 8, 2),
                     'note': 'one 1024-thread workgroup per image: a sort / NMS latency chain, not a streaming kernel; '
                             'bytes = 2 logits per location + the gathered candidates + the kept poses'},
    }
    if world == 1 and not args.no_cpu_baseline and cpu_leg:
        from oracle import decode as OD
        n, t0 = 0, time.perf_counter()
        while n < 4 or (time.perf_counter() - t0 < 10.0 and n < 64):
            b = n % B
            OD.get_poses([c[b:b + 1].permute(0, 3, 1, 2).cpu() for c in cls],
                         [p[b:b + 1].permute(0, 3, 1, 2).cpu() for p in pose],
                         [t[b:b + 1].permute(0, 3, 1, 2).cpu() for t in ctr],
                         [dict(scale_factor=np.ones(4, dtype=np.float32), filename='')], Jm, strides,
                         dict(nms_pre=1000, nms_post=100, nms_thr=0.9, score_thr=0.07))
            n += 1
        line['cpu_baseline'] = dict(value=round(n / (time.perf_counter() - t0), 2), unit='img/s', cores=1, kind='port',
                                    sample=f'{n} images, CPU oracle decode + OKS-NMS (numpy / python), 1 thread')
    if quiet:
        return line
    print(json.dumps(line), flush=True)
    return line


def also_workloads(args, dev, model, opt, data):
    """The two other workloads of BASELINE.json, measured in this process after the train loop (about a second of GPU
    time each) so that the driver's one JSON line carries them too: `infer` = configs[1] (1-stage, B = 8, forward +
    decode, 10 timed steps after 3), `decode` = configs[4] geometry (512 images per step, 10 timed steps after 3). Same
    code paths as `--workload infer` / `--workload decode` (the train model stays resident: 35 of 288 GB)."""
    import gc
    out = {}
    try:
        gc.collect()
        from das_amd.datasets import SyntheticPoseDataset, collate
        m1 = build_model(dev, seed=0, dtype=args.dtype, num_stages=1, train=False)
        ds = SyntheticPoseDataset(num_joints=J, img_shape=(H, W), length=8, seed=0)
        d8 = collate([ds[i] for i in range(8)], device=dev)
        calibrate_scores(m1, d8['img'], d8['img_metas'])
        n = 10

        def timed_infer():
            for _ in range(3):
                r = m1(d8['img'], d8['img_metas'], return_loss=False, rescale=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                r = m1(d8['img'], d8['img_metas'], return_loss=False, rescale=True)
            torch.cuda.synchronize()
            return time.perf_counter() - t0, r
        dt_eager, res = timed_infer()
        graph_note = 'off'
        dt = dt_eager
        if not getattr(args, 'no_infer_graph', False):
            # the fixed-shape eval forward (backbone + neck + head) as one hipGraph, the decode eager behind it
            # (das_amd/graphs.py GraphedInference): what a serving loop of fixed-size batches runs
            try:
                from das_amd.graphs import enable_inference_graph
                enable_inference_graph(m1, d8['img'])
                dt, res = timed_infer()
                graph_note = 'forward (backbone + neck + head) replayed as one hipGraph, decode eager'
            except Exception as e:
                m1._graphed_infer = None
                graph_note = f'off (capture failed: {type(e).__name__}: {e})'
        out['infer'] = dict(metric='imgs/sec forward+decode', value=round(8 * n / dt, 1), unit='img/s',
                            ms_per_step=round(dt / n * 1e3, 3), steps=n, warmup=3, dtype=args.dtype,
                            hip_graphs=graph_note, eager_ms_per_step=round(dt_eager / n * 1e3, 3),
                            model_tflops=round(8 * n * FWD_GFLOP[1] / dt / 1e3, 1),
                            poses_per_step=sum(len(r['scores']) for r in res),
                            workload='BASELINE configs[1]: MSPN-50 1-stage + FPN + DASHead J=15, batch 8 x 3x512x832, '
                                     'forward + decode')
        del m1, d8, res
        torch.cuda.empty_cache()
        line = decode_workload(args, 0, 1, dev, quiet=True, batch=512, steps=10, warmup=3, cpu_leg=False)
        if line:
            out['decode'] = dict(metric=line['metric'], value=line['value'], unit=line['unit'], ms_per_step=line['ms_per_step'],
                                 steps=10, warmup=3, poses_per_sec=line['poses_per_sec'],
                                 us_per_img=line['roofline']['us_per_img'],
                                 us_per_img_at_batch_8=line['roofline']['us_per_img_at_batch_8'],
                                 workload=line['config']['workload'])
    except Exception as e:   # the train line must not be lost to a failure here
        out['error'] = f'{type(e).__name__}: {e}'
    return out


def self_launch(n):
    """`python bench.py --gpus N` without torchrun: start one child process per GPU (the reference's
    tools/dist_train.sh:8-9 does the same through torch.distributed.launch) with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set, relay their output (rank 0 prints the JSON line) and return the worst exit code. The parent never
    touches the GPU; children are separate processes (no exec of a GPU-initialised process)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        live = list(procs)
        while live:
            time.sleep(0.2)
            for pr in list(live):
                code = pr.poll()
                if code is None:
                    continue
                live.remove(pr)
                rc = max(rc, abs(code))
            if rc and live:          # a rank died: the others would wait for it in the next collective
                for pr in live:
                    pr.kill()
                for pr in live:
                    pr.wait()
                live = []
    except BaseException:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
        raise
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None)
    ap.add_argument('--warmup', type=int, default=None)
    ap.add_argument('--workload', default='train', choices=['train', 'infer', 'decode'])
    ap.add_argument('--batch', type=int, default=None, help='images per GPU per step (train 16, infer 8)')
    ap.add_argument('--stages', type=int, default=None, help='MSPN stages (train 4, infer 1)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-baseline-full', action='store_true',
                    help='BASELINE.md section 4 protocol for the CPU leg: 5 warm-ups + 20 iterations (minutes)')
    ap.add_argument('--cpu-threads', type=int, default=None, help='override the CPU leg\'s thread count')
    ap.add_argument('--cpu-baseline-only', action='store_true', help='run only the CPU leg and print it')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--norm', default='BN', choices=['BN', 'SyncBN'],
                    help='norm_cfg type of backbone and neck: SyncBN is what the reference ships (exp_panoptic.py:20,28); '
                         'with one rank it is plain BatchNorm arithmetic')
    ap.add_argument('--grad-comm', default='f32', choices=['f32', 'bf16'],
                    help='dtype the gradient buckets travel in over RCCL (f32 = the reference\'s DDP; bf16 halves the bytes)')
    ap.add_argument('--no-also', action='store_true',
                    help='skip the short infer / decode measurements appended to the train line (`also`)')
    ap.add_argument('--graph-side-stream', action='store_true', help='A/B: capture the weight gradients on their side streams')
    ap.add_argument('--graphs', action='store_true',
                    help='replay the backbone + neck forward / backward as two hipGraphs (train, 1 GPU; das_amd/graphs.py): '
                         '38 instead of 62 ms of host time per step, but 1 % LESS throughput than queueing every launch '
                         '(nine A/B pairs on three boxes: 177.2 vs 179.3 img/s) — the GPU, not the host, bounds the step')
    ap.add_argument('--no-graphs', action='store_true', help='(the default; kept for older command lines)')
    ap.add_argument('--no-infer-graph', action='store_true',
                    help='A/B: the inference workload launch by launch (default: its forward replayed as one hipGraph)')
    ap.add_argument('--no-wgrad-stream', action='store_true', help='A/B: weight gradients on the main stream')
    ap.add_argument('--wgrad-streams', type=int, default=None, help='A/B: number of weight-gradient side streams')
    ap.add_argument('--wgrad-batch', type=int, default=None, help='A/B: weight gradients per batched launch (1 = off)')
    ap.add_argument('--no-fused-bn', action='store_true',
                    help='A/B: one autograd node per conv+BN unit (separate BatchNorm-backward reduction passes)')
    ap.add_argument('--no-early-targets', action='store_true',
                    help='A/B: target assignment inside the loss (after the forward pass) instead of ahead of the backbone')
    ap.add_argument('--tune', nargs='*', default=[], metavar='KEY=VALUE',
                    help='A/B: dispatch thresholds through das_tuning_set, e.g. conv.tail_split=0')
    ap.add_argument('--share-gpu', action='store_true',
                    help='testing only: run all ranks on cuda:0 with the gloo transport (not a measurement)')
    args = ap.parse_args()
    train = args.workload == 'train'
    batch = args.batch or (16 if train else 8)
    stages = args.stages or (4 if train else 1)
    steps = args.steps or (10 if train else 20)
    warmup = args.warmup if args.warmup is not None else (3 if train else 5)

    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args.workload, full=args.cpu_baseline_full, threads=args.cpu_threads)), flush=True)
        return
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args.gpus))   # (no GPU call has been made in this process)
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit(f'bench.py: WORLD_SIZE={world} but --gpus {args.gpus} (launch with --nproc-per-node {args.gpus}, '
                         f'or without torchrun: bench.py starts its own ranks)')
    import torch.distributed as dist
    if args.share_gpu:   # plumbing test on a 1-GPU box: all ranks on cuda:0 over gloo (RCCL needs a GPU per rank)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.share_gpu:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    comm = None
    if world > 1:   # ranks the transport really spans: an all-reduce of ones
        one = torch.ones(1, device=dev)
        dist.all_reduce(one)
        comm = dict(backend=dist.get_backend(), ranks=int(one.item()), world_size=dist.get_world_size())

    if args.workload == 'decode':
        decode_workload(args, rank, world, dev)
        if world > 1:
            dist.destroy_process_group()
        return

    from das_amd import autograd as ag, backbones, ops
    from das_amd.datasets import SyntheticPoseDataset, collate
    for kv in args.tune:
        k, v = kv.split('=')
        from das_amd import _lib
        _lib.check(_lib.load().das_tuning_set(k.encode(), int(v)), f'das_tuning_set({k})')
    if args.no_wgrad_stream:
        ag.WGRAD_SIDE_STREAM = False
    if args.wgrad_streams:
        ag.set_wgrad_streams(args.wgrad_streams)
    if args.wgrad_batch:
        ag.WGRAD_BATCH = args.wgrad_batch
    if args.no_fused_bn:
        backbones.FUSED_LAYER_BACKWARD = False
    if args.no_early_targets:
        from das_amd import detectors
        detectors.EARLY_TARGETS = False
    model = build_model(dev, seed=0, dtype=args.dtype, num_stages=stages, train=train, norm=args.norm)
    ds = SyntheticPoseDataset(num_joints=J, img_shape=(H, W), length=batch * world, seed=0)
    data = collate([ds[rank * batch + i] for i in range(batch)], device=dev)
    metas = data['img_metas']
    extra = {}
    if train:
        from das_amd.optim import FlatSGD, step_lr, train_iteration
        opt = FlatSGD(model, lr=2e-3, momentum=0.9, weight_decay=1e-4, bias_lr_mult=2.0, bias_decay_mult=0.0,
                      max_grad_norm=35.0, grad_comm_dtype=args.grad_comm)
        it = [0]

        def step():
            out = train_iteration(model, opt, data, step_lr(2e-3, 0, it[0]))
            it[0] += 1
            return out
    else:
        calibrate_scores(model, data['img'], metas)

        def step():
            return model(data['img'], metas, return_loss=False, rescale=True)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    graphs = None
    if train and world == 1 and args.graphs and not args.no_graphs:
        # (one eager step first: the optimizer's packed-weight buffers, the weight-gradient schedules and the kernels'
        # attributes exist before anything is captured; it counts as one of the warm-up steps)
        res = step()
        del res
        from das_amd import graphs as _g
        from das_amd.graphs import enable_trunk_graphs
        if args.graph_side_stream:
            _g.CAPTURE_SIDE_STREAM = True
        try:
            graphs = enable_trunk_graphs(model, opt, data['img'])
            extra['hip_graphs'] = 'backbone + neck: forward graph, backward graph (das_amd/graphs.py); head and losses eager'
        except RuntimeError as e:      # capture refused: the same kernels, queued launch by launch (the line says so)
            model._graphed_trunk = None
            extra['hip_graphs'] = f'off (capture failed: {e})'
    elif train:
        extra['hip_graphs'] = 'off (default: every launch queued by hand; --graphs replays the trunk as two hipGraphs)'
    elif world == 1 and not args.no_infer_graph:
        # inference, fixed batch shape: the forward as one hipGraph (das_amd/graphs.py GraphedInference), the decode eager
        res = step()
        del res
        try:
            from das_amd.graphs import enable_inference_graph
            graphs = enable_inference_graph(model, data['img'])
            extra['hip_graphs'] = 'forward (backbone + neck + head) replayed as one hipGraph, decode eager'
        except Exception as e:
            model._graphed_infer = None
            graphs = None
            extra['hip_graphs'] = f'off (capture failed: {type(e).__name__}: {e})'
    else:
        extra['hip_graphs'] = 'off'
    for _ in range(warmup - (1 if graphs is not None else 0)):
        res = step()
    # The interpreter's cyclic garbage collector is parked for the timed region (and the per-launch passes after it): a
    # generation-2 collection over the autograd graphs of a few steps is a 20-100 ms host pause (seen as single 106 / 191 ms
    # steps among 83 ms ones, tools/dev/first_process_steps.py) — over 20 timed steps one such pause is 1-5 ms per step of
    # noise that has nothing to do with the path measured. Reference counting still frees every step's tensors.
    # The SAME policy as the shipped loop (das_amd.optim.GcPark, used by tools/train.py): parked after warm-up, collected at
    # the loop's quiet points (logging intervals, checkpoints) — none of which fall inside 20 timed steps.
    from das_amd.optim import GcPark
    gcp = GcPark(warmup=0)
    gcp.park()
    sync_all()
    # (one event behind every timed step: the spread of the steps inside the timed region goes into the line beside their mean)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    host = [t0]
    for i in range(steps):
        res = step()
        marks[i + 1].record()
        host.append(time.perf_counter())
    sync_all()
    dt = time.perf_counter() - t0
    series = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
    per = sorted(series)
    extra['timed_steps_ms'] = dict(min=round(per[0], 3), median=round(per[len(per) // 2], 3), max=round(per[-1], 3),
                                   in_order=[round(x, 2) for x in series[:64]],
                                   host_in_order=[round((host[i + 1] - host[i]) * 1e3, 1) for i in range(min(steps, 64))],
                                   note='GPU time between the ends of consecutive timed steps (events on the launch stream); '
                                        'host_in_order: the time the host took to QUEUE each step (it runs up to three steps '
                                        'ahead); ms_per_step is the wall clock of the whole region / steps')
    if graphs is not None:     # the per-launch measurement passes below need every launch queued by hand
        model._graphed_trunk = None
        model._graphed_infer = None
    if world > 1:
        # per-rank wall clock of the timed region, gathered so that the first real multi-GPU line explains itself (a slow
        # rank, a straggling bucket): value / ms_per_step use the MAX, as the contract says
        mine = torch.tensor([dt, per[len(per) // 2] * 1e-3], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        dts = [float(a[0]) for a in allr]
        extra['ranks_ms_per_step'] = dict(min=round(min(dts) / steps * 1e3, 3), max=round(max(dts) / steps * 1e3, 3),
                                          per_rank=[round(x / steps * 1e3, 3) for x in dts],
                                          per_rank_median_gpu_step=[round(float(a[1]) * 1e3, 3) for a in allr])
        dt = max(dts)
    if train and comm is not None:
        comm.update(grad_comm_dtype=str(opt.grad_comm_dtype).replace('torch.', ''), buckets=len(opt.buckets),
                    bucket_mb=[round((e - s) * (2 if opt.grad_comm_dtype == torch.bfloat16 else 4) / 2 ** 20, 1)
                               for s, e in opt.buckets],
                    end_only_buckets=int(sum(opt._endonly)), overlapped_launches_total=opt.overlapped_launches,
                    reserved_cus_while_in_flight=opt.comm_reserved_cus)
    if train:
        extra['last_losses'] = {k: round(float(v), 4) for k, v in res['log_vars'].items()}
        extra['peak_mem_gb'] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)
    else:
        extra['poses_per_step_rank0'] = sum(len(r['scores']) for r in res)

    roof, roof_mfma, roof_hbm, roof_bn, priced = roofline_from_profile(ops, step, args.dtype, ms_per_step=dt / steps * 1e3)
    for r in (roof, roof_mfma, roof_hbm, roof_bn):
        if r is not None:
            attach_traffic(r, 'train' if train else 'infer', batch)

    gcp.release()
    also = None
    if train and world == 1 and not args.no_also:
        also = also_workloads(args, dev, model, opt, data)

    if rank == 0:
        total_imgs = batch * world * steps
        gflop = (TRAIN_GFLOP if train else FWD_GFLOP)[stages]
        wl = (f'BASELINE configs[2]/[3]: MSPN-50 {stages}-stage + FPN(4 lvls) + DASHead J=15, batch {batch} x 3x{H}x{W} '
              f'per GPU, full train step (forward, 4 losses, backward, all-reduce, clip, SGD)' if train else
              f'BASELINE configs[1]: MSPN-50 {stages}-stage + FPN(4 lvls) + DASHead J=15, batch {batch} x 3x{H}x{W} '
              f'per GPU, forward + decode (inference)')
        out = {
            'metric': 'imgs/sec train' if train else 'imgs/sec forward+decode', 'value': round(total_imgs / dt, 3),
            'unit': 'img/s', 'n_gpus': world, 'steps': steps, 'warmup': warmup,
            'ms_per_step': round(dt / steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': wl, 'per_gpu_batch': batch, 'global_batch': batch * world,
                       'parallelism': f'dp{world}', 'mspn_stages': stages, 'algorithmic_gflop_per_img': gflop},
            'model_tflops': round(total_imgs * gflop / dt / 1e3, 2),
            'roofline': roof,
        }
        for key, r in (('roofline_mfma', roof_mfma), ('roofline_hbm', roof_hbm), ('roofline_bn', roof_bn), ('priced_step', priced)):
            if r is not None:
                out[key] = r
        out.update(extra)
        if comm is not None:
            out['comm'] = comm      # (backend 'nccl' = RCCL on ROCm; ranks = what an all-reduce of ones returned)
        if also is not None:
            out['also'] = also
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args.workload, full=args.cpu_baseline_full, threads=args.cpu_threads)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
