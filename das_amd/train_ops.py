"""ctypes wrappers of the loss / target / optimizer entry points (train_loss.hip)."""
import ctypes as C

import torch
from torch.autograd import Function

from . import _lib
from .ops import Ragged, _levels, _need_gpu, _ptr, _stream


def _target_desc(strides, regress_ranges, J, radius, alpha, background):
    d = _lib.DasTargetDesc(J=J, background=background, radius=radius, alpha=alpha)
    for l, (s, r) in enumerate(zip(strides, regress_ranges)):
        d.stride[l], d.range_lo[l], d.range_hi[l] = int(s), float(r[0]), float(r[1])
    return d


def positive_rows(geom_like, strides, regress_ranges, J, pos, targets, ctr_t, z_norm, depth_factor, nvis_scale):
    """das_positive_rows: what the pose losses need of the ground truth for the positive rows `pos` (int64 / int32, ascending)
    of assign_targets' outputs. Returns dict(real (n, J, 3), vis (n, J), is2d int32 (n), slot int32 (n), depth_t (n), ctr_t (n),
    nvis 0-dim) — das_head.py:385-409."""
    _need_gpu(pos, targets, ctr_t)
    n, dev = pos.numel(), targets.device
    lv = _levels(geom_like)
    d = _target_desc(strides, regress_ranges, J, 0.0, 0.0, 1)
    p32 = pos.to(torch.int32)
    f = torch.empty(n * J * 4 + 2 * n + 1, dtype=torch.float32, device=dev)     # real | vis | depth_t | ctr | nvis in one buffer
    real, vis = f[:n * J * 3].view(n, J, 3), f[n * J * 3:n * J * 4].view(n, J)
    depth_t, ctr, nvis = f[n * J * 4:n * J * 4 + n], f[n * J * 4 + n:n * J * 4 + 2 * n], f[n * J * 4 + 2 * n:]
    ii = torch.empty(2 * n, dtype=torch.int32, device=dev)
    _lib.check(_lib.load().das_positive_rows(_ptr(p32), n, _ptr(targets), _ptr(ctr_t), C.byref(lv), C.byref(d), float(z_norm),
                                             float(depth_factor), float(nvis_scale), _ptr(real), _ptr(vis), _ptr(ii[:n]),
                                             _ptr(ii[n:]), _ptr(depth_t), _ptr(ctr), _ptr(nvis), _stream()), 'das_positive_rows')
    return dict(real=real, vis=vis, is2d=ii[:n], slot=ii[n:], depth_t=depth_t, ctr_t=ctr, nvis=nvis.reshape(()))


def assign_targets(geom_like, strides, regress_ranges, gt_rows, gt_start, J, radius=1.5, alpha=2.5, background=1,
                   centers=None, counts=None):
    """geom_like: a Ragged giving (B, level sizes). gt_rows (sum G, 3+4J) f32, gt_start (B+1,) int32, both on
    the GPU; centers (sum G, 3) f32 = [centers2d, depths] or None (= gt_rows[:, :3]).
    Returns labels int32 (rows,), targets f32 (rows, 3+4J), centerness f32 (rows,). counts: a ZEROED f32[3] that receives
    [positives, positives with a depth annotation, sum of the positives' joint visibilities]."""
    _need_gpu(gt_rows, gt_start)
    lv = _levels(geom_like)
    d = _target_desc(strides, regress_ranges, J, radius, alpha, background)
    rows, dev = geom_like.rows, geom_like.device
    labels = torch.empty(rows, dtype=torch.int32, device=dev)
    targets = torch.empty(rows, 3 + 4 * J, dtype=torch.float32, device=dev)
    ctr = torch.empty(rows, dtype=torch.float32, device=dev)
    assert gt_rows.dtype == torch.float32 and gt_start.dtype == torch.int32 and gt_start.numel() == geom_like.B + 1
    assert gt_rows.numel() == 0 or (gt_rows.is_contiguous() and gt_rows.shape[1] == 3 + 4 * J)
    if centers is not None:
        assert centers.dtype == torch.float32 and centers.is_contiguous() and centers.shape == (gt_rows.shape[0], 3)
        _need_gpu(centers)
    if counts is not None:
        assert counts.dtype == torch.float32 and counts.numel() == 3 and counts.is_contiguous()
    _lib.check(_lib.load().das_assign_targets(C.byref(lv), C.byref(d), _ptr(gt_rows), _ptr(centers), _ptr(gt_start), _ptr(labels),
                                              _ptr(targets), _ptr(ctr), _ptr(counts), _stream()), 'das_assign_targets')
    return labels, targets, ctr


def _weight(w, n):
    """mmdet's per-element `weight` -> dense f32 of n elements (or None)."""
    if w is None:
        return None
    w = w.detach().float().expand(n) if w.numel() == 1 else w.detach().float().reshape(-1)
    assert w.numel() == n, (w.shape, n)
    return w.contiguous()


class FocalLossSumFn(Function):
    """sum_i focal(logit_i, label_i); logits (rows, 1) view with any row stride."""

    @staticmethod
    def forward(ctx, logits, labels, gamma, alpha, weight=None):
        _need_gpu(logits, labels, weight)
        rows = logits.shape[0]
        grad = torch.empty(rows, dtype=torch.float32, device=logits.device)
        out = torch.empty(1, dtype=torch.float32, device=logits.device)
        assert logits.dtype == torch.float32 and labels.dtype == torch.int32
        weight = _weight(weight, rows)
        _lib.check(_lib.load().das_sigmoid_focal_loss(_ptr(logits), logits.stride(0), _ptr(labels), _ptr(weight), rows,
                                                      gamma, alpha, _ptr(grad), _ptr(out), _stream()),
                   'das_sigmoid_focal_loss')
        ctx.save_for_backward(grad)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad * g).unsqueeze(1), None, None, None, None


class SmoothL1SumFn(Function):
    @staticmethod
    def forward(ctx, pred, target, beta, weight=None):
        _need_gpu(pred, target, weight)
        pred, target = pred.contiguous(), target.contiguous()
        grad = torch.empty_like(pred)
        out = torch.empty(1, dtype=torch.float32, device=pred.device)
        weight = _weight(weight, pred.numel())
        _lib.check(_lib.load().das_smooth_l1_loss(_ptr(pred), _ptr(target), _ptr(weight), pred.numel(), beta,
                                                  _ptr(grad), _ptr(out), _stream()), 'das_smooth_l1_loss')
        ctx.save_for_backward(grad)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None


class BCELogitsSumFn(Function):
    @staticmethod
    def forward(ctx, logits, target, weight=None):
        _need_gpu(logits, target, weight)
        logits, target = logits.contiguous(), target.contiguous()
        grad = torch.empty_like(logits)
        out = torch.empty(1, dtype=torch.float32, device=logits.device)
        weight = _weight(weight, logits.numel())
        _lib.check(_lib.load().das_bce_logits_loss(_ptr(logits), _ptr(target), _ptr(weight), logits.numel(), _ptr(grad),
                                                   _ptr(out), _stream()), 'das_bce_logits_loss')
        ctx.save_for_backward(grad)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None


def grad_sumsq(flat_grad, out=None, zero_first=True):
    _need_gpu(flat_grad)
    out = out if out is not None else torch.empty(1, dtype=torch.float32, device=flat_grad.device)
    _lib.check(_lib.load().das_grad_sumsq(_ptr(flat_grad), flat_grad.numel(), _ptr(out), int(zero_first), _stream()),
               'das_grad_sumsq')
    return out


def sgd_momentum_step(p, g, buf, lr, momentum, weight_decay, grad_scale=1.0, max_norm=0.0, grad_sumsq_t=None,
                      first_step=False):
    _need_gpu(p, g, buf)
    assert p.dtype == g.dtype == buf.dtype == torch.float32 and p.is_contiguous() and g.is_contiguous()
    _lib.check(_lib.load().das_sgd_momentum_step(_ptr(p), _ptr(g), _ptr(buf), p.numel(), lr, momentum, weight_decay,
                                                 grad_scale, max_norm, _ptr(grad_sumsq_t), int(first_step), _stream()),
               'das_sgd_momentum_step')


# ------------------------------------------------------------------ RealNVP log-density (RLE pose loss)
def flow_param_list(flow):
    """The flow's parameters in the kernel's order: per coupling layer [t-net | s-net], per net
    W1, b1, W2, b2, W3, b3 (the nn.Linear tensors of `flow.t[i]` / `flow.s[i]`)."""
    ps = []
    for i in range(len(flow.t)):
        for net in (flow.t[i], flow.s[i]):
            for k in (0, 2, 4):
                ps += [net[k].weight, net[k].bias]
    return ps


def flow_mask_bits(flow):
    D = flow.mask.shape[1]
    bits = 0
    for i, row in enumerate(flow.mask.tolist()):
        for d, v in enumerate(row):
            if v:
                bits |= 1 << (i * D + d)
    return bits


_FLOW_ALIGN = 256   # rows per workgroup of the flow kernels: every job starts on this boundary


def _flow_state(flow, device):
    """Per-flow cache: parameter list in kernel order, mask bits, the parameters packed into the kernel layout
    (once per optimizer step) and, with the flat optimizer, the device table of gradient destinations."""
    from .nn import PARAM_EPOCH
    c = flow.__dict__.get('_das_flow')
    if c is None:
        c = dict(plist=flow_param_list(flow), bits=flow_mask_bits(flow), epoch=None, packed=None, table=None, tkey=None)
        flow.__dict__['_das_flow'] = c
    plist = c['plist']
    if c['epoch'] != PARAM_EPOCH[0] or c['packed'] is None or c['packed'].device != device:
        with torch.no_grad():
            c['packed'] = torch.cat([p.detach().reshape(-1).float() for p in plist])
        c['epoch'] = PARAM_EPOCH[0]
    table = None
    if torch.is_grad_enabled() and all(getattr(p, '_das_slot', None) is not None and p.grad is not None for p in plist):
        key = tuple(p.grad.data_ptr() for p in plist)
        if c['tkey'] != key:
            c['table'] = torch.tensor(key, dtype=torch.int64, device=device)
            c['tkey'] = key
        table = c['table']
    return c, table


class RealNVPLogProbFn(Function):
    """log p(x) of several flows of one dimension in ONE forward and ONE backward launch
    (das_realnvp_log_prob_multi*), instead of ~80 GEMM / elementwise launches per flow and direction.
    x_cat (rows, D): the jobs' rows back to back, each job starting on a 256-row boundary.
    jobs: list of (row_start, row_end, packed params, gradient-destination table or None, parameter list);
    the parameters also come in as inputs so that autograd can deliver their gradients when there is no table."""

    @staticmethod
    def forward(ctx, x_cat, layers, mask_bits, jobs, *flat_plist):
        _need_gpu(x_cat)
        x_cat = x_cat.contiguous().float()
        rows, D = x_cat.shape
        arr = (_lib.DasFlowJob * len(jobs))()
        for q, (r0, r1, packed, table, plist) in enumerate(jobs):
            arr[q].params, arr[q].dparams, arr[q].dst_table = packed.data_ptr(), None, None
            arr[q].row_start, arr[q].row_end = r0, r1
        logp = torch.zeros(rows, dtype=torch.float32, device=x_cat.device)
        z = torch.zeros(rows, D, dtype=torch.float32, device=x_cat.device)
        _lib.check(_lib.load().das_realnvp_log_prob_multi(_ptr(x_cat), rows, D, arr, len(jobs), layers, mask_bits,
                                                          _ptr(logp), _ptr(z), _stream()), 'das_realnvp_log_prob_multi')
        ctx.save_for_backward(z, *[j[2] for j in jobs])
        ctx.cfg = (layers, mask_bits, jobs)
        return logp

    @staticmethod
    def backward(ctx, g):
        z = ctx.saved_tensors[0]
        layers, mask_bits, jobs = ctx.cfg
        rows, D = z.shape
        g = g.contiguous().float()
        dx = torch.zeros_like(z)
        arr = (_lib.DasFlowJob * len(jobs))()
        outs = []
        for q, (r0, r1, packed, table, plist) in enumerate(jobs):
            arr[q].params, arr[q].row_start, arr[q].row_end = packed.data_ptr(), r0, r1
            if table is not None:   # the optimizer's flat gradient: the kernel adds every tensor's gradient in place
                arr[q].dparams, arr[q].dst_table = None, table.data_ptr()
                outs.append(None)
            else:
                dp = torch.empty_like(packed)
                arr[q].dparams, arr[q].dst_table = dp.data_ptr(), None
                outs.append(dp)
        _lib.check(_lib.load().das_realnvp_log_prob_multi_backward(_ptr(z), _ptr(g), rows, D, arr, len(jobs), layers,
                                                                   mask_bits, _ptr(dx), _stream()),
                   'das_realnvp_log_prob_multi_backward')
        grads = []
        for (r0, r1, packed, table, plist), dp in zip(jobs, outs):
            if dp is None:
                for p in plist:
                    p._das_slot.fired()
                grads += [None] * len(plist)
            else:
                off = 0
                for p in plist:
                    n = p.numel()
                    grads.append(dp[off:off + n].view(p.shape))
                    off += n
        return (dx, None, None, None) + tuple(grads)


class RLEPoseLossFn(Function):
    """[sum of vis * (log sigma - log_phi + logQ) over (positive, joint, prediction set, dim) | smooth-L1 sum of the
    depth term] of the positive rows — the RLE pose loss and the depth loss before code weights and normalisers
    (das_head.py:375-381, 385-466; residual_log_likelihood_loss.py:17-37) — as three elementwise kernels
    (das_rle_prepare / das_rle_loss / das_rle_backward) around the two RealNVP launches, forward and backward.
    pose (rows, >= 3 + 6J) and aux (rows, >= 3J): the head's dense f32 outputs. meta: dict with the ground-truth half
    (losses.das_head_targets): pos, real, vis, is2d, slot, depth_t, J, sets, n2d, n3d, amp, beta and
    flows = {2: [flow of set 0, (set 1)], 3: [...]}. The flows' parameters also come in as inputs so that autograd can
    deliver their gradients when the flat optimizer's tables are not in use."""

    @staticmethod
    def forward(ctx, pose, aux, meta, *flat_plist):
        _need_gpu(pose, aux)
        assert pose.dtype == aux.dtype == torch.float32 and pose.stride(1) == 1 and aux.stride(1) == 1
        dev, J, sets = pose.device, meta['J'], meta['sets']
        d = _lib.DasRleDesc()
        d.J, d.sets, d.npos, d.pose_ps, d.aux_ps = J, sets, meta['pos'].numel(), pose.stride(0), aux.stride(0)
        d.amp, d.beta = meta['amp'], meta['beta']
        count = {2: meta['n2d'], 3: meta['n3d']}
        stride = {D: -(-count[D] * J // _FLOW_ALIGN) * _FLOW_ALIGN for D in (2, 3)}
        d.stride2, d.stride3 = stride[2], stride[3]
        x, w, logp, z, jobs = {}, {}, {}, {}, {}
        # (the eight zero-initialised work tensors — flow inputs, weights, log-densities, latent codes for the 2-D and the 3-D
        # rows — are slices of ONE zero fill: each was a launch of its own)
        nrow = {D: sets * stride[D] for D in (2, 3)}
        pool = torch.zeros(sum(nrow[D] * (2 * D + 2) for D in (2, 3)), dtype=torch.float32, device=dev)
        cut = [0]

        def part(n, *shape):
            t = pool[cut[0]:cut[0] + n].view(*shape)
            cut[0] += n
            return t
        for D in (2, 3):
            rows = nrow[D]
            x[D] = part(rows * D, rows, D) if rows else None
            w[D] = part(rows, rows) if rows else None
        lib = _lib.load()
        gt = [meta[k] for k in ('pos', 'real', 'vis', 'is2d', 'slot')]
        _lib.check(lib.das_rle_prepare(_ptr(pose), _ptr(aux), *[_ptr(t) for t in gt], C.byref(d), _ptr(x[2]), _ptr(w[2]),
                                       _ptr(x[3]), _ptr(w[3]), _stream()), 'das_rle_prepare')
        for D in (2, 3):
            if x[D] is None:
                continue
            flows = meta['flows'][D]
            jobs[D] = []
            for q, (c, table) in enumerate(meta['flow_states'][D]):   # (prepared outside: see rle_pose_loss_sums)
                jobs[D].append((q * stride[D], q * stride[D] + count[D] * J, c['packed'], table, c['plist']))
            arr = (_lib.DasFlowJob * sets)()
            for q, (r0, r1, packed, table, plist) in enumerate(jobs[D]):
                arr[q].params, arr[q].dparams, arr[q].dst_table = packed.data_ptr(), None, None
                arr[q].row_start, arr[q].row_end = r0, r1
            rows = sets * stride[D]
            logp[D] = part(rows, rows)
            z[D] = part(rows * D, rows, D)
            c0 = meta['flow_states'][D][0][0]
            _lib.check(lib.das_realnvp_log_prob_multi(_ptr(x[D]), rows, D, arr, sets, len(flows[0].t), c0['bits'],
                                                      _ptr(logp[D]), _ptr(z[D]), _stream()), 'das_realnvp_log_prob_multi')
        partials = torch.empty(lib.das_rle_blocks(C.byref(d)), 2, dtype=torch.float32, device=dev)
        _lib.check(lib.das_rle_loss(_ptr(pose), _ptr(aux), *[_ptr(t) for t in gt], _ptr(meta['depth_t']),
                                    _ptr(logp.get(2)), _ptr(logp.get(3)), C.byref(d), _ptr(partials), _stream()), 'das_rle_loss')
        ctx.cfg = (d, meta, jobs, w, z)
        ctx.save_for_backward(pose, aux)
        return partials.sum(0)

    @staticmethod
    def backward(ctx, g):
        d, meta, jobs, w, z = ctx.cfg
        pose, aux = ctx.saved_tensors
        g = g.contiguous().float()
        lib = _lib.load()
        dx, outs = {}, {}
        for D in (2, 3):
            if D not in jobs:
                continue
            sets, rows = len(jobs[D]), z[D].shape[0]
            dx[D] = torch.zeros_like(z[D])
            arr = (_lib.DasFlowJob * sets)()
            outs[D] = []
            for q, (r0, r1, packed, table, plist) in enumerate(jobs[D]):
                arr[q].params, arr[q].row_start, arr[q].row_end = packed.data_ptr(), r0, r1
                if table is not None:   # the optimizer's flat gradient: the kernel adds every tensor's gradient in place
                    arr[q].dparams, arr[q].dst_table = None, table.data_ptr()
                    outs[D].append(None)
                else:
                    dp = torch.empty_like(packed)
                    arr[q].dparams, arr[q].dst_table = dp.data_ptr(), None
                    outs[D].append(dp)
            f0 = meta['flows'][D][0]
            _lib.check(lib.das_realnvp_log_prob_multi_backward(_ptr(z[D]), _ptr(w[D] * g[0]), rows, D, arr, sets, len(f0.t),
                                                               meta['flow_states'][D][0][0]['bits'], _ptr(dx[D]), _stream()),
                       'das_realnvp_log_prob_multi_backward')
        dpose, daux = torch.zeros_like(pose), torch.zeros_like(aux)
        gt = [meta[k] for k in ('pos', 'real', 'vis', 'is2d', 'slot')]
        _lib.check(lib.das_rle_backward(_ptr(pose), _ptr(aux), *[_ptr(t) for t in gt], _ptr(meta['depth_t']), _ptr(dx.get(2)),
                                        _ptr(dx.get(3)), _ptr(g), C.byref(d), _ptr(dpose), _ptr(daux), _stream()),
                   'das_rle_backward')
        grads = []
        for D in (2, 3):
            for (r0, r1, packed, table, plist), dp in zip(jobs.get(D, []), outs.get(D, [])):
                if dp is None:
                    for p in plist:
                        p._das_slot.fired()
                    grads += [None] * len(plist)
                else:
                    off = 0
                    for p in plist:
                        n = p.numel()
                        grads.append(dp[off:off + n].view(p.shape))
                        off += n
        return (dpose, daux, None) + tuple(grads)


def rle_pose_loss_sums(pose, aux, meta):
    """-> tensor [pose sum, depth sum] (see RLEPoseLossFn); the flows' parameters are threaded through autograd."""
    # the flows' packed parameters and — with the flat optimizer — the device tables of their gradient destinations are
    # looked up HERE: inside Function.forward grad mode is off, _flow_state would hand out no table, and the backward
    # kernel's parameter gradients would come back through autograd (one AccumulateGrad add per tensor: ~290 per step)
    flat = []
    meta = dict(meta, flow_states={D: [_flow_state(f, pose.device) for f in meta['flows'][D]] for D in (2, 3)})
    for D in (2, 3):
        if meta['n2d' if D == 2 else 'n3d'] > 0:
            for c, _ in meta['flow_states'][D]:
                flat += c['plist']
    # (dense rows: the gradient tensors share the inputs' row strides)
    return RLEPoseLossFn.apply(pose.contiguous(), aux.contiguous(), meta, *flat)


def realnvp_log_prob_multi(pairs):
    """[(flow, x (N_i, D)), ...] with one common D -> [log p_i (N_i)]: RealNVP.log_prob (real_nvp.py:60-80) of
    every pair in one forward (and one backward) launch."""
    pairs = [(f, x) for f, x in pairs]
    res = [None] * len(pairs)
    live = [(i, f, x) for i, (f, x) in enumerate(pairs) if x.shape[0] > 0]
    for i, (f, x) in enumerate(pairs):
        if x.shape[0] == 0:
            res[i] = x.new_zeros(0)
    if not live:
        return res
    D = live[0][2].shape[1]
    dev = live[0][2].device
    chunks, jobs, flat, r = [], [], [], 0
    for i, f, x in live:
        assert x.shape[1] == D and f.mask.shape[1] == D
        c, table = _flow_state(f, dev)
        n = x.shape[0]
        jobs.append((r, r + n, c['packed'], table, c['plist']))
        flat += c['plist']
        chunks.append(x.float())
        pad = (-n) % _FLOW_ALIGN
        if pad and (i, f, x) is not live[-1]:
            chunks.append(x.new_zeros(pad, D, dtype=torch.float32))
        r += n + (pad if (i, f, x) is not live[-1] else 0)
    x_cat = torch.cat(chunks) if len(chunks) > 1 else chunks[0]
    c0, _ = _flow_state(live[0][1], dev)
    logp = RealNVPLogProbFn.apply(x_cat, len(live[0][1].t), c0['bits'], jobs, *flat)
    for (i, f, x), (r0, r1, _, _, _) in zip(live, jobs):
        res[i] = logp[r0:r1]
    return res


def realnvp_log_prob(flow, x):
    """RealNVP.log_prob (real_nvp.py:60-80) through the fused kernels."""
    return realnvp_log_prob_multi([(flow, x)])[0]
