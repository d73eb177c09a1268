"""ctypes wrappers of the loss / target / optimizer entry points (train_loss.hip)."""
import ctypes as C

import torch
from torch.autograd import Function

from . import _lib
from .ops import Ragged, _levels, _need_gpu, _ptr, _stream


def assign_targets(geom_like, strides, regress_ranges, gt_rows, gt_start, J, radius=1.5, alpha=2.5, background=1):
    """geom_like: a Ragged giving (B, level sizes). gt_rows (sum G, 3+4J) f32, gt_start (B+1,) int32, both on
    the GPU. Returns labels int32 (rows,), targets f32 (rows, 3+4J), centerness f32 (rows,)."""
    _need_gpu(gt_rows, gt_start)
    lv = _levels(geom_like)
    d = _lib.DasTargetDesc(J=J, background=background, radius=radius, alpha=alpha)
    for l, (s, r) in enumerate(zip(strides, regress_ranges)):
        d.stride[l], d.range_lo[l], d.range_hi[l] = int(s), float(r[0]), float(r[1])
    rows, dev = geom_like.rows, geom_like.device
    labels = torch.empty(rows, dtype=torch.int32, device=dev)
    targets = torch.empty(rows, 3 + 4 * J, dtype=torch.float32, device=dev)
    ctr = torch.empty(rows, dtype=torch.float32, device=dev)
    assert gt_rows.dtype == torch.float32 and gt_start.dtype == torch.int32 and gt_start.numel() == geom_like.B + 1
    assert gt_rows.numel() == 0 or (gt_rows.is_contiguous() and gt_rows.shape[1] == 3 + 4 * J)
    _lib.check(_lib.load().das_assign_targets(C.byref(lv), C.byref(d), _ptr(gt_rows), _ptr(gt_start), _ptr(labels),
                                              _ptr(targets), _ptr(ctr), _stream()), 'das_assign_targets')
    return labels, targets, ctr


class FocalLossSumFn(Function):
    """sum_i focal(logit_i, label_i); logits (rows, 1) view with any row stride."""

    @staticmethod
    def forward(ctx, logits, labels, gamma, alpha):
        _need_gpu(logits, labels)
        rows = logits.shape[0]
        grad = torch.empty(rows, dtype=torch.float32, device=logits.device)
        out = torch.empty(1, dtype=torch.float32, device=logits.device)
        assert logits.dtype == torch.float32 and labels.dtype == torch.int32
        _lib.check(_lib.load().das_sigmoid_focal_loss(_ptr(logits), logits.stride(0), _ptr(labels), rows, gamma, alpha,
                                                      _ptr(grad), _ptr(out), _stream()), 'das_sigmoid_focal_loss')
        ctx.save_for_backward(grad)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad * g).unsqueeze(1), None, None, None


class SmoothL1SumFn(Function):
    @staticmethod
    def forward(ctx, pred, target, beta):
        pred, target = pred.contiguous(), target.contiguous()
        grad = torch.empty_like(pred)
        out = torch.empty(1, dtype=torch.float32, device=pred.device)
        _lib.check(_lib.load().das_smooth_l1_loss(_ptr(pred), _ptr(target), pred.numel(), beta, _ptr(grad), _ptr(out),
                                                  _stream()), 'das_smooth_l1_loss')
        ctx.save_for_backward(grad)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None


class BCELogitsSumFn(Function):
    @staticmethod
    def forward(ctx, logits, target):
        logits, target = logits.contiguous(), target.contiguous()
        grad = torch.empty_like(logits)
        out = torch.empty(1, dtype=torch.float32, device=logits.device)
        _lib.check(_lib.load().das_bce_logits_loss(_ptr(logits), _ptr(target), logits.numel(), _ptr(grad), _ptr(out),
                                                   _stream()), 'das_bce_logits_loss')
        ctx.save_for_backward(grad)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None


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


class RealNVPLogProbFn(Function):
    """log p(x) under the flow for x (N, D): one forward and one backward kernel (das_realnvp_log_prob*)
    instead of ~80 GEMM / elementwise launches each way. `plist` are the flow's parameters
    (flow_param_list), passed as inputs so that autograd delivers their gradients."""

    @staticmethod
    def forward(ctx, x, layers, mask_bits, packed, table, *plist):
        _need_gpu(x)
        x = x.contiguous().float()
        N, D = x.shape
        params = packed if packed is not None else torch.cat([p.detach().reshape(-1).float() for p in plist])
        logp = torch.empty(N, dtype=torch.float32, device=x.device)
        z = torch.empty(N, D, dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().das_realnvp_log_prob(_ptr(x), N, D, _ptr(params), layers, mask_bits, _ptr(logp), _ptr(z),
                                                    _stream()), 'das_realnvp_log_prob')
        ctx.save_for_backward(z, params)
        ctx.cfg = (layers, mask_bits, plist, table)
        return logp

    @staticmethod
    def backward(ctx, g):
        z, params = ctx.saved_tensors
        layers, mask_bits, plist, table = ctx.cfg
        N, D = z.shape
        g = g.contiguous().float()
        dx = torch.empty_like(z)
        if table is not None:
            # the optimizer's flat gradient: the kernel adds every tensor's gradient in place
            _lib.check(_lib.load().das_realnvp_log_prob_backward(_ptr(z), _ptr(g), N, D, _ptr(params), layers,
                                                                 mask_bits, _ptr(dx), None, _ptr(table), _stream()),
                       'das_realnvp_log_prob_backward')
            for p in plist:
                p._das_slot.fired()
            return (dx, None, None, None, None) + (None,) * len(plist)
        dparams = torch.empty_like(params)
        _lib.check(_lib.load().das_realnvp_log_prob_backward(_ptr(z), _ptr(g), N, D, _ptr(params), layers, mask_bits,
                                                             _ptr(dx), _ptr(dparams), None, _stream()),
                   'das_realnvp_log_prob_backward')
        grads, off = [], 0
        for p in plist:
            n = p.numel()
            grads.append(dparams[off:off + n].view(p.shape))
            off += n
        return (dx, None, None, None, None) + tuple(grads)


def realnvp_log_prob(flow, x):
    """RealNVP.log_prob (real_nvp.py:60-80) through the fused kernels. The flow's 72 tensors are packed into the
    kernel's layout once per optimizer step; with the flat optimizer the backward kernel adds their gradients
    straight into the flat gradient buffer (table of destination pointers)."""
    if x.shape[0] == 0:
        return x.new_zeros(0)
    from .nn import PARAM_EPOCH
    c = flow.__dict__.get('_das_flow')
    if c is None:
        c = dict(plist=flow_param_list(flow), bits=flow_mask_bits(flow), epoch=None, packed=None, table=None, tkey=None)
        flow.__dict__['_das_flow'] = c
    plist = c['plist']
    if c['epoch'] != PARAM_EPOCH[0] or c['packed'] is None or c['packed'].device != x.device:
        with torch.no_grad():
            c['packed'] = torch.cat([p.detach().reshape(-1).float() for p in plist])
        c['epoch'] = PARAM_EPOCH[0]
    table = None
    if torch.is_grad_enabled() and all(getattr(p, '_das_slot', None) is not None and p.grad is not None for p in plist):
        key = tuple(p.grad.data_ptr() for p in plist)
        if c['tkey'] != key:
            c['table'] = torch.tensor(key, dtype=torch.int64, device=x.device)
            c['tkey'] = key
        table = c['table']
    return RealNVPLogProbFn.apply(x, len(flow.t), c['bits'], c['packed'], table, *plist)
