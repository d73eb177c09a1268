"""torch.autograd.Function wrappers: autograd sequences the backward pass (plumbing), every forward
and backward body is a HIP kernel from libdas_hip.so. Activations are NHWC tensors or the 2-D `data`
of an `ops.Ragged`; geometry travels as plain python arguments."""
import os

import torch
from torch.autograd import Function

from . import ops
from .ops import Ragged


def _wrap(t, geom):
    """geom = None (4-D NHWC tensor) or (B, sizes) for Ragged rows"""
    return t if geom is None else Ragged(t, geom[0], geom[1])


def _geom(x):
    return (x.B, x.sizes) if isinstance(x, Ragged) else None


def _d(x):
    return x.data if isinstance(x, Ragged) else x


def _dw_to_oihw(dw, weight):
    O, I = weight.shape[0], weight.shape[1]
    return dw[:O, :, :, :I].permute(0, 3, 1, 2)


# ---- weight gradients on a side stream ----------------------------------------------------------------
# A weight gradient feeds nothing but the optimizer, so it need not sit on backward's critical path
# (BN backward -> data gradient -> next layer's BN backward ...). It is launched on a second HIP stream right after
# its operand dY is ready; the MFMA / LDS-bound weight-gradient kernels then overlap the HBM-bound BatchNorm passes
# and the under-filled mid-size data-gradient launches of the main stream. The main stream joins the side stream when
# the backward pass ends (autograd engine callback), before the gradient all-reduce / optimizer step.
WGRAD_SIDE_STREAM = os.environ.get('DAS_WGRAD_STREAM', '1') != '0'
_side = {}           # device index -> [stream, join callback queued for the running backward?]


def _side_stream(dev):
    ent = _side.get(dev.index)
    if ent is None:
        ent = _side[dev.index] = [torch.cuda.Stream(device=dev), False]
    return ent


def _join_side(dev_index):
    ent = _side[dev_index]
    ent[1] = False
    torch.cuda.current_stream(dev_index).wait_stream(ent[0])


def wgrad_stream():
    """The side stream of the current device if weight gradients are in flight on it (the data-parallel
    all-reduce waits on it too), else None."""
    ent = _side.get(torch.cuda.current_device())
    return ent[0] if ent is not None else None


class _on_side:
    """Context: run the enclosed launches on the side stream, after everything enqueued on the current stream so far.
    The tensors named in `keep` are read there: the caching allocator must not hand their memory to a later main-stream
    allocation before the side stream is done with them."""

    def __init__(self, *keep):
        self.keep = [t for t in keep if t is not None]
        self.ctx = None

    def __enter__(self):
        if not WGRAD_SIDE_STREAM or not self.keep or not self.keep[0].is_cuda:
            return self
        dev = self.keep[0].device
        ent = _side_stream(dev)
        side = ent[0]
        side.wait_stream(torch.cuda.current_stream(dev))
        for t in self.keep:
            t.record_stream(side)
        if not ent[1]:
            try:
                torch.autograd.Variable._execution_engine.queue_callback(lambda i=dev.index: _join_side(i))
                ent[1] = True
            except RuntimeError:      # not inside a backward pass: join right after the launch instead
                ent[1] = None
        self.ctx = torch.cuda.stream(side)
        self.ctx.__enter__()
        self.ent, self.dev = ent, dev
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
            if self.ent[1] is None:
                _join_side(self.dev.index)
        return False


def _wgrad(x, dy, weight, k, s, p):
    """Weight gradient of conv(x, weight). With the flat optimizer the kernel adds straight into the flat
    gradient buffer (same (Cout,KH,KW,Cin) layout) and autograd gets None; otherwise an OIHW view is returned."""
    sl = getattr(weight, '_das_slot', None)
    cin, cout = _d(x).shape[-1], _d(dy).shape[-1]
    if sl is not None and sl.direct(cin, cout) and sl.cl_shape[1] == k:
        with _on_side(_d(x), _d(dy)):
            ops.conv2d_wgrad(x, dy, k, k, s, p, out=sl.grad_cl, accumulate=True)
        sl.fired()
        return None
    return _dw_to_oihw(ops.conv2d_wgrad(x, dy, k, k, s, p), weight)


def _param_acc(param):
    """Flat-gradient slice of a 1-D parameter (or None without the flat optimizer)."""
    sl = getattr(param, '_das_slot', None)
    return None if sl is None or param.grad is None else (sl, param.grad)


def _sync_world(bn):
    """Number of ranks a BatchNorm's statistics span: > 1 only for layers built from `type='SyncBN'` inside an
    initialised process group (mmcv's SyncBN -> torch SyncBatchNorm in the reference)."""
    if not getattr(bn, '_das_sync', False):
        return 1
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _all_reduce(t):
    import torch.distributed as dist
    dist.all_reduce(t)


class ConvBNTrainFn(Function):
    """conv (no bias) -> train-mode BatchNorm (+ residual) (+ ReLU). mspn_mmpose.py:126-157,381-404."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, residual, conv, bn, relu):
        from .nn import bn_stats_buffer, packed_weight, sync_stats
        k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        w = packed_weight(conv, x.dtype, cin_pad=x.shape[-1])
        stats = bn_stats_buffer(x, w.shape[0])
        raw = ops.conv2d(x, w, k, k, s, p, stats=stats)
        mom = bn.momentum if bn.momentum is not None else 0.1
        world = _sync_world(bn)
        stat_count = 0
        if world > 1:   # SyncBN: the statistics are those of all ranks' pixels
            stats = sync_stats(stats, w.shape[0], _all_reduce)
            stat_count = (raw.numel() // raw.shape[-1]) * world
        y, mean, invstd = ops.bn_train_apply(raw, stats, gamma, beta, bn.running_mean, bn.running_var, mom, bn.eps,
                                             residual=residual, relu=relu,
                                             num_batches_tracked=bn.num_batches_tracked, stat_count=stat_count)
        bn.__dict__.pop('_das_cache', None)  # running stats changed under the cache's feet (raw-pointer update)
        ctx.save_for_backward(x, raw, y if residual is not None else None, mean, invstd, gamma, weight, beta)
        ctx.cfg = (k, s, p, relu, residual is not None, conv, bn)
        ctx.world = world
        return y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        from .nn import packed_weight_dgrad
        x, raw, y, mean, invstd, gamma, weight, beta = ctx.saved_tensors
        k, s, p, relu, has_res, conv, bn = ctx.cfg
        dy = dy.contiguous()
        ga, ba = _param_acc(bn.weight), _param_acc(bn.bias)
        direct = ga is not None and ba is not None
        # without a residual the ReLU mask is recomputed from raw: y is not read at all
        if ctx.world > 1:   # SyncBN: reduce, sum over ranks, apply; parameter gradients stay local
            direct = False
            draw, dres, dgamma, dbeta = ops.bn_train_backward_sync(dy, y if (relu and has_res) else None, raw, mean, invstd,
                                                                   gamma, relu, has_res, beta, _all_reduce, ctx.world)
        else:
            draw, dres, dgamma, dbeta = ops.bn_train_backward(dy, y if (relu and has_res) else None, raw, mean, invstd,
                                                              gamma, relu, has_res, beta=beta,
                                                              dgamma_acc=ga[1] if direct else None,
                                                              dbeta_acc=ba[1] if direct else None)
        if direct:
            dgamma = dbeta = None
            ga[0].fired()
            ba[0].fired()
        # (the weight gradient goes out first: on its side stream it waits for draw only, not for the data gradient)
        dw = _wgrad(x, draw, conv.weight, k, s, p) if ctx.needs_input_grad[1] else None
        dx = None
        if ctx.needs_input_grad[0]:
            if dskip is not None:
                dskip = dskip.contiguous()
            dx = ops.conv2d_dgrad(draw, packed_weight_dgrad(conv, x.dtype), k, k, s, p, (x.shape[1], x.shape[2]),
                                  residual=dskip)
            if dx.shape[-1] != x.shape[-1]:
                dx = dx[..., :x.shape[-1]]
        return dx, dw, dgamma, dbeta, dres, None, None, None


class ConvBNTrainSkipFn(Function):
    """ConvBNTrainFn that also hands its input through as a second output: y, x_skip = f(x).

    In a bottleneck x feeds conv1 AND the identity path. Routing the identity through this node makes both
    gradients of x arrive together, so the data-gradient kernel adds the skip gradient in its epilogue
    (`residual`) instead of autograd running a separate elementwise add over the whole tensor."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, conv, bn, relu):
        y = ConvBNTrainFn.forward(ctx, x, weight, gamma, beta, None, conv, bn, relu)
        return y, x

    @staticmethod
    def backward(ctx, dy, dskip):
        dx, dw, dgamma, dbeta, _, _, _, _ = ConvBNTrainFn.backward(ctx, dy, dskip)
        return dx, dw, dgamma, dbeta, None, None, None


class ConvFn(Function):
    """conv + bias (shift) [+ ReLU], output dtype T or f32; NHWC or ragged rows.
    `w_packed`/`shift` are prepared by the caller (fused heads concatenate several nn.Conv2d)."""

    @staticmethod
    def forward(ctx, x, weight, bias, conv_like, geom, relu, out_dtype, out):
        from .nn import bias_shift, packed_weight
        k, s, p = conv_like.kernel_size[0], conv_like.stride[0], conv_like.padding[0]
        xin = _wrap(x, geom)
        w = packed_weight(conv_like, x.dtype, cin_pad=x.shape[-1])
        y = ops.conv2d(xin, w, k, k, s, p, shift=bias_shift(conv_like), relu=relu, out_dtype=out_dtype,
                       out=_wrap(out, geom) if out is not None else None)
        yd = _d(y)
        ctx.save_for_backward(x, weight, yd if relu else None)
        ctx.cfg = (k, s, p, relu, geom, conv_like, bias is not None)
        if out is not None:
            ctx.mark_dirty(out)
        return yd

    @staticmethod
    def backward(ctx, dy):
        from .nn import packed_weight_dgrad
        x, weight, y = ctx.saved_tensors
        k, s, p, relu, geom, conv, has_bias = ctx.cfg
        if relu:
            dy = dy * (y > 0).to(dy.dtype)  # only the stem-free plain convs with ReLU (none on the DAS path)
        dz = dy if dy.dtype == x.dtype else dy.to(x.dtype)
        if not (dz.stride(-1) == 1 and (dz.dim() != 2 or dz.stride(0) % 8 == 0)):
            dz = dz.contiguous()
        if dz.dim() == 4 and not dz.is_contiguous():
            dz = dz.contiguous()
        dzr, xr = _wrap(dz, geom), _wrap(x, geom)
        wp = conv.weight if getattr(conv, 'weight', None) is not None and conv.weight.shape == weight.shape else weight
        dw = _wgrad(xr, dzr, wp, k, s, p) if ctx.needs_input_grad[1] else None
        dx = None
        if ctx.needs_input_grad[0]:
            hw = None if geom is not None else (x.shape[1], x.shape[2])
            dx = _d(ops.conv2d_dgrad(dzr, packed_weight_dgrad(conv, x.dtype), k, k, s, p, hw))
            if dx.shape[-1] != x.shape[-1]:
                dx = dx[..., :x.shape[-1]]
        db = ops.colsum(dzr)[:weight.shape[0]] if has_bias else None
        return dx, dw, db, None, None, None, None, None


class GroupNormReLUFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, geom, G, eps, relu):
        xin = _wrap(x, geom)
        out = xin.new(x.shape[-1]) if geom is not None else torch.empty_like(x)
        y, st = ops.groupnorm(xin, gamma, beta, G, eps, relu=relu, out=out, return_stats=True)
        yd = _d(y)
        ctx.save_for_backward(x, yd, st, gamma)
        ctx.cfg = (geom, G, eps, relu)
        return yd

    @staticmethod
    def backward(ctx, dy):
        x, y, st, gamma = ctx.saved_tensors
        geom, G, eps, relu = ctx.cfg
        dy = dy.contiguous()
        dx, dgamma, dbeta = ops.groupnorm_backward(_wrap(dy, geom), _wrap(y, geom), _wrap(x, geom), st, gamma, G, eps,
                                                   relu)
        return _d(dx), dgamma, dbeta, None, None, None, None


class MaxPoolFn(Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.maxpool3x3s2(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.maxpool3x3s2_backward(x, dy.contiguous())


class BilinearUpFn(Function):
    @staticmethod
    def forward(ctx, x, Ho, Wo):
        ctx.hw = (x.shape[1], x.shape[2])
        return ops.upsample_bilinear_ac(x, Ho, Wo)

    @staticmethod
    def backward(ctx, dy):
        return ops.upsample_bilinear_ac_backward(dy.contiguous(), *ctx.hw), None, None


class AddNearestFn(Function):
    """a + nearest_upsample(b)"""

    @staticmethod
    def forward(ctx, a, b):
        ctx.hw = (b.shape[1], b.shape[2])
        return ops.add_upsample_nearest(a, b)

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        return dy, ops.upsample_nearest_backward(dy, *ctx.hw)


class Add3Fn(Function):
    """a + b (+ c): the gradient is the identity on every operand"""

    @staticmethod
    def forward(ctx, a, b, c):
        ctx.n = 3 if c is not None else 2
        return ops.add3(a, b, c)

    @staticmethod
    def backward(ctx, dy):
        return dy, dy, (dy if ctx.n == 3 else None)


class DeformIm2colFn(Function):
    """DCNv2 sampling: x rows (T), om rows f32 [dy,dx per tap | mask logits] -> col rows x 9C (T)."""

    @staticmethod
    def forward(ctx, x, om, geom):
        col = ops.deform_im2col3x3(_wrap(x, geom), _wrap(om, geom))
        ctx.save_for_backward(x, om)
        ctx.geom = geom
        return _d(col)

    @staticmethod
    def backward(ctx, dcol):
        x, om = ctx.saved_tensors
        geom = ctx.geom
        dx, dom = ops.deform_im2col3x3_backward(_wrap(x, geom), _wrap(om, geom), _wrap(dcol.contiguous(), geom))
        return _d(dx).to(x.dtype), _d(dom), None


class DcnGemmFn(Function):
    """The GEMM half of DCNv2: y = col (rows, 9C) x W^T + bias, W = dcn.weight (O, C, 3, 3)."""

    @staticmethod
    def forward(ctx, col, weight, bias, dcn, geom):
        from .nn import _cache_of, _pad8
        O, Cc = weight.shape[0], weight.shape[1]
        w = _cache_of(dcn).get(('w', col.dtype), (weight,),
                               lambda: ops.pack_weight(weight, col.dtype).reshape(-1, 1, 1, 9 * col.shape[-1] // 9))
        shift = None
        if bias is not None:
            shift = _cache_of(dcn).get(('b',), (bias,), lambda: _pad8(bias, bias.numel()))
        y = ops.conv2d(_wrap(col, geom), w, 1, 1, shift=shift)
        ctx.save_for_backward(col, weight)
        ctx.cfg = (dcn, geom, bias is not None)
        return _d(y)

    @staticmethod
    def backward(ctx, dy):
        from .nn import _cache_of
        col, weight = ctx.saved_tensors
        dcn, geom, has_bias = ctx.cfg
        O, Cc = weight.shape[0], weight.shape[1]
        dy = dy.contiguous()
        dyr = _wrap(dy, geom)

        def make_wt():  # (9C_pad.., 1, 1, O_pad) = transpose of the packed GEMM weight
            wp = ops.pack_weight(weight, col.dtype).reshape(-1, 9 * col.shape[-1] // 9)
            return wp.t().contiguous().reshape(wp.shape[1], 1, 1, wp.shape[0])
        wt = _cache_of(dcn).get(('wt', col.dtype), (weight,), make_wt)
        dcol = _d(ops.conv2d(dyr, wt, 1, 1))
        sl = getattr(dcn.weight, '_das_slot', None)
        cpad = col.shape[-1] // 9
        if sl is not None and sl.direct(cpad, dy.shape[-1]):
            # (O,3,3,C) channels-last storage == the (O,1,1,9C) GEMM weight: add straight into the flat gradient
            with _on_side(col, dy):
                ops.conv2d_wgrad(_wrap(col, geom), dyr, 1, 1, 1, 0, out=sl.grad_cl, accumulate=True)
            sl.fired()
            dw = None
        else:
            dwp = ops.conv2d_wgrad(_wrap(col, geom), dyr, 1, 1, 1, 0)          # (O_pad, 1, 1, 9*Cin_pad)
            dw = dwp.reshape(dwp.shape[0], 3, 3, cpad)[:O, :, :, :Cc].permute(0, 3, 1, 2)
        db = ops.colsum(dyr)[:O] if has_bias else None
        return dcol, dw, db, None, None


class OffsetSampleFn(Function):
    @staticmethod
    def forward(ctx, uvd, so, conf, geom, J, heads):
        out = ops.offset_sample(_wrap(uvd, geom), _wrap(so, geom), _wrap(conf, geom), J, heads)
        ctx.save_for_backward(uvd, so, conf)
        ctx.cfg = (geom, J, heads)
        return _d(out)

    @staticmethod
    def backward(ctx, g):
        uvd, so, conf = ctx.saved_tensors
        geom, J, heads = ctx.cfg
        d_uvd, d_so, d_conf = ops.offset_sample_backward(_wrap(uvd, geom), _wrap(so, geom), _wrap(conf, geom),
                                                         _wrap(g.contiguous(), geom), J, heads)
        return _d(d_uvd), _d(d_so), _d(d_conf), None, None, None


class SigmoidBlendFn(Function):
    @staticmethod
    def forward(ctx, off, w, nxt, geom):
        out = ops.sigmoid_blend(_wrap(off, geom), _wrap(w, geom), _wrap(nxt, geom))
        ctx.save_for_backward(off, w, nxt)
        ctx.geom = geom
        return _d(out)

    @staticmethod
    def backward(ctx, g):
        off, w, nxt = ctx.saved_tensors
        geom = ctx.geom
        d_off, d_w, d_nxt = ops.sigmoid_blend_backward(_wrap(off, geom), _wrap(w, geom), _wrap(nxt, geom),
                                                       _wrap(g.contiguous(), geom))
        return d_off, d_w, d_nxt, None


class HeadAssembleFn(Function):
    """raw (rows, raw_ps) f32 + the per-level Scale parameters (L,4) -> pose_pred, initial uvd."""

    @staticmethod
    def forward(ctx, raw, scales, geom, desc, level_ids):
        pose, uvd = ops.head_assemble(_wrap(raw, geom), desc)
        ctx.save_for_backward(raw)
        ctx.cfg = (geom, desc, level_ids, scales.shape)
        return _d(pose), _d(uvd)

    @staticmethod
    def backward(ctx, d_pose, d_uvd):
        (raw,) = ctx.saved_tensors
        geom, desc, level_ids, sshape = ctx.cfg
        d_raw, d_scale = ops.head_assemble_backward(_wrap(raw, geom), _wrap(d_pose.contiguous(), geom),
                                                    _wrap(d_uvd.contiguous(), geom), desc)
        ds = torch.zeros(sshape, dtype=torch.float32, device=raw.device)
        for i, l in enumerate(level_ids):
            ds[l] = d_scale[i]
        return d_raw, ds, None, None, None


def grad_mode(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)
