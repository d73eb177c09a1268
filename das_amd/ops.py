"""Tensor-level wrappers over the C ABI of libdas_hip.so.

PyTorch is plumbing here: it owns device memory and the stream; every op below hands raw
device pointers + sizes to a HIP kernel. Activations are NHWC tensors of shape (B, H, W, C),
bf16 or f32, or `Ragged` row sets (all FPN levels back to back, the head's native layout).
Ops raise (never fall back) when a tensor is not on a GPU.
"""
import ctypes as C

import torch

from . import _lib

_DT = {torch.float32: _lib.DAS_F32, torch.bfloat16: _lib.DAS_BF16}

# Per-launch timing for bench.py and the tools under tools/dev: profile_begin() makes this a list and switches on the
# library's own event pairs (das_prof_*, recorded INSIDE each C entry point right around its launches, on the stream
# they go to); every wrapper below then appends (kernel family tag, algorithmic FLOPs, span, span, shape, ops,
# algorithmic bytes, launches), where `span` says which of the library's records the call produced.
PROFILE = None
_PROF_MS = None        # per-record milliseconds of the last finished pass (numpy f32), filled by profile_end()
_PROF_NAMES = None


class _Span:
    """Records [i0, i1) of the library's event pairs. elapsed_time() has torch.cuda.Event's shape (ms), so a PROFILE
    entry reads the same whether a tool expects events or spans."""
    __slots__ = ('i0', 'i1')

    def __init__(self, i0, i1):
        self.i0, self.i1 = i0, i1

    def elapsed_time(self, other=None):
        assert _PROF_MS is not None, 'profile_end() first'
        return float(_PROF_MS[self.i0:self.i1].sum())


def _prof_mark():
    return _lib.load().das_prof_count()


def profile_begin():
    global PROFILE, _PROF_MS, _PROF_NAMES
    _lib.check(_lib.load().das_prof_begin(), 'das_prof_begin')
    PROFILE, _PROF_MS, _PROF_NAMES = [], None, None


def profile_end():
    """Stops recording, waits for the recorded events and returns the pass's entries (their spans now resolve)."""
    global PROFILE, _PROF_MS, _PROF_NAMES
    import numpy as np
    lib = _lib.load()
    lib.das_prof_end()
    n = lib.das_prof_count()
    ms = np.zeros(max(n, 1), dtype=np.float32)
    names = np.zeros((max(n, 1), 64), dtype=np.uint8)
    _lib.check(lib.das_prof_read(ms.ctypes.data, names.ctypes.data, 64, n), 'das_prof_read')
    if n and float(ms[:n].min()) < 0:
        raise _lib.DasHipError('das_prof_read: a record without a valid event pair')
    _PROF_MS = ms[:n]
    _PROF_NAMES = [bytes(r).split(b'\0', 1)[0].decode() for r in names[:n]]
    ent, PROFILE = PROFILE, None
    return ent


def profile_records():
    """(name, ms) of every record of the last pass, in launch order: name = the kernel a conv / weight-gradient
    launcher picked, or the entry point."""
    return list(zip(_PROF_NAMES, (float(v) for v in _PROF_MS)))


class _timed:
    """Per-call record for the non-conv families (BatchNorm passes) while PROFILE is a list.
    nbytes = algorithmic bytes of the op (every operand of every pass once)."""

    def __init__(self, tag, nbytes, launches=1, shape=()):
        self.tag, self.nbytes, self.launches, self.shape = tag, float(nbytes), launches, ('bn',) + tuple(shape)

    def __enter__(self):
        if PROFILE is not None:
            self.i0 = _prof_mark()
        return self

    def __exit__(self, *exc):
        if PROFILE is not None and exc[0] is None:
            sp = _Span(self.i0, _prof_mark())
            PROFILE.append((self.tag, 0.0, sp, sp, self.shape, 1, self.nbytes, self.launches))
        return False


class tuning:
    """Context manager over das_tuning_set: `with ops.tuning(**{'conv.glds4_minblocks': 1}): ...` (tests, A/B runs)."""

    def __init__(self, **kv):
        self.kv, self.old = kv, {}

    def __enter__(self):
        lib = _lib.load()
        for k, v in self.kv.items():
            cur = C.c_longlong()
            _lib.check(lib.das_tuning_get(k.encode(), C.byref(cur)), f'das_tuning_get({k})')
            self.old[k] = cur.value
            _lib.check(lib.das_tuning_set(k.encode(), int(v)), f'das_tuning_set({k})')
        return self

    def __exit__(self, *exc):
        lib = _lib.load()
        for k, v in self.old.items():
            lib.das_tuning_set(k.encode(), int(v))
        return False


def last_kernel():
    """Kernel name picked by the last conv2d / conv2d_wgrad call of this thread."""
    return _lib.load().das_last_kernel().decode()


def last_tile_rows():
    """Rows per pixel tile of this thread's last conv2d launch on a 256-row tile kernel (das_conv_last_tile_rows)."""
    return int(_lib.load().das_conv_last_tile_rows())


def last_wgrad_plan():
    """Schedule of this thread's last weight-gradient launch (das_wgrad_last_plan): dict of kernel class, grid, units,
    direct (units stored straight into dW), partial (tiles through the workspace), reduced, longest list, groups, schedules
    built so far, shape (wave arrangement of the launch's first op: 0 = 128 x 128, 1 = 64 x 256, 2 = 256 x 64)."""
    out = (C.c_longlong * 10)()
    _lib.check(_lib.load().das_wgrad_last_plan(out, 10), 'das_wgrad_last_plan')
    return dict(zip(('cls', 'grid', 'units', 'direct', 'partial', 'reduced', 'longest', 'groups', 'schedules_built', 'shape'),
                    list(out)))


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _need_gpu(*ts):
    for t in ts:
        if isinstance(t, Ragged):
            t = t.data
        if t is not None and not t.is_cuda:
            raise _lib.DasHipError('das_amd ops run on the GPU only (no CPU fallback); got a CPU tensor')


class Ragged:
    """Pixel rows of several FPN levels back to back: `data` is (rows, C) (possibly a channel-slice
    view), level l owns B*H_l*W_l consecutive rows in (b, h, w) order."""

    def __init__(self, data, B, sizes):
        assert data.dim() == 2
        self.data, self.B, self.sizes = data, B, [tuple(s) for s in sizes]
        self.starts = [0]
        for h, w in self.sizes:
            self.starts.append(self.starts[-1] + B * h * w)
        assert data.shape[0] == self.starts[-1], (data.shape, self.starts)

    @property
    def dtype(self):
        return self.data.dtype

    @property
    def device(self):
        return self.data.device

    @property
    def C(self):
        return self.data.shape[1]

    @property
    def rows(self):
        return self.data.shape[0]

    def like(self, data):
        return Ragged(data, self.B, self.sizes)

    def new(self, C_, dtype=None):
        return self.like(torch.empty(self.rows, C_, dtype=dtype or self.dtype, device=self.device))

    def cslice(self, c0, c1):
        return self.like(self.data[:, c0:c1])

    def level(self, l):
        h, w = self.sizes[l]
        return self.data[self.starts[l]:self.starts[l + 1]].unflatten(0, (self.B, h, w))

    @staticmethod
    def from_levels(tensors):
        """list of NHWC (B,H,W,C) tensors -> Ragged (one concatenation copy)."""
        B = tensors[0].shape[0]
        return Ragged(torch.cat([t.reshape(-1, t.shape[-1]) for t in tensors], 0), B, [t.shape[1:3] for t in tensors])


def _ps(t):
    """pixel stride (elements) of an NHWC tensor / Ragged / channel-slice view"""
    if isinstance(t, Ragged):
        t = t.data
    assert t.shape[-1] == 1 or t.stride(-1) == 1, 'channel dim must be contiguous'
    ps = t.stride(-2)
    if t.dim() == 4:
        assert t.stride(1) == ps * t.shape[2] and t.stride(0) == ps * t.shape[1] * t.shape[2], 'rows must be dense'
    return ps


def _levels(x):
    lv = _lib.DasLevels()
    if isinstance(x, Ragged):
        lv.num_levels, lv.B = len(x.sizes), x.B
        for l, (h, w) in enumerate(x.sizes):
            lv.H[l], lv.W[l] = h, w
    else:
        lv.num_levels, lv.B, lv.H[0], lv.W[0] = 1, x.shape[0], x.shape[1], x.shape[2]
    return lv


def _data(x):
    return x.data if isinstance(x, Ragged) else x


def _empty_like_rows(x, C_, dtype):
    if isinstance(x, Ragged):
        return x.new(C_, dtype)
    return torch.empty(*x.shape[:-1], C_, dtype=dtype, device=x.device)


def pack_weight(w, dtype, cin_pad=None, cout_pad=None, out=None):
    """OIHW f32 parameter -> (Cout_pad, KH, KW, Cin_pad) K-contiguous tensor in `dtype`.
    out: a tensor this function returned earlier for a parameter of this shape — rewritten in place by ONE strided,
    casting copy (the zero padding is already there)."""
    O, I, KH, KW = w.shape
    if out is not None:
        out[:O, :, :, :I].copy_(w.detach().permute(0, 2, 3, 1))
        return out
    cin_pad = cin_pad or (I + 7) // 8 * 8
    cout_pad = cout_pad or (O + 7) // 8 * 8
    p = torch.zeros(cout_pad, KH, KW, cin_pad, dtype=dtype, device=w.device)
    p[:O, :, :, :I].copy_(w.detach().permute(0, 2, 3, 1))
    return p


def pack_weight_dgrad(w, dtype, out=None):
    """OIHW f32 parameter -> data-gradient weights (Cin_pad, KH, KW, Cout_pad): taps flipped, I/O swapped.
    out: as in pack_weight."""
    O, I, KH, KW = w.shape
    p = out if out is not None else torch.zeros((I + 7) // 8 * 8, KH, KW, (O + 7) // 8 * 8, dtype=dtype, device=w.device)
    src = w.detach().permute(1, 2, 3, 0)
    p[:I, :, :, :O].copy_(src.flip(1, 2) if KH * KW > 1 else src)
    return p


def pack_conv_weights(flat_src, fwd_dst, dgrad_dst, table_dev, n_entries, total_tiles, dgrad_s2_dst=None):
    """One launch packing every conv weight listed in the device table (see das_pack_conv_weights). dgrad_s2_dst: the
    buffer the stride-2 layers' parity-class operands go to (entries with s2_pad >= 0), or None."""
    _need_gpu(flat_src, dgrad_dst, dgrad_s2_dst)
    _lib.check(_lib.load().das_pack_conv_weights(_ptr(flat_src), _ptr(fwd_dst), _ptr(dgrad_dst), _ptr(dgrad_s2_dst),
                                                 _DT[dgrad_dst.dtype], _ptr(table_dev), n_entries, total_tiles, _stream()),
               'das_pack_conv_weights')


def _s2_classes(k, pad):
    """Stride-2 data gradient by output parity. With the flipped weights Wf (pad' = k-1-pad), dX[h] sums
    dY_up[h - pad' + a] * Wf[a] over the taps a whose dY_up sample is not an inserted zero: a = a0, a0 + 2, ... with
    a0 = (h + pad') % 2, reading dY[i + a' + c] for h = 2 i + ph, a = a0 + 2 a'. Per parity: (a0, taps, pad = -c)."""
    padp, out = k - 1 - pad, []
    for ph in (0, 1):
        a0 = (ph + padp) % 2
        nt = max(0, (k - a0 + 1) // 2)
        c2 = ph - padp + a0
        assert c2 % 2 == 0
        out.append((a0, nt, -(c2 // 2)))
    return out


def dgrad_s2_weights(w_dgrad, k, pad):
    """The flipped weights (Cin, k, k, Cout) of a stride-2 conv split by output parity: {(ph, pw): (Cin, nth, ntw, Cout)
    contiguous} for the classes that have taps (see _s2_classes)."""
    cls, out = _s2_classes(k, pad), {}
    for ph, (ah, nh, _) in enumerate(cls):
        for pw, (aw, nw, _) in enumerate(cls):
            if nh and nw:
                out[(ph, pw)] = w_dgrad if (k == 1) else w_dgrad[:, ah::2, aw::2, :].contiguous()
    return out


def s2_decomposable(k, pad):
    """stride-2 data gradient through the parity classes: taps exist for some class, non-negative class padding"""
    cls = _s2_classes(k, pad)
    return any(n for _, n, _ in cls) and all(p >= 0 for _, n, p in cls if n) and len({p for _, n, p in cls if n}) == 1


def conv2d_dgrad(dy, w_dgrad, KH, KW, stride, pad, in_hw, residual=None, bn_bwd=None, stats=None, w_classes=None,
                 accumulate=None):
    """dX of conv(x, w, stride, pad): a stride-1 conv of the (zero-upsampled) dY with flipped weights.
    residual: another gradient of the same input, added in the epilogue (x that also feeds a skip path).
    bn_bwd + stats: x is the output of a train-mode BatchNorm (+ReLU) — see `conv2d`.
    Stride 2: one launch per output parity over the taps that hit dY samples (a quarter of the multiplies of the
    zero-upsampled form); w_classes = dgrad_s2_weights(...) if the caller caches them.
    accumulate: a tensor of dX's shape that already holds another gradient of x; the result is added into it in place
    (only the parities that have taps are touched: a stride-2 1x1 conv reaches a quarter of the pixels)."""
    if isinstance(dy, Ragged):
        assert accumulate is None
        return conv2d(dy, w_dgrad, KH, KW, 1, KH - 1 - pad, residual=residual, bn_bwd=bn_bwd, stats=stats)
    if stride == 2 and KH == KW and s2_decomposable(KH, pad):
        assert not isinstance(residual, tuple), 'masked residuals: stride-1 data gradients only'
        cls = _s2_classes(KH, pad)
        every = all(n for _, n, _ in cls)
        if every or (bn_bwd is None and stats is None):
            H, W = in_hw
            B, Cx = dy.shape[0], w_dgrad.shape[0]
            if accumulate is not None:
                assert residual is None and accumulate.shape == (B, H, W, Cx) and accumulate.is_contiguous()
                out, residual = accumulate, accumulate
            else:
                out = torch.empty(B, H, W, Cx, dtype=dy.dtype, device=dy.device)
            wc = w_classes if w_classes is not None else dgrad_s2_weights(w_dgrad, KH, pad)
            for ph, (ah, nh, padc) in enumerate(cls):
                for pw, (aw, nw, _) in enumerate(cls):
                    hs, ws = (H - ph + 1) // 2, (W - pw + 1) // 2
                    if hs < 1 or ws < 1:
                        continue
                    if nh and nw:
                        conv2d(dy, wc[(ph, pw)], nh, nw, 1, padc, residual=residual, bn_bwd=bn_bwd, stats=stats, out=out,
                               out_hw=(hs, ws), out_sub=(ph, pw))
                    elif residual is not out:   # no tap reaches this parity: the other gradient alone, or zero
                        if residual is not None:
                            out[:, ph::2, pw::2] = residual[:, ph::2, pw::2]
                        else:
                            out[:, ph::2, pw::2] = 0
            return out
    if accumulate is not None:
        assert residual is None
        residual = accumulate
    return conv2d(dy, w_dgrad, KH, KW, 1, KH - 1 - pad, in_up=stride, out_hw=in_hw, residual=residual, bn_bwd=bn_bwd,
                  stats=stats)


class BnBwd:
    """What a data-gradient conv needs to fold the BatchNorm-backward reduction of the layer that PRODUCED its output
    tensor into its epilogue (DasConvDesc.bnb_*): raw = that layer's pre-norm tensor, y = its post-ReLU output (only
    when a residual entered before the ReLU; None = recompute the mask from raw), per-channel mean / invstd / gamma /
    beta, relu flag."""
    __slots__ = ('raw', 'y', 'mean', 'invstd', 'gamma', 'beta', 'relu', 'bits')

    def __init__(self, raw, y, mean, invstd, gamma, beta, relu, bits=None):
        """bits: instead of y, its ReLU mask as recorded by the forward apply pass (relu_bits_buffer): 1/16 of y's bytes."""
        self.raw, self.y, self.mean, self.invstd, self.gamma, self.beta, self.relu = raw, y, mean, invstd, gamma, beta, relu
        self.bits = bits if y is None else None


def bn_backward_apply(dz, raw, mean, invstd, gamma, sums, dgamma_acc=None, dbeta_acc=None, stat_rows=0):
    """BatchNorm backward's apply pass alone (das_bn_backward_apply): dz already masked, sums f32[slots * 2C] from the
    producing data-gradient conv. Returns d_raw."""
    _need_gpu(dz, raw, sums)
    assert dz.is_contiguous() and raw.is_contiguous() and dz.shape == raw.shape and dz.dtype == raw.dtype
    Cc = raw.shape[-1]
    rows = raw.numel() // Cc
    draw = torch.empty_like(raw)
    with _timed('bn_bwd_apply_dz_kernel', 3 * raw.numel() * raw.element_size(), shape=(rows, Cc)):     # dZ, raw -> d raw
        _lib.check(_lib.load().das_bn_backward_apply(_ptr(dz), _ptr(raw), _DT[raw.dtype], rows, Cc, _ptr(mean), _ptr(invstd),
                                                     _ptr(gamma), _ptr(sums), sums.numel() // (2 * Cc), _ptr(draw),
                                                     _ptr(dgamma_acc), _ptr(dbeta_acc), stat_rows or rows, _stream()),
                   'das_bn_backward_apply')
    return draw


def conv2d_wgrad(x, dy, KH, KW, stride, pad, out=None, accumulate=False):
    """dW (Cout, KH, KW, Cin) f32 of conv(x, w, stride, pad) given dy; x/dy NHWC or Ragged.
    out + accumulate: add into an existing buffer of that layout (the optimizer's flat gradient)."""
    _need_gpu(x, dy)
    xd, dyd = _data(x), _data(dy)
    Cin, Cout = xd.shape[-1], dyd.shape[-1]
    ragged = isinstance(x, Ragged)
    if ragged:
        B, (H, W) = x.B, x.sizes[0]
        Ho, Wo = H, W
    else:
        B, H, W, _ = x.shape
        Ho, Wo = dy.shape[1], dy.shape[2]
    if out is None:
        assert not accumulate
        dw = torch.empty(Cout, KH, KW, Cin, dtype=torch.float32, device=xd.device)
    else:
        dw = out
        Cout = _wgrad_rows(dw, Cout, KH, KW, Cin)
    d = _lib.DasConvDesc(dtype=_DT[xd.dtype], out_dtype=_lib.DAS_F32, B=B, H=H, W=W, Cin=Cin, x_pix_stride=_ps(x),
                         Ho=Ho, Wo=Wo, Cout=Cout, y_pix_stride=_ps(dy), KH=KH, KW=KW, stride=stride, pad=pad,
                         num_levels=len(x.sizes) if ragged else 0)
    if ragged:
        for l, (h, w_) in enumerate(x.sizes):
            d.lvl_H[l], d.lvl_W[l] = h, w_
    assert dyd.dtype == xd.dtype
    if PROFILE is not None:
        i0 = _prof_mark()
    _lib.check(_lib.load().das_conv2d_wgrad_nhwc(_ptr(xd), _ptr(dyd), _ptr(dw), C.byref(d), int(accumulate), _stream()),
               'das_conv2d_wgrad_nhwc')
    if PROFILE is not None:
        e0 = e1 = _Span(i0, _prof_mark())
        rows = x.rows if ragged else B * Ho * Wo
        tag = f'conv_wgrad_kernel<{"bf16" if xd.dtype == torch.bfloat16 else "float"}>'
        nby = (xd.numel() + dyd.numel()) * xd.element_size() + dw.numel() * 4
        PROFILE.append((tag, 2.0 * rows * Cout * KH * KW * Cin, e0, e1, (B, H, W, Cin, Cout, KH, stride, 0), 1, float(nby)))
    return dw


def _wgrad_rows(out, width, KH, KW, Cin):
    """Output channels of a weight gradient written into `out`: the gradient tensor's width, or fewer when that width is the
    layer's channel count padded to a multiple of 8 (out = the rows that exist; the kernel stores no others)."""
    assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() % (KH * KW * Cin) == 0
    cout = out.numel() // (KH * KW * Cin)
    assert cout <= width < cout + 8 and (cout == width or width % 8 == 0), (cout, width)
    return cout


def _wgrad_desc(x, dy, KH, KW, stride, pad, out=None):
    xd, dyd = _data(x), _data(dy)
    Cin, Cout = xd.shape[-1], dyd.shape[-1]
    if out is not None:
        Cout = _wgrad_rows(out, Cout, KH, KW, Cin)
    ragged = isinstance(x, Ragged)
    if ragged:
        B, (H, W) = x.B, x.sizes[0]
        Ho, Wo = H, W
    else:
        B, H, W, _ = x.shape
        Ho, Wo = dy.shape[1], dy.shape[2]
    d = _lib.DasConvDesc(dtype=_DT[xd.dtype], out_dtype=_lib.DAS_F32, B=B, H=H, W=W, Cin=Cin, x_pix_stride=_ps(x),
                         Ho=Ho, Wo=Wo, Cout=Cout, y_pix_stride=_ps(dy), KH=KH, KW=KW, stride=stride, pad=pad,
                         num_levels=len(x.sizes) if ragged else 0)
    if ragged:
        for l, (h, w_) in enumerate(x.sizes):
            d.lvl_H[l], d.lvl_W[l] = h, w_
    assert dyd.dtype == xd.dtype
    rows = x.rows if ragged else B * Ho * Wo
    return d, 2.0 * rows * Cout * KH * KW * Cin


def _is_pp_wgrad(x, dy, KH, KW):
    """Kernel class das_conv2d_wgrad_batch will pick for this op (csrc/train_ops.hip `classify`): the 256 x 256
    ping-pong kernel for bf16 ops with K >= wgrad.pp_mink and Cout >= 256, the 128 x 128 kernel otherwise."""
    mink = C.c_longlong()
    _lib.load().das_tuning_get(b'wgrad.pp_mink', C.byref(mink))
    xd, dyd = _data(x), _data(dy)
    return xd.dtype == torch.bfloat16 and mink.value > 0 and KH * KW * xd.shape[-1] >= mink.value and dyd.shape[-1] >= 256


def conv2d_wgrad_batch(items, accumulate=True):
    """items: [(x, dy, KH, KW, stride, pad, out)], out = f32 (Cout,KH,KW,Cin) buffers (distinct) the results are ADDED
    to (accumulate=False: written). One das_conv2d_wgrad_batch call: the ops of one kernel class share a persistent,
    host-scheduled launch (see include/das_hip.h)."""
    n = len(items)
    if n == 0:
        return
    if PROFILE is not None:
        # measuring: one call per kernel class, so that the event pair around it times that kernel (+ its reduce pass)
        # alone; the launches themselves are the ones the single call would have made
        pp = [it for it in items if _is_pp_wgrad(it[0], it[1], it[2], it[3])]
        rest = [it for it in items if not _is_pp_wgrad(it[0], it[1], it[2], it[3])]
        if pp and rest:
            conv2d_wgrad_batch(pp, accumulate)
            conv2d_wgrad_batch(rest, accumulate)
            return
    descs = (_lib.DasConvDesc * n)()
    xs, dys, dws = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_void_p * n)()
    flops, nby = 0.0, 0.0
    for i, (x, dy, KH, KW, stride, pad, out) in enumerate(items):
        _need_gpu(x, dy, out)
        d, fl = _wgrad_desc(x, dy, KH, KW, stride, pad, out=out)
        descs[i] = d
        xs[i], dys[i], dws[i] = _data(x).data_ptr(), _data(dy).data_ptr(), out.data_ptr()
        flops += fl
        nby += (_data(x).numel() + _data(dy).numel()) * _data(x).element_size() + out.numel() * 4
    if PROFILE is not None:
        i0 = _prof_mark()
    _lib.check(_lib.load().das_conv2d_wgrad_batch(n, xs, dys, dws, descs, 1 if accumulate else 0, _stream()),
               'das_conv2d_wgrad_batch')
    if PROFILE is not None:
        e0 = e1 = _Span(i0, _prof_mark())
        x0, dy0, KH0, KW0 = items[0][:4]
        dt = 'bf16' if _data(x0).dtype == torch.bfloat16 else 'float'
        tag = 'conv_wgrad_pp_kernel' if _is_pp_wgrad(x0, dy0, KH0, KW0) else f'conv_wgrad_kernel<{dt}>'
        plan = last_wgrad_plan()
        # (tag, flops, events, shape key, ops in the launch, algorithmic bytes, kernel launches: + the reduce pass)
        PROFILE.append((tag, flops, e0, e1, ('batch', n), n, nby, 1 + (1 if plan['partial'] > 0 else 0)))


def colsum(x, acc=None):
    """f32[C] column sums over all rows of an NHWC tensor / Ragged (bias gradient).
    acc: f32[C'] the sums are ADDED to instead (the bias's slice of the flat gradient buffer); returns None then. C' may be
    up to 7 less than x's width (a layer whose channel count is not a multiple of 8: the padding columns are not summed)."""
    _need_gpu(x)
    xd = _data(x)
    Cc = xd.shape[-1]
    rows = 1
    for s in xd.shape[:-1]:
        rows *= s
    if acc is not None:
        n = acc.numel()
        assert acc.dtype == torch.float32 and acc.is_contiguous() and n <= Cc < n + 8 and (n == Cc or Cc % 8 == 0), (n, Cc)
        _lib.check(_lib.load().das_colsum_acc(_ptr(xd), _DT[xd.dtype], rows, n, _ps(x), _ptr(acc), _stream()), 'das_colsum_acc')
        return None
    out = torch.empty(Cc, dtype=torch.float32, device=xd.device)
    _lib.check(_lib.load().das_colsum(_ptr(xd), _DT[xd.dtype], rows, Cc, _ps(x), _ptr(out), _stream()), 'das_colsum')
    return out


def relu_bits_buffer(x):
    """u8 buffer for the ReLU mask of a tensor shaped like x: one byte per 16-byte vector (das_bn_train_apply relu_bits_out)."""
    assert (x.numel() * x.element_size()) % 16 == 0
    return torch.empty(x.numel() * x.element_size() // 16, dtype=torch.uint8, device=x.device)


def bn_train_backward(dy, y, raw, mean, invstd, gamma, relu, want_dres, beta=None, dgamma_acc=None, dbeta_acc=None, bits=None):
    """Returns d_raw, d_residual (or None), dgamma, dbeta. y=None with relu: the ReLU mask is recomputed
    from raw (needs beta; only valid when no residual entered before the ReLU). dgamma_acc/dbeta_acc:
    f32[C] buffers the parameter gradients are also added to. bits (with y=None, relu): the mask as recorded by the forward
    (relu_bits_buffer) — valid with a residual, and neither pass reads y."""
    _need_gpu(dy, raw, bits)
    if bits is not None:
        assert y is None and relu and bits.dtype == torch.uint8 and bits.numel() * 16 == raw.numel() * raw.element_size()
    assert dy.is_contiguous() and raw.is_contiguous() and (y is None or y.is_contiguous())
    Cc = raw.shape[-1]
    rows = raw.numel() // Cc
    draw = torch.empty_like(raw)
    dres = torch.empty_like(raw) if want_dres else None
    if dgamma_acc is not None:
        # the sums are scratch here (the gradients go to the accumulators): a slice of the zeroed arena
        from .nn import zeroed_stats
        sums, prezeroed = zeroed_stats(2 * Cc, raw.device), 1
    else:
        sums, prezeroed = torch.empty(2 * Cc, dtype=torch.float32, device=raw.device), 0
    assert not (relu and y is None and bits is None and want_dres), 'the recomputed mask ignores a residual'
    # two passes (the sums must be complete before anything can be applied): reduce reads dY, raw (, y); apply reads them
    # again and writes d raw (, d residual)
    nin = 3 if y is not None else 2
    with _timed('bn_bwd_reduce_kernel + bn_bwd_apply_kernel', (2 * nin + 1 + (1 if want_dres else 0)) * raw.numel() *
                raw.element_size() + (bits.numel() * 2 if bits is not None else 0), launches=2,
                shape=(rows, Cc, nin if bits is None else 'bits', int(bool(want_dres)))):
        if bits is not None:
            _lib.check(_lib.load().das_bn_train_backward_bits(_ptr(dy), _ptr(bits), _ptr(raw), _DT[raw.dtype], rows, Cc, _ptr(mean),
                                                              _ptr(invstd), _ptr(gamma), _ptr(draw), _ptr(dres), _ptr(sums),
                                                              prezeroed, _ptr(dgamma_acc), _ptr(dbeta_acc), _stream()),
                       'das_bn_train_backward_bits')
            return draw, dres, sums[Cc:], sums[:Cc]
        _lib.check(_lib.load().das_bn_train_backward(_ptr(dy), _ptr(y), _ptr(raw), _DT[raw.dtype], rows, Cc, _ptr(mean),
                                                     _ptr(invstd), _ptr(gamma), _ptr(beta), int(relu), _ptr(draw),
                                                     _ptr(dres), _ptr(sums), prezeroed, _ptr(dgamma_acc),
                                                     _ptr(dbeta_acc), _stream()), 'das_bn_train_backward')
    return draw, dres, sums[Cc:], sums[:Cc]


def bn_train_backward_sync(dy, y, raw, mean, invstd, gamma, relu, want_dres, beta, all_reduce, world, bits=None):
    """SyncBN backward in two phases around a cross-rank sum of the per-channel sums (what torch's
    SyncBatchNorm does with its all_reduce of sum_dy / sum_dy_xmu). `all_reduce(t)` sums t over the ranks in
    place. Returns d_raw, d_residual, dgamma, dbeta (the LOCAL parameter gradients: the data-parallel
    gradient all-reduce adds the ranks' contributions, exactly as for every other parameter). bits (with y=None, relu): the
    ReLU mask as recorded by the forward (relu_bits_buffer) instead of y."""
    _need_gpu(dy, raw, bits)
    assert dy.is_contiguous() and raw.is_contiguous() and (y is None or y.is_contiguous())
    Cc = raw.shape[-1]
    rows = raw.numel() // Cc
    draw = torch.empty_like(raw)
    dres = torch.empty_like(raw) if want_dres else None
    sums = torch.empty(2 * Cc, dtype=torch.float32, device=raw.device)
    lib = _lib.load()
    if bits is not None:
        assert y is None and relu and bits.dtype == torch.uint8 and bits.numel() * 16 == raw.numel() * raw.element_size()
        args = (_ptr(dy), _ptr(bits), _ptr(raw), _DT[raw.dtype], rows, Cc, _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(draw),
                _ptr(dres), _ptr(sums), 0, None, None)
        fn, name = lib.das_bn_train_backward_bits_phase, 'das_bn_train_backward_bits_phase'
    else:
        args = (_ptr(dy), _ptr(y), _ptr(raw), _DT[raw.dtype], rows, Cc, _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta),
                int(relu), _ptr(draw), _ptr(dres), _ptr(sums), 0, None, None)
        fn, name = lib.das_bn_train_backward_phase, 'das_bn_train_backward_phase'
    _lib.check(fn(*args, 1, rows, _stream()), name)
    local = sums.clone()
    all_reduce(sums)
    _lib.check(fn(*args, 2, rows * world, _stream()), name)
    return draw, dres, local[Cc:], local[:Cc]


def conv2d(x, w, KH, KW, stride=1, pad=0, scale=None, shift=None, residual=None, relu=False, relu_in=False,
           out_dtype=None, stats=None, out=None, in_up=1, out_hw=None, bn_bwd=None, out_sub=None):
    """x (B,H,W,Cin[view]) or Ragged; w packed (Cout,KH,KW,Cin). Returns y (B,Ho,Wo,Cout) / Ragged.
    in_up / out_hw: data-gradient mode (x zero-upsampled by in_up, explicit output size).
    bn_bwd (BnBwd) + stats: the output (conv + residual) is the gradient wrt a BatchNorm(+ReLU) layer's output; the
    kernel stores dZ = masked gradient and adds [sum dZ | sum dZ * xhat] into stats (slots as for the forward).
    out_sub = (ph, pw): `out` (required), residual and the bn_bwd tensors are (B, oH, oW, Cout); conv output pixel
    (i, j) of the out_hw grid lands on their pixel (2 i + ph, 2 j + pw) (DasConvDesc.out_sub)."""
    _need_gpu(x, w)
    lib = _lib.load()
    ragged = isinstance(x, Ragged)
    res_bits = None
    if isinstance(residual, tuple):    # (tensor, bits): the residual enters masked by recorded ReLU bits (bn_bwd launches only)
        residual, res_bits = residual
        assert bn_bwd is not None and res_bits.dtype == torch.uint8 and \
            res_bits.numel() * 16 == residual.numel() * residual.element_size() and residual.is_contiguous()
        _need_gpu(res_bits)
    xd = _data(x)
    Cin = xd.shape[-1]
    Cout = w.shape[0]
    assert w.shape[1:] == (KH, KW, Cin), (w.shape, (KH, KW, Cin))
    assert w.dtype == xd.dtype and w.is_contiguous()
    out_dtype = out_dtype or xd.dtype
    if ragged:
        assert stride == 1 and pad == KH // 2 and KH == KW
        B, (H, W) = x.B, x.sizes[0]
        Ho, Wo = H, W
        if out is None:
            out = x.new(Cout, out_dtype)
        assert isinstance(out, Ragged) and out.data.shape == (x.rows, Cout) and out.dtype == out_dtype
        rows = x.rows
    else:
        B, H, W, _ = x.shape
        if out_hw is not None:
            Ho, Wo = out_hw
        else:
            Ho = (H + 2 * pad - KH) // stride + 1
            Wo = (W + 2 * pad - KW) // stride + 1
        if out_sub is not None:
            assert out is not None and out.is_contiguous() and out.shape[0] == B and out.shape[3] == Cout and stride == 1
            full_rows = out.shape[0] * out.shape[1] * out.shape[2]
        else:
            if out is None:
                out = torch.empty(B, Ho, Wo, Cout, dtype=out_dtype, device=x.device)
            assert out.shape == (B, Ho, Wo, Cout)
        assert out.dtype == out_dtype
        rows = B * Ho * Wo
    od, rd = _data(out), _data(residual) if residual is not None else None
    for t in (scale, shift):
        assert t is None or (t.dtype == torch.float32 and t.numel() == Cout and t.is_contiguous())
    d = _lib.DasConvDesc(
        dtype=_DT[xd.dtype], out_dtype=_DT[out_dtype], B=B, H=H, W=W, Cin=Cin, x_pix_stride=_ps(x), Ho=Ho, Wo=Wo,
        Cout=Cout, y_pix_stride=_ps(out), KH=KH, KW=KW, stride=stride, pad=pad, relu_in=int(relu_in), relu=int(relu),
        scale=scale.data_ptr() if scale is not None else None, shift=shift.data_ptr() if shift is not None else None,
        residual=rd.data_ptr() if rd is not None else None, res_pix_stride=_ps(residual) if rd is not None else 0,
        stats=stats.data_ptr() if stats is not None else None, num_levels=len(x.sizes) if ragged else 0,
        in_up=in_up, stats_slots=stats.numel() // (2 * Cout) if stats is not None else 0)
    if out_sub is not None:
        assert not ragged
        d.out_sub, d.out_ph, d.out_pw, d.out_H, d.out_W = 1, out_sub[0], out_sub[1], out.shape[1], out.shape[2]
    if bn_bwd is not None:
        b = bn_bwd
        assert stats is not None and not relu and scale is None and shift is None and out_dtype == xd.dtype
        assert b.raw.is_contiguous() and b.raw.shape[-1] == Cout and b.raw.dtype == out_dtype
        assert b.raw.numel() == (full_rows if out_sub is not None else rows) * Cout
        assert b.y is None or (b.y.is_contiguous() and b.y.shape == b.raw.shape and b.y.dtype == out_dtype)
        d.bnb_raw, d.bnb_y = b.raw.data_ptr(), (b.y.data_ptr() if b.y is not None else None)
        d.bnb_mean, d.bnb_invstd = b.mean.data_ptr(), b.invstd.data_ptr()
        d.bnb_gamma, d.bnb_beta = b.gamma.data_ptr(), b.beta.data_ptr()
        d.bnb_relu, d.bnb_pix_stride = int(b.relu), Cout
        if b.bits is not None:
            assert b.bits.dtype == torch.uint8 and b.bits.numel() * 16 == b.raw.numel() * b.raw.element_size()
            d.bnb_mask_bits = b.bits.data_ptr()
        if res_bits is not None:
            d.residual_mask_bits = res_bits.data_ptr()
        _need_gpu(b.raw, b.y, b.mean, b.invstd, b.gamma, b.beta, b.bits)
    if ragged:
        for l, (h, w_) in enumerate(x.sizes):
            d.lvl_H[l], d.lvl_W[l] = h, w_
    if rd is not None:
        assert rd.shape == od.shape and rd.dtype == od.dtype
    if PROFILE is not None:
        i0 = _prof_mark()
    _lib.check(lib.das_conv2d_nhwc(_ptr(xd), _ptr(w), _ptr(od), C.byref(d), _stream()), 'das_conv2d_nhwc')
    if PROFILE is not None:
        e0 = e1 = _Span(i0, _prof_mark())
        tag = last_kernel()   # the kernel the launcher picked (das_last_kernel): equals the rocprof kernel family
        # algorithmic bytes: every operand once (x, weights, y, + residual, + the BatchNorm-backward operands)
        eo = od.element_size()
        nby = xd.numel() * xd.element_size() + Cout * KH * KW * Cin * xd.element_size() + rows * Cout * eo
        nby += rows * Cout * eo if rd is not None else 0
        if bn_bwd is not None:
            nby += rows * Cout * eo * (2 if bn_bwd.y is not None else 1) + (rows * Cout * eo // 16 if bn_bwd.bits is not None else 0)
        mode = ('s' if stats is not None and bn_bwd is None else '') + ('r' if rd is not None else '') + \
            ('' if bn_bwd is None else ('by' if bn_bwd.y is not None else 'bm' if bn_bwd.bits is not None else 'bx')) + \
            ('a' if scale is not None else '') + \
            ('u' if out_sub is not None else '')
        PROFILE.append((tag, 2.0 * rows * Cout * KH * KW * Cin, e0, e1,
                        (B, H, W, Cin, Cout, KH, stride, len(x.sizes) if ragged else 1, mode), 1, float(nby)))
    return out


def pack_image(img, dtype, cpad=8):
    """NCHW f32 -> NHWC dtype with channels zero-padded to cpad."""
    _need_gpu(img)
    B, Cc, H, W = img.shape
    img = img.contiguous().float()
    y = torch.empty(B, H, W, cpad, dtype=dtype, device=img.device)
    _lib.check(_lib.load().das_pack_nchw_to_nhwc(_ptr(img), _ptr(y), _DT[dtype], B, Cc, H, W, cpad, _stream()),
               'das_pack_nchw_to_nhwc')
    return y


def to_nchw_f32(x, c0=0, Cn=None):
    """NHWC (slice) -> dense NCHW f32."""
    _need_gpu(x)
    B, H, W, Cc = x.shape
    Cn = Cn or Cc - c0
    y = torch.empty(B, Cn, H, W, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().das_unpack_nhwc_to_nchw(_ptr(x), _ptr(y), _DT[x.dtype], B, Cn, H, W, _ps(x), c0, _stream()),
               'das_unpack_nhwc_to_nchw')
    return y


def maxpool3x3s2(x, return_argmax=False):
    """return_argmax: also the winning tap of every output element (u8, same shape as y) for maxpool3x3s2_backward_argmax."""
    _need_gpu(x)
    B, H, W, Cc = x.shape
    assert x.is_contiguous()
    y = torch.empty(B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Cc, dtype=x.dtype, device=x.device)
    if return_argmax:
        idx = torch.empty(y.shape, dtype=torch.uint8, device=x.device)
        _lib.check(_lib.load().das_maxpool3x3s2_argmax(_ptr(x), _ptr(y), _ptr(idx), _DT[x.dtype], B, H, W, Cc, _stream()),
                   'das_maxpool3x3s2_argmax')
        return y, idx
    _lib.check(_lib.load().das_maxpool3x3s2(_ptr(x), _ptr(y), _DT[x.dtype], B, H, W, Cc, _stream()), 'das_maxpool3x3s2')
    return y


def upsample_bilinear_ac(x, Ho, Wo, stats=None, stats_only=False):
    """stats: zeroed f32[slots * 2C] — also reduce the BatchNorm statistics of the OUTPUT (das_upsample_bilinear_ac_stats);
    stats_only: do not write the output at all (returns None)."""
    _need_gpu(x)
    B, H, W, Cc = x.shape
    assert x.is_contiguous()
    y = None if stats_only else torch.empty(B, Ho, Wo, Cc, dtype=x.dtype, device=x.device)
    if stats is not None:
        assert stats.dtype == torch.float32 and stats.numel() % (2 * Cc) == 0
        _lib.check(_lib.load().das_upsample_bilinear_ac_stats(_ptr(x), _ptr(y) if y is not None else None, _DT[x.dtype],
                                                              B, H, W, Cc, Ho, Wo, _ptr(stats),
                                                              stats.numel() // (2 * Cc), _stream()), 'das_upsample_bilinear_ac_stats')
        return y
    _lib.check(_lib.load().das_upsample_bilinear_ac(_ptr(x), _ptr(y), _DT[x.dtype], B, H, W, Cc, Ho, Wo, _stream()),
               'das_upsample_bilinear_ac')
    return y


def add_upsample_nearest(a, b):
    _need_gpu(a, b)
    B, H, W, Cc = a.shape
    assert a.is_contiguous() and b.is_contiguous() and a.dtype == b.dtype
    y = torch.empty_like(a)
    _lib.check(_lib.load().das_add_upsample_nearest(_ptr(a), _ptr(b), _ptr(y), _DT[a.dtype], B, H, W, Cc, b.shape[1],
                                                    b.shape[2], _stream()), 'das_add_upsample_nearest')
    return y


def add3(a, b, c=None, relu=False):
    """Elementwise a + b (+ c) on dense tensors (or dense Ragged rows)."""
    _need_gpu(a, b, c)
    ad, bd, cd = _data(a), _data(b), _data(c) if c is not None else None
    assert ad.is_contiguous() and bd.is_contiguous() and (cd is None or cd.is_contiguous())
    yd = torch.empty_like(ad)
    _lib.check(_lib.load().das_add3(_ptr(ad), _ptr(bd), _ptr(cd), _ptr(yd), _DT[ad.dtype], ad.numel(), int(relu),
                                    _stream()), 'das_add3')
    return a.like(yd) if isinstance(a, Ragged) else yd


def bn_train_apply(x, stats, gamma, beta, running_mean, running_var, momentum=0.1, eps=1e-5, residual=None,
                   relu=False, num_batches_tracked=None, stat_count=0, finalize_only=False, bits_out=None):
    """x (B,H,W,C) raw conv output, stats f32[slots][2C] from the conv epilogue. Returns y, mean, invstd.
    num_batches_tracked: the BatchNorm's int64 counter buffer, incremented on the device by the same launch.
    stat_count: global row count when `stats` was all-reduced over ranks (SyncBN); 0 = this tensor's rows."""
    _need_gpu(x, stats)
    assert x.is_contiguous()
    Cc = x.shape[-1]
    count = x.numel() // Cc
    y = None if finalize_only else torch.empty_like(x)     # (finalize_only: statistics published / advanced, no output)
    mi = torch.empty(2, Cc, dtype=torch.float32, device=x.device)
    mean, invstd = mi[0], mi[1]
    if num_batches_tracked is not None:
        assert num_batches_tracked.dtype == torch.int64 and num_batches_tracked.is_cuda
    with _timed('bn_apply_kernel', 0 if finalize_only else (3 if residual is not None else 2) * x.numel() * x.element_size(),
                shape=(count, Cc, int(residual is not None))):
        _lib.check(_lib.load().das_bn_train_apply(_ptr(x), _ptr(y), _DT[x.dtype], count, Cc, _ptr(stats), _ptr(gamma),
                                                  _ptr(beta), _ptr(running_mean), _ptr(running_var), momentum, eps,
                                                  _ptr(residual), int(relu), _ptr(mean), _ptr(invstd),
                                                  _ptr(num_batches_tracked), int(stat_count),
                                                  stats.numel() // (2 * Cc), _ptr(bits_out), _stream()),
                   'das_bn_train_apply')
    return y, mean, invstd


def bn_finalize_many(layers, outs=None):
    """bn_train_apply(finalize_only=True) of up to four layers in one launch (das_bn_finalize_many). layers: [(stats
    f32[slots][2C], C, count, running_mean, running_var, momentum, eps, num_batches_tracked)]; count = the rows behind the
    statistics (all ranks' for SyncBN). Returns [(mean, invstd)]; outs: [(2, C) f32 buffer or None] to publish them into."""
    n = len(layers)
    assert 1 <= n <= 4
    arr = (_lib.DasBnFinalize * n)()
    out = []
    for i, (stats, Cc, count, rm, rv, momentum, eps, nbt) in enumerate(layers):
        _need_gpu(stats)
        assert stats.dtype == torch.float32 and stats.is_contiguous() and stats.numel() % (2 * Cc) == 0
        assert nbt is None or (nbt.dtype == torch.int64 and nbt.is_cuda)
        mi = outs[i] if outs is not None and outs[i] is not None else torch.empty(2, Cc, dtype=torch.float32, device=stats.device)
        assert mi.shape == (2, Cc) and mi.dtype == torch.float32 and mi.is_contiguous()
        out.append((mi[0], mi[1]))
        f = arr[i]
        f.stats, f.stats_slots, f.C, f.count = stats.data_ptr(), stats.numel() // (2 * Cc), Cc, int(count)
        f.running_mean = rm.data_ptr() if rm is not None else None
        f.running_var = rv.data_ptr() if rv is not None else None
        f.momentum, f.eps = momentum, eps
        f.save_mean, f.save_invstd = mi[0].data_ptr(), mi[1].data_ptr()
        f.num_batches_tracked = nbt.data_ptr() if nbt is not None else None
    with _timed('bn_apply_kernel', 0, shape=(int(layers[0][2]), int(layers[0][1]), 0)):
        _lib.check(_lib.load().das_bn_finalize_many(arr, n, _stream()), 'das_bn_finalize_many')
    return out


def groupnorm_backward(dy, y, x, fwd_stats, gamma, G, eps=1e-5, relu=True, dgamma_acc=None, dbeta_acc=None, beta=None, ws=None):
    """Returns dx (same container as x), dgamma, dbeta. dgamma_acc / dbeta_acc (both): f32[C] slices of the flat gradient
    buffer the parameter gradients are ADDED to instead (dgamma, dbeta come back as None). y=None with relu (needs beta):
    the ReLU mask is recomputed from x, y is not read."""
    assert not (relu and y is None and beta is None)
    _need_gpu(dy, x)
    xd, dyd = _data(x), _data(dy)
    yd = _data(y) if y is not None else None
    lv = _levels(x)
    Cc = xd.shape[-1]
    dx = _empty_like_rows(x, Cc, xd.dtype)
    assert _ps(dy) == _ps(x) == _ps(dx) and (yd is None or _ps(y) == _ps(x)), 'operands must share the pixel stride'
    # (the three accumulators back to back: the call zeroes them with one fill)
    ngs = lv.num_levels * lv.B * G * 2
    if dgamma_acc is not None and dbeta_acc is not None:
        # ws: a ZEROED f32 buffer of at least ngs values for the group sums (a slice of a buffer the caller fills once for many
        # layers), only read and written by this call's launches
        assert ws is None or (ws.dtype == torch.float32 and ws.numel() >= ngs and ws.is_contiguous())
        gs = ws if ws is not None else torch.empty(ngs, dtype=torch.float32, device=xd.device)
        _lib.check(_lib.load().das_groupnorm_backward_acc(_ptr(dyd), _ptr(yd), _ptr(xd), _ptr(_data(dx)), _DT[xd.dtype],
                                                          C.byref(lv), Cc, _ps(x), G, _ptr(fwd_stats), _ptr(gamma), _ptr(beta), eps,
                                                          int(relu), _ptr(gs), _ptr(dgamma_acc), _ptr(dbeta_acc),
                                                          int(ws is not None), _stream()),
                   'das_groupnorm_backward_acc')
        return dx, None, None
    acc = torch.empty(ngs + 2 * Cc, dtype=torch.float32, device=xd.device)
    gs, dgamma, dbeta = acc[:ngs], acc[ngs:ngs + Cc], acc[ngs + Cc:]
    _lib.check(_lib.load().das_groupnorm_backward(_ptr(dyd), _ptr(yd), _ptr(xd), _ptr(_data(dx)), _DT[xd.dtype],
                                                  C.byref(lv), Cc, _ps(x), G, _ptr(fwd_stats), _ptr(gamma), _ptr(beta), eps,
                                                  int(relu), _ptr(gs), _ptr(dgamma), _ptr(dbeta), _stream()),
               'das_groupnorm_backward')
    return dx, dgamma, dbeta


def maxpool3x3s2_backward(x, dy):
    _need_gpu(x, dy)
    B, H, W, Cc = x.shape
    assert x.is_contiguous() and dy.is_contiguous()
    dx = torch.empty_like(x)
    _lib.check(_lib.load().das_maxpool3x3s2_backward(_ptr(x), _ptr(dy), _ptr(dx), _DT[x.dtype], B, H, W, Cc, _stream()),
               'das_maxpool3x3s2_backward')
    return dx


def maxpool3x3s2_backward_argmax(dy, idx, H, W):
    """dx (B, H, W, C) from dy and the forward's winning taps (maxpool3x3s2(x, return_argmax=True))."""
    _need_gpu(dy, idx)
    B, Ho, Wo, Cc = dy.shape
    assert dy.is_contiguous() and idx.is_contiguous() and idx.shape == dy.shape and idx.dtype == torch.uint8
    assert Ho == (H - 1) // 2 + 1 and Wo == (W - 1) // 2 + 1
    dx = torch.empty(B, H, W, Cc, dtype=dy.dtype, device=dy.device)
    _lib.check(_lib.load().das_maxpool3x3s2_backward_argmax(_ptr(dy), _ptr(idx), _ptr(dx), _DT[dy.dtype], B, H, W, Cc, _stream()),
               'das_maxpool3x3s2_backward_argmax')
    return dx


_UP_TABLES = {}


def _upsample_tables(H, Ho, device):
    """upsample_bilinear (align_corners) along one axis as a matrix U (Ho x H), built with the kernels' f32 arithmetic; returns
    the three diagonals of U^T U as f32[H][3] (offsets -1, 0, +1) and U^T 1 as f32[H] (das_upmerge_backward_lowres)."""
    key = (H, Ho, str(device))
    if key not in _UP_TABLES:
        import numpy as np
        s = np.float32(H - 1) / np.float32(Ho - 1) if Ho > 1 else np.float32(0)
        r = (s * np.arange(Ho, dtype=np.float32)).astype(np.float32)
        i0 = r.astype(np.int32)
        l1 = (r - i0.astype(np.float32)).astype(np.float32)
        l0 = (np.float32(1) - l1).astype(np.float32)
        ip = (i0 < H - 1).astype(np.int32)
        U = np.zeros((Ho, H), np.float64)
        np.add.at(U, (np.arange(Ho), i0), l0)
        np.add.at(U, (np.arange(Ho), i0 + ip), l1)
        A = U.T @ U
        a = np.zeros((H, 3), np.float32)
        for d in (-1, 0, 1):
            idx = np.arange(max(0, -d), min(H, H - d))
            a[idx, d + 1] = A[idx, idx + d]
        assert abs(A - np.triu(np.tril(A, 1), -1)).max() == 0
        _UP_TABLES[key] = (torch.from_numpy(a).to(device), torch.from_numpy(U.sum(0).astype(np.float32)).to(device))
    return _UP_TABLES[key]


def _bn_ptrs(bn1, bn2):
    arr = (C.c_void_p * 8)(*[_ptr(t) for t in (*bn1, *bn2)])
    return arr


def bn_dual_apply(raw1, bn1, raw2, bn2, relu=True, bits_out=None):
    """relu(BN1(raw1) + BN2(raw2)); bn = (mean, invstd, gamma, beta) f32[C] (das_bn_dual_apply)."""
    _need_gpu(raw1, raw2)
    assert raw1.is_contiguous() and raw2.is_contiguous() and raw1.shape == raw2.shape and raw1.dtype == raw2.dtype
    Cc = raw1.shape[-1]
    out = torch.empty_like(raw1)
    arr = _bn_ptrs(bn1, bn2)
    with _timed('bn_apply_kernel', 3 * raw1.numel() * raw1.element_size(), shape=(raw1.numel() // Cc, Cc, 'dual')):
        _lib.check(_lib.load().das_bn_dual_apply(_ptr(raw1), _ptr(raw2), _ptr(out), _DT[raw1.dtype], raw1.numel() // Cc, Cc, arr,
                                                 int(relu), _ptr(bits_out), _stream()), 'das_bn_dual_apply')
    return out


def bn_relu_add3_forward(x, raw1, bn1, raw2, bn2):
    """x + relu(BN1(raw1)) + relu(BN2(raw2)); bn = (mean, invstd, gamma, beta) f32[C] (das_bn_relu_add3_forward)."""
    _need_gpu(x, raw1, raw2)
    assert x.is_contiguous() and raw1.is_contiguous() and raw2.is_contiguous() and x.shape == raw1.shape == raw2.shape
    assert x.dtype == raw1.dtype == raw2.dtype
    Cc = x.shape[-1]
    out = torch.empty_like(x)
    arr = _bn_ptrs(bn1, bn2)
    with _timed('bn_apply_kernel', 4 * x.numel() * x.element_size(), shape=(x.numel() // Cc, Cc, 'skipadd')):
        _lib.check(_lib.load().das_bn_relu_add3_forward(_ptr(x), _ptr(raw1), _ptr(raw2), _ptr(out), _DT[x.dtype], x.numel() // Cc, Cc,
                                                        arr, _stream()), 'das_bn_relu_add3_forward')
    return out


def bn_relu_add3_backward(g, raw1, bn1, raw2, bn2, acc=None, all_reduce=None, world=1):
    """Returns d raw1, d raw2, sums f32[4C] = [dbeta1 | dgamma1 | dbeta2 | dgamma2] (this rank's); acc = (dgamma1, dbeta1,
    dgamma2, dbeta2) accumulators the parameter gradients are also added to (das_bn_relu_add3_backward).
    SyncBN (world > 1): the two passes run around all_reduce(sums) — the apply pass sees the sums over all ranks' rows."""
    _need_gpu(g, raw1, raw2)
    assert g.is_contiguous() and g.shape == raw1.shape == raw2.shape and g.dtype == raw1.dtype == raw2.dtype
    Cc = g.shape[-1]
    rows = g.numel() // Cc
    d1, d2 = torch.empty_like(raw1), torch.empty_like(raw2)
    from .nn import zeroed_stats
    sums = zeroed_stats(4 * Cc, g.device)
    arr = _bn_ptrs(bn1, bn2)
    a = acc if acc is not None else (None, None, None, None)
    lib = _lib.load()
    if world > 1:
        assert acc is None, 'the accumulators take LOCAL sums: add them from the returned sums'
        _lib.check(lib.das_bn_relu_add3_backward(_ptr(g), _ptr(raw1), _ptr(raw2), None, None, _DT[g.dtype], rows, Cc, arr, _ptr(sums),
                                                 1, rows, None, None, None, None, 1, _stream()), 'das_bn_relu_add3_backward')
        glob = sums.clone()
        all_reduce(glob)
        _lib.check(lib.das_bn_relu_add3_backward(_ptr(g), _ptr(raw1), _ptr(raw2), _ptr(d1), _ptr(d2), _DT[g.dtype], rows, Cc, arr,
                                                 _ptr(glob), 1, rows * world, None, None, None, None, 2, _stream()),
                   'das_bn_relu_add3_backward')
        return d1, d2, sums
    with _timed('bn_bwd_reduce_kernel + bn_bwd_apply_kernel', 8 * g.numel() * g.element_size(), launches=2, shape=(rows, Cc, 'skipadd')):
        _lib.check(lib.das_bn_relu_add3_backward(_ptr(g), _ptr(raw1), _ptr(raw2), _ptr(d1), _ptr(d2), _DT[g.dtype], rows, Cc,
                                                 arr, _ptr(sums), 1, rows, _ptr(a[0]), _ptr(a[1]), _ptr(a[2]), _ptr(a[3]), 0,
                                                 _stream()), 'das_bn_relu_add3_backward')
    return d1, d2, sums


def upsample_stats_lowres(z, Ho, Wo, stats):
    """Batch statistics of upsample_bilinear_ac(z, Ho, Wo) (f32, unrounded) added into the zeroed stats f32[slots * 2C], from
    z alone (das_upsample_stats_lowres)."""
    _need_gpu(z, stats)
    B, H, W, Cc = z.shape
    assert z.is_contiguous() and stats.dtype == torch.float32 and stats.numel() % (2 * Cc) == 0
    ah, wh = _upsample_tables(H, Ho, z.device)
    aw, ww = _upsample_tables(W, Wo, z.device)
    with _timed('bn_apply_kernel', z.numel() * z.element_size(), shape=(z.numel() // Cc, Cc, 'upstats')):
        _upsample_stats_lowres(z, B, H, W, Cc, ah, aw, wh, ww, stats)


def _upsample_stats_lowres(z, B, H, W, Cc, ah, aw, wh, ww, stats):
    _lib.check(_lib.load().das_upsample_stats_lowres(_ptr(z), _DT[z.dtype], B, H, W, Cc, _ptr(ah), _ptr(aw), _ptr(wh), _ptr(ww),
                                                     _ptr(stats), stats.numel() // (2 * Cc), _stream()), 'das_upsample_stats_lowres')


def upmerge_forward(raw1, z, bn1, bn2, bits_out=None):
    """relu(BN1(raw1) + BN2(upsample(z))); bn = (mean, invstd, gamma, beta) f32[C] each (das_upmerge_forward)."""
    _need_gpu(raw1, z)
    B, Ho, Wo, Cc = raw1.shape
    assert raw1.is_contiguous() and z.is_contiguous() and z.shape[0] == B and z.shape[3] == Cc and z.dtype == raw1.dtype
    out = torch.empty_like(raw1)
    with _timed('bn_apply_kernel', 2 * raw1.numel() * raw1.element_size() + z.numel() * z.element_size(),
                shape=(raw1.numel() // Cc, Cc, 'upmerge')):
        _lib.check(_lib.load().das_upmerge_forward(_ptr(raw1), _ptr(z), _ptr(out), _DT[raw1.dtype], B, z.shape[1], z.shape[2], Cc, Ho,
                                                   Wo, *[_ptr(t) for t in bn1], *[_ptr(t) for t in bn2], _ptr(bits_out), _stream()),
                   'das_upmerge_forward')
    return out


def upmerge_backward_reduce(dy, out, raw1, z, mean1, invstd1, mean2, invstd2, bits=None):
    """Returns dzm = dy * (out > 0) and sums f32[3C] = [sum dZ | sum dZ xhat1 | sum dZ xhat2] (das_upmerge_backward_reduce)."""
    _need_gpu(dy, raw1, z)
    B, Ho, Wo, Cc = raw1.shape
    assert dy.is_contiguous() and dy.shape == raw1.shape and dy.dtype == raw1.dtype == z.dtype
    assert (out is None) != (bits is None) and (out is None or (out.shape == raw1.shape and out.dtype == raw1.dtype))
    dzm = torch.empty_like(raw1)
    from .nn import zeroed_stats
    sums = zeroed_stats(3 * Cc, raw1.device)
    with _timed('bn_bwd_reduce_kernel + bn_bwd_apply_kernel', (4 if bits is None else 3) * raw1.numel() * raw1.element_size() +
                z.numel() * z.element_size() + (bits.numel() if bits is not None else 0), shape=(raw1.numel() // Cc, Cc, 'upmerge')):
        _lib.check(_lib.load().das_upmerge_backward_reduce(_ptr(dy), _ptr(out), _ptr(bits), _ptr(raw1), _ptr(z), _ptr(dzm), _DT[raw1.dtype], B,
                                                           z.shape[1], z.shape[2], Cc, Ho, Wo, _ptr(mean1), _ptr(invstd1),
                                                           _ptr(mean2), _ptr(invstd2), _ptr(sums), 1, _stream()),
                   'das_upmerge_backward_reduce')
    return dzm, sums


def upmerge_backward_lowres(P, z, Ho, Wo, sums, gamma2, mean2, invstd2, stat_rows, dgamma2_acc=None, dbeta2_acc=None):
    """dz = upsample^T(d raw2) at low resolution (das_upmerge_backward_lowres)."""
    _need_gpu(P, z, sums)
    B, H, W, Cc = z.shape
    assert P.shape == z.shape and P.dtype == z.dtype and P.is_contiguous() and z.is_contiguous()
    ah, wh = _upsample_tables(H, Ho, z.device)
    aw, ww = _upsample_tables(W, Wo, z.device)
    dz = torch.empty_like(z)
    with _timed('bn_bwd_reduce_kernel + bn_bwd_apply_kernel', 3 * z.numel() * z.element_size(), shape=(z.numel() // Cc, Cc, 'uplow')):
        _lib.check(_lib.load().das_upmerge_backward_lowres(_ptr(P), _ptr(z), _ptr(dz), _DT[z.dtype], B, H, W, Cc, _ptr(ah), _ptr(aw),
                                                           _ptr(wh), _ptr(ww), _ptr(sums), _ptr(gamma2), _ptr(mean2), _ptr(invstd2),
                                                           int(stat_rows), _ptr(dgamma2_acc), _ptr(dbeta2_acc), _stream()),
                   'das_upmerge_backward_lowres')
    return dz


def upsample_bilinear_ac_backward(dy, H, W):
    _need_gpu(dy)
    B, Ho, Wo, Cc = dy.shape
    assert dy.is_contiguous()
    dx = torch.empty(B, H, W, Cc, dtype=dy.dtype, device=dy.device)
    _lib.check(_lib.load().das_upsample_bilinear_ac_backward(_ptr(dy), _ptr(dx), _DT[dy.dtype], B, H, W, Cc, Ho, Wo,
                                                             _stream()), 'das_upsample_bilinear_ac_backward')
    return dx


def upsample_nearest_backward(dy, Hb, Wb):
    _need_gpu(dy)
    B, H, W, Cc = dy.shape
    assert dy.is_contiguous()
    db = torch.empty(B, Hb, Wb, Cc, dtype=dy.dtype, device=dy.device)
    _lib.check(_lib.load().das_upsample_nearest_backward(_ptr(dy), _ptr(db), _DT[dy.dtype], B, H, W, Cc, Hb, Wb,
                                                         _stream()), 'das_upsample_nearest_backward')
    return db


def groupnorm_stats_size(x, G):
    """f32 values of a GroupNorm call's statistics workspace (sum, sum of squares per level, image and group)"""
    lv = _levels(x)
    return lv.num_levels * lv.B * G * 2


def groupnorm(x, gamma, beta, G, eps=1e-5, relu=True, out=None, return_stats=False, ws=None):
    """In/out NHWC or Ragged (may be channel-slice views with a pixel stride); default in place.
    ws: a ZEROED f32 buffer of groupnorm_stats_size values for the statistics (a slice of a buffer the caller fills once for
    many layers); it is what return_stats hands back, so it must live as long as the backward needs it."""
    _need_gpu(x)
    out = x if out is None else out
    xd, od = _data(x), _data(out)
    lv = _levels(x)
    n = lv.num_levels * lv.B * G * 2
    zeroed = ws is not None
    if zeroed:
        assert ws.dtype == torch.float32 and ws.numel() == n and ws.is_contiguous()
    else:
        ws = torch.empty(n, dtype=torch.float32, device=xd.device)
    assert _ps(out) == _ps(x)
    _lib.check(_lib.load().das_groupnorm_nhwc(_ptr(xd), _ptr(od), _DT[xd.dtype], C.byref(lv), xd.shape[-1], _ps(x), G,
                                              _ptr(gamma), _ptr(beta), eps, int(relu), _ptr(ws), int(zeroed), _stream()),
               'das_groupnorm_nhwc')
    return (out, ws) if return_stats else out


def deform_im2col3x3(x, om):
    """x rows x C (NHWC / Ragged, may be a slice view); om rows x >=27 f32 -> col rows x 9C."""
    _need_gpu(x, om)
    xd, omd = _data(x), _data(om)
    Cc = xd.shape[-1]
    assert omd.dtype == torch.float32
    col = _empty_like_rows(x, 9 * Cc, xd.dtype)
    lv = _levels(x)
    _lib.check(_lib.load().das_deform_im2col3x3(_ptr(xd), _ptr(omd), _ptr(_data(col)), _DT[xd.dtype], C.byref(lv), Cc,
                                                _ps(x), _ps(om), _stream()), 'das_deform_im2col3x3')
    return col


def dcn3x3_fused(x, om, w, bias=None, want_col=False):
    """DCNv2 forward in one kernel (das_dcn3x3_fused): x NHWC / Ragged bf16 (C % 64 == 0), om f32 offsets + mask logits,
    w (Cout, 1, 1, 9 C) bf16 GEMM weight (Cout <= 256), bias f32 padded to 8 or None. Returns y, or (y, col) with
    want_col (the sampled operand as das_deform_im2col3x3 would have written it: kept for the weight gradient)."""
    _need_gpu(x, om, w)
    xd, omd = _data(x), _data(om)
    Cc, Cout = xd.shape[-1], w.shape[0]
    assert xd.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and omd.dtype == torch.float32
    assert w.numel() == Cout * 9 * Cc and w.is_contiguous()
    lv = _levels(x)
    y = _empty_like_rows(x, Cout, xd.dtype)
    col = _empty_like_rows(x, 9 * Cc, xd.dtype) if want_col else None
    i0 = _prof_mark() if PROFILE is not None else 0
    _lib.check(_lib.load().das_dcn3x3_fused(_ptr(xd), _ptr(omd), _ptr(w), _ptr(bias), _ptr(_data(y)),
                                            _ptr(_data(col)) if col is not None else None, _DT[xd.dtype], C.byref(lv), Cc,
                                            Cout, _ps(x), _ps(om), _ps(y), _stream()), 'das_dcn3x3_fused')
    if PROFILE is not None:   # a conv family of its own: the GEMM's FLOPs, every operand once (x, offsets, weights -> y, col)
        e0 = e1 = _Span(i0, _prof_mark())
        rows = xd.numel() // Cc
        nby = (xd.numel() + w.numel() + _data(y).numel() + (_data(col).numel() if want_col else 0)) * 2 + omd.numel() * 4
        ragged = isinstance(x, Ragged)
        PROFILE.append(('dcn3x3_fused_kernel', 2.0 * rows * Cout * 9 * Cc, e0, e1,
                        (x.B if ragged else xd.shape[0], 0, 0, Cc, Cout, 3, 1, len(x.sizes) if ragged else 1,
                         'dcn+col' if want_col else 'dcn'), 1, float(nby)))
    return (y, col) if want_col else y


def offset_sample(uvd, samp_off, conf, J, heads=4):
    _need_gpu(uvd, samp_off, conf)
    out = _empty_like_rows(uvd, 3 * J, torch.float32)
    lv = _levels(uvd)
    _lib.check(_lib.load().das_offset_sample(_ptr(_data(uvd)), _ptr(_data(samp_off)), _ptr(_data(conf)),
                                             _ptr(_data(out)), C.byref(lv), J, heads, _ps(uvd), _ps(samp_off),
                                             _ps(conf), _ps(out), _stream()), 'das_offset_sample')
    return out


def sigmoid_blend(off, w, nxt):
    _need_gpu(off, w, nxt)
    od = _data(off)
    Cc = od.shape[-1]
    out = _empty_like_rows(off, Cc, torch.float32)
    npix = 1
    for s in od.shape[:-1]:
        npix *= s
    _lib.check(_lib.load().das_sigmoid_blend(_ptr(od), _ptr(_data(w)), _ptr(_data(nxt)), _ptr(_data(out)), npix, Cc,
                                             _ps(off), _ps(w), _ps(nxt), _ps(out), _stream()), 'das_sigmoid_blend')
    return out


def head_desc(J, root_idx, raw_ps, off_c, depth_c, uvd_c, sigma_c, scales, strides, z_norm, depth_factor, scale_dev=None):
    """scales: per level [offset, depth, uv, d] as host floats, or None with scale_dev = the same as an f32 (levels, 4) DEVICE
    tensor (kept alive by the descriptor object: `_scale_dev`); strides: per level head stride."""
    d = _lib.DasHeadDesc(J=J, root_idx=root_idx, raw_ps=raw_ps, off_c=off_c, depth_c=depth_c, uvd_c=uvd_c,
                         sigma_c=sigma_c, z_norm=float(z_norm), depth_factor=float(depth_factor))
    if scale_dev is not None:
        _need_gpu(scale_dev)
        assert scales is None and scale_dev.dtype == torch.float32 and scale_dev.is_contiguous() and scale_dev.shape == (len(strides), 4)
        d.scale_dev = scale_dev.data_ptr()
        d._scale_dev = scale_dev
        scales = ()
    for l, sc in enumerate(scales):
        for k in range(4):
            d.scale[l][k] = float(sc[k])
    for l, s in enumerate(strides):
        d.level_stride[l] = float(s)
    return d


def head_assemble(raw, desc):
    """raw rows x raw_ps f32 -> pose_pred rows x (3+6J), uvd rows x 3J."""
    _need_gpu(raw)
    J = desc.J
    pose = _empty_like_rows(raw, 3 + 6 * J, torch.float32)
    uvd = _empty_like_rows(raw, 3 * J, torch.float32)
    lv = _levels(raw)
    _lib.check(_lib.load().das_head_assemble(_ptr(_data(raw)), _ptr(_data(pose)), _ptr(_data(uvd)), C.byref(lv),
                                             C.byref(desc), _stream()), 'das_head_assemble')
    return pose, uvd


def head_finalize(pose, ref, desc, eval_mode):
    _need_gpu(pose, ref)
    lv = _levels(pose)
    _lib.check(_lib.load().das_head_finalize(_ptr(_data(pose)), _ptr(_data(ref)), C.byref(lv), C.byref(desc), _ps(ref),
                                             int(eval_mode), _stream()), 'das_head_finalize')
    return pose, ref


def deform_im2col3x3_backward(x, om, dcol):
    """Returns dx (rows, C) f32 and dom (rows, om channels) f32 (same containers as x / om)."""
    _need_gpu(x, om, dcol)
    xd, omd, dcd = _data(x), _data(om), _data(dcol)
    Cc = xd.shape[-1]
    dx = _empty_like_rows(x, Cc, torch.float32)
    dom = _empty_like_rows(om, omd.shape[-1], torch.float32)
    _data(dom).zero_()   # (dx is written in full by the call)
    lv = _levels(x)
    assert dcd.is_contiguous() and dcd.dtype == xd.dtype
    _lib.check(_lib.load().das_deform_im2col3x3_backward(_ptr(xd), _ptr(omd), _ptr(dcd), _ptr(_data(dx)),
                                                         _ptr(_data(dom)), _DT[xd.dtype], C.byref(lv), Cc, _ps(x),
                                                         _ps(om), _ps(dom), _stream()), 'das_deform_im2col3x3_backward')
    return dx, dom


def offset_sample_backward(uvd, samp_off, conf, gout, J, heads=4):
    _need_gpu(uvd, samp_off, conf, gout)
    d_uvd = _empty_like_rows(uvd, 3 * J, torch.float32)
    d_so = _empty_like_rows(uvd, 8 * J, torch.float32)
    d_conf = _empty_like_rows(uvd, 3 * J, torch.float32)
    for t in (d_uvd, d_so, d_conf):
        _data(t).zero_()
    lv = _levels(uvd)
    _lib.check(_lib.load().das_offset_sample_backward(_ptr(_data(uvd)), _ptr(_data(samp_off)), _ptr(_data(conf)),
                                                      _ptr(_data(gout)), _ptr(_data(d_uvd)), _ptr(_data(d_so)),
                                                      _ptr(_data(d_conf)), C.byref(lv), J, heads, _ps(uvd),
                                                      _ps(samp_off), _ps(conf), _ps(gout), _stream()),
               'das_offset_sample_backward')
    return d_uvd, d_so, d_conf


def sigmoid_blend_backward(off, w, nxt, gout):
    _need_gpu(off, w, nxt, gout)
    od, gd = _data(off), _data(gout)
    Cc = od.shape[-1]
    assert gd.is_contiguous()
    d_off, d_w, d_nxt = (torch.empty_like(gd) for _ in range(3))
    npix = gd.numel() // Cc
    _lib.check(_lib.load().das_sigmoid_blend_backward(_ptr(od), _ptr(_data(w)), _ptr(_data(nxt)), _ptr(gd), _ptr(d_off),
                                                      _ptr(d_w), _ptr(d_nxt), npix, Cc, _ps(off), _ps(w), _ps(nxt),
                                                      _stream()), 'das_sigmoid_blend_backward')
    return d_off, d_w, d_nxt


def head_assemble_backward(raw, d_pose, d_uvd, desc):
    """Returns d_raw (same shape as raw, zero outside the off/depth/uvd/sigma slices) and d_scale (5,4)."""
    _need_gpu(raw, d_pose, d_uvd)
    rd = _data(raw)
    d_raw = torch.zeros_like(rd)
    assert rd.is_contiguous() and _data(d_pose).is_contiguous() and _data(d_uvd).is_contiguous()
    d_scale = torch.empty(5, 4, dtype=torch.float32, device=rd.device)
    lv = _levels(raw)
    _lib.check(_lib.load().das_head_assemble_backward(_ptr(rd), _ptr(_data(d_pose)), _ptr(_data(d_uvd)), _ptr(d_raw),
                                                      _ptr(d_scale), C.byref(lv), C.byref(desc), _stream()),
               'das_head_assemble_backward')
    return d_raw, d_scale


def decode(cls_list, ctr_list, pose_list, strides, scale_factors, J, nms_pre, nms_post, score_thr, nms_thr, nms_soft=False):
    """Eval-mode head outputs per level, NHWC f32: cls/ctr (B,H,W,c>=1) logits in channel 0,
    pose (B,H,W,>=3+3J). scale_factors: (B,2) f32 device tensor. Returns dict of device tensors:
    count (B,), scores (B,nms_post), poses (B,nms_post,J,3), centers (B,nms_post,3), index (B,nms_post).
    nms_soft: soft OKS-NMS (`nms_type='soft'`, pose_nms.py:128-194) instead of the greedy one."""
    _need_gpu(*cls_list, *ctr_list, *pose_list, scale_factors)
    lib = _lib.load()
    B = cls_list[0].shape[0]
    L = len(cls_list)
    d = _lib.DasDecodeDesc(B=B, J=J, num_levels=L, nms_pre=nms_pre, nms_post=nms_post, score_thr=score_thr,
                           nms_thr=nms_thr, scale_factor=scale_factors.data_ptr(), nms_soft=1 if nms_soft else 0)
    for l in range(L):
        d.H[l], d.W[l], d.stride[l] = cls_list[l].shape[1], cls_list[l].shape[2], strides[l]
        d.cls[l], d.ctr[l], d.pose[l] = cls_list[l].data_ptr(), ctr_list[l].data_ptr(), pose_list[l].data_ptr()
        d.cls_ps[l], d.ctr_ps[l], d.pose_ps[l] = _ps(cls_list[l]), _ps(ctr_list[l]), _ps(pose_list[l])
        for t in (cls_list[l], ctr_list[l], pose_list[l]):
            assert t.dtype == torch.float32
    assert scale_factors.dtype == torch.float32 and scale_factors.shape == (B, 2) and scale_factors.is_contiguous()
    cap = lib.das_decode_cap(C.byref(d))
    dev = cls_list[0].device
    ws = torch.empty(lib.das_decode_ws_bytes(B, cap, J), dtype=torch.uint8, device=dev)
    out = dict(count=torch.zeros(B, dtype=torch.int32, device=dev),
               scores=torch.zeros(B, nms_post, dtype=torch.float32, device=dev),
               poses=torch.zeros(B, nms_post, J, 3, dtype=torch.float32, device=dev),
               centers=torch.zeros(B, nms_post, 3, dtype=torch.float32, device=dev),
               index=torch.zeros(B, nms_post, dtype=torch.int32, device=dev))
    _lib.check(lib.das_decode(C.byref(d), _ptr(out['scores']), _ptr(out['poses']), _ptr(out['centers']),
                              _ptr(out['index']), _ptr(out['count']), _ptr(ws), _stream()), 'das_decode')
    return out
