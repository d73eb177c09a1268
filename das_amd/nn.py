"""Layer containers and fused-op helpers shared by the backbone, neck and head.

Parameters live in ordinary `nn.Conv2d` / `nn.BatchNorm2d` / `nn.GroupNorm` modules (OIHW, f32)
so that state-dict keys and shapes equal the reference's checkpoints
(`backbone.top.top.0.conv.weight`, `...bn.running_mean`, `bbox_head.cls_convs.1.conv.conv_offset.weight`
...). Those modules are never *called*: the forward path packs their parameters once into the
NHWC/K-contiguous layout the HIP kernels want (cache keyed on parameter version) and launches
libdas_hip.so through `das_amd.ops`.
"""
import math

import torch
import torch.nn as nn

from . import ops


# ------------------------------------------------------------------ layout helpers
def as_nhwc(t, dtype):
    """Accept an NCHW-shaped tensor (dense NCHW, or a channels-last view produced by `to_nchw_view`)
    and return an NHWC (B,H,W,C) tensor of `dtype` without copying when possible."""
    assert t.dim() == 4
    v = t.permute(0, 2, 3, 1)
    if v.is_contiguous() and t.dtype == dtype and t.shape[1] % 8 == 0:
        return v
    if t.dtype == torch.float32 and t.is_contiguous():
        return ops.pack_image(t, dtype, (t.shape[1] + 7) // 8 * 8)
    return ops.pack_image(t.float().contiguous(), dtype, (t.shape[1] + 7) // 8 * 8)


def to_nchw_view(x):
    """NHWC (B,H,W,C) -> logical NCHW view (channels-last strides), no copy."""
    return x.permute(0, 3, 1, 2)


# ------------------------------------------------------------------ parameter packing cache
# The fused optimizer updates parameters through raw pointers (no autograd version bump), so it
# advances this counter instead; it is part of every cache key.
PARAM_EPOCH = [0]


def bump_param_epoch():
    PARAM_EPOCH[0] += 1


def _versions(*ts):
    return (PARAM_EPOCH[0],) + tuple((t.data_ptr(), t._version) for t in ts if t is not None)


class _Cache:
    """Per-module cache of packed tensors, invalidated when any source tensor changes."""

    def __init__(self):
        self.store = {}

    def get(self, key, sources, make, refresh=None):
        """refresh(value): rewrite a stale cached value IN PLACE (same shape, same memory) instead of building a new one —
        the parameters change every optimisation step, so a training step refreshes every entry once; a rebuild costs
        an allocation + fill + cast + copy (four to six small launches), a refresh one strided copy.
        CONSTRAINT (the same one the optimizer's once-per-step packed buffers impose, optim.FlatSGD / _Slot.packed): packed
        operands are views of memory that is rewritten when the parameters change, and the backward functions fetch the
        data-gradient weights at backward time — so a backward pass must run BEFORE the parameters it differentiates are
        updated (forward, backward, step: what train_iteration and tools/train.py do). Two forward passes with an
        optimizer step between them followed by the first one's backward, or parameter edits (EMA, manual surgery)
        between a forward and its backward, would differentiate the NEW weights. Not reachable from train_iteration;
        stated here because a rebuilt tensor (the behaviour without `refresh`) would have kept the old values alive."""
        ver = _versions(*sources)
        hit = self.store.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        with torch.no_grad():
            if hit is not None and refresh is not None:
                val = hit[1]
                refresh(val)
            else:
                val = make()
        self.store[key] = (ver, val)
        return val


def _cache_of(mod):
    c = mod.__dict__.get('_das_cache')
    if c is None:
        c = _Cache()
        mod.__dict__['_das_cache'] = c
    return c


def _slot_of(weight, cin=None):
    """The optimizer's record of a conv weight if its flat storage is directly usable as a kernel operand
    (channels multiples of 8, no input-channel padding needed)."""
    sl = getattr(weight, '_das_slot', None)
    if sl is None or not sl.packable or (cin is not None and cin != sl.cl_shape[3]):
        return None
    return sl


def packed_weight(conv, dtype, cin_pad=None):
    sl = _slot_of(conv.weight, cin_pad)
    if sl is not None:
        return sl.packed(dtype)  # view of the optimizer's once-per-step packed buffer
    return _cache_of(conv).get(('w', dtype, cin_pad), (conv.weight,),
                               lambda: ops.pack_weight(conv.weight, dtype, cin_pad=cin_pad),
                               refresh=lambda buf: ops.pack_weight(conv.weight, dtype, out=buf))


def packed_weight_dgrad(conv, dtype):
    sl = _slot_of(conv.weight)
    if sl is not None:
        return sl.packed(dtype, dgrad=True)
    return _cache_of(conv).get(('wd', dtype), (conv.weight,), lambda: ops.pack_weight_dgrad(conv.weight, dtype),
                               refresh=lambda buf: ops.pack_weight_dgrad(conv.weight, dtype, out=buf))


def packed_weight_dgrad_s2(conv, dtype):
    """The flipped weights of a stride-2 conv split by output parity (ops.dgrad_s2_weights), or None when the conv's
    data gradient does not take that path."""
    k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
    if s != 2 or conv.kernel_size[0] != conv.kernel_size[1] or not ops.s2_decomposable(k, p):
        return None
    sl = _slot_of(conv.weight)
    if sl is not None and getattr(sl, 's2_pad', -1) == p and k == 3:
        return sl.packed(dtype, dgrad='s2')   # views of the optimizer's once-per-step packed buffer
    return _cache_of(conv).get(('wd2', dtype), (conv.weight,),
                               lambda: ops.dgrad_s2_weights(packed_weight_dgrad(conv, dtype), k, p))


def _pad8(v, n, fill=0.0, out=None):
    """f32 vector of n values padded to a multiple of 8 (the kernels read per-channel constants in vectors of 8). A
    vector that already is f32, contiguous and a multiple of 8 long is used as it is: padding it anyway cost two small
    launches per biased conv and step (the parameters change every step, so nothing here can be cached across steps)."""
    vd = v.detach()
    if n % 8 == 0 and vd.numel() == n and vd.dtype == torch.float32 and vd.is_contiguous():
        return vd
    sl = getattr(v, '_das_slot', None)
    if sl is not None and fill == 0.0 and vd.numel() == n and sl.numel == n and sl.span == (n + 7) // 8 * 8:
        return sl.padded()     # (the flat optimizer stores a 1-D parameter with its zero padding: nothing to copy)
    if out is not None and out.numel() == (n + 7) // 8 * 8 and out.data_ptr() != vd.data_ptr():
        out[:n].copy_(vd)      # (refresh of a cached padded copy: the padding keeps its fill value)
        return out
    out = torch.full(((n + 7) // 8 * 8,), fill, dtype=torch.float32, device=v.device)
    out[:n] = vd.float()
    return out


def bn_eval_affine(conv, bn):
    """BatchNorm in eval mode is y = x*scale + shift with running stats (+ optional conv bias)."""
    def make():
        scale = bn.weight.float() * torch.rsqrt(bn.running_var.float() + bn.eps)
        shift = bn.bias.float() - bn.running_mean.float() * scale
        if conv.bias is not None:
            shift = shift + conv.bias.float() * scale
        n = scale.numel()
        return _pad8(scale, n, 1.0), _pad8(shift, n)
    return _cache_of(bn).get(('affine', id(conv)), (bn.weight, bn.bias, bn.running_mean, bn.running_var, conv.bias), make)


def bias_shift(conv):
    if conv.bias is None:
        return None
    n = conv.bias.numel()
    if n % 8 == 0 or getattr(conv.bias, '_das_slot', None) is not None:
        return _pad8(conv.bias, n)      # (used as it is, or with the padding its flat storage carries: nothing to cache)
    return _cache_of(conv).get(('bias',), (conv.bias,), lambda: _pad8(conv.bias, n),
                               refresh=lambda buf: _pad8(conv.bias, n, out=buf))


def _channels(x):
    return x.C if isinstance(x, ops.Ragged) else x.shape[-1]


# ------------------------------------------------------------------ fused units
def conv_bn(x, conv, bn, relu=False, residual=None, relu_in=False, skip_through=False):
    """ConvModule(conv, BN[, ReLU]) (+ residual add before the ReLU) on an NHWC tensor.

    eval: one kernel (BN folded into the conv epilogue). train: conv with fused per-channel
    sum / sum-of-squares, then the BN apply kernel (batch statistics, running-stat update).
    skip_through: return (y, x) with x routed through the same autograd node (see ConvBNTrainSkipFn) —
    use the returned x for the skip connection.
    """
    w = packed_weight(conv, x.dtype, cin_pad=x.shape[-1])
    k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
    if not bn.training:
        scale, shift = bn_eval_affine(conv, bn)
        y = ops.conv2d(x, w, k, k, s, p, scale=scale, shift=shift, residual=residual, relu=relu, relu_in=relu_in)
        return (y, x) if skip_through else y
    assert conv.bias is None
    from . import autograd as ag
    if ag.grad_mode(x, conv.weight, bn.weight, residual):
        assert not relu_in
        if skip_through:
            assert residual is None
            return ag.ConvBNTrainSkipFn.apply(x, conv.weight, bn.weight, bn.bias, conv, bn, relu)
        return ag.ConvBNTrainFn.apply(x, conv.weight, bn.weight, bn.bias, residual, conv, bn, relu)
    if skip_through:
        return conv_bn(x, conv, bn, relu=relu, residual=residual, relu_in=relu_in), x
    cout = w.shape[0]
    stats = bn_stats_buffer(x, cout)
    raw = ops.conv2d(x, w, k, k, s, p, relu_in=relu_in, stats=stats)
    mom = bn.momentum if bn.momentum is not None else 0.1
    world, stat_count = ag._sync_world(bn), 0
    if world > 1:   # SyncBN
        stats = sync_stats(stats, cout, ag._all_reduce)
        stat_count = (raw.numel() // raw.shape[-1]) * world
    y, mean, invstd = ops.bn_train_apply(raw, stats, bn.weight, bn.bias, bn.running_mean, bn.running_var, mom, bn.eps,
                                         residual=residual, relu=relu, num_batches_tracked=bn.num_batches_tracked,
                                         stat_count=stat_count)
    bn.__dict__.pop('_das_cache', None)  # running stats were updated through raw pointers
    return y


# up_conv(upsample(x)) evaluated as upsample(up_conv(x)) in train mode: see autograd.UpConvBNTrainFn. A switch for A/B runs
# and for the test that pins the two orders against each other.
UPCONV_AT_LOW_RES = True


def upsample_conv_bn(x_lo, Ho, Wo, conv, bn, relu=False, residual=None):
    """ConvModule(1x1 conv, BN)(bilinear_upsample(x_lo, (Ho, Wo))) (+ residual) (+ ReLU): MSPN's `up_conv` branch,
    mspn_mmpose.py:385-389. Train mode with autograd: the conv runs BEFORE the upsampling (one autograd node,
    UpConvBNTrainFn); otherwise the reference's order."""
    from . import autograd as ag
    if (UPCONV_AT_LOW_RES and bn.training and conv.bias is None and conv.kernel_size[0] == 1 and conv.stride[0] == 1
            and ag.grad_mode(x_lo, conv.weight, bn.weight, residual)):
        return ag.UpConvBNTrainFn.apply(x_lo, conv.weight, bn.weight, bn.bias, residual, conv, bn, relu, Ho, Wo)
    return conv_bn(upsample_bilinear(x_lo, Ho, Wo), conv, bn, relu=relu, residual=residual)


# MSPN's cross-stage skips (out_skip1 / out_skip2 -> the next stage's add) with the BatchNorm apply deferred to the add
# (autograd.ConvStatsFn + BnReluAdd3Fn); switch for A/B runs and tests
DEFERRED_SKIPS = True


class DeferredBN:
    """A ConvModule(conv, BN, ReLU) evaluated up to the BatchNorm's statistics: the consumer (skip_add) normalises."""
    __slots__ = ('raw', 'mean', 'invstd', 'bn')

    def __init__(self, raw, mean, invstd, bn):
        self.raw, self.mean, self.invstd, self.bn = raw, mean, invstd, bn


def conv_bn_deferred(x, module, skip_through=False, partner=None, hold=False):
    """module(x) for a ConvModule with BN + ReLU whose only consumer is skip_add: in train mode with autograd a DeferredBN
    (the normalised tensor is never written), otherwise the tensor. skip_through as conv_bn. partner: the ConvModule whose
    output meets this one in skip_add — the fused add's backward sums BOTH layers' reductions over one set of ranks, so a
    pair of which only one layer is SyncBN (several ranks) is evaluated layer by layer instead."""
    from . import autograd as ag
    conv, bn = module.conv, module.norm
    same_span = partner is None or ag._sync_world(partner.norm) == ag._sync_world(bn)
    if (DEFERRED_SKIPS and same_span and module.with_activation and isinstance(bn, nn.modules.batchnorm._BatchNorm)
            and bn.training and conv.bias is None
            and ag.grad_mode(x, conv.weight, bn.weight)):
        # (hold: this layer's finalize launch waits for the partner's, computed right after: one launch for the two)
        res = ag.ConvStatsFn.apply(x, conv.weight, conv, bn, skip_through, hold)
        d = DeferredBN(res[0], res[1], res[2], bn)
        return (d, res[3]) if skip_through else d
    return conv_bn(x, conv, bn, relu=module.with_activation, skip_through=skip_through)


def skip_add(x, s1, s2):
    """x + s1 + s2 where s1, s2 are tensors or DeferredBN (both of one kind)."""
    if isinstance(s1, DeferredBN) or isinstance(s2, DeferredBN):
        from . import autograd as ag
        assert isinstance(s1, DeferredBN) and isinstance(s2, DeferredBN)
        ag.flush_held_finalize()      # (a held finalize launch whose partner never came: nothing left behind by here)
        assert ag._sync_world(s1.bn) == ag._sync_world(s2.bn), \
            'skip_add: one SyncBN and one plain BatchNorm layer (build both with conv_bn_deferred(..., partner=the other))'
        return ag.BnReluAdd3Fn.apply(x, s1.raw, s1.mean, s1.invstd, s1.bn.weight, s1.bn.bias,
                                     s2.raw, s2.mean, s2.invstd, s2.bn.weight, s2.bn.bias, s1.bn, s2.bn)
    return add3(x, s1, s2)


# the whole merge of an MSPN upsample unit as one autograd node (autograd.UpMergeTrainFn); switch for A/B runs and tests
UPMERGE_FUSED = True


def up_merge(x, up_x, in_skip, up_conv, skip_through=False):
    """relu(in_skip(x) + up_conv(upsample(up_x))) of an MSPN upsample unit (mspn_mmpose.py:381-404); in_skip / up_conv are
    ConvModules (1x1 conv + BN, no activation). Returns out, or (out, x) with skip_through (see conv_bn)."""
    from . import autograd as ag
    c1, bn1, c2, bn2 = in_skip.conv, in_skip.bn, up_conv.conv, up_conv.bn
    fused = (UPMERGE_FUSED and UPCONV_AT_LOW_RES and bn1.training and bn2.training and c1.bias is None and c2.bias is None
             and c1.kernel_size[0] == c2.kernel_size[0] == 1 and c1.stride[0] == c2.stride[0] == 1
             and c1.out_channels == c2.out_channels and ag._sync_world(bn1) == ag._sync_world(bn2)
             and ag.grad_mode(x, up_x, c1.weight, bn1.weight, c2.weight, bn2.weight))
    if fused:
        return ag.UpMergeTrainFn.apply(x, up_x, c1.weight, bn1.weight, bn1.bias, c2.weight, bn2.weight, bn2.bias, in_skip, up_conv,
                                       skip_through)
    lat = conv_bn(x, c1, bn1, skip_through=skip_through)
    if skip_through:
        lat, x = lat
    out = upsample_conv_bn(up_x, x.shape[1], x.shape[2], c2, bn2, relu=True, residual=lat)
    return (out, x) if skip_through else out


def upsample_conv_bn_stats_only(x_lo, Ho, Wo, conv, bn):
    """`conv_bn_stats_only` of upsample_conv_bn: the 1x1 conv at low resolution, then the upsampling kernel reduces the
    statistics of the tensor it does not write."""
    from . import autograd as ag
    if not (UPCONV_AT_LOW_RES and conv.bias is None and conv.kernel_size[0] == 1 and conv.stride[0] == 1):
        return conv_bn_stats_only(ops.upsample_bilinear_ac(x_lo, Ho, Wo), conv, bn)
    assert bn.training
    w = packed_weight(conv, x_lo.dtype, cin_pad=x_lo.shape[-1])
    cout = w.shape[0]
    lo = ops.conv2d(x_lo, w, 1, 1, 1, 0)
    rows = x_lo.shape[0] * Ho * Wo
    stats = bn_stats_buffer_rows(rows, cout, x_lo.device)
    ops.upsample_bilinear_ac(lo, Ho, Wo, stats=stats, stats_only=True)
    world, stat_count = ag._sync_world(bn), rows
    if world > 1:
        stats = sync_stats(stats, cout, ag._all_reduce)
        stat_count = rows * world
    mom = bn.momentum if bn.momentum is not None else 0.1
    ops.bn_train_apply(lo, stats, bn.weight, bn.bias, bn.running_mean, bn.running_var, mom, bn.eps,
                       num_batches_tracked=bn.num_batches_tracked, stat_count=stat_count, finalize_only=True)
    bn.__dict__.pop('_das_cache', None)


def conv_bn_stats_only(x, conv, bn):
    """Train-mode ConvModule(conv, BN) whose OUTPUT nobody reads: the conv runs (its epilogue reduces the batch
    statistics), the BatchNorm publishes mean / invstd and advances running_mean / running_var / num_batches_tracked exactly
    as `conv_bn` would, and the normalised tensor is never written (das_bn_train_apply with y = NULL: finalize only)."""
    assert bn.training and conv.bias is None
    from . import autograd as ag
    w = packed_weight(conv, x.dtype, cin_pad=x.shape[-1])
    k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
    cout = w.shape[0]
    stats = bn_stats_buffer(x, cout)
    raw = ops.conv2d(x, w, k, k, s, p, stats=stats)
    rows = raw.numel() // cout
    world, stat_count = ag._sync_world(bn), rows
    if world > 1:
        stats = sync_stats(stats, cout, ag._all_reduce)
        stat_count = rows * world
    mom = bn.momentum if bn.momentum is not None else 0.1
    ops.bn_train_apply(raw, stats, bn.weight, bn.bias, bn.running_mean, bn.running_var, mom, bn.eps,
                       num_batches_tracked=bn.num_batches_tracked, stat_count=stat_count, finalize_only=True)
    bn.__dict__.pop('_das_cache', None)


class _ZeroArena:
    """Zero-filled f32 scratch handed out in slices (per-channel statistics accumulators of the conv epilogues, reduction
    workspaces): one fill per ~4M floats instead of one per layer. TWO buffers used alternately: when the current one is
    used up, the OTHER one is filled and taken over — its slices were handed out a whole buffer of takes ago, so every
    launch that used them precedes the fill in stream order, and the slices of the buffer being left stay intact until the
    next switch. (Rounds 3-4 had ONE buffer refilled in place: a layer pair that takes its second slice before the first
    one's consumer is launched — the two convs of an upsample-unit merge, since round 4 — lost the first layer's sums
    whenever the refill fell between the two takes: mean 0 / variance 0 for that layer in that step, a few times per
    hundred steps. Found in round 5 by comparing the running variances of two identical runs, tools/dev/rv_probe.py.)
    All takes and their consumers must be issued on ONE stream (the training stream): the fills are ordered by it."""

    def __init__(self, cap=1 << 22):
        self.cap, self.bufs, self.cur, self.off = cap, None, 0, 0

    def take(self, n, device):
        n = (n + 63) // 64 * 64
        assert n <= self.cap, n
        if self.bufs is None or self.bufs[0].device != device:
            self.bufs = [torch.zeros(self.cap, dtype=torch.float32, device=device) for _ in range(2)]
            self.cur, self.off = 0, 0
        if self.off + n > self.cap:
            self.cur ^= 1
            self.bufs[self.cur].zero_()
            self.off = 0
        s = self.bufs[self.cur][self.off:self.off + n]
        self.off += n
        return s

    def reset(self):
        """Zero the current buffer and start over (the first node of a captured graph: das_amd/graphs.py)."""
        self.bufs[self.cur].zero_()
        self.off = 0


_STATS_ARENA = _ZeroArena()


class _KeptZeros:
    """Zeroed f32 slices that must LIVE (GroupNorm statistics: the backward reads them much later): handed out from a buffer
    that was filled once; when it is used up a NEW buffer is filled — the slices keep the old one alive. One fill per ~60
    layers instead of one per layer (each a launch of its own: 14 per training step)."""

    def __init__(self, cap=1 << 18):
        self.cap, self.buf, self.off = cap, None, 0

    def take(self, n, device):
        step = (n + 63) // 64 * 64
        if self.buf is None or self.buf.device != device or self.off + step > self.buf.numel():
            self.buf, self.off = torch.zeros(max(self.cap, step), dtype=torch.float32, device=device), 0
        s = self.buf[self.off:self.off + n]
        self.off += step
        return s


_KEPT_ZEROS = _KeptZeros()
ZEROED_GN_WS = True     # GroupNorm statistics workspaces come zeroed from the pools (False: every call fills its own); A/B switch


def kept_zeros(n, device):
    return _KEPT_ZEROS.take(n, device) if ZEROED_GN_WS else None


def zeroed_stats(n, device):
    return _STATS_ARENA.take(n, device)[:n]


_SLOT_ROWS = 16384          # rows from which the statistics of a conv epilogue are spread over slots:
_FULL_SLOT_ROWS = 65536     # _FULL_SLOTS from here up,
_FULL_SLOTS = 8             # (8 cost the apply passes 0.2 ms per step less than 16 and no conv kernel more)
_MID_SLOTS = 4              # this many in between (the 32x52 stage: the apply pass folds the slots in every workgroup,
                            # 7 us for 16 slots x 1024 channels, while 4 slots already cut the atomic chain to 16 per line)


def bn_stats_buffer(x, cout):
    """Zeroed [slots, 2*cout] accumulators for the conv epilogue's BatchNorm statistics. Thousands of workgroups
    adding into one [2*cout] array serialise on the same words, so large-M layers spread them over slots
    (DasConvDesc.stats_slots); das_bn_train_apply sums the slots."""
    xd = x.data if hasattr(x, 'sizes') else x
    return bn_stats_buffer_rows(xd.numel() // xd.shape[-1], cout, xd.device)


def bn_stats_buffer_rows(rows, cout, device):
    slots = (_FULL_SLOTS if rows >= _FULL_SLOT_ROWS else _MID_SLOTS) if rows >= _SLOT_ROWS else 1
    return zeroed_stats(slots * 2 * cout, device)


_STATS_GROUP = [None, None]     # (the statistics' process group, the default group it was created beside)


def stats_group():
    """The SyncBN statistics' own process group over all ranks (the reference ships SyncBN: configs/das/exp_panoptic.py:20,28;
    its launcher is tools/dist_train.sh:8-9). Created ONCE per default group, by the first call — every rank reaches its
    first SyncBN layer (or FlatSGD's constructor, which calls this when the model holds SyncBN layers) at the same point
    of the program, and dist.new_group is itself a collective. A group of its own = an RCCL communicator of its own with
    its own internal stream: a layer's two-vector message is not ordered behind the gradient buckets the default group is
    carrying (optim.FlatSGD._launch) — on ONE communicator collectives execute in issue order whatever streams they
    were called from, which would serialise the overlap away exactly in the configuration the reference ships."""
    import torch.distributed as dist
    default = dist.distributed_c10d._get_default_group()
    if _STATS_GROUP[1] is not default:
        _STATS_GROUP[0] = dist.new_group()
        _STATS_GROUP[1] = default
    return _STATS_GROUP[0]


def sync_stats_many(stats_list, couts, all_reduce):
    """SyncBN statistics of several layers whose conv outputs are available together (a first bottleneck's projection
    shortcut and its conv3; the upsample unit's two branches): slots folded, the [2 C] vectors concatenated, ONE message."""
    folded = [st if st.numel() == 2 * c else st.view(-1, 2 * c).sum(0) for st, c in zip(stats_list, couts)]
    both = torch.cat(folded)
    all_reduce(both)
    return list(both.split([2 * c for c in couts]))


def sync_stats(stats, cout, all_reduce):
    """SyncBN: fold the slots, then sum [sum, sum of squares] over the ranks."""
    if stats.numel() != 2 * cout:
        stats = stats.view(-1, 2 * cout).sum(0)
    all_reduce(stats)
    return stats


# A tensor with several consumers is handed THROUGH all but the last of them (conv_plain / ConvModule / dcn_v2 with
# skip_through=True return (y, x)): the consumers' input gradients then meet in data-gradient epilogues instead of autograd's
# elementwise adds (14 three-pass adds over 72 MB tensors per step in the head); switch for A/B runs and tests
CHAIN_CONSUMERS = True


def conv_plain(x, conv, relu=False, out_dtype=None, out=None, skip_through=False):
    """nn.Conv2d with bias, no norm (the 1x1 predictors). skip_through: returns (y, x) with x routed through the conv's
    autograd node (autograd.ConvFn) for its other consumers; without a graph, or with CHAIN_CONSUMERS off, x itself."""
    from . import autograd as ag
    if ag.grad_mode(_tensor(x), conv.weight, conv.bias):
        assert out is None, 'writing into a slice is an inference-only shortcut'
        g = ag._geom(x)
        if skip_through and CHAIN_CONSUMERS and _tensor(x).requires_grad:
            y, xs = ag.ConvFn.apply(_tensor(x), conv.weight, conv.bias, conv, g, relu, out_dtype, None, True)
            return ag._wrap(y, g), ag._wrap(xs, g)
        y = ag.ConvFn.apply(_tensor(x), conv.weight, conv.bias, conv, g, relu, out_dtype, None)
        return (ag._wrap(y, g), x) if skip_through else ag._wrap(y, g)
    w = packed_weight(conv, x.dtype, cin_pad=_channels(x))
    k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
    y = ops.conv2d(x, w, k, k, s, p, shift=bias_shift(conv), relu=relu, out_dtype=out_dtype, out=out)
    return (y, x) if skip_through else y


def _tensor(x):
    return x.data if isinstance(x, ops.Ragged) else x


def dcn_v2(x, dcn):
    """ModulatedDeformConv2dPack.forward: offset/mask conv (f32 out) -> deformable im2col -> GEMM."""
    C = _channels(x)
    from . import autograd as ag
    # (x feeds the offset conv AND the sampling: it is handed through the offset conv's node, so that the sampling's input
    # gradient is added in the offset conv's data-gradient epilogue)
    om, x = conv_plain(x, dcn.conv_offset, out_dtype=torch.float32, skip_through=True)  # 27 -> 32 padded channels
    # one kernel (sampling into the GEMM's LDS stage, das_dcn3x3_fused) for the shapes it takes: bf16, C % 64 == 0, Cout <= 256
    fusable = (_tensor(x).dtype == torch.bfloat16 and C % 64 == 0 and dcn.weight.shape[0] % 8 == 0
               and dcn.weight.shape[0] <= 256 and dcn.weight.shape[1] == C)
    if ag.grad_mode(_tensor(x), dcn.weight, _tensor(om)):
        g = ag._geom(x)
        if fusable and ag.DCN_FUSED:
            return ag._wrap(ag.DcnFusedFn.apply(_tensor(x), _tensor(om), dcn.weight, dcn.bias, dcn, g), g)
        col = ag.DeformIm2colFn.apply(_tensor(x), _tensor(om), g)
        return ag._wrap(ag.DcnGemmFn.apply(col, dcn.weight, dcn.bias, dcn, g), g)
    w = _cache_of(dcn).get(('w', x.dtype), (dcn.weight,),
                           lambda: ops.pack_weight(dcn.weight, x.dtype).reshape(dcn.weight.shape[0], 1, 1, 9 * C))
    shift = None
    if dcn.bias is not None:
        shift = _cache_of(dcn).get(('b',), (dcn.bias,), lambda: _pad8(dcn.bias, dcn.bias.numel()))
    if fusable:
        import ctypes
        from . import _lib
        minrows = ctypes.c_longlong()
        _lib.load().das_tuning_get(b'dcn.fused_minrows', ctypes.byref(minrows))
        rows = _tensor(x).numel() // C
        if ag.DCN_FUSED and minrows.value > 0 and rows >= minrows.value:   # (ag.DCN_FUSED: the A/B switch, off = never)
            return ops.dcn3x3_fused(x, om, w, shift)
    col = ops.deform_im2col3x3(x, om)
    return ops.conv2d(col, w, 1, 1, shift=shift)


def group_norm_relu(x, gn, relu=True):
    from . import autograd as ag
    if ag.grad_mode(_tensor(x), gn.weight):
        g = ag._geom(x)
        return ag._wrap(ag.GroupNormReLUFn.apply(_tensor(x), gn.weight, gn.bias, g, gn.num_groups, gn.eps, relu, gn), g)
    ws = zeroed_stats(ops.groupnorm_stats_size(x, gn.num_groups), _tensor(x).device) if ZEROED_GN_WS else None   # (used by this call's launches only)
    return ops.groupnorm(x, gn.weight, gn.bias, gn.num_groups, gn.eps, relu=relu, ws=ws)


def max_pool(x):
    from . import autograd as ag
    return ag.MaxPoolFn.apply(x) if ag.grad_mode(x) else ops.maxpool3x3s2(x)


def upsample_bilinear(x, Ho, Wo):
    from . import autograd as ag
    return ag.BilinearUpFn.apply(x, Ho, Wo) if ag.grad_mode(x) else ops.upsample_bilinear_ac(x, Ho, Wo)


def add_nearest(a, b):
    from . import autograd as ag
    return ag.AddNearestFn.apply(a, b) if ag.grad_mode(a, b) else ops.add_upsample_nearest(a, b)


def add3(a, b, c=None):
    from . import autograd as ag
    ta, tb, tc = _tensor(a), _tensor(b), _tensor(c) if c is not None else None
    if ag.grad_mode(ta, tb, tc):
        return ag._wrap(ag.Add3Fn.apply(ta, tb, tc), ag._geom(a))
    return ops.add3(a, b, c)


# ------------------------------------------------------------------ parameter containers
class ModulatedDeformConv2dPack(nn.Module):
    """Parameter container with mmcv's DCNv2 key names: weight, bias, conv_offset.{weight,bias}."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=True):
        super().__init__()
        assert kernel_size == 3 and stride == 1 and padding == 1, 'the DAS path only uses 3x3/s1/p1 DCNv2'
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = (3, 3), (1, 1), (1, 1)
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, 3, 3))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        self.conv_offset = nn.Conv2d(in_channels, 27, 3, 1, 1, bias=True)
        self.init_weights()

    def init_weights(self):
        stdv = 1.0 / math.sqrt(self.in_channels * 9)
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.zero_()
        self.conv_offset.weight.data.zero_()
        self.conv_offset.bias.data.zero_()


def build_norm(cfg, num_features):
    """(name, module) like mmcv build_norm_layer: BN/SyncBN -> 'bn', GN -> 'gn'."""
    cfg = dict(cfg)
    t = cfg.pop('type')
    requires_grad = cfg.pop('requires_grad', True)
    if t in ('BN', 'SyncBN', 'BN2d'):
        # SyncBN: `_das_sync` makes ConvBNTrainFn all-reduce the batch statistics (forward) and the two
        # per-channel gradient sums (backward) over the process group, as torch.nn.SyncBatchNorm does
        name, layer = 'bn', nn.BatchNorm2d(num_features, **cfg)
        layer._das_sync = (t == 'SyncBN')
    elif t == 'GN':
        name, layer = 'gn', nn.GroupNorm(num_channels=num_features, **cfg)
    else:
        raise KeyError(f'unsupported norm type {t}')
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return name, layer


class ConvModule(nn.Module):
    """conv -> norm -> ReLU container with mmcv's attribute names (`conv`, `bn`/`gn`, `activate`).

    `bias='auto'` means bias iff there is no norm. `conv_cfg=dict(type='DCNv2')` selects the
    deformable conv. forward() takes and returns NHWC tensors.
    """

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias='auto', conv_cfg=None,
                 norm_cfg=None, act_cfg=dict(type='ReLU'), inplace=True):
        super().__init__()
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        if bias == 'auto':
            bias = not self.with_norm
        self.is_dcn = conv_cfg is not None and conv_cfg.get('type') == 'DCNv2'
        if self.is_dcn:
            self.conv = ModulatedDeformConv2dPack(in_channels, out_channels, kernel_size, stride, padding, bias=bias)
        else:
            self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=bias)
        self.norm_name = None
        if self.with_norm:
            self.norm_name, norm = build_norm(norm_cfg, out_channels)
            self.add_module(self.norm_name, norm)
        if self.with_activation:
            self.activate = nn.ReLU(inplace=inplace)
        self.init_weights()

    @property
    def norm(self):
        return getattr(self, self.norm_name) if self.norm_name else None

    def init_weights(self):
        if not self.is_dcn:
            nn.init.kaiming_normal_(self.conv.weight, a=0, mode='fan_out', nonlinearity='relu')
            if self.conv.bias is not None:
                nn.init.constant_(self.conv.bias, 0)
        if self.with_norm:
            nn.init.constant_(self.norm.weight, 1)
            nn.init.constant_(self.norm.bias, 0)

    def forward(self, x, residual=None, relu_in=False, skip_through=False):
        """skip_through (plain conv + GroupNorm modules, the head's): returns (y, x) with x routed through the conv's
        autograd node for the tensor's other consumers (conv_plain)."""
        relu = self.with_activation
        if self.norm_name == 'bn':
            assert not skip_through
            return conv_bn(x, self.conv, self.norm, relu=relu, residual=residual, relu_in=relu_in)
        assert residual is None and not relu_in
        if skip_through and not self.is_dcn:
            y, x = conv_plain(x, self.conv, relu=relu and self.norm_name != 'gn', skip_through=True)
            return (group_norm_relu(y, self.norm, relu=relu) if self.norm_name == 'gn' else y), x
        if self.norm_name == 'gn':
            y = dcn_v2(x, self.conv) if self.is_dcn else conv_plain(x, self.conv)
            y = group_norm_relu(y, self.norm, relu=relu)
        else:
            y = dcn_v2(x, self.conv) if self.is_dcn else conv_plain(x, self.conv, relu=relu)
        return (y, x) if skip_through else y


class Scale(nn.Module):
    def __init__(self, scale=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor(scale, dtype=torch.float))
