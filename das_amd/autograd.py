"""torch.autograd.Function wrappers: autograd sequences the backward pass (plumbing), every forward
and backward body is a HIP kernel from libdas_hip.so. Activations are NHWC tensors or the 2-D `data`
of an `ops.Ragged`; geometry travels as plain python arguments."""
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
WGRAD_SIDE_STREAM = True   # (bench.py --no-wgrad-stream and the tests flip it for A/B runs)
# Several weight gradients in flight at once, each on its own stream with a grid of 1 / WGRAD_STREAMS of the chip: the
# pixel reduction of one op is then split WGRAD_STREAMS times less, and the [split][tile] f32 workspace every split
# costs (a 256 x 256 tile = 256 KiB written and read back per workgroup: half of an average op's time when 256
# workgroups share one small Cout x K result) shrinks with it. set_wgrad_streams() sizes the kernels' grids to match.
WGRAD_STREAMS = 1
_side = {}           # device index -> [streams, next stream, join callback queued for the running backward?]


def set_wgrad_streams(n):
    """n side streams, weight-gradient grids of 1/n of the chip (das_tuning_set wgrad.blocks / wgrad.pp_blocks)."""
    global WGRAD_STREAMS
    from . import _lib
    lib = _lib.load()
    WGRAD_STREAMS = max(1, int(n))
    _lib.check(lib.das_tuning_set(b'wgrad.pp_blocks', 0 if WGRAD_STREAMS == 1 else 256 // WGRAD_STREAMS), 'das_tuning_set')
    _lib.check(lib.das_tuning_set(b'wgrad.blocks', 0 if WGRAD_STREAMS == 1 else 768 // WGRAD_STREAMS), 'das_tuning_set')
    _side.clear()


def _side_streams(dev):
    ent = _side.get(dev.index)
    if ent is None or len(ent[0]) != WGRAD_STREAMS:
        ent = _side[dev.index] = [[torch.cuda.Stream(device=dev) for _ in range(WGRAD_STREAMS)], 0, False]
    return ent


def _join_side(dev_index):
    ent = _side[dev_index]
    ent[2] = False
    cur = torch.cuda.current_stream(dev_index)
    for st in ent[0]:
        cur.wait_stream(st)


def wgrad_streams():
    """The side streams of the current device (the data-parallel all-reduce waits on them too)."""
    ent = _side.get(torch.cuda.current_device())
    return ent[0] if ent is not None else []


# Grid of conv_wgrad_pp_kernel (one 128 KiB-LDS workgroup per CU) while it runs on the side stream BESIDE the backward's main-stream
# kernels: HALF the usable CUs (device CUs minus comm.reserved_cus, / SIDE_PP_SHARE) instead of one workgroup per CU. A full-chip
# persistent grid and the main stream's kernels time-slice every CU; on half the chip each the two run side by side: -0.8 / -1.0 ms
# per step on two boxes (tools/dev/tune_step.py wgrad.pp_blocks = 64 ... 224 of 256, interleaved: 96 -0.3, 112 -0.5, 128 -0.8,
# 144 +0.1, 192 -0.3). Launched on the main stream (WGRAD_SIDE_STREAM off, the per-launch pricing passes of bench.py) the kernel
# keeps the whole chip. 1 = off. An explicit wgrad.pp_blocks (set_wgrad_streams, das_tuning_set) wins over the share.
SIDE_PP_SHARE = 2


class _on_side:
    """Context: run the enclosed launches on the next side stream, after everything enqueued on the current stream so
    far. The tensors named in `keep` are read there: the caching allocator must not hand their memory to a later
    main-stream allocation before the side stream is done with them."""

    def __init__(self, *keep):
        self.keep = [t for t in keep if t is not None]
        self.ctx = None

    def __enter__(self):
        if not WGRAD_SIDE_STREAM or not self.keep or not self.keep[0].is_cuda:
            return self
        dev = self.keep[0].device
        ent = _side_streams(dev)
        side = ent[0][ent[1]]
        ent[1] = (ent[1] + 1) % len(ent[0])
        side.wait_stream(torch.cuda.current_stream(dev))
        for t in self.keep:
            t.record_stream(side)
        if not ent[2]:
            try:
                torch.autograd.Variable._execution_engine.queue_callback(lambda i=dev.index: _join_side(i))
                ent[2] = True
            except RuntimeError:      # not inside a backward pass: join right after the launch instead
                ent[2] = None
        self.ctx = torch.cuda.stream(side)
        self.ctx.__enter__()
        self.ent, self.dev = ent, dev
        # the ping-pong weight-gradient kernel on HALF the usable CUs while it runs beside the main stream (SIDE_PP_SHARE):
        # a per-thread setting of the library, not a process-global tuning key — other threads' launches never see it
        self.grid = None
        if SIDE_PP_SHARE > 1 and len(ent[0]) == 1:
            from . import _lib
            lib = _lib.load()
            if lib.das_wgrad_pp_share(SIDE_PP_SHARE) == 0:
                self.grid = lib
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            if self.grid is not None:
                self.grid.das_wgrad_pp_share(1)
            self.ctx.__exit__(*exc)
            if self.ent[2] is None:
                _join_side(self.dev.index)
        return False


# ---- deferred, batched weight gradients ----------------------------------------------------------------
# With the flat optimizer a weight gradient only adds into the flat gradient buffer, so backward queues it and hands
# WGRAD_BATCH of them at a time to das_conv2d_wgrad_batch: ops of one kernel class then share ONE launch, each on a share
# of the grid (less split workspace per op — see include/das_hip.h). The queue is flushed when the backward pass ends.
WGRAD_BATCH = 32     # 1 = launch every weight gradient on its own, as it arises (at most 64 per launch)
_pending = []        # [(x, dy, k, k, stride, pad, out, slot)]
_flush_queued = [False]


def flush_wgrads():
    if not _pending:
        return
    items, _pending[:] = list(_pending), []
    keep = []
    for it in items:
        keep += [_d(it[0]), _d(it[1])]
    with _on_side(*keep):
        ops.conv2d_wgrad_batch([it[:7] for it in items])
    for it in items:
        it[7].fired()


def _end_of_backward():
    _flush_queued[0] = False
    flush_wgrads()


def finish_backward():
    """Launch whatever weight gradients are still queued and make the current stream wait for the side streams.
    The autograd engine's end-of-backward callbacks normally did both already; the engine skips them when backward
    raises, so the consumers of the gradients (FlatSGD.all_reduce_grads / step) call this unconditionally."""
    flush_wgrads()
    for dev_index, ent in _side.items():
        ent[2] = False
        cur = torch.cuda.current_stream(dev_index)
        for st in ent[0]:
            cur.wait_stream(st)


def reset_step_state():
    """Start of an optimisation step (FlatSGD.zero_grad): forget everything a previous backward left behind. After a
    backward that raised (out of memory, an assert inside a Function) the queue still holds that step's weight-gradient
    operands and the 'callback queued' flags are still set; launched now, those stale operands would add into the next
    step's gradient, and with the flags set no later backward would flush or join again."""
    _pending.clear()
    _held_finalize.clear()
    _flush_queued[0] = False
    _rows_checked[0] = False
    for dev_index, ent in _side.items():
        if ent[2]:
            cur = torch.cuda.current_stream(dev_index)
            for st in ent[0]:
                cur.wait_stream(st)
        ent[2] = False


def _wgrad(x, dy, weight, k, s, p):
    """Weight gradient of conv(x, weight). With the flat optimizer the kernel adds straight into the flat
    gradient buffer (same (Cout,KH,KW,Cin) layout) and autograd gets None; otherwise an OIHW view is returned."""
    sl = getattr(weight, '_das_slot', None)
    cin, cout = _d(x).shape[-1], _d(dy).shape[-1]
    if sl is not None and sl.direct(cin, cout) and sl.cl_shape[1] == k:
        if WGRAD_BATCH <= 1:
            with _on_side(_d(x), _d(dy)):
                ops.conv2d_wgrad(x, dy, k, k, s, p, out=sl.grad_cl, accumulate=True)
            sl.fired()
            return None
        if any(it[6].data_ptr() == sl.grad_cl.data_ptr() for it in _pending):
            flush_wgrads()      # (a weight used twice: its two gradients must not meet in one launch)
        _pending.append((x, dy, k, k, s, p, sl.grad_cl, sl))
        if not _flush_queued[0]:
            try:
                torch.autograd.Variable._execution_engine.queue_callback(_end_of_backward)
                _flush_queued[0] = True
            except RuntimeError:
                flush_wgrads()          # not inside a backward pass
                return None
        if len(_pending) >= WGRAD_BATCH:
            flush_wgrads()
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
    """Sum a SyncBN statistics vector over the ranks — on the statistics' OWN process group (nn.stats_group): one process
    group is one RCCL communicator, and a communicator runs its collectives in issue order whatever stream they were
    called from, so on the default group a layer's two-vector message would queue behind whichever 64 MiB gradient
    bucket is in flight (optim.FlatSGD._launch) and the main stream would wait for it."""
    import torch.distributed as dist
    from .nn import stats_group
    dist.all_reduce(t, group=stats_group())


# SyncBN normalises with `rows x world` as the statistics' count, i.e. it assumes every rank holds the same number of
# pixel rows (torch's SyncBatchNorm all-gathers the counts instead). The DAS pipelines pad every batch to one size, so
# that holds; it is CHECKED once per optimisation step — at the step's first SyncBN layer, one 2-element MAX all-reduce
# on the statistics' group — rather than assumed: ranks with different padded shapes (mixed-aspect data, a short last
# batch) raise instead of silently normalising with the wrong count. The host never waits for the answer inside the
# forward pass: the two numbers travel to page-locked memory asynchronously and are looked at when they have arrived
# (at a later step's first SyncBN layer, at most ROWS_CHECK_LAG steps late, or in verify_rows()).
_rows_checked = [False]
_rows_pending = []          # [(host tensor [max rows, max -rows], event or None, this rank's rows)]
ROWS_CHECK_LAG = 3          # = optim.MAX_RUN_AHEAD: never the reason the launch thread stops running ahead


def verify_rows(wait=False):
    """Look at the row-count checks whose answers have arrived (all of them with wait=True); raises on a mismatch."""
    while _rows_pending:
        host, ev, rows = _rows_pending[0]
        if ev is not None and not (wait or len(_rows_pending) > ROWS_CHECK_LAG) and not ev.query():
            return
        if ev is not None:
            ev.synchronize()
        _rows_pending.pop(0)
        hi, lo = host.tolist()
        if hi != -lo:
            _rows_pending.clear()
            raise RuntimeError(f'SyncBN: ranks hold different numbers of pixel rows (between {int(-lo)} and {int(hi)}; this '
                               f'rank {rows}): pad the batches of all ranks to one size (Pad(size=...)) or use norm_cfg '
                               f'type BN')


_ROWS_SIGN = {}
_ROWS_HOST, _ROWS_NEXT = [], [0]


def _check_equal_rows(rows):
    if _rows_checked[0]:
        return
    _rows_checked[0] = True
    import torch.distributed as dist
    from .nn import stats_group
    verify_rows()
    if dist.get_backend() == 'nccl':
        # (built on the device from a cached [1, -1]: a host tensor copied up here is a pageable host-to-device copy on the
        # training stream — the host would wait for the stream to reach it, every step)
        dev = torch.device('cuda', torch.cuda.current_device())
        sign = _ROWS_SIGN.get(dev)
        if sign is None:
            sign = _ROWS_SIGN[dev] = torch.tensor([1.0, -1.0], dtype=torch.float64).to(dev)
        t = sign * float(rows)
    else:
        t = torch.tensor([float(rows), -float(rows)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=stats_group())
    if t.is_cuda:
        # (page-locked landing buffers reused round-robin — at most ROWS_CHECK_LAG + 1 answers are pending: allocating one per
        # step waits for the device to drain whenever the allocator cannot reuse the previous step's, see detectors.LazyLogVars)
        if len(_ROWS_HOST) < 8:
            _ROWS_HOST.append(torch.empty(2, dtype=torch.float64, pin_memory=True))
            host = _ROWS_HOST[-1]
        else:
            host = _ROWS_HOST[_ROWS_NEXT[0] % 8]
            _ROWS_NEXT[0] += 1
        host.copy_(t, non_blocking=True)
        ev = torch.cuda.Event(blocking=True)
        ev.record()
        _rows_pending.append((host, ev, rows))
    else:            # gloo: the answer is on the host already
        _rows_pending.append((t, None, rows))
        verify_rows()


# A gradient that only has to be masked by recorded bits before it is added as a residual is masked inside the adding
# data-gradient epilogue instead of being written masked first; switch for A/B runs and tests
RES_BITS = True

# The ReLU mask of a BatchNorm + residual + ReLU layer recorded as bits by the forward apply pass (1/16 of the output) and
# read by the backward instead of the output itself (ops.relu_bits_buffer); switch for A/B runs and tests
MASK_BITS = True


def _convbn_train_forward(x, conv, bn, gamma, beta, relu, residual, bits=None):
    """conv (BatchNorm statistics in its epilogue) -> finalize + apply (+ residual, + ReLU). Returns
    y, raw, mean, invstd, world (number of ranks the statistics span). bits: a list that receives the ReLU mask of y as
    bits (or None) when the layer has a residual and a ReLU."""
    from .nn import bn_stats_buffer, packed_weight, sync_stats
    k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
    w = packed_weight(conv, x.dtype, cin_pad=x.shape[-1])
    stats = bn_stats_buffer(x, w.shape[0])
    raw = ops.conv2d(x, w, k, k, s, p, stats=stats)
    mom = bn.momentum if bn.momentum is not None else 0.1
    world = _sync_world(bn)
    stat_count = 0
    if world > 1:   # SyncBN: the statistics are those of all ranks' pixels
        _check_equal_rows(raw.numel() // raw.shape[-1])
        stats = sync_stats(stats, w.shape[0], _all_reduce)
        stat_count = (raw.numel() // raw.shape[-1]) * world
    mask = ops.relu_bits_buffer(raw) if (bits is not None and MASK_BITS and relu and residual is not None) else None
    y, mean, invstd = ops.bn_train_apply(raw, stats, gamma, beta, bn.running_mean, bn.running_var, mom, bn.eps,
                                         residual=residual, relu=relu,
                                         num_batches_tracked=bn.num_batches_tracked, stat_count=stat_count, bits_out=mask)
    if bits is not None:
        bits.append(mask)
    bn.__dict__.pop('_das_cache', None)  # running stats changed under the cache's feet (raw-pointer update)
    return y, raw, mean, invstd, world


_held_finalize = []     # finalize jobs waiting for the next _conv_stats_forward call: (bn, stats, count, C, (2, C) output)


def flush_held_finalize():
    if _held_finalize:
        jobs, _held_finalize[:] = list(_held_finalize), []
        _finalize_many([j[:4] for j in jobs], outs=[j[4] for j in jobs])


def _conv_stats_forward(x, conv, bn, gamma, beta, hold=False):
    """conv (BatchNorm statistics in its epilogue) -> finalize only (mean / invstd published, running buffers advanced): the
    normalisation is left to a consumer. SyncBN: the statistics are summed over the ranks first. Returns raw, mean, invstd.
    hold: the finalize launch is left to the NEXT call of this function, which publishes both layers' statistics in one
    launch (the two cross-stage skip convs of an upsample unit, computed back to back and consumed together much later);
    the mean / invstd tensors returned here are filled by then."""
    from .nn import bn_stats_buffer, packed_weight, sync_stats
    k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
    w = packed_weight(conv, x.dtype, cin_pad=x.shape[-1])
    stats = bn_stats_buffer(x, w.shape[0])
    raw = ops.conv2d(x, w, k, k, s, p, stats=stats)
    world, stat_count = _sync_world(bn), 0
    if world > 1:
        rows = raw.numel() // raw.shape[-1]
        _check_equal_rows(rows)
        stats = sync_stats(stats, w.shape[0], _all_reduce)
        stat_count = rows * world
    if FINALIZE_MANY and (hold or _held_finalize):
        Cc = raw.shape[-1]
        mi = torch.empty(2, Cc, dtype=torch.float32, device=raw.device)
        _held_finalize.append((bn, stats, stat_count or raw.numel() // Cc, Cc, mi))
        if not hold or len(_held_finalize) == 4:
            flush_held_finalize()
        return raw, mi[0], mi[1]
    _, mean, invstd = ops.bn_train_apply(raw, stats, gamma, beta, bn.running_mean, bn.running_var,
                                         bn.momentum if bn.momentum is not None else 0.1, bn.eps,
                                         num_batches_tracked=bn.num_batches_tracked, stat_count=stat_count, finalize_only=True)
    bn.__dict__.pop('_das_cache', None)
    return raw, mean, invstd


def _conv_stats_forward_pair(a, b):
    """_conv_stats_forward of two (x, conv, bn, gamma, beta) layers whose outputs are consumed together (a first
    bottleneck's projection shortcut and its conv3, both read by ONE dual apply pass). When both are SyncBN over the same
    ranks their statistics travel in ONE message (nn.sync_stats_many) instead of two."""
    from .nn import bn_stats_buffer, packed_weight, sync_stats_many
    wa, wb = _sync_world(a[2]), _sync_world(b[2])
    if wa != wb:
        return _conv_stats_forward(*a), _conv_stats_forward(*b)
    raws, stats, couts = [], [], []
    for x, conv, bn, gamma, beta in (a, b):
        k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        w = packed_weight(conv, x.dtype, cin_pad=x.shape[-1])
        st = bn_stats_buffer(x, w.shape[0])
        raws.append(ops.conv2d(x, w, k, k, s, p, stats=st))
        stats.append(st)
        couts.append(w.shape[0])
    if wa > 1:
        _check_equal_rows(raws[0].numel() // raws[0].shape[-1])
        stats = sync_stats_many(stats, couts, _all_reduce)
    fin = _finalize_many([(bn, st, raw.numel() // raw.shape[-1] * wa, raw.shape[-1])
                          for (x, conv, bn, gamma, beta), raw, st in zip((a, b), raws, stats)])
    return (raws[0],) + fin[0], (raws[1],) + fin[1]


FINALIZE_MANY = True    # one finalize launch for the layers of a fused consumer (False: one per layer); A/B switch


def _finalize_many(items, outs=None):
    """mean / invstd published and running statistics advanced for [(bn, stats, count, C)] in ONE launch (layers whose
    statistics are complete together and whose normalisation one fused consumer does). Returns [(mean, invstd)]."""
    if not FINALIZE_MANY:
        fin = []
        for i, it in enumerate(items):
            fin += _finalize_many_impl([it], None if outs is None else [outs[i]])
        return fin
    return _finalize_many_impl(items, outs)


def _finalize_many_impl(items, outs=None):
    fin = ops.bn_finalize_many([(st, Cc, count, bn.running_mean, bn.running_var,
                                 bn.momentum if bn.momentum is not None else 0.1, bn.eps, bn.num_batches_tracked)
                                for bn, st, count, Cc in items], outs=outs)
    for bn, _, _, _ in items:
        bn.__dict__.pop('_das_cache', None)   # running stats changed under the cache's feet (raw-pointer update)
    return fin


# a bottleneck's projection shortcut normalised inside bn3's apply pass (ops.bn_dual_apply); switch for A/B runs and tests
DUAL_APPLY = True


class ConvBNTrainFn(Function):
    """conv (no bias) -> train-mode BatchNorm (+ residual) (+ ReLU). mspn_mmpose.py:126-157,381-404."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, residual, conv, bn, relu):
        k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        bits = []
        y, raw, mean, invstd, world = _convbn_train_forward(x, conv, bn, gamma, beta, relu, residual, bits=bits)
        ctx.bits = bits[0]   # (the mask of y as bits: the backward then never reads y)
        ctx.save_for_backward(x, raw, y if (residual is not None and ctx.bits is None) else None, mean, invstd, gamma, weight, beta)
        ctx.cfg = (k, s, p, relu, residual is not None, conv, bn)
        ctx.world = world
        return y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        from .nn import packed_weight_dgrad, packed_weight_dgrad_s2
        x, raw, y, mean, invstd, gamma, weight, beta = ctx.saved_tensors
        k, s, p, relu, has_res, conv, bn = ctx.cfg
        dy = dy.contiguous()
        ga, ba = _param_acc(bn.weight), _param_acc(bn.bias)
        direct = ga is not None and ba is not None
        # without a residual the ReLU mask is recomputed from raw: y is not read at all
        if ctx.world > 1:   # SyncBN: reduce, sum over ranks, apply; parameter gradients stay local
            direct = False
            draw, dres, dgamma, dbeta = ops.bn_train_backward_sync(dy, y if (relu and has_res and ctx.bits is None) else None, raw,
                                                                   mean, invstd, gamma, relu, has_res, beta, _all_reduce, ctx.world,
                                                                   bits=ctx.bits if (relu and has_res) else None)
        else:
            draw, dres, dgamma, dbeta = ops.bn_train_backward(dy, y if (relu and has_res and ctx.bits is None) else None, raw,
                                                              mean, invstd, gamma, relu, has_res, beta=beta,
                                                              dgamma_acc=ga[1] if direct else None,
                                                              dbeta_acc=ba[1] if direct else None,
                                                              bits=ctx.bits if (relu and has_res) else None)
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
                                  residual=dskip, w_classes=packed_weight_dgrad_s2(conv, x.dtype))
            if dx.shape[-1] != x.shape[-1]:
                dx = dx[..., :x.shape[-1]]
        return dx, dw, dgamma, dbeta, dres, None, None, None


class UpConvBNTrainFn(Function):
    """bilinear upsample (align_corners) -> bias-free 1x1 conv -> train-mode BatchNorm (+ residual) (+ ReLU): the `up_conv`
    branch of MSPN's upsample units, mspn_mmpose.py:385-389, with the first two steps exchanged. Both are linear maps, one
    over pixels and one over channels, so conv(upsample(x)) == upsample(conv(x)) exactly in real arithmetic (in floating
    point up to the order of summation, ~1e-6 relative in f32): the conv, its data gradient and its weight gradient then
    run on the quarter-size tensor, and the upsampling kernel — which now writes the pre-norm tensor — reduces the
    BatchNorm statistics the conv epilogue used to reduce (das_upsample_bilinear_ac_stats)."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, residual, conv, bn, relu, Ho, Wo):
        from .nn import bn_stats_buffer_rows, packed_weight, sync_stats
        w = packed_weight(conv, x.dtype, cin_pad=x.shape[-1])
        cout = w.shape[0]
        lo = ops.conv2d(x, w, 1, 1, 1, 0)
        rows = x.shape[0] * Ho * Wo
        stats = bn_stats_buffer_rows(rows, cout, x.device)
        raw = ops.upsample_bilinear_ac(lo, Ho, Wo, stats=stats)
        mom = bn.momentum if bn.momentum is not None else 0.1
        world, stat_count = _sync_world(bn), 0
        if world > 1:
            _check_equal_rows(rows)
            stats = sync_stats(stats, cout, _all_reduce)
            stat_count = rows * world
        y, mean, invstd = ops.bn_train_apply(raw, stats, gamma, beta, bn.running_mean, bn.running_var, mom, bn.eps,
                                             residual=residual, relu=relu,
                                             num_batches_tracked=bn.num_batches_tracked, stat_count=stat_count)
        bn.__dict__.pop('_das_cache', None)
        ctx.save_for_backward(x, raw, y if residual is not None else None, mean, invstd, gamma, weight, beta)
        ctx.cfg = (relu, residual is not None, conv, bn)
        ctx.world = world
        return y

    @staticmethod
    def backward(ctx, dy):
        from .nn import packed_weight_dgrad
        x, raw, y, mean, invstd, gamma, weight, beta = ctx.saved_tensors
        relu, has_res, conv, bn = ctx.cfg
        dy = dy.contiguous()
        ga, ba = _param_acc(bn.weight), _param_acc(bn.bias)
        direct = ga is not None and ba is not None
        if ctx.world > 1:
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
        dlo = ops.upsample_bilinear_ac_backward(draw, x.shape[1], x.shape[2])
        dw = _wgrad(x, dlo, conv.weight, 1, 1, 0) if ctx.needs_input_grad[1] else None
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.conv2d_dgrad(dlo, packed_weight_dgrad(conv, x.dtype), 1, 1, 1, 0, (x.shape[1], x.shape[2]))
            if dx.shape[-1] != x.shape[-1]:
                dx = dx[..., :x.shape[-1]]
        return dx, dw, dgamma, dbeta, dres, None, None, None, None, None


class ConvStatsFn(Function):
    """conv (no bias) with the train-mode BatchNorm's statistics finalised (running buffers advanced), the normalisation
    itself left to the consumer (BnReluAdd3Fn): returns raw, mean, invstd (+ x handed through, see ConvBNTrainSkipFn).
    The gradient that arrives for `raw` is the BatchNorm's complete backward."""

    @staticmethod
    def forward(ctx, x, weight, conv, bn, skip_through, hold=False):
        k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        raw, mean, invstd = _conv_stats_forward(x, conv, bn, bn.weight, bn.bias, hold=hold)   # (SyncBN: statistics of all ranks)
        ctx.save_for_backward(x)
        ctx.cfg = (k, s, p, conv)
        ctx.mark_non_differentiable(mean, invstd)
        ctx.set_materialize_grads(False)   # (no zero tensors for the gradients of mean / invstd: 4 launches per call)
        return (raw, mean, invstd, x) if skip_through else (raw, mean, invstd)

    @staticmethod
    def backward(ctx, draw, _dmean, _dinvstd, dskip=None):
        from .nn import packed_weight_dgrad, packed_weight_dgrad_s2
        (x,) = ctx.saved_tensors
        k, s, p, conv = ctx.cfg
        if draw is None:        # only the handed-through x took part in this backward (autograd.grad over selected outputs)
            return dskip, None, None, None, None, None
        draw = draw.contiguous()
        dw = _wgrad(x, draw, conv.weight, k, s, p) if ctx.needs_input_grad[1] else None
        dx = None
        if ctx.needs_input_grad[0]:
            if dskip is not None:
                dskip = dskip.contiguous()
            dx = ops.conv2d_dgrad(draw, packed_weight_dgrad(conv, x.dtype), k, k, s, p, (x.shape[1], x.shape[2]),
                                  residual=dskip, w_classes=packed_weight_dgrad_s2(conv, x.dtype))
            if dx.shape[-1] != x.shape[-1]:
                dx = dx[..., :x.shape[-1]]
        elif dskip is not None:
            dx = dskip
        return dx, dw, None, None, None, None


class BnReluAdd3Fn(Function):
    """x + relu(BN1(raw1)) + relu(BN2(raw2)) over ConvStatsFn's outputs: MSPN's cross-stage merge (mspn_mmpose.py:254-275)
    as one pass forward and two backward (csrc/skipadd.hip)."""

    @staticmethod
    def forward(ctx, x, raw1, mean1, invstd1, g1, b1, raw2, mean2, invstd2, g2, b2, bn1, bn2):
        out = ops.bn_relu_add3_forward(x, raw1, (mean1, invstd1, g1, b1), raw2, (mean2, invstd2, g2, b2))
        ctx.save_for_backward(raw1, mean1, invstd1, g1, b1, raw2, mean2, invstd2, g2, b2)
        ctx.mods = (bn1, bn2)
        return out

    @staticmethod
    def backward(ctx, g):
        raw1, mean1, invstd1, g1, b1, raw2, mean2, invstd2, g2, b2 = ctx.saved_tensors
        bn1, bn2 = ctx.mods
        g = g.contiguous()
        acc = [_param_acc(p) for p in (bn1.weight, bn1.bias, bn2.weight, bn2.bias)]
        direct = all(a is not None for a in acc)
        world = max(_sync_world(bn1), _sync_world(bn2))
        Cc = raw1.shape[-1]
        if world > 1:
            # SyncBN: reduce, sum over the ranks, apply; the parameter gradients are this rank's sums (the gradient
            # all-reduce adds the ranks' contributions)
            d1, d2, sums = ops.bn_relu_add3_backward(g, raw1, (mean1, invstd1, g1, b1), raw2, (mean2, invstd2, g2, b2),
                                                     all_reduce=_all_reduce, world=world)
            if direct:
                for a, i in zip(acc, (1, 0, 3, 2)):      # acc order: dgamma1, dbeta1, dgamma2, dbeta2
                    a[1].add_(sums[i * Cc:(i + 1) * Cc])
        else:
            d1, d2, sums = ops.bn_relu_add3_backward(g, raw1, (mean1, invstd1, g1, b1), raw2, (mean2, invstd2, g2, b2),
                                                     acc=tuple(a[1] for a in acc) if direct else None)
        if direct:
            for a in acc:
                a[0].fired()
            db1 = dg1 = db2 = dg2 = None
        else:
            db1, dg1, db2, dg2 = (sums[i * Cc:(i + 1) * Cc].clone() for i in range(4))
        return g, d1, None, None, dg1, db1, d2, None, None, dg2, db2, None, None


class UpMergeTrainFn(Function):
    """One MSPN upsample unit's merge in train mode (mspn_mmpose.py:381-404):
    out = relu(BN1(in_skip(x)) + BN2(up_conv(upsample(up_x)))), as one autograd node over the kernels of
    das_amd/csrc/upmerge.hip: up_conv runs before the upsampling (UpConvBNTrainFn's exchange), and neither normalised
    branch nor the upsampled tensor is written — forward or backward. skip_through: x is handed through as a second
    output (ConvBNTrainSkipFn), its other consumers' gradient arrives as `dskip` and is added in in_skip's data-gradient
    epilogue."""

    @staticmethod
    def forward(ctx, x, up_x, w1, g1, b1, w2, g2, b2, in_skip, up_conv, skip_through):
        from .nn import bn_stats_buffer, bn_stats_buffer_rows, packed_weight
        c1, bn1, c2, bn2 = in_skip.conv, in_skip.bn, up_conv.conv, up_conv.bn
        B, Ho, Wo = x.shape[0], x.shape[1], x.shape[2]
        rows = B * Ho * Wo
        wp1 = packed_weight(c1, x.dtype, cin_pad=x.shape[-1])
        cout = wp1.shape[0]
        stats1 = bn_stats_buffer(x, cout)
        raw1 = ops.conv2d(x, wp1, 1, 1, 1, 0, stats=stats1)
        z = ops.conv2d(up_x, packed_weight(c2, up_x.dtype, cin_pad=up_x.shape[-1]), 1, 1, 1, 0)
        stats2 = bn_stats_buffer_rows(z.numel() // cout, cout, x.device)
        ops.upsample_stats_lowres(z, Ho, Wo, stats2)
        world = max(_sync_world(bn1), _sync_world(bn2))
        if world > 1:   # SyncBN: both layers' statistics over all ranks' pixels, ONE collective for the two
            _check_equal_rows(rows)
            both = torch.cat([stats1.view(-1, 2 * cout).sum(0), stats2.view(-1, 2 * cout).sum(0)])
            _all_reduce(both)
            stats1, stats2 = both[:2 * cout], both[2 * cout:]
        (mean1, invstd1), (mean2, invstd2) = _finalize_many([(bn1, stats1, rows * world, cout), (bn2, stats2, rows * world, cout)])
        mask = ops.relu_bits_buffer(raw1) if MASK_BITS else None
        out = ops.upmerge_forward(raw1, z, (mean1, invstd1, g1, b1), (mean2, invstd2, g2, b2), bits_out=mask)
        ctx.save_for_backward(x, up_x, raw1, z, out if mask is None else None, mean1, invstd1, mean2, invstd2, g1, g2, mask)
        ctx.mods = (c1, bn1, c2, bn2)
        ctx.skip_through, ctx.world = skip_through, world
        if skip_through:
            ctx.set_materialize_grads(False)   # (see BottleneckChainFn: no zero tensor for a handed-through input nobody took)
        return (out, x) if skip_through else out

    @staticmethod
    def backward(ctx, dy, dskip=None):
        from .nn import packed_weight_dgrad
        if dy is None:
            return (dskip,) + (None,) * 10
        x, up_x, raw1, z, out, mean1, invstd1, mean2, invstd2, g1, g2, mask = ctx.saved_tensors
        c1, bn1, c2, bn2 = ctx.mods
        Cc = raw1.shape[-1]
        rows = raw1.numel() // Cc
        dzm, sums = ops.upmerge_backward_reduce(dy.contiguous(), out, raw1, z, mean1, invstd1, mean2, invstd2, bits=mask)
        acc = [_param_acc(p) for p in (bn1.weight, bn1.bias, bn2.weight, bn2.bias)]
        direct = all(a is not None for a in acc)
        world, local = ctx.world, sums
        if world > 1:
            # SyncBN: the apply passes need the sums over all ranks' rows; the parameter gradients stay this rank's sums
            sums = local.clone()
            _all_reduce(sums)
            if direct:
                acc[0][1].add_(local[Cc:2 * Cc]); acc[1][1].add_(local[:Cc])
                acc[2][1].add_(local[2 * Cc:]); acc[3][1].add_(local[:Cc])
        kernel_acc = direct and world == 1
        # BatchNorm 1: the apply pass every other layer uses (dZ, raw1 -> d raw1); sums[:2C] is its [sum dZ | sum dZ xhat]
        draw1 = ops.bn_backward_apply(dzm, raw1, mean1, invstd1, g1, sums[:2 * Cc],
                                      dgamma_acc=acc[0][1] if kernel_acc else None, dbeta_acc=acc[1][1] if kernel_acc else None,
                                      stat_rows=rows * world)
        dw1 = _wgrad(x, draw1, c1.weight, 1, 1, 0) if ctx.needs_input_grad[2] else None
        # BatchNorm 2 + upsampling: d raw2 is never formed; dz = upsample^T(d raw2) from upsample^T(dZ) and low-resolution terms
        P = ops.upsample_bilinear_ac_backward(dzm, z.shape[1], z.shape[2])
        dz = ops.upmerge_backward_lowres(P, z, raw1.shape[1], raw1.shape[2], sums, g2, mean2, invstd2, rows * world,
                                         dgamma2_acc=acc[2][1] if kernel_acc else None,
                                         dbeta2_acc=acc[3][1] if kernel_acc else None)
        dw2 = _wgrad(up_x, dz, c2.weight, 1, 1, 0) if ctx.needs_input_grad[5] else None
        if direct:
            for a in acc:
                a[0].fired()
            dg1 = db1 = dg2 = db2 = None
        else:
            db1, dg1, dg2 = local[:Cc].clone(), local[Cc:2 * Cc].clone(), local[2 * Cc:].clone()
            db2 = db1.clone()
        dx = dup = None
        if ctx.needs_input_grad[0]:
            if dskip is not None:
                dskip = dskip.contiguous()
            dx = ops.conv2d_dgrad(draw1, packed_weight_dgrad(c1, x.dtype), 1, 1, 1, 0, (x.shape[1], x.shape[2]), residual=dskip)
            if dx.shape[-1] != x.shape[-1]:
                dx = dx[..., :x.shape[-1]]
        elif dskip is not None:
            dx = dskip
        if ctx.needs_input_grad[1]:
            dup = ops.conv2d_dgrad(dz, packed_weight_dgrad(c2, up_x.dtype), 1, 1, 1, 0, (up_x.shape[1], up_x.shape[2]))
            if dup.shape[-1] != up_x.shape[-1]:
                dup = dup[..., :up_x.shape[-1]]
        return dx, dup, dw1, dg1, db1, dw2, dg2, db2, None, None, None


class ConvBNTrainSkipFn(Function):
    """ConvBNTrainFn that also hands its input through as a second output: y, x_skip = f(x).

    In a bottleneck x feeds conv1 AND the identity path. Routing the identity through this node makes both
    gradients of x arrive together, so the data-gradient kernel adds the skip gradient in its epilogue
    (`residual`) instead of autograd running a separate elementwise add over the whole tensor."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, conv, bn, relu):
        y = ConvBNTrainFn.forward(ctx, x, weight, gamma, beta, None, conv, bn, relu)
        ctx.set_materialize_grads(False)       # (see BottleneckChainFn)
        return y, x

    @staticmethod
    def backward(ctx, dy, dskip):
        if dy is None:
            return dskip, None, None, None, None, None, None
        dx, dw, dgamma, dbeta, _, _, _, _ = ConvBNTrainFn.backward(ctx, dy, dskip)
        return dx, dw, dgamma, dbeta, None, None, None


class BottleneckChainFn(Function):
    """A whole ResNet layer (`nn.Sequential` of Bottlenecks, mspn_mmpose.py:17-157,254-275) as ONE autograd node, so
    that backward can be scheduled by hand: the data-gradient conv that produces the gradient of a BatchNorm layer's
    output also masks it (ReLU) and reduces [sum dZ | sum dZ * xhat] in its epilogue (DasConvDesc.bnb_*), and the layer
    is finished by the apply pass alone — the separate reduction pass over (dY, raw) that every BatchNorm backward
    otherwise needs disappears for bn1 / bn2 of every block and for bn3 of every block but the last (whose gradient
    arrives from outside). Forward runs the same kernels as the per-unit path. Arithmetic per layer is unchanged
    (same mask, same sums up to f32 summation order)."""

    @staticmethod
    def forward(ctx, x, blocks, skip_through, *params):
        x_in = x
        saved, plan = [x], []
        it = iter(params)
        for blk in blocks:
            xin = x
            ent = {}
            w1, g1, b1, w2, g2, b2, w3, g3, b3 = (next(it) for _ in range(9))
            y1, raw1, m1, i1, _ = _convbn_train_forward(xin, blk.conv1, blk.bn1, g1, b1, True, None)
            y2, raw2, m2, i2, _ = _convbn_train_forward(y1, blk.conv2, blk.bn2, g2, b2, True, None)
            mb = []   # the ReLU mask of y3 as bits
            if blk.downsample is not None:
                wd, gd, bd = (next(it) for _ in range(3))
                ds = blk.downsample
                if DUAL_APPLY:
                    # the shortcut's normalised tensor is read by bn3's apply pass only: never written
                    (rawd, md, idd), (raw3, m3, i3) = _conv_stats_forward_pair((xin, ds.conv, ds.bn, gd, bd),
                                                                               (y2, blk.conv3, blk.bn3, g3, b3))
                    mb.append(ops.relu_bits_buffer(raw3) if MASK_BITS else None)
                    y3 = ops.bn_dual_apply(raw3, (m3, i3, g3, b3), rawd, (md, idd, gd, bd), relu=True, bits_out=mb[0])
                else:
                    idn, rawd, md, idd, _ = _convbn_train_forward(xin, ds.conv, ds.bn, gd, bd, False, None)
                    y3, raw3, m3, i3, _ = _convbn_train_forward(y2, blk.conv3, blk.bn3, g3, b3, True, idn, bits=mb)
            else:
                y3, raw3, m3, i3, _ = _convbn_train_forward(y2, blk.conv3, blk.bn3, g3, b3, True, xin, bits=mb)
            ent['u'] = len(saved)
            saved += [raw1, m1, i1, y1, raw2, m2, i2, y2, raw3, m3, i3, y3, mb[0]]
            if blk.downsample is not None:
                ent['d'] = len(saved)
                saved += [rawd, md, idd]
            plan.append(ent)
            x = y3
        ctx.save_for_backward(*saved, *params)
        ctx.plan, ctx.blocks, ctx.nsaved = plan, blocks, len(saved)
        # skip_through: the layer's input is handed through as a second output (see ConvBNTrainSkipFn): its other
        # consumers take it from there, and their gradient arrives here as `dskip`, added in a data-gradient epilogue
        if skip_through:
            # (a handed-through input nobody took — the last stage's units generate no cross-stage skips — must not come
            # back as a materialised zero tensor: a fill of the whole map and its read as a residual)
            ctx.set_materialize_grads(False)
        return (x, x_in) if skip_through else x

    @staticmethod
    def backward(ctx, dy, dskip=None):
        from .nn import bn_stats_buffer, packed_weight_dgrad, packed_weight_dgrad_s2
        if dy is None:      # (only the handed-through input took part in the backward)
            return (dskip, None, None) + (None,) * (len(ctx.saved_tensors) - ctx.nsaved)
        if dskip is not None:
            dskip = dskip.contiguous()
        saved, params = ctx.saved_tensors[:ctx.nsaved], ctx.saved_tensors[ctx.nsaved:]
        blocks, plan = ctx.blocks, ctx.plan
        grads = [None] * len(params)
        # parameter offsets per block
        offs, o = [], 0
        for blk in blocks:
            offs.append(o)
            o += 12 if blk.downsample is not None else 9

        def deliver(bn, pi, dgamma, dbeta):
            """LOCAL parameter gradients of a SyncBN layer (the gradient all-reduce sums the ranks' contributions)"""
            ga, ba = _param_acc(bn.weight), _param_acc(bn.bias)
            if ga is not None and ba is not None:
                ga[1].add_(dgamma)
                ba[1].add_(dbeta)
                ga[0].fired()
                ba[0].fired()
            else:
                grads[pi], grads[pi + 1] = dgamma, dbeta

        def finish_bn(bn, gamma, beta, pi, dz, raw, mean, invstd, sums):
            """apply pass of a layer whose dZ and sums are ready; parameter gradients -> flat accumulators or grads[]"""
            world = _sync_world(bn)
            if world > 1:
                # SyncBN: the data-gradient epilogue reduced this rank's [sum dZ | sum dZ * xhat] (xhat from the global
                # statistics); the apply pass needs the sums over ALL ranks' rows (what torch's SyncBatchNorm does with
                # its all_reduce of sum_dy / sum_dy_xmu), the parameter gradients are the local sums
                C_ = raw.shape[-1]
                local = sums.view(-1, 2 * C_).sum(0)
                glob = local.clone()
                _all_reduce(glob)
                rows = raw.numel() // C_
                draw = ops.bn_backward_apply(dz, raw, mean, invstd, gamma, glob, stat_rows=rows * world)
                deliver(bn, pi, local[C_:], local[:C_])
                return draw
            ga, ba = _param_acc(bn.weight), _param_acc(bn.bias)
            direct = ga is not None and ba is not None
            draw = ops.bn_backward_apply(dz, raw, mean, invstd, gamma, sums,
                                         dgamma_acc=ga[1] if direct else None, dbeta_acc=ba[1] if direct else None)
            if direct:
                ga[0].fired()
                ba[0].fired()
            else:
                C_ = raw.shape[-1]
                f = sums.view(-1, 2 * C_).sum(0)
                grads[pi], grads[pi + 1] = f[C_:], f[:C_]
            return draw

        def classic_bn(bn, gamma, beta, pi, dyv, y, raw, mean, invstd, relu, want_dres, bits=None):
            world = _sync_world(bn)
            if world > 1:
                draw, dres, dgamma, dbeta = ops.bn_train_backward_sync(dyv, None if bits is not None else y, raw, mean, invstd,
                                                                       gamma, relu, want_dres, beta, _all_reduce, world, bits=bits)
                deliver(bn, pi, dgamma, dbeta)
                return draw, dres
            ga, ba = _param_acc(bn.weight), _param_acc(bn.bias)
            direct = ga is not None and ba is not None
            draw, dres, dgamma, dbeta = ops.bn_train_backward(dyv, None if bits is not None else y, raw, mean, invstd, gamma, relu,
                                                              want_dres, beta=beta, dgamma_acc=ga[1] if direct else None,
                                                              dbeta_acc=ba[1] if direct else None, bits=bits)
            if direct:
                ga[0].fired()
                ba[0].fired()
            else:
                grads[pi], grads[pi + 1] = dgamma, dbeta
            return draw, dres

        def wgrad(conv, pi, xin, draw):
            k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
            grads[pi] = _wgrad(xin, draw, conv.weight, k, s, p)

        def dgrad(conv, draw, xin, residual=None, fuse=None, accumulate=None):
            """data gradient of `conv` wrt xin; fuse = (raw, y, mean, invstd, gamma, beta[, mask bits instead of y]) of the
            BatchNorm+ReLU layer that produced xin -> returns (dZ, sums) of that layer instead of the plain gradient;
            accumulate = a tensor already holding another gradient of xin, added to in place"""
            k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
            w = packed_weight_dgrad(conv, xin.dtype)
            wc = packed_weight_dgrad_s2(conv, xin.dtype)
            if fuse is None:
                return ops.conv2d_dgrad(draw, w, k, k, s, p, (xin.shape[1], xin.shape[2]), residual=residual, w_classes=wc,
                                        accumulate=accumulate)
            sums = bn_stats_buffer(xin, xin.shape[-1])
            dz = ops.conv2d_dgrad(draw, w, k, k, s, p, (xin.shape[1], xin.shape[2]), residual=residual,
                                  bn_bwd=ops.BnBwd(*fuse[:6], True, bits=fuse[6] if len(fuse) > 6 else None), stats=sums,
                                  w_classes=wc)
            return dz, sums

        x0 = saved[0]
        carry_dy, carry_dz = dy.contiguous(), None     # gradient wrt the current block's output: raw, or (dZ, sums)
        for bi in range(len(blocks) - 1, -1, -1):
            blk, ent, po = blocks[bi], plan[bi], offs[bi]
            raw1, m1, i1, y1, raw2, m2, i2, y2, raw3, m3, i3, y3, bits3 = saved[ent['u']:ent['u'] + 13]
            xin = x0 if bi == 0 else saved[plan[bi - 1]['u'] + 11]
            w1, g1, b1, w2, g2, b2, w3, g3, b3 = params[po:po + 9]
            # ---- bn3 (+ identity, ReLU)
            if carry_dz is None:
                # with the mask as bits the identity branch's gradient dY * mask need not be written: the conv1 data gradient
                # below takes (dY, bits) as its residual and masks on the fly (DasConvDesc.residual_mask_bits)
                lazy = RES_BITS and bits3 is not None and bi > 0 and blk.downsample is None and blk.conv1.stride[0] == 1
                draw3, dz3 = classic_bn(blk.bn3, g3, b3, po + 7, carry_dy, y3, raw3, m3, i3, True, not lazy, bits=bits3)
                if lazy:
                    dz3 = (carry_dy, bits3)
            else:
                dz3, sums3 = carry_dz
                draw3 = finish_bn(blk.bn3, g3, b3, po + 7, dz3, raw3, m3, i3, sums3)
            wgrad(blk.conv3, po + 6, y2, draw3)
            dz2, sums2 = dgrad(blk.conv3, draw3, y2, fuse=(raw2, None, m2, i2, g2, b2))
            draw2 = finish_bn(blk.bn2, g2, b2, po + 4, dz2, raw2, m2, i2, sums2)
            wgrad(blk.conv2, po + 3, y1, draw2)
            dz1, sums1 = dgrad(blk.conv2, draw2, y1, fuse=(raw1, None, m1, i1, g1, b1))
            draw1 = finish_bn(blk.bn1, g1, b1, po + 1, dz1, raw1, m1, i1, sums1)
            wgrad(blk.conv1, po, xin, draw1)
            if blk.downsample is not None:
                ds = blk.downsample
                rawd, md, idd = saved[ent['d']:ent['d'] + 3]
                wd, gd, bd = params[po + 9:po + 12]
                drawd, _ = classic_bn(ds.bn, gd, bd, po + 10, dz3, None, rawd, md, idd, False, False)
                wgrad(ds.conv, po + 9, xin, drawd)
                dx1 = dgrad(blk.conv1, draw1, xin, residual=dskip if bi == 0 else None)
                carry_dy, carry_dz = dgrad(ds.conv, drawd, xin, accumulate=dx1), None
                if bi == 0:
                    dskip = None
            elif bi > 0:   # xin is the previous block's output: mask by it, reduce for its bn3
                pent, ppo = plan[bi - 1], offs[bi - 1]
                praw3, pm3, pi3 = saved[pent['u'] + 8:pent['u'] + 11]
                pbits3 = saved[pent['u'] + 12]
                pg3, pb3 = params[ppo + 7], params[ppo + 8]
                # (the previous block's ReLU mask: its recorded bits, or its output xin itself)
                fuse = (praw3, None, pm3, pi3, pg3, pb3, pbits3) if pbits3 is not None else (praw3, xin, pm3, pi3, pg3, pb3)
                carry_dy, carry_dz = None, dgrad(blk.conv1, draw1, xin, residual=dz3, fuse=fuse)
            else:
                carry_dy, carry_dz = dgrad(blk.conv1, draw1, xin, residual=dz3), None
        assert carry_dz is None
        if dskip is not None:   # (first block without a downsample branch: its epilogue's residual slot is taken)
            carry_dy = carry_dy + dskip
        return (carry_dy, None, None) + tuple(grads)


def bottleneck_chain(x, blocks, skip_through=False):
    """Run a layer of Bottlenecks through BottleneckChainFn if that applies (training, gradients on); else None.
    SyncBN layers (any mix with plain ones: the reference's `_make_layer` makes only a layer's first block SyncBN) run
    the same fused backward: the sums a data-gradient epilogue reduced are all-reduced before their apply pass.
    skip_through: returns (y, x) with x routed through the node for its other consumers."""
    blocks = list(blocks)
    bns = []
    for b in blocks:
        bns += [b.bn1, b.bn2, b.bn3] + ([b.downsample.bn] if b.downsample is not None else [])
        if b.downsample is not None and (not getattr(b.downsample, 'norm_name', None) == 'bn' or b.downsample.with_activation):
            return None
    if not all(bn.training for bn in bns):
        return None
    if any(b.downsample is not None for b in blocks[1:]) or x.shape[-1] % 8:
        return None
    params = []
    for b in blocks:
        params += [b.conv1.weight, b.bn1.weight, b.bn1.bias, b.conv2.weight, b.bn2.weight, b.bn2.bias,
                   b.conv3.weight, b.bn3.weight, b.bn3.bias]
        if b.downsample is not None:
            params += [b.downsample.conv.weight, b.downsample.bn.weight, b.downsample.bn.bias]
    if not grad_mode(x, *params):
        return None
    return BottleneckChainFn.apply(x, blocks, skip_through, *params)


class ConvFn(Function):
    """conv + bias (shift) [+ ReLU], output dtype T or f32; NHWC or ragged rows.
    `w_packed`/`shift` are prepared by the caller (fused heads concatenate several nn.Conv2d)."""

    @staticmethod
    def forward(ctx, x, weight, bias, conv_like, geom, relu, out_dtype, out, skip_through=False):
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
        if skip_through:
            # x is handed through as a second output (see ConvBNTrainSkipFn): the tensor's other consumers take it from
            # there, their gradient arrives here as `dskip` and is added in the data-gradient epilogue — the head's feature
            # maps have three or four consumers each, and autograd summed their gradients with one 3-pass add per pair
            ctx.set_materialize_grads(False)
            return yd, x
        return yd

    @staticmethod
    def backward(ctx, dy, dskip=None):
        from .nn import packed_weight_dgrad, packed_weight_dgrad_s2
        x, weight, y = ctx.saved_tensors
        k, s, p, relu, geom, conv, has_bias = ctx.cfg
        if dy is None:      # this consumer's output took no part in the backward (the root-offset branch): pass dskip on
            return dskip, None, None, None, None, None, None, None, None
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
            wd = packed_weight_dgrad(conv, x.dtype)
            fuse = dskip is not None and s == 1 and dskip.dtype == x.dtype and dskip.shape == x.shape and wd.shape[0] == x.shape[-1]
            res = _wrap(dskip.contiguous(), geom) if fuse else None
            dx = _d(ops.conv2d_dgrad(dzr, wd, k, k, s, p, hw, residual=res, w_classes=packed_weight_dgrad_s2(conv, x.dtype)))
            if dx.shape[-1] != x.shape[-1]:
                dx = dx[..., :x.shape[-1]]
            if dskip is not None and not fuse:
                dx = dx + dskip
        elif dskip is not None:
            dx = dskip
        db = None
        if has_bias:
            ba = _param_acc(conv.bias) if getattr(conv, 'bias', None) is not None else None
            if ba is not None and 0 <= _d(dzr).shape[-1] - conv.bias.numel() < 8:
                # straight into the flat gradient: no temporary, fill or add (a gradient as wide as the layer's channel count
                # padded to a multiple of 8: only the columns that exist are summed)
                ops.colsum(dzr, acc=ba[1].reshape(-1))
                ba[0].fired()
            else:
                db = ops.colsum(dzr)[:weight.shape[0]]
        return dx, dw, db, None, None, None, None, None, None


# GroupNorm + ReLU backward with the mask recomputed from the input instead of read from the output; switch for A/B runs / tests
GN_REMASK = True


class GroupNormReLUFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, geom, G, eps, relu, gn=None):
        ctx.gn = gn      # (the nn.GroupNorm module: its parameters' flat-gradient slices take the gradients directly)
        xin = _wrap(x, geom)
        out = xin.new(x.shape[-1]) if geom is not None else torch.empty_like(x)
        from .nn import kept_zeros
        y, st = ops.groupnorm(xin, gamma, beta, G, eps, relu=relu, out=out, return_stats=True,
                              ws=kept_zeros(ops.groupnorm_stats_size(xin, G), x.device))
        yd = _d(y)
        # (the ReLU mask is recomputed from x in the backward — no residual enters these layers — so y is not kept)
        ctx.save_for_backward(x, None if GN_REMASK else yd, st, gamma, beta)
        ctx.cfg = (geom, G, eps, relu)
        return yd

    @staticmethod
    def backward(ctx, dy):
        x, y, st, gamma, beta = ctx.saved_tensors
        geom, G, eps, relu = ctx.cfg
        dy = dy.contiguous()
        ga, ba = (_param_acc(ctx.gn.weight), _param_acc(ctx.gn.bias)) if ctx.gn is not None else (None, None)
        direct = ga is not None and ba is not None
        from .nn import zeroed_stats
        from . import nn as _nn
        xr = _wrap(x, geom)
        dx, dgamma, dbeta = ops.groupnorm_backward(_wrap(dy, geom), _wrap(y, geom) if y is not None else None, xr, st,
                                                   gamma, G, eps, relu, dgamma_acc=ga[1] if direct else None,
                                                   dbeta_acc=ba[1] if direct else None, beta=beta,
                                                   ws=zeroed_stats(ops.groupnorm_stats_size(xr, G), x.device)
                                                   if (direct and _nn.ZEROED_GN_WS) else None)
        if direct:
            ga[0].fired()
            ba[0].fired()
        return _d(dx), dgamma, dbeta, None, None, None, None, None


class MaxPoolFn(Function):
    @staticmethod
    def forward(ctx, x):
        y, idx = ops.maxpool3x3s2(x, return_argmax=True)     # (one byte per output: the backward gathers, x is not kept)
        ctx.save_for_backward(idx)
        ctx.hw = (x.shape[1], x.shape[2])
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        return ops.maxpool3x3s2_backward_argmax(dy.contiguous(), idx, *ctx.hw)


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


def _dcn_gemm_weight_c(dcn, weight, dtype, cpad):
    """_dcn_gemm_weight for callers that have no col tensor (the fused forward): same cache entries."""
    from .nn import _cache_of, _slot_of
    sl = _slot_of(weight, cpad)
    if sl is not None:
        return sl.packed(dtype).reshape(sl.o_pad, 1, 1, 9 * cpad)
    return _cache_of(dcn).get(('w', dtype), (weight,),
                              lambda: ops.pack_weight(weight, dtype).reshape(-1, 1, 1, 9 * cpad),
                              refresh=lambda buf: ops.pack_weight(weight, dtype, out=buf.view(-1, 3, 3, cpad)))


def _dcn_gemm_weight(dcn, weight, col):
    """(O_pad, 1, 1, 9 C_pad) GEMM weight of a DCNv2 layer in col's dtype: the optimizer's once-per-step packed copy when the
    layer's flat storage is usable as it is (channels-last (O, 3, 3, C) == (O, 1, 1, 9C)), else a cached pack refreshed in place."""
    from .nn import _cache_of, _slot_of
    cpad = col.shape[-1] // 9
    sl = _slot_of(weight, cpad)
    if sl is not None:
        return sl.packed(col.dtype).reshape(sl.o_pad, 1, 1, 9 * cpad)
    return _cache_of(dcn).get(('w', col.dtype), (weight,),
                              lambda: ops.pack_weight(weight, col.dtype).reshape(-1, 1, 1, 9 * cpad),
                              refresh=lambda buf: ops.pack_weight(weight, col.dtype, out=buf.view(-1, 3, 3, cpad)))


class DcnGemmFn(Function):
    """The GEMM half of DCNv2: y = col (rows, 9C) x W^T + bias, W = dcn.weight (O, C, 3, 3)."""

    @staticmethod
    def forward(ctx, col, weight, bias, dcn, geom):
        from .nn import _cache_of, _pad8
        O, Cc = weight.shape[0], weight.shape[1]
        w = _dcn_gemm_weight(dcn, weight, col)
        shift = None
        if bias is not None:
            shift = _cache_of(dcn).get(('b',), (bias,), lambda: _pad8(bias, bias.numel()))
        y = ops.conv2d(_wrap(col, geom), w, 1, 1, shift=shift)
        ctx.save_for_backward(col, weight)
        ctx.cfg = (dcn, geom, bias is not None)
        return _d(y)

    @staticmethod
    def backward(ctx, dy):
        col, weight = ctx.saved_tensors
        dcn, geom, has_bias = ctx.cfg
        dcol, dw, db = _dcn_gemm_backward(col, weight, dcn, geom, has_bias, dy)
        return dcol, dw, db, None, None


def _dcn_gemm_backward(col, weight, dcn, geom, has_bias, dy):
    """Backward of y = col x W^T + bias: d col (GEMM with the transposed weight), dW (weight gradient over col, straight
    into the flat gradient when the optimizer's storage allows), d bias. Shared by DcnGemmFn and DcnFusedFn."""
    from .nn import _cache_of
    O, Cc = weight.shape[0], weight.shape[1]
    dy = dy.contiguous()
    dyr = _wrap(dy, geom)

    def make_wt():  # (9C_pad.., 1, 1, O_pad) = transpose of the packed GEMM weight
        wp = _dcn_gemm_weight(dcn, weight, col).reshape(-1, col.shape[-1])
        return wp.t().contiguous().reshape(wp.shape[1], 1, 1, wp.shape[0])

    def refresh_wt(buf):   # one strided copy into the cached tensor (the weights change every step)
        wp = _dcn_gemm_weight(dcn, weight, col).reshape(-1, col.shape[-1])
        buf.view(wp.shape[1], wp.shape[0]).copy_(wp.t())
    wt = _cache_of(dcn).get(('wt', col.dtype), (weight,), make_wt, refresh=refresh_wt)
    dcol = _d(ops.conv2d(dyr, wt, 1, 1))
    sl = getattr(dcn.weight, '_das_slot', None)
    cpad = col.shape[-1] // 9
    if sl is not None and sl.direct(cpad, dy.shape[-1]):
        # (O,3,3,C) channels-last storage == the (O,1,1,9C) GEMM weight: add straight into the flat gradient
        flush_wgrads()
        with _on_side(col, dy):
            ops.conv2d_wgrad(_wrap(col, geom), dyr, 1, 1, 1, 0, out=sl.grad_cl, accumulate=True)
        sl.fired()
        dw = None
    else:
        dwp = ops.conv2d_wgrad(_wrap(col, geom), dyr, 1, 1, 1, 0)          # (O_pad, 1, 1, 9*Cin_pad)
        dw = dwp.reshape(dwp.shape[0], 3, 3, cpad)[:O, :, :, :Cc].permute(0, 3, 1, 2)
    db = ops.colsum(dyr)[:O] if has_bias else None
    return dcol, dw, db


# DCNv2 forward as one kernel in the training graph (ops.dcn3x3_fused with `col` as a side output for the weight gradient): 441 us
# against 499 us for im2col + GEMM on a head layer at B = 16, -0.35 ms per step. The eval forward (no col: 398 us)
# uses it from das_tuning key dcn.fused_minrows rows up. Switch for A/B runs and tests.
DCN_FUSED = True


class DcnFusedFn(Function):
    """ModulatedDeformConv2d forward in ONE kernel; backward = DcnGemmFn's + DeformIm2colFn's on the saved col."""

    @staticmethod
    def forward(ctx, x, om, weight, bias, dcn, geom):
        from .nn import _cache_of, _pad8
        w = _dcn_gemm_weight_c(dcn, weight, x.dtype, x.shape[-1])
        shift = None
        if bias is not None:
            shift = _cache_of(dcn).get(('b',), (bias,), lambda: _pad8(bias, bias.numel()))
        y, col = ops.dcn3x3_fused(_wrap(x, geom), _wrap(om, geom), w, shift, want_col=True)
        ctx.save_for_backward(x, om, _d(col), weight)
        ctx.cfg = (dcn, geom, bias is not None)
        return _d(y)

    @staticmethod
    def backward(ctx, dy):
        x, om, col, weight = ctx.saved_tensors
        dcn, geom, has_bias = ctx.cfg
        dcol, dw, db = _dcn_gemm_backward(col, weight, dcn, geom, has_bias, dy)
        dx, dom = ops.deform_im2col3x3_backward(_wrap(x, geom), _wrap(om, geom), _wrap(dcol.contiguous(), geom))
        return _d(dx).to(x.dtype), _d(dom), dw, db, None, None


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
    """raw (rows, raw_ps) f32 + the per-level Scale parameters (level-major, four per level: 0-dim tensors) -> pose_pred,
    initial uvd. The scales enter the kernel by value (`desc`); they are inputs here so that autograd knows about them. With
    the flat optimizer their sixteen gradients are added into the flat buffer by ONE multi-tensor launch (before: a stack of
    stacks in the forward, a fill, four row copies and sixteen scalar adds in the backward)."""

    @staticmethod
    def forward(ctx, raw, geom, desc, level_ids, *scale_params):
        pose, uvd = ops.head_assemble(_wrap(raw, geom), desc)
        ctx.save_for_backward(raw)
        ctx.cfg = (geom, desc, level_ids)
        ctx.scale_params = scale_params
        return _d(pose), _d(uvd)

    @staticmethod
    def backward(ctx, d_pose, d_uvd):
        (raw,) = ctx.saved_tensors
        geom, desc, level_ids = ctx.cfg
        d_raw, d_scale = ops.head_assemble_backward(_wrap(raw, geom), _wrap(d_pose.contiguous(), geom),
                                                    _wrap(d_uvd.contiguous(), geom), desc)
        params = ctx.scale_params
        grads = [None] * len(params)
        for i, l in enumerate(level_ids):
            for j in range(4):
                if 4 * l + j < len(params):
                    grads[4 * l + j] = d_scale[i][j]
        accs = [_param_acc(p) for p in params]
        if params and all(a is not None for a in accs):
            live = [(a, g) for a, g in zip(accs, grads) if g is not None]
            torch._foreach_add_([a[1] for a, _ in live], [g.reshape(a[1].shape) for a, g in live])
            for a in accs:
                a[0].fired()
            grads = [None] * len(params)
        else:
            grads = [None if g is None else g.reshape(p.shape) for g, p in zip(grads, params)]
        return (d_raw, None, None, None) + tuple(grads)


def grad_mode(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)
