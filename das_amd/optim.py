"""Optimizer / schedule / data-parallel side of the train step, following the reference's configs
(SURVEY.md section 3.4): SGD momentum 0.9, wd 1e-4, lr 2e-3 with `paramwise_cfg(bias_lr_mult=2,
bias_decay_mult=0)`, step LR with linear warm-up, global-norm clipping at 35 (configs/_base_/schedules/
mmdet_schedule_1x.py:2, configs/das/exp_panoptic.py:201-212).

MI355X-first layout: all parameters live in ONE flat f32 buffer per group (parameters become views of it)
and so do their gradients, so that
  * the optimizer step is one fused HIP kernel per group (clip coefficient folded in, no host sync),
  * the data-parallel exchange is a handful of large RCCL all-reduces over contiguous slices of the flat
    gradient buffer (xGMI rings are per-link bound: few, large messages), launched on a side stream.
"""
import torch
import torch.distributed as dist

from . import train_ops as T


def _is_bias_param(name, module_of):
    """mmcv DefaultOptimizerConstructor: `bias_lr_mult` / `bias_decay_mult` apply to parameters literally
    named 'bias' that are not in a norm layer and not the DCN offset conv."""
    if not name.endswith('.bias') and name != 'bias':
        return False
    mod = module_of[name]
    if isinstance(mod, (torch.nn.modules.batchnorm._BatchNorm, torch.nn.GroupNorm)):
        return False
    if '.conv_offset.' in name or name.endswith('conv_offset.bias'):
        return False
    return True


# zero padding of odd channel counts inside the flat storage (FlatSGD docstring); switch for A/B runs (tools/dev/ab_launches.py)
PAD_ODD_CHANNELS = True


class _Slot:
    """Where one parameter lives in the flat buffers (attached to the parameter as `_das_slot`)."""
    __slots__ = ('opt', 'off', 'numel', 'span', 'o_pad', 'bucket', 'cl_shape', 'grad_cl', 'packable', 'index', 's2_pad')

    def fired(self):
        """Tell the optimizer that this parameter's gradient of the current backward is complete."""
        self.opt._fired(self)

    def direct(self, cin, cout):
        """True if a weight-gradient kernel over (cout, cin)-channel operands can add straight into the flat gradient: cout
        is the gradient tensor's width, i.e. the layer's output channels or their padding to a multiple of 8 (o_pad: the
        kernel then stores the rows that exist, ops.conv2d_wgrad takes the row count from `out`)."""
        return self.cl_shape is not None and cout in (self.cl_shape[0], self.o_pad) and self.cl_shape[3] == cin

    def padded(self):
        """A 1-D parameter with its zero padding to a multiple of 8 elements: f32 view of the flat buffer (what the kernels
        read per-channel constants from, in vectors of 8)."""
        return self.opt.flat_p[self.off:self.off + self.span]

    def packed(self, dtype, dgrad=False):
        return self.opt._packed_view(self, dtype, dgrad)


class FlatSGD:
    """SGD with momentum over flat f32 buffers (parameters, gradients, momentum).

    Conv weights are STORED in (Cout, KH, KW, Cin) order (the parameter is a channels-last strided OIHW view):
    that is the forward kernels' operand layout and the weight-gradient kernels' output layout, so backward
    adds straight into the flat gradient and one launch per step packs the bf16 / data-gradient copies of
    every layer (`das_pack_conv_weights`).
    A weight whose output channels are not a multiple of 8 (the head's predictors: 45 / 27 / 2 / 1 channels) is followed by
    zero rows up to the next multiple, and a 1-D parameter by zeros up to a multiple of 8 elements: parameters, gradients
    and momentum of the padding are zero and stay zero (zero gradient, weight decay of zero), the parameter itself is a
    view of the rows that exist — and the padded block IS the operand the kernels want (8-channel vectors), so these
    layers take the one packing launch and the direct weight-gradient path like every other (before: a pack, a bias pad,
    a data-gradient pack, a fill and two gradient adds per layer and step).

    Data parallel: gradients are all-reduced in param-aligned buckets of >= bucket_mb, launched on a side
    stream DURING backward as soon as a bucket's gradients are complete (hooks count completions; the
    expected counts are learned in the first iteration). Buckets always launch in the same order (from the
    end of the buffer = the layers backward reaches first), so the collective sequence is identical on all
    ranks whatever the timing."""

    def __init__(self, model, lr, momentum=0.9, weight_decay=1e-4, bias_lr_mult=1.0, bias_decay_mult=1.0,
                 max_grad_norm=0.0, bucket_mb=64, overlap=True, force_collectives=False, comm_reserved_cus=None,
                 grad_comm_dtype='f32'):
        self.model, self.base_lr, self.momentum, self.max_grad_norm = model, lr, momentum, max_grad_norm
        # dtype the gradient buckets TRAVEL in: 'f32' (the reference: torch DDP all-reduces the f32 gradients,
        # tools/dist_train.sh:8-9) or 'bf16' — a bucket is rounded to bf16 into a staging buffer, summed over the ranks in
        # bf16 and converted back: half the bytes per xGMI link (516 -> 258 MB per step at four stages). Every rank's
        # gradient loses its low 16 mantissa bits and the sum of W ranks carries ~sqrt(W) * 2^-9 relative error per element
        # (band asserted in tests/test_ddp_cpu.py); momentum, parameters and the clip norm stay f32.
        assert grad_comm_dtype in ('f32', 'bf16'), grad_comm_dtype
        self.grad_comm_dtype = torch.bfloat16 if grad_comm_dtype == 'bf16' else torch.float32
        self._comm_buf, self._copy_back = None, []
        module_of = {}
        for mname, mod in model.named_modules():
            for pname, _ in mod.named_parameters(recurse=False):
                module_of[(mname + '.' if mname else '') + pname] = mod
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        groups = {'main': [], 'bias': []}
        for n, p in named:
            groups['bias' if _is_bias_param(n, module_of) else 'main'].append((n, p))
        dev = named[0][1].device
        # layout: [bias group | main group]; 4-D tensors start on 64-element boundaries: whole 256-byte f32 /
        # 128-byte bf16 lines, so that the weight-gradient atomics and the operand loads never straddle lines
        plan, off = [], 0
        bounds = {}
        for key in ('bias', 'main'):
            start = off
            for n, p in groups[key]:
                if p.dim() == 4:
                    off = (off + 63) // 64 * 64
                    O, I, KH, KW = p.shape
                    span = ((O + 7) // 8 * 8 if (I % 8 == 0 and PAD_ODD_CHANNELS) else O) * I * KH * KW
                else:
                    off = (off + 7) // 8 * 8
                    span = (p.numel() + 7) // 8 * 8 if PAD_ODD_CHANNELS else p.numel()
                plan.append((n, p, off, span))
                off += span
            bounds[key] = (start, off)
        total = off
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.slots, self._conv_slots = [], []
        self._params, self._names = [p for _, p, _, _ in plan], [n for n, _, _, _ in plan]
        for n, p, o, span in plan:
            k = p.numel()
            sl = _Slot()
            sl.opt, sl.off, sl.numel, sl.bucket, sl.cl_shape, sl.grad_cl, sl.packable = self, o, k, 0, None, None, False
            sl.s2_pad, sl.span, sl.o_pad = -1, span, 0
            if p.dim() == 4:
                O, I, KH, KW = p.shape
                vp = self.flat_p[o:o + k].view(O, KH, KW, I)
                vp.copy_(p.data.permute(0, 2, 3, 1))
                p.data = vp.permute(0, 3, 1, 2)                       # OIHW-shaped view, channels-last storage
                sl.grad_cl = self.flat_g[o:o + k].view(O, KH, KW, I)
                p.grad = sl.grad_cl.permute(0, 3, 1, 2)
                sl.cl_shape = (O, KH, KW, I)
                sl.o_pad = span // (KH * KW * I)
                sl.packable = I % 8 == 0 and sl.o_pad % 8 == 0    # (output channels: padded by the storage itself)
                if sl.packable:
                    self._conv_slots.append(sl)
                    mod = module_of.get(n)
                    from . import ops as _ops
                    if (isinstance(mod, torch.nn.Conv2d) and n.endswith('weight') and mod.stride[0] == 2 and KH == KW == 3
                            and mod.padding[0] == mod.padding[1] and _ops.s2_decomposable(3, mod.padding[0])):
                        sl.s2_pad = int(mod.padding[0])   # (its parity-class operands come out of the one packing launch)
            else:
                self.flat_p[o:o + k].copy_(p.data.reshape(-1))
                p.data = self.flat_p[o:o + k].view_as(p)               # parameters become views of the flat buffer
                p.grad = self.flat_g[o:o + k].view_as(p)              # autograd accumulates into the flat buffer
            p._das_slot = sl
            self.slots.append(sl)
            p.register_post_accumulate_grad_hook(lambda t, sl=sl: sl.opt._fired(sl))
        self.groups = [dict(key='main', start=bounds['main'][0], end=bounds['main'][1], lr_mult=1.0, wd=weight_decay),
                       dict(key='bias', start=bounds['bias'][0], end=bounds['bias'][1], lr_mult=bias_lr_mult,
                            wd=weight_decay * bias_decay_mult)]
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.steps = 0
        self._sig = self._param_signature()
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        # `force_collectives`: run the multi-rank gradient path (bucket launches from the completion hooks, comm stream,
        # flag exchange) in a process group of ONE rank too — how a one-GPU box exercises it over RCCL
        self._multi = self.world > 1 or (bool(force_collectives) and dist.is_available() and dist.is_initialized())
        # param-aligned buckets in buffer order
        n_bucket = max(1, bucket_mb * (1 << 20) // 4)
        self.buckets, bstart = [], 0
        for i, sl in enumerate(self.slots):
            sl.bucket = len(self.buckets)
            end = sl.off + sl.span
            if end - bstart >= n_bucket or i == len(self.slots) - 1:
                self.buckets.append((bstart, total if i == len(self.slots) - 1 else end))
                bstart = end
        self.overlap = overlap
        # CUs the persistent one-workgroup-per-CU kernels (weight gradients, streaming 1x1, 3x3 c64) leave free WHILE
        # gradient buckets are in flight (das_tuning key comm.reserved_cus, set at the first bucket of a backward and
        # cleared when the sum is complete): a collective's workgroups that cannot share a CU with a 128-150 KiB-LDS
        # workgroup otherwise push part of a one-wave grid into a second pass (measured with das_dev_occupy_cus,
        # 16 occupied CUs cost +10 % of a step without the reserve). Default: 32 with several ranks.
        self.comm_reserved_cus = int(comm_reserved_cus if comm_reserved_cus is not None else (32 if self.world > 1 else 0))
        self._reserve_on = False
        self.comm_stream = torch.cuda.Stream() if self._multi and dev.type == 'cuda' else None
        if self.world > 1 and any(getattr(m, '_das_sync', False) for m in model.modules()):
            # the SyncBN statistics' own communicator exists before the first step (dist.new_group is a collective:
            # every rank constructs its optimizer at the same point of the program)
            from .nn import stats_group
            stats_group()
        for i, sl in enumerate(self.slots):
            sl.index = i
        self._pexp = None                          # completions per parameter in one backward (learned in iteration 0)
        self._pfires = [0] * len(self.slots)
        self._endonly = [False] * len(self.buckets)   # buckets holding parameters that may or may not get gradients
        self._remaining = [0] * len(self.buckets)  # parameters of the bucket still incomplete in this backward
        self._launched = [False] * len(self.buckets)
        self._next = len(self.buckets) - 1         # next bucket to launch (descending)
        self._works = []
        self._late = False
        self.overlapped_launches = 0               # buckets launched before all_reduce_grads() (diagnostic)
        # one-launch weight packing
        self._table = None
        self._fwd, self._dgrad, self._dgrad_s2, self._packed_epoch = {}, {}, {}, {}
        self._has_s2 = False
        if hasattr(model, 'register_load_state_dict_post_hook'):
            from .nn import bump_param_epoch
            model.register_load_state_dict_post_hook(lambda m, keys: bump_param_epoch())

    # ------------------------------------------------------------------ packed weights
    def _build_table(self):
        import numpy as np
        dt = np.dtype([('off', '<i8'), ('O', '<i4'), ('I', '<i4'), ('KH', '<i4'), ('KW', '<i4'), ('tile_start', '<i4'),
                       ('s2_pad', '<i4')], align=True)
        assert dt.itemsize == 32
        tab = np.zeros(len(self._conv_slots), dtype=dt)
        tiles = 0
        for i, sl in enumerate(self._conv_slots):
            _, KH, KW, I = sl.cl_shape
            O = sl.o_pad
            tab[i] = (sl.off, O, I, KH, KW, tiles, sl.s2_pad)
            tiles += KH * KW * ((O + 63) // 64) * ((I + 63) // 64)
        self._has_s2 = any(sl.s2_pad >= 0 for sl in self._conv_slots)
        self._table = torch.from_numpy(tab.view(np.uint8).copy()).to(self.flat_p.device)
        self._tiles = tiles

    def _packed_view(self, sl, dtype, dgrad):
        from .nn import PARAM_EPOCH
        from . import ops
        if self._packed_epoch.get(dtype) != PARAM_EPOCH[0]:
            if self._table is None:
                self._build_table()
            if dtype not in self._dgrad:
                self._dgrad[dtype] = torch.zeros(self.flat_p.numel(), dtype=dtype, device=self.flat_p.device)
                if dtype != torch.float32:
                    self._fwd[dtype] = torch.zeros_like(self._dgrad[dtype])
                if self._has_s2:
                    self._dgrad_s2[dtype] = torch.zeros_like(self._dgrad[dtype])
            ops.pack_conv_weights(self.flat_p, self._fwd.get(dtype), self._dgrad[dtype], self._table,
                                  len(self._conv_slots), self._tiles, dgrad_s2_dst=self._dgrad_s2.get(dtype))
            self._packed_epoch[dtype] = PARAM_EPOCH[0]
        _, KH, KW, I = sl.cl_shape
        O = sl.o_pad         # (the rows of the stored block: the layer's output channels and their zero padding)
        if dgrad == 's2':
            # the stride-2 data gradient's operands by output parity: {(ph, pw): (I, nth, ntw, O)} (ops.dgrad_s2_weights)
            out, o = {}, sl.off
            cls = ops._s2_classes(KH, sl.s2_pad)
            for ph, (_, nh, _) in enumerate(cls):
                for pw, (_, nw, _) in enumerate(cls):
                    if nh and nw:
                        out[(ph, pw)] = self._dgrad_s2[dtype][o:o + I * nh * nw * O].view(I, nh, nw, O)
                        o += I * nh * nw * O
            return out
        if dgrad:
            return self._dgrad[dtype][sl.off:sl.off + sl.span].view(I, KH, KW, O)
        src = self.flat_p if dtype == torch.float32 else self._fwd[dtype]
        return src[sl.off:sl.off + sl.span].view(O, KH, KW, I)

    # ------------------------------------------------------------------ gradients
    def _param_signature(self):
        """Sum of the parameters' autograd version counters: changes whenever anybody writes a parameter in place
        through torch (init_weights(), p.normal_(), copy_(), EMA, a second optimizer ...). The fused step itself
        goes through raw pointers and bumps PARAM_EPOCH instead."""
        return sum(p._version for p in self._params)

    def zero_grad(self):
        """Clears the flat gradient. Also the once-per-iteration guard of the packed weight copies: an in-place
        parameter edit since the last call invalidates them (they are keyed on PARAM_EPOCH only)."""
        from .autograd import reset_step_state
        reset_step_state()          # (a backward that raised leaves queued weight gradients and sticky flags behind)
        if self._multi and any(self._pfires):
            self._reset_iteration()
        self.flat_g.zero_()
        sig = self._param_signature()
        if sig != self._sig:
            from .nn import bump_param_epoch
            bump_param_epoch()
            self._sig = sig

    def _check_grad_aliasing(self):
        for p, sl in zip(self._params, self.slots):
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * sl.off:
                raise RuntimeError('FlatSGD: a parameter\'s .grad no longer aliases the flat gradient buffer '
                                   '(model.zero_grad(set_to_none=True) or p.grad = ... detached it); use '
                                   'optimizer.zero_grad()')

    def _set_reserve(self, on):
        if self.comm_reserved_cus <= 0 or on == self._reserve_on or self.flat_g.device.type != 'cuda':
            return
        from . import _lib
        _lib.check(_lib.load().das_tuning_set(b'comm.reserved_cus', self.comm_reserved_cus if on else 0), 'das_tuning_set')
        self._reserve_on = on

    def _launch(self, b):
        s, e = self.buckets[b]
        self._set_reserve(True)
        half = self.grad_comm_dtype != torch.float32
        if half and self._comm_buf is None:    # (one staging buffer for the whole flat gradient: buckets never overlap)
            self._comm_buf = torch.empty(self.flat_g.numel(), dtype=self.grad_comm_dtype, device=self.flat_g.device)
        if self.comm_stream is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            from .autograd import wgrad_streams
            for side in wgrad_streams():   # weight gradients of this bucket may still be running on a side stream
                self.comm_stream.wait_stream(side)
            with torch.cuda.stream(self.comm_stream):
                if half:      # round -> sum in bf16 -> back, all on the communication stream, in order
                    buf = self._comm_buf[s:e]
                    buf.copy_(self.flat_g[s:e])
                    dist.all_reduce(buf)
                    self.flat_g[s:e].copy_(buf)
                else:
                    dist.all_reduce(self.flat_g[s:e])
        elif half:            # host backend (gloo): the conversion back waits for the collective in all_reduce_grads()
            buf = self._comm_buf[s:e]
            buf.copy_(self.flat_g[s:e])
            self._works.append(dist.all_reduce(buf, async_op=True))
            self._copy_back.append((s, e))
        else:
            self._works.append(dist.all_reduce(self.flat_g[s:e], async_op=True))

    def _advance(self):
        """Launch, in descending bucket order, every bucket whose parameters are all complete; buckets that
        hold a parameter without a guaranteed gradient (`_endonly`) are skipped here and go out at the end."""
        while self._next >= 0 and (self._endonly[self._next] or self._remaining[self._next] == 0):
            if not self._endonly[self._next]:
                self._launch(self._next)
                self._launched[self._next] = True
                self.overlapped_launches += 1
            self._next -= 1

    def _fired(self, sl):
        if not self._multi or not self.overlap:
            return
        i = sl.index
        self._pfires[i] += 1
        if self._pexp is None:
            return
        b = sl.bucket
        if self._launched[b]:
            self._late = True                      # gradient arrived after its bucket went out
            return
        if self._pfires[i] == self._pexp[i]:
            self._remaining[b] -= 1
            self._advance()

    def mark_complete(self, slots):
        """The gradients of `slots`' parameters are complete although their completion hooks did not run: a replayed
        backward graph (das_amd/graphs.py) adds them into the flat buffer without returning to Python. Same bookkeeping
        as `_fired` up to the expected count, then the ready buckets go out (descending order, as always)."""
        if not self._multi or not self.overlap:
            return
        for sl in slots:
            i = sl.index
            if self._pexp is None:          # first iteration: still learning the expected counts
                self._pfires[i] = max(self._pfires[i], 1)
                continue
            if self._pexp[i] > self._pfires[i]:
                self._pfires[i] = self._pexp[i]
                if not self._launched[sl.bucket]:
                    self._remaining[sl.bucket] -= 1
        if self._pexp is not None:
            self._advance()

    def _learn(self):
        """After the first backward: expected completions per parameter. A bucket is launched early only if every
        one of its parameters produced a gradient on EVERY rank (one MAX all-reduce of the flags); the others
        (the reference's unused root-offset branch, the 2-D flows when a batch has no 2-D-only person, ...) are
        summed at the end of backward, so a parameter that wakes up later can never be late."""
        self._pexp = list(self._pfires)
        flags = torch.zeros(len(self.buckets), dtype=torch.int32, device=self.flat_g.device)
        bad = sorted({sl.bucket for sl in self.slots if self._pexp[sl.index] == 0})
        if bad:
            flags[bad] = 1
        dist.all_reduce(flags, op=dist.ReduceOp.MAX)
        self._endonly = [bool(v) for v in flags.tolist()]

    def _reset_iteration(self):
        self._pfires = [0] * len(self.slots)
        self._launched = [False] * len(self.buckets)
        self._next = len(self.buckets) - 1
        if self._pexp is not None:
            self._remaining = [0] * len(self.buckets)
            for sl in self.slots:
                if self._pexp[sl.index] > 0:
                    self._remaining[sl.bucket] += 1

    def all_reduce_grads(self):
        """Finish the gradient sum over ranks (the mean is folded into the step's grad_scale): launch the
        buckets that backward did not complete (unused parameters, first iteration), then wait."""
        from .autograd import finish_backward
        finish_backward()           # queued weight gradients launched, side streams joined — whatever backward did
        if not self._multi:
            return
        if self._late:
            raise RuntimeError('a gradient was produced after its bucket had been all-reduced (a parameter '
                               'produced more gradients than in the first iteration); use overlap=False')
        # whatever is left: first the remaining early-launch buckets, then the end-only ones, each in descending
        # order — so that the sequence of collectives is the same on every rank however far its backward got
        for endonly in (False, True):
            for b in range(len(self.buckets) - 1, -1, -1):
                if not self._launched[b] and self._endonly[b] == endonly:
                    self._launch(b)
                    self._launched[b] = True
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        for w in self._works:
            w.wait()
        self._works = []
        for s, e in self._copy_back:
            self.flat_g[s:e].copy_(self._comm_buf[s:e])
        self._copy_back = []
        self._set_reserve(False)
        if self._pexp is None and self.overlap:
            self._learn()
        self._reset_iteration()

    @property
    def param_groups(self):
        """torch.optim-style view for LR hooks (mmcv's LrUpdaterHook writes `group['lr']`): one dict per flat group;
        `step()` without an argument applies the lr stored here (group lr = base lr x its lr_mult)."""
        if not hasattr(self, '_param_groups'):
            self._param_groups = [dict(lr=self.base_lr * g['lr_mult'], initial_lr=self.base_lr * g['lr_mult'],
                                       momentum=self.momentum, weight_decay=g['wd'], lr_mult=g['lr_mult'])
                                  for g in self.groups]
        return self._param_groups

    def step(self, lr=None):
        from .autograd import finish_backward
        finish_backward()
        self._check_grad_aliasing()
        scale = 1.0 / self.world
        sumsq = None
        if self.max_grad_norm > 0:
            sumsq = T.grad_sumsq(self.flat_g, self.sumsq)
        for gi, g in enumerate(self.groups):
            if g['end'] == g['start']:
                continue
            s, e = g['start'], g['end']
            glr = lr * g['lr_mult'] if lr is not None else self.param_groups[gi]['lr']
            T.sgd_momentum_step(self.flat_p[s:e], self.flat_g[s:e], self.flat_m[s:e], glr, self.momentum,
                                g['wd'], grad_scale=scale, max_norm=self.max_grad_norm, grad_sumsq_t=sumsq,
                                first_step=self.steps == 0)
        self.steps += 1
        from .nn import bump_param_epoch
        bump_param_epoch()  # packed bf16 weight copies are rebuilt on the next forward

    # ------------------------------------------------------------------ checkpointing
    def _dense_momentum(self):
        """name -> momentum buffer as a dense tensor of the parameter's own (OIHW) shape, on the host"""
        out = {}
        for n, p, sl in zip(self._names, self._params, self.slots):
            m = self.flat_m[sl.off:sl.off + sl.numel]
            if sl.cl_shape is not None:
                O, KH, KW, I = sl.cl_shape
                m = m.view(O, KH, KW, I).permute(0, 3, 1, 2)
            out[n] = m.reshape(p.shape).detach().contiguous().cpu()
        return out

    def state_dict(self):
        """The layout `torch.optim.SGD.state_dict()` has for the reference's optimizer — mmcv's
        DefaultOptimizerConstructor with `paramwise_cfg` makes one param group PER parameter, in
        `model.named_parameters()` order — so that a checkpoint written here resumes upstream
        (`runner.resume` -> `optimizer.load_state_dict`, tools/train.py:200-210) and an upstream checkpoint resumes here:
        state[i]['momentum_buffer'] dense OIHW, param_groups[i] = lr / momentum / dampening / weight_decay / nesterov.
        The step count and base lr ride along under `das_*` keys (torch ignores unknown top-level keys)."""
        mom = self._dense_momentum()
        # EVERY parameter in named_parameters() order, frozen ones included: mmcv's constructor appends a group for a
        # parameter with requires_grad=False too (it just never gets a momentum buffer), and torch's load_state_dict
        # compares group counts
        order = [n for n, p in self.model.named_parameters()]
        group_of = {}
        for g in self.groups:
            for n, sl in zip(self._names, self.slots):
                if g['start'] <= sl.off < g['end']:
                    group_of[n] = g
        main = next(g for g in self.groups if g['key'] == 'main')
        state, pgs = {}, []
        for i, n in enumerate(order):
            g = group_of.get(n, main)      # (frozen: the plain defaults, no state)
            if self.steps > 0 and n in mom:
                state[i] = dict(momentum_buffer=mom[n])
            pgs.append(dict(lr=self.base_lr * g['lr_mult'], initial_lr=self.base_lr * g['lr_mult'], momentum=self.momentum,
                            dampening=0, weight_decay=g['wd'], nesterov=False, params=[i]))
        return dict(state=state, param_groups=pgs, das_steps=self.steps, das_base_lr=self.base_lr, das_names=order)

    def load_state_dict(self, sd):
        """Accepts this class's layout, the reference's (torch SGD: `state` / `param_groups`, one group per parameter or
        one group for all) and the round-2 layout of this repo (`momentum_buffer` by parameter name)."""
        if 'momentum_buffer' in sd:
            mom, steps = sd['momentum_buffer'], int(sd.get('steps', 1))
        else:
            flat_ids = [i for g in sd['param_groups'] for i in g['params']]
            order = [n for n, p in self.model.named_parameters()]          # upstream / this class: frozen ones included
            if len(flat_ids) != len(order):
                trainable = [n for n, p in self.model.named_parameters() if p.requires_grad]   # (files of rounds 2-3)
                if len(flat_ids) != len(trainable):
                    raise ValueError(f'optimizer state holds {len(flat_ids)} parameters, the model has {len(order)} '
                                     f'({len(trainable)} trainable)')
                order = trainable
            mom = {}
            for n, i in zip(order, flat_ids):
                st = sd['state'].get(i, sd['state'].get(str(i)))
                if st is not None and st.get('momentum_buffer') is not None:
                    mom[n] = st['momentum_buffer']
            steps = int(sd.get('das_steps', 1 if mom else 0))
        for n, p, sl in zip(self._names, self._params, self.slots):
            dst = self.flat_m[sl.off:sl.off + sl.numel]
            if n not in mom:
                if 'momentum_buffer' in sd:
                    raise KeyError(f'optimizer state has no momentum buffer for {n}')
                dst.zero_()        # (torch creates a buffer at a parameter's first gradient: never stepped = none)
                continue
            m = mom[n].to(self.flat_m.device, torch.float32)
            if sl.cl_shape is not None:
                O, KH, KW, I = sl.cl_shape
                dst.view(O, KH, KW, I).copy_(m.permute(0, 2, 3, 1))
            else:
                dst.copy_(m.reshape(-1))
        self.steps = steps

    def grad_norm(self):
        return float(torch.sqrt(T.grad_sumsq(self.flat_g)).item()) / self.world


def step_lr(base_lr, epoch, it, steps=(16, 20), gamma=0.1, warmup_iters=250, warmup_ratio=1.0 / 3):
    """mmcv StepLrUpdaterHook with linear warm-up: lr_it = lr*(1 - (1 - it/warm)*(1 - ratio))."""
    lr = base_lr * (gamma ** sum(1 for s in steps if epoch >= s))
    if it < warmup_iters:
        lr = lr * (1 - (1 - it / warmup_iters) * (1 - warmup_ratio))
    return lr


def build_optimizer(model, cfg, optimizer_config=None):
    """build_optimizer(model, cfg) with the whole config, or — the reference's call shape (mmdet3d/apis/train.py
    via mmcv `build_optimizer(model, cfg.optimizer)`) — build_optimizer(model, cfg.optimizer, cfg.optimizer_config)."""
    if 'optimizer' not in cfg and ('type' in cfg or 'lr' in cfg):
        cfg = dict(optimizer=cfg, optimizer_config=optimizer_config or {})
    opt = dict(cfg.get('optimizer', {}))
    assert opt.get('type', 'SGD') == 'SGD'
    pw = opt.get('paramwise_cfg', {}) or {}
    clip = (cfg.get('optimizer_config', {}) or {}).get('grad_clip') or {}
    return FlatSGD(model, lr=opt.get('lr', 0.02), momentum=opt.get('momentum', 0.9),
                   weight_decay=opt.get('weight_decay', 1e-4), bias_lr_mult=pw.get('bias_lr_mult', 1.0),
                   bias_decay_mult=pw.get('bias_decay_mult', 1.0), max_grad_norm=clip.get('max_norm', 0.0))


class GcPark:
    """The training loop's policy for the interpreter's cyclic garbage collector, shared by tools/train.py and bench.py so
    that the benchmarked step IS the shipped step. A generation-2 collection over the autograd graphs of a few steps is a
    20-100 ms host pause (single 106 / 191 ms steps among 83 ms ones: tools/dev/first_process_steps.py). After `warmup`
    steps — model, optimizer, schedules and workspaces exist and will live for the whole run — everything alive is moved
    to the permanent generation (gc.freeze) and automatic collection is switched off; reference counting still frees every
    step's tensors, and the cycles a step does leave behind are collected at the caller's own quiet points
    (`collect()`: logging intervals, checkpoints, epoch ends), where a pause costs nothing."""

    def __init__(self, warmup=3):
        self.warmup, self.steps, self.parked = warmup, 0, False

    def step(self):
        self.steps += 1
        if not self.parked and self.steps >= self.warmup:
            self.park()

    def park(self):
        import gc
        gc.collect()
        gc.freeze()
        gc.disable()
        self.parked = True

    def collect(self):
        import gc
        if self.parked:
            gc.collect()

    def release(self):
        import gc
        if self.parked:
            gc.enable()
            gc.unfreeze()
            self.parked = False


def finish_checks():
    """Everything a training loop must have looked at before it saves weights, evaluates or ends: the SyncBN row-count
    answers that are still in flight (autograd.verify_rows reads them up to ROWS_CHECK_LAG steps late; a mismatch in the last
    steps before a checkpoint would otherwise be saved unreported). Cheap: the events have long completed."""
    from . import autograd
    autograd.verify_rows(wait=True)


MAX_RUN_AHEAD = 3     # steps the host may queue beyond the one the GPU runs (0 = unbounded; see train_iteration)


def train_iteration(model, optimizer, data, lr):
    """One optimisation step: forward + losses + backward (HIP kernels under autograd), gradient
    all-reduce over RCCL, fused clip + SGD. Returns the detector's `train_step` dict (queued, not waited for)."""
    optimizer.zero_grad()
    # the log variables stay on the device until the whole step has been queued: read back right after the forward
    # pass (as mmdet's `_parse_losses` does) they drain the launch queue, and backward then starts from an empty one
    lazy = 'lazy_log' in getattr(getattr(model.train_step, '__code__', None), 'co_varnames', ())
    out = model.train_step(data, None, lazy_log=True) if lazy else model.train_step(data, None)
    out['loss'].backward()
    optimizer.all_reduce_grads()
    optimizer.step(lr)
    # bounded run-ahead: the host may queue at most MAX_RUN_AHEAD steps beyond the one the GPU is running (a BLOCKING event
    # per step, wait for the one of step n - MAX_RUN_AHEAD). Unbounded, a kernel fault would surface many steps late and a wall-clock "s/it" would
    # measure launch-queue time; two steps of slack keep the queue full and never stall a loader.
    if MAX_RUN_AHEAD > 0 and out['loss'].is_cuda:
        evs = optimizer.__dict__.setdefault('_step_events', [])
        ev = torch.cuda.Event(blocking=True)     # (a spinning wait on a default event cost 8 ms per step: measured)
        ev.record()
        evs.append(ev)
        if len(evs) > MAX_RUN_AHEAD:
            evs.pop(0).synchronize()
    # (`out['log_vars']` is a LazyLogVars: a mapping name -> float whose values reach the host without stalling this
    # thread; reading one waits for this step's forward pass only. The step is NOT synchronised with the host here:
    # the next batch can be fetched and the next step queued while this one still runs.)
    return out
