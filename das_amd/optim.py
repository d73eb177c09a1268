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


class FlatSGD:
    def __init__(self, model, lr, momentum=0.9, weight_decay=1e-4, bias_lr_mult=1.0, bias_decay_mult=1.0,
                 max_grad_norm=0.0, bucket_mb=64):
        self.model, self.base_lr, self.momentum, self.max_grad_norm = model, lr, momentum, max_grad_norm
        module_of = {}
        for mname, mod in model.named_modules():
            for pname, _ in mod.named_parameters(recurse=False):
                module_of[(mname + '.' if mname else '') + pname] = mod
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        groups = {'main': [], 'bias': []}
        for n, p in named:
            groups['bias' if _is_bias_param(n, module_of) else 'main'].append((n, p))
        self.groups = []
        dev = named[0][1].device
        total = sum(p.numel() for _, p in named)
        self.flat_p = torch.empty(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        for key, lr_mult, wd in (('main', 1.0, weight_decay), ('bias', bias_lr_mult, weight_decay * bias_decay_mult)):
            start = off
            for n, p in groups[key]:
                k = p.numel()
                self.flat_p[off:off + k].copy_(p.data.reshape(-1))
                p.data = self.flat_p[off:off + k].view_as(p)           # parameters become views of the flat buffer
                p.grad = self.flat_g[off:off + k].view_as(p)          # autograd accumulates into the flat buffer
                off += k
            self.groups.append(dict(key=key, start=start, end=off, lr_mult=lr_mult, wd=wd))
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.steps = 0
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        n_bucket = max(1, bucket_mb * (1 << 20) // 4)
        self.buckets = [(s, min(s + n_bucket, total)) for s in range(0, total, n_bucket)]
        self.comm_stream = torch.cuda.Stream() if self.world > 1 and dev.type == 'cuda' else None

    def zero_grad(self):
        self.flat_g.zero_()

    def all_reduce_grads(self):
        """Sum gradients over ranks (mean is folded into the step's grad_scale). Large contiguous buckets,
        issued on a side stream so the tail of backward / the next forward's packing can overlap."""
        if self.world == 1:
            return
        if self.comm_stream is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                for s, e in self.buckets:
                    dist.all_reduce(self.flat_g[s:e])
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        else:
            for s, e in self.buckets:
                dist.all_reduce(self.flat_g[s:e])

    def step(self, lr):
        scale = 1.0 / self.world
        sumsq = None
        if self.max_grad_norm > 0:
            sumsq = T.grad_sumsq(self.flat_g, self.sumsq)
        for g in self.groups:
            if g['end'] == g['start']:
                continue
            s, e = g['start'], g['end']
            T.sgd_momentum_step(self.flat_p[s:e], self.flat_g[s:e], self.flat_m[s:e], lr * g['lr_mult'], self.momentum,
                                g['wd'], grad_scale=scale, max_norm=self.max_grad_norm, grad_sumsq_t=sumsq,
                                first_step=self.steps == 0)
        self.steps += 1
        from .nn import bump_param_epoch
        bump_param_epoch()  # packed bf16 weight copies are rebuilt on the next forward

    def grad_norm(self):
        return float(torch.sqrt(T.grad_sumsq(self.flat_g)).item()) / self.world


def step_lr(base_lr, epoch, it, steps=(16, 20), gamma=0.1, warmup_iters=250, warmup_ratio=1.0 / 3):
    """mmcv StepLrUpdaterHook with linear warm-up: lr_it = lr*(1 - (1 - it/warm)*(1 - ratio))."""
    lr = base_lr * (gamma ** sum(1 for s in steps if epoch >= s))
    if it < warmup_iters:
        lr = lr * (1 - (1 - it / warmup_iters) * (1 - warmup_ratio))
    return lr


def build_optimizer(model, cfg):
    """cfg: the reference's `optimizer` + `optimizer_config` dicts."""
    opt = dict(cfg.get('optimizer', {}))
    assert opt.get('type', 'SGD') == 'SGD'
    pw = opt.get('paramwise_cfg', {}) or {}
    clip = (cfg.get('optimizer_config', {}) or {}).get('grad_clip') or {}
    return FlatSGD(model, lr=opt.get('lr', 0.02), momentum=opt.get('momentum', 0.9),
                   weight_decay=opt.get('weight_decay', 1e-4), bias_lr_mult=pw.get('bias_lr_mult', 1.0),
                   bias_decay_mult=pw.get('bias_decay_mult', 1.0), max_grad_norm=clip.get('max_norm', 0.0))


def train_iteration(model, optimizer, data, lr):
    """One optimisation step: forward + losses + backward (HIP kernels under autograd), gradient
    all-reduce over RCCL, fused clip + SGD. Returns the detector's `train_step` dict."""
    optimizer.zero_grad()
    out = model.train_step(data, None)
    out['loss'].backward()
    optimizer.all_reduce_grads()
    optimizer.step(lr)
    return out
