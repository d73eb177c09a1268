"""Load the reference's own hot-path source files by path (authoring container only).

mmcv / mmdet are not installed anywhere, so `import mmdet3d` fails. This module injects a
minimal stand-in for the ~20 third-party symbols those files import,
then loads the reference files from /root/reference under their real dotted names. It is
test scaffolding: it exists to pin ``oracle/`` against the reference and to generate the
fixtures in tests/golden/. Nothing here travels to (or is needed on) the GPU box:
``available()`` is False there and every test that uses it skips.

The DCNv2 stand-in calls ``oracle.nn_ops.modulated_deform_conv2d`` — mmcv's CUDA op cannot
run here, so for DCNv2 the "reference" arithmetic IS the oracle's restatement (parity
unpinned for that op).
"""
import functools
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REF = '/root/reference'


def available():
    return os.path.isdir(os.path.join(REF, 'mmdet3d'))


# ------------------------------------------------------------------ third-party stand-ins
class Registry:
    def __init__(self, name):
        self.name = name
        self.module_dict = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self.module_dict[name or cls.__name__] = cls
            return cls
        return deco

    def build(self, cfg, **default):
        cfg = dict(cfg)
        cfg.update({k: v for k, v in default.items() if k not in cfg})
        return self.module_dict[cfg.pop('type')](**cfg)


BACKBONES, HEADS, LOSSES, DETECTORS = (Registry(n) for n in ('backbone', 'head', 'loss', 'detector'))


def build_loss(cfg):
    return LOSSES.build(cfg)


def build_norm_layer(cfg, num_features, postfix=''):
    cfg = dict(cfg)
    t = cfg.pop('type')
    rg = cfg.pop('requires_grad', True)
    if t in ('BN', 'SyncBN'):
        name, layer = 'bn', nn.BatchNorm2d(num_features, **cfg)
    elif t == 'GN':
        name, layer = 'gn', nn.GroupNorm(num_channels=num_features, **cfg)
    else:
        raise KeyError(t)
    for p in layer.parameters():
        p.requires_grad = rg
    return name + str(postfix), layer


class ModulatedDeformConv2dPack(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deform_groups=1, bias=True):
        super().__init__()
        k = kernel_size
        self.stride, self.padding, self.dilation = stride, padding, dilation
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, k, k))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        self.conv_offset = nn.Conv2d(in_channels, 3 * k * k, k, stride, padding, dilation, bias=True)
        stdv = 1.0 / (in_channels * k * k) ** 0.5
        self.weight.data.uniform_(-stdv, stdv)
        self.conv_offset.weight.data.zero_()
        self.conv_offset.bias.data.zero_()

    def forward(self, x):
        from oracle.nn_ops import modulated_deform_conv2d
        out = self.conv_offset(x)
        o1, o2, m = torch.chunk(out, 3, dim=1)
        y = modulated_deform_conv2d(x, torch.cat((o1, o2), 1), torch.sigmoid(m), self.weight, self.bias,
                                    self.stride, self.padding, self.dilation)
        return y.contiguous()


def build_conv_layer(cfg, *args, **kwargs):
    t = 'Conv2d' if cfg is None else cfg['type']
    if t in ('Conv2d', 'Conv'):
        return nn.Conv2d(*args, **kwargs)
    if t == 'DCNv2':
        return ModulatedDeformConv2dPack(*args, **kwargs)
    raise KeyError(t)


class ConvModule(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias='auto', conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'), inplace=True,
                 with_spectral_norm=False, padding_mode='zeros', order=('conv', 'norm', 'act')):
        super().__init__()
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        if bias == 'auto':
            bias = not self.with_norm
        self.conv = build_conv_layer(conv_cfg, in_channels, out_channels, kernel_size, stride=stride,
                                     padding=padding, dilation=dilation, groups=groups, bias=bias)
        if self.with_norm:
            self.norm_name, norm = build_norm_layer(norm_cfg, out_channels)
            self.add_module(self.norm_name, norm)
        if self.with_activation:
            self.activate = nn.ReLU(inplace=inplace)

    def forward(self, x):
        x = self.conv(x)
        if self.with_norm:
            x = getattr(self, self.norm_name)(x)
        if self.with_activation:
            x = self.activate(x)
        return x


class Scale(nn.Module):
    def __init__(self, scale=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor(scale, dtype=torch.float))

    def forward(self, x):
        return x * self.scale


class BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg

    def init_weights(self):
        pass


def force_fp32(apply_to=None, out_fp16=False):
    return lambda f: f


def multi_apply(func, *args, **kwargs):
    pfunc = functools.partial(func, **kwargs) if kwargs else func
    return tuple(map(list, zip(*map(pfunc, *args))))


def _weight_reduce(loss, weight, reduction, avg_factor):
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        return loss.mean() if reduction == 'mean' else (loss.sum() if reduction == 'sum' else loss)
    return loss.sum() / avg_factor if reduction == 'mean' else loss


@LOSSES.register_module()
class FocalLoss(nn.Module):
    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.gamma, self.alpha, self.reduction, self.loss_weight = gamma, alpha, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        nc = pred.size(1)
        t = F.one_hot(target, num_classes=nc + 1)[:, :nc].type_as(pred)
        p = pred.sigmoid()
        pt = (1 - p) * t + p * (1 - t)
        fw = (self.alpha * t + (1 - self.alpha) * (1 - t)) * pt.pow(self.gamma)
        loss = F.binary_cross_entropy_with_logits(pred, t, reduction='none') * fw
        return self.loss_weight * _weight_reduce(loss, weight, self.reduction, avg_factor)


@LOSSES.register_module()
class SmoothL1Loss(nn.Module):
    def __init__(self, beta=1.0, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.beta, self.reduction, self.loss_weight = beta, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        if target.numel() == 0:
            return pred.sum() * 0
        d = torch.abs(pred - target)
        loss = torch.where(d < self.beta, 0.5 * d * d / self.beta, d - 0.5 * self.beta)
        return self.loss_weight * _weight_reduce(loss, weight, self.reduction, avg_factor)


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):
    def __init__(self, use_sigmoid=False, use_mask=False, reduction='mean', class_weight=None, loss_weight=1.0):
        super().__init__()
        assert use_sigmoid
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        loss = F.binary_cross_entropy_with_logits(cls_score, label.float(), reduction='none')
        return self.loss_weight * _weight_reduce(loss, weight, self.reduction, avg_factor)


def _noop(*a, **k):
    return None


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _pkg(name, path=None):
    m = _mod(name)
    m.__path__ = [path] if path else []
    return m


def _load(name, relpath):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


_LOADED = None


def load():
    """Returns a namespace with the reference classes: MSPN2, DASHead, RealNVP, RealNVP2D,
    RLELoss3D, RecursiveUpdateBranch, offset_sample, oks_nms, oks_iou."""
    global _LOADED
    if _LOADED is not None:
        return _LOADED
    assert available(), 'reference tree not mounted'
    if not hasattr(np, 'float'):
        np.float = float  # pose_nms.py:72 uses the alias numpy >= 1.24 removed
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if repo not in sys.path:
        sys.path.insert(0, repo)

    _pkg('mmcv')
    _mod('mmcv.cnn', ConvModule=ConvModule, MaxPool2d=nn.MaxPool2d, build_conv_layer=build_conv_layer,
         build_norm_layer=build_norm_layer, constant_init=_noop, kaiming_init=_noop, normal_init=_noop,
         Scale=Scale, bias_init_with_prob=lambda p: float(-np.log((1 - p) / p)))
    _pkg('mmcv.runner')
    sys.modules['mmcv.runner'].force_fp32 = force_fp32
    sys.modules['mmcv.runner'].BaseModule = BaseModule
    _mod('mmcv.runner.checkpoint', _load_checkpoint=_noop, load_state_dict=_noop, load_checkpoint=_noop)
    _pkg('mmdet')
    _mod('mmdet.utils', get_root_logger=_noop)
    _mod('mmdet.core', multi_apply=multi_apply)
    _pkg('mmdet.models')
    _mod('mmdet.models.builder', HEADS=HEADS, LOSSES=LOSSES, DETECTORS=DETECTORS, build_loss=build_loss)

    _pkg('mmdet3d', os.path.join(REF, 'mmdet3d'))
    _pkg('mmdet3d.core', os.path.join(REF, 'mmdet3d/core'))
    _pkg('mmdet3d.core.post_processing', os.path.join(REF, 'mmdet3d/core/post_processing'))
    nms = _load('mmdet3d.core.post_processing.pose_nms', 'mmdet3d/core/post_processing/pose_nms.py')
    sys.modules['mmdet3d.core'].oks_nms = nms.oks_nms
    sys.modules['mmdet3d.core'].soft_oks_nms = nms.soft_oks_nms
    _pkg('mmdet3d.models', os.path.join(REF, 'mmdet3d/models'))
    _mod('mmdet3d.models.builder', BACKBONES=BACKBONES, HEADS=HEADS, LOSSES=LOSSES, DETECTORS=DETECTORS)
    _pkg('mmdet3d.models.backbones', os.path.join(REF, 'mmdet3d/models/backbones'))
    _pkg('mmdet3d.models.pose_heads', os.path.join(REF, 'mmdet3d/models/pose_heads'))
    _pkg('mmdet3d.models.losses', os.path.join(REF, 'mmdet3d/models/losses'))

    mspn = _load('mmdet3d.models.backbones.mspn_mmpose', 'mmdet3d/models/backbones/mspn_mmpose.py')
    rle = _load('mmdet3d.models.losses.residual_log_likelihood_loss',
                'mmdet3d/models/losses/residual_log_likelihood_loss.py')
    nvp = _load('mmdet3d.models.pose_heads.real_nvp', 'mmdet3d/models/pose_heads/real_nvp.py')
    ru = _load('mmdet3d.models.pose_heads.recursive_update', 'mmdet3d/models/pose_heads/recursive_update.py')
    _load('mmdet3d.models.pose_heads.base_mono3d_dense_pose_head',
          'mmdet3d/models/pose_heads/base_mono3d_dense_pose_head.py')
    _load('mmdet3d.models.pose_heads.anchor_free_mono3d_pose_head',
          'mmdet3d/models/pose_heads/anchor_free_mono3d_pose_head.py')
    head = _load('mmdet3d.models.pose_heads.das_head', 'mmdet3d/models/pose_heads/das_head.py')

    _LOADED = types.SimpleNamespace(
        MSPN2=mspn.MSPN2, DASHead=head.DASHead, RealNVP=nvp.RealNVP, RealNVP2D=nvp.RealNVP2D,
        RLELoss3D=rle.RLELoss3D, RecursiveUpdateBranch=ru.RecursiveUpdateBranch, offset_sample=ru.offset_sample,
        oks_nms=nms.oks_nms, oks_iou=nms.oks_iou, ConvModule=ConvModule)
    return _LOADED


class RefFPN(nn.Module):
    """mmdet 2.14.0 FPN as configured for DAS, assembled from the stub ConvModule so that
    state-dict keys (`lateral_convs.i.conv/bn`, `fpn_convs.i.conv/bn`) match upstream."""

    def __init__(self, in_channels, out_channels, num_outs, start_level=1, norm_cfg=dict(type='BN')):
        super().__init__()
        self.start_level, self.num_outs = start_level, num_outs
        n = len(in_channels) - start_level
        self.lateral_convs = nn.ModuleList(
            ConvModule(in_channels[i + start_level], out_channels, 1, norm_cfg=norm_cfg, act_cfg=None) for i in range(n))
        self.fpn_convs = nn.ModuleList(
            ConvModule(out_channels, out_channels, 3, padding=1, norm_cfg=norm_cfg, act_cfg=None) for _ in range(n))
        for _ in range(num_outs - n):
            self.fpn_convs.append(ConvModule(out_channels, out_channels, 3, stride=2, padding=1, norm_cfg=norm_cfg,
                                             act_cfg=None))

    def forward(self, inputs):
        lats = [l(inputs[i + self.start_level]) for i, l in enumerate(self.lateral_convs)]
        n = len(lats)
        for i in range(n - 1, 0, -1):
            lats[i - 1] = lats[i - 1] + F.interpolate(lats[i], size=lats[i - 1].shape[2:], mode='nearest')
        outs = [self.fpn_convs[i](lats[i]) for i in range(n)]
        outs.append(self.fpn_convs[n](outs[-1]))  # first extra level: no ReLU (mmdet 2.14.0)
        for i in range(n + 1, self.num_outs):
            outs.append(self.fpn_convs[i](F.relu(outs[-1])))
        return tuple(outs)
