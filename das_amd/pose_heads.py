"""DASHead — distribution-aware dense 3D-pose head (reference: mmdet3d/models/pose_heads/
das_head.py, anchor_free_mono3d_pose_head.py, recursive_update.py, real_nvp.py).

Same constructor surface, module tree and state-dict keys as the reference; forward, decode
and (see losses.py) the training losses run as HIP kernels on NHWC tensors.

Per level (das_head.py:180-267): three conv towers (3x3+GN+ReLU, DCNv2+GN+ReLU), one 3x3+GN+ReLU
"prev" conv and a 1x1 predictor per output group, per-level Scale, root-joint pinning, the
recursive-update refinement branch and, in eval mode, the stride / z_norm / depth_factor rescale.
The 1x1 predictors write channel slices of one f32 "raw" tensor per level
([cls|ctr|off|depth|uvd|sigma], each slice padded to 8 channels) which `das_head_assemble`
turns into `pose_pred`.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .nn import ConvModule, Scale, _cache_of, _pad8, add3, as_nhwc, conv_plain, to_nchw_view
from .registry import HEADS, build_loss

INF = 1e8


def _p8(n):
    return (n + 7) // 8 * 8


def _cs(t, c0, c1):
    """channel slice of an NHWC tensor or a Ragged row set (a view)"""
    return t.cslice(c0, c1) if isinstance(t, ops.Ragged) else t[..., c0:c1]


# ---------------------------------------------------------------------------- RealNVP (train only)
def _mlp(d, tanh):
    layers = [nn.Linear(d, 64), nn.LeakyReLU(), nn.Linear(64, 64), nn.LeakyReLU(), nn.Linear(64, d)]
    if tanh:
        layers.append(nn.Tanh())
    return nn.Sequential(*layers)


class RealNVP(nn.Module):
    """3-D flow: 6 coupling layers, masks [[0,0,1],[1,1,0]]x3 (real_nvp.py:29-88)."""
    dim = 3
    mask_pattern = [[0, 0, 1], [1, 1, 0]]

    def __init__(self):
        super().__init__()
        mask = torch.tensor(self.mask_pattern * 3, dtype=torch.float32)
        self.register_buffer('mask', mask)
        self.t = nn.ModuleList([_mlp(self.dim, False) for _ in range(len(mask))])
        self.s = nn.ModuleList([_mlp(self.dim, True) for _ in range(len(mask))])


class RealNVP2D(RealNVP):
    dim = 2
    mask_pattern = [[0, 1], [1, 0]]


# ---------------------------------------------------------------------------- recursive update
class NextLevelOffset(nn.Module):
    def __init__(self, num_joints, num_heads, in_channels, dim=3, **kwargs):
        super().__init__()
        self.num_joints, self.num_heads, self.dim = num_joints, num_heads, dim
        self.sampling_offset = nn.Conv2d(in_channels, num_joints * num_heads * 2, 1)
        self.sampling_conf = nn.Conv2d(in_channels, num_joints * dim, 1)
        nn.init.normal_(self.sampling_offset.weight.data, 0, 1e-2)
        nn.init.constant_(self.sampling_offset.bias.data, 0)
        self.update_feat_conv = ConvModule(in_channels, in_channels, 3, padding=1, conv_cfg=dict(type='DCNv2'),
                                           norm_cfg=dict(type='GN', num_groups=32))
        self.update_weight = nn.Conv2d(in_channels, num_joints * dim, 1)
        self.update_offset_value = nn.Conv2d(in_channels, num_joints * dim, 1)

    def _fused_heads(self, dtype):
        """The four 1x1 heads share their input: one GEMM, slices padded to 8 channels."""
        convs = (self.sampling_offset, self.sampling_conf, self.update_weight, self.update_offset_value)

        def make():
            ws, bs, offs, o = [], [], [], 0
            for c in convs:
                n = c.weight.shape[0]
                w = torch.zeros(_p8(n), 1, 1, c.weight.shape[1], dtype=dtype, device=c.weight.device)
                w[:n, 0, 0] = c.weight.detach()[:, :, 0, 0].to(dtype)
                ws.append(w)
                bs.append(_pad8(c.bias, n))
                offs.append(o)
                o += _p8(n)
            return torch.cat(ws).contiguous(), torch.cat(bs).contiguous(), offs
        srcs = [p for c in convs for p in (c.weight, c.bias)]
        return _cache_of(self).get(('heads', dtype), srcs, make)

    def forward(self, feat, offset):
        """feat (B,h,w,C) T; offset (B,h,w,3J) f32 -> feat', blended offset, samp_off view, conf view."""
        J = self.num_joints
        feat = add3(feat, self.update_feat_conv(feat))
        from . import autograd as ag
        if ag.grad_mode(feat.data if isinstance(feat, ops.Ragged) else feat, self.sampling_offset.weight):
            # training: four separate GEMMs so that autograd sees the four parameter sets
            g = ag._geom(feat)
            f32 = torch.float32
            # (feat is handed through the first three heads' autograd nodes: one summed gradient, no elementwise adds)
            so, feat = conv_plain(feat, self.sampling_offset, out_dtype=f32, skip_through=True)
            conf, feat = conv_plain(feat, self.sampling_conf, out_dtype=f32, skip_through=True)
            wgt, feat = conv_plain(feat, self.update_weight, out_dtype=f32, skip_through=True)
            so, conf, wgt = _cs(so, 0, J * self.num_heads * 2), _cs(conf, 0, J * self.dim), _cs(wgt, 0, J * self.dim)
            nxt = _cs(conv_plain(feat, self.update_offset_value, out_dtype=f32), 0, J * self.dim)
            offset = ag._wrap(ag.SigmoidBlendFn.apply(ag._d(offset), ag._d(wgt), ag._d(nxt), g), g)
            return feat, offset, so, conf
        w, b, offs = self._fused_heads(feat.dtype)
        out = ops.conv2d(feat, w, 1, 1, shift=b, out_dtype=torch.float32)
        so = _cs(out, offs[0], offs[0] + J * self.num_heads * 2)
        conf = _cs(out, offs[1], offs[1] + J * self.dim)
        wgt = _cs(out, offs[2], offs[2] + J * self.dim)
        nxt = _cs(out, offs[3], offs[3] + J * self.dim)
        offset = ops.sigmoid_blend(offset, wgt, nxt)
        return feat, offset, so, conf


class RecursiveUpdateLayer(nn.Module):
    def __init__(self, num_joints, num_heads, in_channels, dim=3, **kwargs):
        super().__init__()
        assert dim == 3, 'the DAS configs use dim=3'
        self.num_joints, self.num_heads, self.dim = num_joints, num_heads, dim
        self.next_level_offset = NextLevelOffset(num_joints, num_heads, in_channels, dim, **kwargs)

    def forward(self, feat, prev_offset):
        feat, off, so, conf = self.next_level_offset(feat, prev_offset)
        from . import autograd as ag
        if ag.grad_mode(ag._d(off), ag._d(so), ag._d(conf)):
            g = ag._geom(off)
            out = ag.OffsetSampleFn.apply(ag._d(off), ag._d(so), ag._d(conf), g, self.num_joints, self.num_heads)
            return feat, ag._wrap(out, g)
        return feat, ops.offset_sample(off, so, conf, self.num_joints, self.num_heads)


class RecursiveUpdateBranch(nn.Module):
    def __init__(self, num_joints, num_heads, in_channels, feat_channels, num_layers=3, dim=3, **kwargs):
        super().__init__()
        self.num_layers = num_layers
        self.reduction = ConvModule(in_channels, feat_channels, 1,
                                    norm_cfg=dict(type='GN', num_groups=32, requires_grad=True))
        for i in range(num_layers):
            self.add_module(f'layer_{i}', RecursiveUpdateLayer(num_joints, num_heads, feat_channels, dim, **kwargs))

    def forward(self, feat, offset):
        feat = self.reduction(feat)
        for i in range(self.num_layers):
            feat, offset = getattr(self, f'layer_{i}')(feat, offset)
        return offset


class Bias(nn.Module):
    def __init__(self, bias=0.0, use_bias=False):
        super().__init__()
        self.use_bias = use_bias


# ---------------------------------------------------------------------------- the head
@HEADS.register_module()
class DASHead(nn.Module):
    _version = 1

    def __init__(self,
                 num_classes,
                 in_channels,
                 feat_channels=256,
                 stacked_convs=4,
                 strides=(4, 8, 16, 32, 64),
                 dcn_on_last_conv=False,
                 conv_bias='auto',
                 background_label=None,
                 center_sample_radius=1.5,
                 centerness_on_reg=True,
                 centerness_branch=(64,),
                 centerness_alpha=2.5,
                 loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
                 loss_reg=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0),
                 loss_pose=dict(type='RLELoss3D', residual=True, loss_weight=1.0),
                 loss_centerness=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                 norm_cfg=dict(type='GN', num_groups=32, requires_grad=True),
                 regress_ranges=((-1, 48), (48, 96), (96, 192), (192, 384), (384, INF)),
                 recursive_update=None,
                 depth_factor=1,
                 z_norm=1,
                 num_joints=15,
                 root_idx=None,
                 cls_branch=(128, 64),
                 reg_branch=((128, 64), (128, 64), (128, 64), (128, 64)),
                 conv_cfg=None,
                 train_cfg=None,
                 test_cfg=None,
                 init_cfg=None,
                 compute_dtype=None):
        super().__init__()
        assert num_classes == 1, 'DAS detects one class (person)'
        assert centerness_on_reg, 'the DAS configs set centerness_on_reg=True'
        assert root_idx is not None and recursive_update is not None
        self.num_classes = self.cls_out_channels = num_classes
        self.in_channels, self.feat_channels, self.stacked_convs = in_channels, feat_channels, stacked_convs
        self.strides = list(strides)
        self.dcn_on_last_conv = dcn_on_last_conv
        assert conv_bias == 'auto' or isinstance(conv_bias, bool)
        self.conv_bias = conv_bias
        self.center_sample_radius, self.centerness_on_reg = center_sample_radius, centerness_on_reg
        self.centerness_branch, self.centerness_alpha = centerness_branch, centerness_alpha
        self.regress_ranges = regress_ranges
        self.depth_factor, self.z_norm, self.root_idx, self.num_joints = depth_factor, z_norm, root_idx, num_joints
        self.group_reg_dims = [2, 1, num_joints * 3, num_joints * 3]
        self.cls_branch, self.reg_branch = cls_branch, reg_branch
        assert len(reg_branch) == len(self.group_reg_dims)
        self.out_channels = [b[-1] if len(b) > 0 else -1 for b in reg_branch]
        self.train_cfg, self.test_cfg, self.conv_cfg, self.norm_cfg = train_cfg, test_cfg, conv_cfg, norm_cfg
        self.fp16_enabled = False
        self.background_label = num_classes if background_label is None else background_label
        assert self.background_label in (0, num_classes)
        self.loss_cls, self.loss_reg = build_loss(loss_cls), build_loss(loss_reg)
        self.loss_pose, self.loss_centerness = build_loss(loss_pose), build_loss(loss_centerness)
        self.init_cfg = init_cfg

        self._init_layers()
        if 'RLE' in loss_pose['type']:
            self.flow3d, self.flow2d = RealNVP(), RealNVP2D()
            self.flow3d_update, self.flow2d_update = RealNVP(), RealNVP2D()
        self.recursive_update = recursive_update
        ru = dict(recursive_update)
        self.prev_loss = ru.pop('prev_loss', False)
        self.recursive_update_branch = RecursiveUpdateBranch(**ru)
        self.compute_dtype = compute_dtype

    # ------------------------------------------------------------------ construction
    def _tower(self):
        convs = nn.ModuleList()
        for i in range(self.stacked_convs):
            chn = self.in_channels if i == 0 else self.feat_channels
            cfg = dict(type='DCNv2') if (self.dcn_on_last_conv and i == self.stacked_convs - 1) else self.conv_cfg
            convs.append(ConvModule(chn, self.feat_channels, 3, stride=1, padding=1, conv_cfg=cfg,
                                    norm_cfg=self.norm_cfg, bias=self.conv_bias))
        return convs

    def _init_branch(self, conv_channels=(64,), conv_strides=(1,)):
        chans = [self.feat_channels] + list(conv_channels)
        branch = nn.ModuleList()
        for i in range(len(conv_strides)):
            branch.append(ConvModule(chans[i], chans[i + 1], 3, stride=conv_strides[i], padding=1,
                                     conv_cfg=self.conv_cfg, norm_cfg=self.norm_cfg, bias=self.conv_bias))
        return branch

    def _init_layers(self):
        self.pose_convs = self._tower()
        self.cls_convs = self._tower()
        self.reg_convs = self._tower()
        self.conv_cls_prev = self._init_branch(self.cls_branch, (1,) * len(self.cls_branch))
        self.conv_cls = nn.Conv2d(self.cls_branch[-1], self.cls_out_channels, 1)
        self.conv_reg_prevs, self.conv_regs = nn.ModuleList(), nn.ModuleList()
        self.conv_pose_prevs, self.conv_poses = nn.ModuleList(), nn.ModuleList()
        for i in range(4):
            prevs, preds = (self.conv_reg_prevs, self.conv_regs) if i < 2 else (self.conv_pose_prevs, self.conv_poses)
            br = self.reg_branch[i]
            if len(br) > 0:
                prevs.append(self._init_branch(br, (1,) * len(br)))
                preds.append(nn.Conv2d(self.out_channels[i], self.group_reg_dims[i], 1))
            else:
                prevs.append(None)
                preds.append(nn.Conv2d(self.feat_channels, self.group_reg_dims[i], 1))
        self.conv_centerness_prev = self._init_branch(self.centerness_branch, (1,) * len(self.centerness_branch))
        self.conv_centerness = nn.Conv2d(self.centerness_branch[-1], 1, 1)
        self.scales = nn.ModuleList([nn.ModuleList([Scale(1.0) for _ in self.group_reg_dims]) for _ in self.strides])
        self._level_sizes = {}   # input (H, W) -> feature-map sizes per level, learnt from the first forward pass
        self.biases = nn.ModuleList([Bias(0.0, use_bias=False) for _ in self.strides])

    def init_weights(self):
        """Normal(0, 0.01) for every nn.Conv2d, conv_cls bias = -log((1-p)/p), p = 0.01
        (das_head.py:86-92); DCNv2 offset convs zero; sampling_offset N(0, 1e-2)."""
        for name, m in self.named_modules():
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, 0, 0.01)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        nn.init.constant_(self.conv_cls.bias, float(-np.log((1 - 0.01) / 0.01)))
        for m in self.modules():
            if hasattr(m, 'conv_offset'):
                m.init_weights()
            if isinstance(m, (nn.GroupNorm,)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    # ------------------------------------------------------------------ forward
    def raw_layout(self):
        P = _p8(3 * self.num_joints)
        return dict(cls=0, ctr=8, off=16, depth=24, uvd=32, sigma=32 + P, total=32 + 2 * P)

    @staticmethod
    def _run(mods, x):
        if mods is not None:
            for m in mods:
                x = m(x)
        return x

    @staticmethod
    def _run_thru(mods, x):
        """_run for a tensor that has further consumers: returns (mods(x), x handed through the first module's autograd
        node) — see nn.conv_plain(skip_through=True)."""
        if not mods:
            return x, x
        y, x = mods[0](x, skip_through=True)
        for m in list(mods)[1:]:
            y = m(y)
        return y, x

    def forward_rows(self, x, level_ids):
        """x: `ops.Ragged` rows of the FPN levels `level_ids` (all levels in one launch per layer; the
        head's weights are shared across levels, das_head.py:176-178). Returns Ragged f32:
        cls (rows,1 view), pose_pred (rows,3+6J), centerness (rows,1 view)[, ref_uvd (rows,3J)]."""
        J, L = self.num_joints, self.raw_layout()
        cls_feat, x = self._run_thru(self.cls_convs, x)     # (x has three consumers: handed through the first two)
        reg_feat, x = self._run_thru(self.reg_convs, x)
        pose_feat = self._run(self.pose_convs, x)
        from . import autograd as ag
        train_graph = ag.grad_mode(x.data, self.conv_cls.weight)
        raw = None if train_graph else x.new(L['total'], torch.float32)
        parts = []

        def predict(feat, prevs, pred, c0, more=False):
            """more: `feat` has further consumers after this predictor — returns it handed through (train graph)."""
            n = _p8(pred.weight.shape[0])
            if train_graph:  # slices are concatenated (in the raw_layout order) so autograd sees each predictor
                mid, feat = self._run_thru(prevs, feat) if (more and prevs) else (self._run(prevs, feat), feat)
                parts.append(conv_plain(mid, pred, out_dtype=torch.float32).data)
            else:
                conv_plain(self._run(prevs, feat), pred, out_dtype=torch.float32, out=_cs(raw, c0, c0 + n))
            return feat
        predict(cls_feat, self.conv_cls_prev, self.conv_cls, L['cls'])
        reg_feat = predict(reg_feat, self.conv_centerness_prev, self.conv_centerness, L['ctr'], more=True)
        reg_feat = predict(reg_feat, self.conv_reg_prevs[0], self.conv_regs[0], L['off'], more=True)
        predict(reg_feat, self.conv_reg_prevs[1], self.conv_regs[1], L['depth'])
        pose_feat = predict(pose_feat, self.conv_pose_prevs[0], self.conv_poses[0], L['uvd'], more=True)
        pose_feat = predict(pose_feat, self.conv_pose_prevs[1], self.conv_poses[1], L['sigma'], more=True)

        if train_graph and x.data.is_cuda:
            # training: the kernels read the Scale values from DEVICE memory (one gather launch) — as host floats they cost a
            # device-to-host copy of parameters the optimizer wrote at the end of the previous step, i.e. the host waited for the
            # GPU to finish that step before it could queue the head of this one, every step
            dev_sc = torch.stack([s.scale.detach() for lv in self.scales for s in lv]).float().view(len(self.scales), 4)
            if tuple(level_ids) != tuple(range(len(self.scales))):
                dev_sc = dev_sc[list(level_ids)]
            desc = ops.head_desc(J, self.root_idx, L['total'], L['off'], L['depth'], L['uvd'], L['sigma'], None,
                                 [self.strides[l] for l in level_ids], self.z_norm, self.depth_factor,
                                 scale_dev=dev_sc.contiguous())
        else:
            sc = self._scale_values()
            desc = ops.head_desc(J, self.root_idx, L['total'], L['off'], L['depth'], L['uvd'], L['sigma'],
                                 [sc[l] for l in level_ids], [self.strides[l] for l in level_ids], self.z_norm,
                                 self.depth_factor)
        if train_graph:
            raw = x.like(torch.cat(parts, 1))
            pose_d, uvd_d = ag.HeadAssembleFn.apply(raw.data, ag._geom(raw), desc, tuple(level_ids),
                                                    *[s.scale for lv in self.scales for s in lv])
            pose_pred, uvd0 = x.like(pose_d), x.like(uvd_d)
            ref = self.recursive_update_branch(pose_feat, uvd0)
            # (the mask is built ONCE per device: `zmask[i] = 0` on a device tensor is a host-to-device copy of a scalar on the
            # training stream, and the host waits for the stream to reach it — 15 ms per step in lockstep with the GPU)
            zkey = ('_zmask', str(ref.device), J, self.root_idx)
            zmask = self.__dict__.get(zkey)
            if zmask is None:
                zm = torch.ones(3 * J, dtype=torch.float32)
                zm[self.root_idx * 3 + 2] = 0
                zmask = self.__dict__[zkey] = zm.to(ref.device)
            ref = ref.like(ref.data * zmask)  # ref_uvd[:, root z] = 0 (das_head.py:254)
            assert self.training, 'gradients through the eval-mode rescale are not part of the DAS path'
        else:
            pose_pred, uvd0 = ops.head_assemble(raw, desc)
            ref = self.recursive_update_branch(pose_feat, uvd0)
            ops.head_finalize(pose_pred, ref, desc, eval_mode=not self.training)
        cls, ctr = _cs(raw, L['cls'], L['cls'] + 1), _cs(raw, L['ctr'], L['ctr'] + 1)
        if self.training:
            return cls, pose_pred, ctr, ref
        return cls, pose_pred, ctr

    def forward_single(self, x, lvl):
        """One level, NHWC in / NHWC out (reference signature `forward_single`, das_head.py:232-267)."""
        r = ops.Ragged(x.reshape(-1, x.shape[-1]), x.shape[0], [x.shape[1:3]])
        return tuple(t.level(0) for t in self.forward_rows(r, [lvl]))

    def prefetch_scales(self):
        """Start the device-to-host copy of the per-level Scale parameters NOW (the kernels take them by value): called
        by the detector before the backbone is queued. Read back only when the head needs them, the copy — which depends
        on nothing but the previous optimizer step — has long finished; read back at that point (as rounds 1-2 did:
        `.cpu()` inside forward_rows) the host waits for the whole queued backbone and the GPU then waits for the host
        to queue the rest of the step."""
        params = [s.scale for lv in self.scales for s in lv]
        if not params or not params[0].is_cuda:
            return
        if self.training and torch.is_grad_enabled() and params[0].requires_grad:
            return      # (the training graph reads the scales from device memory: forward_rows)
        from .nn import _versions
        ver = _versions(*params)
        pend = self.__dict__.get('_scale_prefetch')
        if pend is not None and pend[0] == ver:
            return
        host = torch.empty(len(params), dtype=torch.float32, pin_memory=True)
        host.copy_(torch.stack([p.detach().float().reshape(()) for p in params]), non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        self.__dict__['_scale_prefetch'] = (ver, host, done)

    def _scale_values(self):
        """Per-level Scale parameters as python floats (one small D2H copy per parameter version: see prefetch_scales)."""
        params = [s.scale for lv in self.scales for s in lv]

        def make():
            from .nn import _versions
            pend = self.__dict__.get('_scale_prefetch')
            if pend is not None and pend[0] == _versions(*params):
                pend[2].synchronize()
                v = pend[1].tolist()
            else:
                v = torch.stack([p.detach().float() for p in params]).cpu().tolist()
            return [v[i * 4:(i + 1) * 4] for i in range(len(self.scales))]
        return _cache_of(self).get(('scales',), params, make)

    def forward(self, feats):
        """feats: tuple of NCHW-shaped tensors (one per level). Returns the reference's tuple of
        per-output lists, each tensor an NCHW-shaped view of NHWC f32 storage."""
        assert len(feats) == len(self.strides)
        dtype = self.compute_dtype or feats[0].dtype
        x = ops.Ragged.from_levels([as_nhwc(f, dtype) for f in feats])
        outs = self.forward_rows(x, list(range(len(feats))))
        return tuple([to_nchw_view(t.level(l)) for l in range(len(feats))] for t in outs)

    # ------------------------------------------------------------------ decode
    def get_points(self, featmap_sizes, dtype, device, flatten=False):
        pts = []
        for (h, w), s in zip(featmap_sizes, self.strides):
            ys, xs = torch.meshgrid(torch.arange(h, dtype=dtype, device=device),
                                    torch.arange(w, dtype=dtype, device=device), indexing='ij')
            pts.append(torch.stack((xs.reshape(-1) * s, ys.reshape(-1) * s), -1) + s // 2)
        return pts

    def get_poses(self, cls_scores, pose_preds, centernesses, img_metas, cfg=None, rescale=None, return_index=False):
        """das_head.py:653-796. One fused HIP kernel per batch; a single D2H copy of the counts."""
        assert len(cls_scores) == len(pose_preds) == len(centernesses)
        cfg = self.test_cfg if cfg is None else cfg
        J = self.num_joints
        dev = cls_scores[0].device
        f32 = torch.float32
        cls = [as_nhwc_f32(t) for t in cls_scores]
        ctr = [as_nhwc_f32(t) for t in centernesses]
        pose = [as_nhwc_f32(t) for t in pose_preds]
        sf = torch.tensor(np.stack([np.asarray(m['scale_factor'], dtype=np.float32)[:2] for m in img_metas]),
                          dtype=f32, device=dev)
        nms_post = cfg.get('nms_post', 100)
        out = ops.decode(cls, ctr, pose, self.strides, sf, J, cfg.get('nms_pre', -1), nms_post,
                         cfg.get('score_thr', 0.), cfg.get('nms_thr', 0.9),
                         nms_soft=cfg.get('nms_type', 'hard') != 'hard')   # (das_head.py:784-790)
        # ONE device-to-host copy per batch (counts and scores together): a copy per image is a host sync per image
        host = torch.cat([out['count'].to(f32).unsqueeze(1), out['scores']], 1).cpu().numpy()
        vis = torch.ones(len(img_metas), nms_post, J, dtype=f32, device=dev)
        results = []
        for b, meta in enumerate(img_metas):
            K = int(host[b, 0])
            r = {'poses': out['poses'][b, :K], 'vis': vis[b, :K],
                 'centers': out['centers'][b, :K], 'image_paths': [meta.get('filename')],
                 'scores': host[b, 1:1 + K].tolist()}
            if return_index:
                r['index'] = out['index'][b, :K]
            results.append(r)
        return results

    # ------------------------------------------------------------------ train
    def forward_train(self, x, img_metas, gt_bboxes, gt_labels=None, gt_poses_3d=None, gt_labels_3d=None,
                      centers2d=None, depths=None, gt_bboxes_ignore=None, proposal_cfg=None, **kwargs):
        # same as `self.loss(*self(x), ...)` (base_mono3d_dense_pose_head.py:21-38) but the outputs stay in
        # the ragged all-levels layout, which is already the loss's flatten-and-concat order
        from .losses import das_head_loss_rows
        assert len(x) == len(self.strides)
        dtype = self.compute_dtype or x[0].dtype
        rows = ops.Ragged.from_levels([as_nhwc(f, dtype) for f in x])
        cls, pose, ctr, ref = self.forward_rows(rows, list(range(len(x))))
        prep = kwargs.get('targets')
        if 'input_hw' in kwargs:   # remembered for prepare_targets: the level sizes this input size leads to
            self._level_sizes[tuple(kwargs['input_hw'])] = [tuple(s) for s in rows.sizes]
        return das_head_loss_rows(self, cls, pose, ctr, ref, gt_poses_3d, centers2d, depths, prep=prep)

    def prepare_targets(self, input_hw, batch, device, gt_poses_3d, centers2d=None, depths=None):
        """The ground-truth half of the loss (losses.das_head_targets), computed before the backbone runs so that its
        device-to-host copy of the counts does not drain a queued forward pass. Needs the level sizes of this input
        size: known from the second step on (None before — the loss then computes its targets itself)."""
        from .losses import das_head_targets
        sizes = self._level_sizes.get(tuple(input_hw))
        if sizes is None or not self.training:
            return None
        return das_head_targets(self, batch, sizes, device, gt_poses_3d, centers2d, depths)

    def loss(self, cls_scores, pose_preds, centernesses, aux_pose_preds, gt_bboxes, gt_labels, gt_poses_3d,
             gt_labels_3d, centers2d, depths, img_metas, gt_bboxes_ignore=None):
        from .losses import das_head_loss
        return das_head_loss(self, cls_scores, pose_preds, centernesses, aux_pose_preds, gt_labels_3d, gt_poses_3d,
                             centers2d, depths)


def as_nhwc_f32(t):
    """NCHW-shaped f32 tensor (possibly a channels-last / sliced view) -> NHWC view or copy."""
    v = t.permute(0, 2, 3, 1)
    if t.dtype == torch.float32 and v.stride(-1) == 1 and v.stride(1) == v.stride(2) * v.shape[2] \
            and v.stride(0) == v.stride(1) * v.shape[1]:
        return v
    return v.float().contiguous()
