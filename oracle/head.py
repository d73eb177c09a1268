"""Oracle: DASHead forward incl. the recursive-update branch (test infrastructure only).

Follows /root/reference/mmdet3d/models/pose_heads/das_head.py
  _forward_single :180-216, _forward_centerness :218-230, forward_single :232-267
and /root/reference/mmdet3d/models/pose_heads/recursive_update.py
  offset_sample_core :9-31, offset_sample :34-82, NextLevelOffset.forward :186-197,
  RecursiveUpdateLayer.forward :220-235, RecursiveUpdateBranch.forward :250-255.
"""
import torch
import torch.nn.functional as F

from .nn_ops import conv_gn_relu, dcn_pack, group_norm


def _tower(sd, p, x, n, dcn_last=True):
    for i in range(n):
        x = conv_gn_relu(sd, f'{p}.{i}', x, 1, dcn=(dcn_last and i == n - 1))
    return x


def _branch(sd, p, x, gn_groups=32):
    """`_init_branch` tower: 3x3 conv+GN+ReLU layers named <p>.0, <p>.1, ..."""
    i = 0
    while f'{p}.{i}.conv.weight' in sd:
        x = conv_gn_relu(sd, f'{p}.{i}', x, 1, dcn=False, groups=gn_groups)
        i += 1
    return x


def _conv1x1(sd, p, x):
    return F.conv2d(x, sd[p + '.weight'], sd.get(p + '.bias'))


def offset_sample(uvd, samp_off, conf, J, heads, dim=3):
    """recursive_update.py:34-82 restated with explicit (B,J,...) axes.

    uvd (B, J*dim, h, w); samp_off (B, J*heads*2, h, w); conf (B, J*dim, h, w).
    """
    B, _, h, w = uvd.shape
    dev, dt = uvd.device, uvd.dtype
    ys, xs = torch.meshgrid(torch.arange(h, dtype=dt, device=dev), torch.arange(w, dtype=dt, device=dev), indexing='ij')
    pts = torch.stack((xs, ys), 0) + 0.5  # (2,h,w) pixel centres (recursive_update.py:217)
    norm = uvd.new_tensor([w, h]).view(1, 2, 1, 1)

    u = uvd.reshape(B * J, dim, h, w)
    off_uv = u[:, :2]
    # stage 1: the 4 head offsets seen from the current target location
    tgt = ((pts + off_uv) / norm).permute(0, 2, 3, 1)
    so = samp_off.reshape(B * J, heads * 2, h, w)
    # the reference casts to float (recursive_update.py:25,56) to leave fp16; f64 stays f64 here so that
    # the oracle can also serve as a high-precision gradient reference
    up = (lambda t: t) if uvd.dtype == torch.float64 else (lambda t: t.float())
    from_tgt = F.grid_sample(up(so), 2 * tgt - 1, mode='bilinear', padding_mode='zeros', align_corners=False)
    from_tgt = from_tgt.view(B * J, heads, 2, h, w) + off_uv[:, None]
    from_src = so.view(B * J, heads, 2, h, w)
    s = torch.cat([from_tgt, from_src], 1).reshape(B * J * 2 * heads, 2, h, w)
    loc = ((pts + s) / norm).permute(0, 2, 3, 1)
    # stage 2: sample [offset(dim), conf(dim)] at each of the 2*heads locations
    H2 = 2 * heads
    feat = torch.cat([u.repeat_interleave(H2, 0), conf.reshape(B * J, dim, h, w).repeat_interleave(H2, 0)], 1)
    samp = F.grid_sample(up(feat), 2 * loc - 1, mode='bilinear', padding_mode='zeros', align_corners=False)
    s_off, s_conf = samp[:, :dim], samp[:, dim:]
    diff = torch.cat([s, s.new_zeros(s.size(0), 1, h, w)], 1) if dim == 3 else s
    s_off = (s_off + diff).reshape(B * J, H2, dim, h, w)
    wgt = s_conf.reshape(B * J, H2, dim, h, w).softmax(1)
    return (s_off * wgt).sum(1).reshape(B, J * dim, h, w)


def recursive_update_branch(sd, p, pose_feat, uvd, J, heads, num_layers, dim=3):
    feat = F.relu(group_norm(sd, p + '.reduction.gn', F.conv2d(pose_feat, sd[p + '.reduction.conv.weight'])))
    off = uvd
    for i in range(num_layers):
        q = f'{p}.layer_{i}.next_level_offset'
        upd = F.relu(group_norm(sd, q + '.update_feat_conv.gn', dcn_pack(sd, q + '.update_feat_conv.conv', feat)))
        feat = feat + upd
        samp_off = _conv1x1(sd, q + '.sampling_offset', feat)
        conf = _conv1x1(sd, q + '.sampling_conf', feat)
        wgt = torch.sigmoid(_conv1x1(sd, q + '.update_weight', feat))
        nxt = _conv1x1(sd, q + '.update_offset_value', feat)
        off = (1 - wgt) * off + wgt * nxt
        off = offset_sample(off, samp_off, conf, J, heads, dim)
    return off


def head_forward_single(sd, p, x, lvl, stride, cfg, train):
    """One FPN level. cfg keys: num_joints, root_idx, depth_factor, z_norm, stacked_convs,
    num_heads, num_layers. Returns (cls, pose_pred, centerness[, ref_uvd])."""
    J, root = cfg['num_joints'], cfg['root_idx']
    n = cfg.get('stacked_convs', 2)
    cls_feat = _tower(sd, p + 'cls_convs', x, n)
    cls_score = _conv1x1(sd, p + 'conv_cls', _branch(sd, p + 'conv_cls_prev', cls_feat))
    reg_feat = _tower(sd, p + 'reg_convs', x, n)
    pose_feat = _tower(sd, p + 'pose_convs', x, n)
    parts = [
        _conv1x1(sd, p + 'conv_regs.0', _branch(sd, p + 'conv_reg_prevs.0', reg_feat)),
        _conv1x1(sd, p + 'conv_regs.1', _branch(sd, p + 'conv_reg_prevs.1', reg_feat)),
        _conv1x1(sd, p + 'conv_poses.0', _branch(sd, p + 'conv_pose_prevs.0', pose_feat)),
        _conv1x1(sd, p + 'conv_poses.1', _branch(sd, p + 'conv_pose_prevs.1', pose_feat)),
    ]
    # centerness_on_reg=True (configs/_base_/models/das.py:39)
    ctr = _conv1x1(sd, p + 'conv_centerness', _branch(sd, p + 'conv_centerness_prev', reg_feat))

    sc = [sd[f'{p}scales.{lvl}.{k}.scale'] for k in range(4)]
    offset = parts[0] * sc[0]
    depth = parts[1] * sc[1]
    B, _, h, w = x.shape
    uvd = parts[2].reshape(B, J, 3, h, w)
    uvd = torch.stack([uvd[:, :, 0] * sc[2], uvd[:, :, 1] * sc[2], uvd[:, :, 2] * sc[3]], 2)
    # root joint: relative depth 0, sigma *logit* 1 (das_head.py:249-250)
    zmask = torch.ones(J, 3, dtype=x.dtype, device=x.device)
    zmask[root, 2] = 0
    uvd = (uvd * zmask[None, :, :, None, None]).reshape(B, 3 * J, h, w)
    sigma = parts[3].reshape(B, J, 3, h, w) * zmask[None, :, :, None, None] + (1 - zmask)[None, :, :, None, None]
    sigma = sigma.reshape(B, 3 * J, h, w)

    ref = recursive_update_branch(sd, p + 'recursive_update_branch', pose_feat, uvd, J,
                                  cfg.get('num_heads', 4), cfg.get('num_layers', 1))
    ref = (ref.reshape(B, J, 3, h, w) * zmask[None, :, :, None, None]).reshape(B, 3 * J, h, w)

    if train:
        return cls_score, torch.cat([offset, depth, uvd, sigma], 1), ctr, ref
    out = ref.reshape(B, J, 3, h, w)
    out = torch.stack([out[:, :, 0] * stride, out[:, :, 1] * stride, out[:, :, 2] * cfg['z_norm']], 2)
    out = out.reshape(B, 3 * J, h, w)
    return cls_score, torch.cat([offset, depth / cfg['depth_factor'], out, sigma], 1), ctr


def head_forward(sd, feats, cfg, prefix='', train=False):
    strides = cfg['strides']
    outs = [head_forward_single(sd, prefix, x, i, strides[i], cfg, train) for i, x in enumerate(feats)]
    return tuple(list(t) for t in zip(*outs))
