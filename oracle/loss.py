"""Oracle: point grid, target assignment and the four DASHead losses (test infrastructure only).

Follows /root/reference/mmdet3d/models/pose_heads/das_head.py
  _get_points_single :269-279, get_targets :488-549, _get_target_single :551-651, loss :281-486;
/root/reference/mmdet3d/models/pose_heads/real_nvp.py backward_p/log_prob :60-80;
/root/reference/mmdet3d/models/losses/residual_log_likelihood_loss.py :21-37;
mmdet 2.14.0 FocalLoss / SmoothL1Loss / CrossEntropyLoss(use_sigmoid) (source absent —
published formulas).
"""
import math

import torch
import torch.nn.functional as F

INF = 1e8


def get_points(featmap_sizes, strides, dtype=torch.float32, device='cpu'):
    pts = []
    for (h, w), s in zip(featmap_sizes, strides):
        ys, xs = torch.meshgrid(torch.arange(h, dtype=dtype, device=device),
                                torch.arange(w, dtype=dtype, device=device), indexing='ij')
        pts.append(torch.stack((xs.reshape(-1) * s, ys.reshape(-1) * s), -1) + s // 2)
    return pts


def target_single(points, lvl_strides, lvl_ranges, gt_labels_3d, gt_poses_3d, centers2d, depths, J,
                  radius=1.5, alpha=2.5, background=1):
    """points (P,2) all levels concatenated; lvl_strides/lvl_ranges per point (P,), (P,2).

    Returns labels (P,), pose_targets (P, 3+4J) [dx,dy un-normalised], centerness (P,).
    """
    P, G = points.size(0), gt_labels_3d.size(0)
    if G == 0:
        return (gt_labels_3d.new_full((P,), background), gt_poses_3d.new_zeros((P, 3 + 4 * J)),
                gt_poses_3d.new_zeros((P,)))
    xs, ys = points[:, 0:1], points[:, 1:2]  # (P,1)
    dx = xs - centers2d[None, :, 0]
    dy = ys - centers2d[None, :, 1]  # (P,G)
    uvd = gt_poses_3d[:, 3:3 + 3 * J].reshape(G, J, 3)
    duvd = uvd - gt_poses_3d[:, None, :3]
    duvd = torch.cat([duvd[..., :2], uvd[..., 2:]], -1)  # z stays as given
    vis = gt_poses_3d[:, 3 + 3 * J:]
    reach = torch.sqrt((duvd[..., :2] ** 2).sum(-1)) * vis  # (G,J)
    max_reach = reach.max(-1)[0][None].expand(P, G)

    r = (lvl_strides * radius)[:, None]  # (P,1)
    cx, cy = centers2d[None, :, 0], centers2d[None, :, 1]
    left = xs - (cx - r)
    right = (cx + r) - xs
    top = ys - (cy - r)
    bottom = (cy + r) - ys
    inside = torch.stack((left, top, right, bottom), -1).min(-1)[0] > 0
    in_range = (max_reach >= lvl_ranges[:, None, 0]) & (max_reach <= lvl_ranges[:, None, 1])

    dist = torch.sqrt(dx ** 2 + dy ** 2)
    dist = torch.where(inside & in_range, dist, torch.full_like(dist, INF))
    min_dist, idx = dist.min(1)
    labels = gt_labels_3d[idx].clone()
    labels[min_dist == INF] = background

    ar = torch.arange(P)
    tgt = torch.cat([dx[ar, idx][:, None], dy[ar, idx][:, None], depths[idx][:, None],
                     duvd.reshape(G, 3 * J)[idx], vis[idx]], 1)
    rel = torch.sqrt((tgt[:, :2] ** 2).sum(-1)) / (1.414 * r[:, 0])
    return labels, tgt, torch.exp(-alpha * rel)


def get_targets(points, strides, regress_ranges, gt_labels_3d_list, gt_poses_3d_list, centers2d_list,
                depths_list, J, radius=1.5, alpha=2.5, background=1):
    """Returns per-level lists, each concatenated over images (image-major inside a level),
    root offsets divided by the level stride (das_head.py:547)."""
    n = [p.size(0) for p in points]
    allp = torch.cat(points)
    st = torch.cat([p.new_full((p.size(0),), float(s)) for p, s in zip(points, strides)])
    rg = torch.cat([p.new_tensor(r)[None].expand(p.size(0), 2) for p, r in zip(points, regress_ranges)])
    per_img = [target_single(allp, st, rg, l, g, c, d, J, radius, alpha, background)
               for l, g, c, d in zip(gt_labels_3d_list, gt_poses_3d_list, centers2d_list, depths_list)]
    labels, targets, ctrs = [], [], []
    for i, s in enumerate(strides):
        a, b = sum(n[:i]), sum(n[:i + 1])
        labels.append(torch.cat([t[0][a:b] for t in per_img]))
        tg = torch.cat([t[1][a:b] for t in per_img]).clone()
        tg[:, :2] = tg[:, :2] / s
        targets.append(tg)
        ctrs.append(torch.cat([t[2][a:b] for t in per_img]))
    return labels, targets, ctrs


# ----------------------------------------------------------------------------- losses
def sigmoid_focal_loss(logits, labels, num_classes, gamma=2.0, alpha=0.25):
    """Per-element focal loss; labels in [0, num_classes] with num_classes = background."""
    t = F.one_hot(labels, num_classes + 1)[:, :num_classes].type_as(logits)
    p = logits.sigmoid()
    pt = (1 - p) * t + p * (1 - t)
    fw = (alpha * t + (1 - alpha) * (1 - t)) * pt.pow(gamma)
    return F.binary_cross_entropy_with_logits(logits, t, reduction='none') * fw


def smooth_l1(pred, target, beta):
    d = (pred - target).abs()
    return torch.where(d < beta, 0.5 * d * d / beta, d - 0.5 * beta)


def realnvp_log_prob(sd, p, x):
    """x (N,d). 6 coupling layers run in reverse, then the N(0,I) prior (real_nvp.py:60-80)."""
    mask = sd[p + '.mask']
    d = x.shape[1]
    z = x
    logdet = x.new_zeros(x.shape[0])

    def mlp(q, v, tanh):
        v = F.leaky_relu(F.linear(v, sd[q + '.0.weight'], sd[q + '.0.bias']))
        v = F.leaky_relu(F.linear(v, sd[q + '.2.weight'], sd[q + '.2.bias']))
        v = F.linear(v, sd[q + '.4.weight'], sd[q + '.4.bias'])
        return torch.tanh(v) if tanh else v

    for i in reversed(range(mask.shape[0])):
        m = mask[i]
        z_ = m * z
        s = mlp(f'{p}.s.{i}', z_, True) * (1 - m)
        t = mlp(f'{p}.t.{i}', z_, False) * (1 - m)
        z = (1 - m) * (z - t) * torch.exp(-s) + z_
        logdet = logdet - s.sum(1)
    prior = -0.5 * (z ** 2).sum(1) - 0.5 * d * math.log(2 * math.pi)
    return prior + logdet


def rle_loss3d(nf, pred, sigma, gt, vis_w, weight):
    """RLELoss3D.forward with residual=True, avg_factor flag False."""
    amp = 1 / math.sqrt(2 * math.pi)
    nf = nf * vis_w
    if vis_w[..., 0].sum() < 1:
        return vis_w[..., 0].sum()
    q = (torch.log(sigma / amp) + (gt - pred).abs() / (math.sqrt(2) * sigma + 1e-9)) * vis_w
    return ((nf + q) * weight).sum() / vis_w[..., 0].sum()


def head_loss(sd, prefix, cls_scores, pose_preds, centernesses, aux_uvds, gts, cfg):
    """gts = dict(gt_labels_3d, gt_poses_3d, centers2d, depths) lists per image.
    cfg keys: num_joints, strides, regress_ranges, depth_factor, z_norm, code_weight, prev_loss."""
    J = cfg['num_joints']
    strides = cfg['strides']
    sizes = [c.shape[-2:] for c in cls_scores]
    points = get_points(sizes, strides, pose_preds[0].dtype, pose_preds[0].device)
    labels, targets, ctr_t = get_targets(points, strides, cfg['regress_ranges'], gts['gt_labels_3d'],
                                         gts['gt_poses_3d'], gts['centers2d'], gts['depths'], J,
                                         cfg.get('center_sample_radius', 1.5), cfg.get('centerness_alpha', 2.5))
    B = cls_scores[0].size(0)
    D = 3 + 6 * J
    f_cls = torch.cat([c.permute(0, 2, 3, 1).reshape(-1, 1) for c in cls_scores])
    f_pose = torch.cat([c.permute(0, 2, 3, 1).reshape(-1, D) for c in pose_preds])
    f_ctr = torch.cat([c.permute(0, 2, 3, 1).reshape(-1) for c in centernesses])
    f_aux = torch.cat([c.permute(0, 2, 3, 1).reshape(-1, 3 * J) for c in aux_uvds])
    f_str = torch.cat([c.new_full((c.numel(),), float(s)) for c, s in zip(centernesses, strides)])
    f_lab = torch.cat(labels)
    f_tgt = torch.cat(targets)
    f_ct = torch.cat(ctr_t)

    pos = ((f_lab >= 0) & (f_lab < 1)).nonzero().reshape(-1)
    npos = len(pos)
    if npos == 0:
        z = (f_cls[0, 0] - f_cls[0, 0]).clone()
        return dict(loss_cls=z, loss_depth=z, loss_pose=z, loss_centerness=z)

    loss_cls = sigmoid_focal_loss(f_cls, f_lab, 1).sum() / (npos + B)

    pp, pc, ps = f_pose[pos], f_ctr[pos], f_str[pos]
    pt, pct, paux = f_tgt[pos], f_ct[pos], f_aux[pos]
    cw = pp.new_tensor(cfg['code_weight'])
    gt_uvd = pt[:, 3:3 + 3 * J]
    is2d = (gt_uvd[:, 2::3] == 0).all(1)
    is3d = ~is2d
    if is3d.sum() > 0:
        l = smooth_l1(pp[is3d, 2], pt[is3d, 2] * cfg['depth_factor'], 1.0 / 9.0) * cw[2]
        loss_depth = l.sum() / is3d.sum()
    else:
        loss_depth = pp[0, 2] - pp[0, 2]

    uvd = pp[:, 3:3 + 3 * J].reshape(npos, J, 3).clone()
    upd = paux.reshape(npos, J, 3).clone()
    sig = pp[:, 3 + 3 * J:].reshape(npos, J, 3).clone()
    uvd[is2d, :, 2] = 0
    upd[is2d, :, 2] = 0
    sig[is2d, :, 2] = 1
    root = pt[:, :3] * ps[:, None]
    root = torch.cat([root[:, :2], torch.zeros_like(root[:, :1])], 1)
    real = gt_uvd.reshape(npos, J, 3) - root[:, None]
    real = torch.cat([real[..., :2] / ps[:, None, None], real[..., 2:] / cfg['z_norm']], -1)
    vis_w = pt[:, 3 + 3 * J:].reshape(npos, J, 1).expand(npos, J, 3)
    sig = sig.sigmoid() + 1e-9

    if cfg.get('prev_loss', True):
        pred = torch.cat([upd, uvd], 1)
        real2, sig2, vis2 = real.repeat(1, 2, 1), sig.repeat(1, 2, 1), vis_w.repeat(1, 2, 1)
        flows = [('_update', slice(0, J)), ('', slice(J, 2 * J))]
    else:
        pred, real2, sig2, vis2 = upd, real, sig, vis_w
        flows = [('', slice(0, J))]
    bar = (pred - real2) / sig2
    two_d = (real2[..., 2] == 0).all(1)
    log_phi = bar.new_zeros(npos, bar.size(1), 1)
    for suffix, sl in flows:
        if two_d.any():
            v = realnvp_log_prob(sd, f'{prefix}flow2d{suffix}', bar[two_d][:, sl, :2].reshape(-1, 2))
            tmp = log_phi[two_d]
            tmp[:, sl, 0] = v.view(-1, J)
            log_phi[two_d] = tmp
        if (~two_d).any():
            v = realnvp_log_prob(sd, f'{prefix}flow3d{suffix}', bar[~two_d][:, sl].reshape(-1, 3))
            tmp = log_phi[~two_d]
            tmp[:, sl, 0] = v.view(-1, J)
            log_phi[~two_d] = tmp
    nf = torch.log(sig2) - log_phi
    loss_pose = rle_loss3d(nf, pred, sig2, real2, vis2, cw[3])
    loss_ctr = F.binary_cross_entropy_with_logits(pc, pct, reduction='mean')
    return dict(loss_cls=loss_cls, loss_depth=loss_depth, loss_pose=loss_pose, loss_centerness=loss_ctr)
