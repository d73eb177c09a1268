"""Loss modules with the reference's registry names (FocalLoss / SmoothL1Loss / CrossEntropyLoss from
mmdet 2.14.0, RLELoss3D from mmdet3d/models/losses/residual_log_likelihood_loss.py) and the DASHead loss
assembly (das_head.py:281-486).

Dense work runs in HIP kernels over all rows of all levels in one launch each: target assignment
(`das_assign_targets`), sigmoid focal loss (`das_sigmoid_focal_loss`); SmoothL1 and centerness BCE on the
positives are HIP kernels too, and so is the RealNVP log-density of the RLE term (`das_realnvp_log_prob`,
forward and backward fused over all positives x joints). What remains in torch tensor ops on the GPU is
the gather of the positive rows and the elementwise algebra around the flow (a few dozen small kernels).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import train_ops as T
from .ops import Ragged
from .registry import LOSSES


def _reduce(total, n_elems, reduction, avg_factor):
    """mmdet `weight_reduce_loss` for a loss whose elementwise values were already summed by the kernel."""
    if avg_factor is None:
        if reduction == 'mean':
            return total / max(n_elems, 1)
        if reduction == 'sum':
            return total
    elif reduction == 'mean':
        return total / avg_factor
    elif reduction == 'sum':
        raise ValueError('avg_factor can not be used with reduction="sum"')
    raise NotImplementedError("reduction='none' needs the per-element losses; the HIP kernels return their sum "
                              '(and per-element gradients)')


@LOSSES.register_module()
class FocalLoss(nn.Module):
    """mmdet 2.14 FocalLoss (sigmoid) for ONE class, label 0 = positive, label num_classes = 1 = background
    (configs/_base_/models/das.py:40-45, call site das_head.py:341-344): `das_sigmoid_focal_loss`."""

    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction='mean', loss_weight=1.0):
        super().__init__()
        assert use_sigmoid, 'only sigmoid focal loss is supported'
        self.use_sigmoid, self.gamma, self.alpha = use_sigmoid, gamma, alpha
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        if pred.dim() != 2 or pred.shape[1] != 1:
            raise NotImplementedError('FocalLoss on the DAS path has num_classes = 1: pred must be (N, 1)')
        total = T.FocalLossSumFn.apply(pred.float(), target.to(torch.int32), self.gamma, self.alpha, weight)
        return self.loss_weight * _reduce(total, pred.numel(), reduction_override or self.reduction, avg_factor)


@LOSSES.register_module()
class SmoothL1Loss(nn.Module):
    """mmdet 2.14 SmoothL1Loss (call site das_head.py:375-379): `das_smooth_l1_loss`."""

    def __init__(self, beta=1.0, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.beta, self.reduction, self.loss_weight = beta, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        if target.numel() == 0:
            return pred.sum() * 0
        total = T.SmoothL1SumFn.apply(pred.float(), target.float(), self.beta, weight)
        return self.loss_weight * _reduce(total, pred.numel(), reduction_override or self.reduction, avg_factor)


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):
    """mmdet 2.14 CrossEntropyLoss(use_sigmoid=True) = BCE with logits (call site das_head.py:470-471):
    `das_bce_logits_loss`."""

    def __init__(self, use_sigmoid=False, use_mask=False, reduction='mean', class_weight=None, loss_weight=1.0):
        super().__init__()
        assert use_sigmoid and not use_mask and class_weight is None
        self.use_sigmoid, self.reduction, self.loss_weight = use_sigmoid, reduction, loss_weight

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        total = T.BCELogitsSumFn.apply(cls_score.float(), label.float(), weight)
        return self.loss_weight * _reduce(total, cls_score.numel(), reduction_override or self.reduction, avg_factor)


@LOSSES.register_module()
class RLELoss3D(nn.Module):
    """residual_log_likelihood_loss.py:7-37. `loss_weight` is accepted and ignored, as the reference's
    `**kwargs` does (configs/_base_/models/das.py:49)."""

    def __init__(self, residual=True, avg_factor=False, **kwargs):
        super().__init__()
        self.residual, self.avg_factor = residual, avg_factor
        self.amp = 1 / math.sqrt(2 * math.pi)

    def logQ(self, gt_uv, pred_jts, sigma):
        return torch.log(sigma / self.amp) + torch.abs(gt_uv - pred_jts) / (math.sqrt(2) * sigma + 1e-9)

    def forward(self, nf_loss, uvd, sigma, gt_uvd, gt_uv_weight, weight=None, avg_factor=None):
        """Generic entry (elementwise tensor algebra on the caller's device). DASHead.loss does not come through
        here: its RLE term is fused with the positive-row gather and the flows (`das_head_loss_rows`)."""
        gt_uv_weight = gt_uv_weight.expand_as(gt_uvd)
        nf_loss = nf_loss * gt_uv_weight
        nvis = gt_uv_weight[..., 0].sum()
        if nvis < 1:
            return nvis
        loss = nf_loss + self.logQ(gt_uvd, uvd, sigma) * gt_uv_weight if self.residual else nf_loss
        if weight is not None:
            loss = loss * weight
        if avg_factor is not None and self.avg_factor:
            return loss.sum() / avg_factor
        return loss.sum() / nvis


def realnvp_log_prob(flow, x):
    """RealNVP.log_prob (real_nvp.py:60-80) on the GPU: the fused forward / backward kernels."""
    return T.realnvp_log_prob(flow, x)


def realnvp_log_prob_torch(flow, x):
    """The same density spelled with torch ops, layer by layer (kept as a cross-check for the tests; the
    training path does not call it)."""
    d = x.shape[1]
    z, logdet = x, x.new_zeros(x.shape[0])
    for i in reversed(range(flow.mask.shape[0])):
        m = flow.mask[i]
        z_ = m * z
        s = flow.s[i](z_) * (1 - m)
        t = flow.t[i](z_) * (1 - m)
        z = (1 - m) * (z - t) * torch.exp(-s) + z_
        logdet = logdet - s.sum(1)
    return -0.5 * (z ** 2).sum(1) - 0.5 * d * math.log(2 * math.pi) + logdet


def _pack_gt(gt_poses_3d, device):
    starts = [0]
    for g in gt_poses_3d:
        starts.append(starts[-1] + int(g.shape[0]))
    D = gt_poses_3d[0].shape[1] if len(gt_poses_3d) else 0
    rows = torch.cat([g.to(device=device, dtype=torch.float32) for g in gt_poses_3d]) if starts[-1] > 0 \
        else torch.zeros(0, D, dtype=torch.float32, device=device)
    return rows.contiguous(), torch.tensor(starts, dtype=torch.int32, device=device)


def _pack_centers(centers2d, depths, n, device):
    """[centers2d | depths] rows for das_assign_targets, or None when the caller did not pass them."""
    if centers2d is None or depths is None:
        return None
    if n == 0:
        return torch.zeros(0, 3, dtype=torch.float32, device=device)
    # (three concatenations for the batch, not one per image)
    c = torch.cat([t.reshape(-1, 2) for t in centers2d]).to(device=device, dtype=torch.float32)
    d = torch.cat([t.reshape(-1, 1) for t in depths]).to(device=device, dtype=torch.float32)
    return torch.cat([c, d], 1).contiguous()


# the ground-truth half of DASHead.loss through das_assign_targets' counts + das_positive_rows (False: the ATen formulation it
# replaces, kept as the cross-check of tests/test_loss_gpu.py); switch for A/B runs and tests
FUSED_TARGETS = True


def das_head_targets(head, B, sizes, device, gt_poses_3d, centers2d=None, depths=None):
    """Everything of DASHead.loss (das_head.py:283-478) that depends on the ground truth and the feature-map geometry
    only: target assignment, the positive rows, the 2-D / 3-D split, the RLE targets and weights, and the host-side
    branch conditions (counts). The detector calls this BEFORE the backbone runs (the one device-to-host copy below then
    waits for a few small kernels instead of draining a queued forward pass), the prediction-dependent half
    (`das_head_loss_rows`) runs without any host synchronisation."""
    J = head.num_joints
    geom = Ragged(torch.empty(sum(B * h * w for h, w in sizes), 0, device=device), B, sizes)
    gt_rows, gt_start = _pack_gt(gt_poses_3d, device)
    sets = 2 if head.prev_loss else 1
    if FUSED_TARGETS and torch.device(device).type == 'cuda':
        # counts out of the assignment kernel itself, the positives' rows from two launches (das_positive_rows): the mask /
        # sum / gather / concatenate / cumsum formulation below is ~35 launches of 3-6 us at the head of a GPU-bound step
        # (its own fill: this runs on the detector's side stream next to the backbone — the per-step zero arena belongs to the
        # main stream's order, and with the host several steps ahead its wrap-around fill may not have RUN yet over there)
        counts = torch.zeros(3, dtype=torch.float32, device=device)
        labels, targets, ctr_t = T.assign_targets(geom, head.strides, head.regress_ranges, gt_rows, gt_start, J,
                                                  head.center_sample_radius, head.centerness_alpha, head.background_label,
                                                  centers=_pack_centers(centers2d, depths, gt_rows.shape[0], device),
                                                  counts=counts)
        prep = dict(B=B, sizes=[tuple(s) for s in sizes], labels=labels)
        npos, n3d, vis_all = counts.tolist()
        npos, n3d = int(npos), int(n3d)
        prep.update(npos=npos, n3d=n3d, nvis_host=vis_all * sets)
        if npos == 0:
            return prep
        pos = torch.nonzero_static(labels == 0, size=npos).reshape(-1)
        prep.update(pos=pos, **T.positive_rows(geom, head.strides, head.regress_ranges, J, pos, targets, ctr_t, head.z_norm,
                                               head.depth_factor, float(sets)))
        return prep
    labels, targets, ctr_t = T.assign_targets(geom, head.strides, head.regress_ranges, gt_rows, gt_start, J,
                                              head.center_sample_radius, head.centerness_alpha, head.background_label,
                                              centers=_pack_centers(centers2d, depths, gt_rows.shape[0], device))
    prep = dict(B=B, sizes=[tuple(s) for s in sizes], labels=labels)
    is_pos = labels == 0
    row_stride = torch.cat([torch.full((geom.starts[l + 1] - geom.starts[l],), float(s), device=device)
                            for l, s in enumerate(head.strides)])
    # counts first (one copy): positives, 3-D positives, visible joints
    vis_all = (targets[:, 3 + 3 * J:] * is_pos[:, None]).sum() * (2 if head.prev_loss else 1)
    is3d_all = is_pos & ~(targets[:, 5:3 + 3 * J:3] == 0).all(1)
    npos, n3d, nvis_host = torch.stack([is_pos.sum().float(), is3d_all.sum().float(), vis_all]).tolist()
    npos, n3d = int(npos), int(n3d)
    prep.update(npos=npos, n3d=n3d, nvis_host=nvis_host)
    if npos == 0:
        return prep
    pos = torch.nonzero_static(is_pos, size=npos).reshape(-1)
    pt, pct, ps = targets[pos], ctr_t[pos], row_stride[pos]
    gt_uvd = pt[:, 3:3 + 3 * J].reshape(npos, J, 3)
    is2d = (gt_uvd[..., 2] == 0).all(1)                              # sample without depth annotation (das_head.py:388)
    # pixel-to-joint targets: image offsets in units of the level's stride, depth in units of z_norm (:393-409)
    root = torch.cat([pt[:, :2] * ps[:, None], torch.zeros_like(pt[:, :1])], 1)
    real = gt_uvd - root[:, None]
    real = torch.cat([real[..., :2] / ps[:, None, None], real[..., 2:] / head.z_norm], -1).contiguous()
    vis = pt[:, 3 + 3 * J:].contiguous()
    kind = is2d.to(torch.int32)
    # rank of every positive among the positives of its kind: its block of J rows in the flows' input
    slot = torch.where(is2d, torch.cumsum(kind, 0), torch.cumsum(1 - kind, 0)).to(torch.int32) - 1
    prep.update(pos=pos, ctr_t=pct, real=real, vis=vis, is2d=kind, slot=slot,
                depth_t=(pt[:, 2] * head.depth_factor).contiguous(),
                nvis=vis.sum() * (2 if head.prev_loss else 1))
    return prep


def das_head_loss_rows(head, cls, pose, ctr, aux, gt_poses_3d, centers2d=None, depths=None, prep=None):
    """cls (rows,1), pose (rows,3+6J), ctr (rows,1), aux = refined uvd (rows,3J): Ragged f32, rows ordered
    level-major / image / (h,w) exactly like the reference's flatten-and-concat (das_head.py:306-333).
    gt_poses_3d: list per image of (G, 3+4J) [cx,cy,depth, J x (u,v,dz), J x vis].
    prep: das_head_targets(...) of the same batch and geometry if the caller computed it ahead of the forward pass."""
    J, B, dev = head.num_joints, cls.B, cls.device
    if prep is None or prep['B'] != B or prep['sizes'] != [tuple(s) for s in cls.sizes]:
        prep = das_head_targets(head, B, cls.sizes, dev, gt_poses_3d, centers2d, depths)
    npos, n3d = prep['npos'], prep['n3d']
    if npos == 0:  # das_head.py:473-478
        z = (cls.data[0, 0] - cls.data[0, 0]).clone()
        return dict(loss_cls=z, loss_depth=z, loss_pose=z, loss_centerness=z)

    lc = head.loss_cls
    loss_cls = T.FocalLossSumFn.apply(cls.data, prep['labels'], lc.gamma, lc.alpha) * (lc.loss_weight / (npos + B))

    pos = prep['pos']
    cw = [float(v) for v in head.train_cfg['code_weight']] if head.train_cfg and head.train_cfg.get('code_weight') \
        else [1.0] * (3 + 6 * J)
    # depth term and RLE pose loss of the positive rows: three elementwise kernels around the two RealNVP launches
    # (train_ops.RLEPoseLossFn), forward and backward; the gradients land in the positives' rows of dense zero tensors
    lr, lp = head.loss_reg, head.loss_pose
    sets = 2 if head.prev_loss else 1
    meta = dict(J=J, sets=sets, n2d=npos - n3d, n3d=n3d, amp=float(lp.amp), beta=float(lr.beta),
                flows={D: ([getattr(head, f'flow{D}d_update')] if sets == 2 else []) + [getattr(head, f'flow{D}d')]
                       for D in (2, 3)},
                **{k: prep[k] for k in ('pos', 'real', 'vis', 'is2d', 'slot', 'depth_t')})
    sums = T.rle_pose_loss_sums(pose.data, aux.data, meta)
    loss_depth = sums[1] * (cw[2] * lr.loss_weight / n3d) if n3d > 0 else sums[1] * 0.0
    nvis = prep['nvis']
    if prep['nvis_host'] < 1:  # residual_log_likelihood_loss.py:24-25
        loss_pose = nvis
    else:
        loss_pose = sums[0] * cw[3] / nvis
    pc = ctr.data.index_select(0, pos)[:, 0]

    lctr = head.loss_centerness
    loss_ctr = T.BCELogitsSumFn.apply(pc, prep['ctr_t']) * (lctr.loss_weight / npos)
    return dict(loss_cls=loss_cls, loss_depth=loss_depth, loss_pose=loss_pose, loss_centerness=loss_ctr)


def das_head_loss(head, cls_scores, pose_preds, centernesses, aux_pose_preds, gt_labels_3d, gt_poses_3d, centers2d,
                  depths):
    """Public signature (lists of NCHW-shaped per-level tensors, das_head.py:283-295)."""
    B = cls_scores[0].shape[0]
    sizes = [t.shape[-2:] for t in cls_scores]

    def rag(lst):
        return Ragged(torch.cat([t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]) for t in lst], 0).float(), B, sizes)
    return das_head_loss_rows(head, rag(cls_scores), rag(pose_preds), rag(centernesses), rag(aux_pose_preds),
                              gt_poses_3d, centers2d, depths)
