"""Oracle: decode = per-level top-k -> threshold -> OKS-NMS (test infrastructure only).

Follows /root/reference/mmdet3d/models/pose_heads/das_head.py get_poses :653-688,
_get_poses_single :690-796 and /root/reference/mmdet3d/core/post_processing/pose_nms.py
oks_iou :51-89, oks_nms :92-126, _rescore / soft_oks_nms :128-194.

Tie-break: the reference's `torch.topk` / `np.argsort()[::-1]` leave equal scores in an
undefined order; the oracle defines "higher score first, then lower flat index" (stable
descending sort), which coincides with the reference on tie-free inputs. `return_index=True`
also returns each kept candidate's flat location index (level-concatenated, fine->coarse).
"""
import numpy as np
import torch

from .loss import get_points


def oks_iou(g, d, a_g, a_d):
    """g (3J,) float32, d (n,3J) float32, areas float32. sigma_k = 0.08 (J != 17)."""
    J = len(g) // 3
    if J == 17:
        sigmas = np.array([.26, .25, .25, .35, .35, .79, .79, .72, .72, .62, .62, 1.07, 1.07, .87, .87, .89, .89]) / 10.0
    else:
        sigmas = np.ones(J, dtype=np.float64) * 0.08
    var = (sigmas * 2) ** 2
    ious = np.zeros(len(d), dtype=np.float32)
    for n in range(len(d)):
        dx = d[n, 0::3] - g[0::3]
        dy = d[n, 1::3] - g[1::3]
        e = (dx ** 2 + dy ** 2) / var / ((a_g + a_d[n]) / 2 + np.spacing(1)) / 2
        ious[n] = np.sum(np.exp(-e)) / len(e) if len(e) != 0 else 0.0
    return ious


def oks_nms(scores, kpts, areas, thr, limit=None):
    """scores (n,), kpts (n,3J), areas (n,) numpy float32 -> kept indices, score order.
    limit: stop after this many kept poses. The reference runs the greedy loop to the end and truncates afterwards
    (`keep[:nms_post]`, das_head.py:783-788); a kept pose is never un-kept by later rounds, so the first `limit` entries are
    the same either way — the early stop only keeps the oracle's cost at limit x n instead of kept x n pair evaluations
    (tens of thousands of candidates at 1080p with nms_pre <= 0)."""
    if len(scores) == 0:
        return np.zeros((0,), dtype=np.int64)
    order = np.argsort(-scores.astype(np.float64), kind='stable')
    keep = []
    while len(order) > 0 and (limit is None or len(keep) < limit):
        i = order[0]
        keep.append(i)
        ovr = oks_iou(kpts[i], kpts[order[1:]], areas[i], areas[order[1:]])
        order = order[np.where(ovr <= thr)[0] + 1]
    return np.array(keep, dtype=np.int64)


def soft_oks_nms(scores, kpts, areas, thr, max_dets=20):
    """pose_nms.py:128-194 (`_rescore` gaussian + `soft_oks_nms`): float32 arithmetic as numpy does with these arrays.
    The reference re-sorts the rescored scores every round with an unstable argsort; the oracle defines ties as
    "the candidate that came first in the initial (score desc, index asc) order", which coincides on tie-free input."""
    if len(scores) == 0:
        return np.zeros((0,), dtype=np.int64)
    order = np.argsort(-scores.astype(np.float64), kind='stable')
    sc = scores[order].astype(np.float32)
    pos = np.arange(len(order))      # position in the initial order (the tie rule)
    keep = []
    while len(order) > 0 and len(keep) < max_dets:
        i = order[0]
        ovr = oks_iou(kpts[i], kpts[order[1:]], areas[i], areas[order[1:]])
        order, pos = order[1:], pos[1:]
        sc = sc[1:] * np.exp(-ovr ** 2 / thr)
        assert sc.dtype == np.float32
        tmp = np.lexsort((pos, -sc.astype(np.float64)))
        order, pos, sc = order[tmp], pos[tmp], sc[tmp]
        keep.append(i)
    return np.array(keep, dtype=np.int64)


def decode_single(cls_scores, pose_preds, centernesses, points, scale_factor, J, test_cfg, return_index=False):
    """Per image: lists over levels of (1,h,w), (3+6J,h,w), (1,h,w) eval-mode head outputs."""
    nms_pre = test_cfg.get('nms_pre', -1)
    scale = pose_preds[0].new_tensor(scale_factor[:2])
    zs = torch.sqrt(scale.prod())
    C, P, S, K, I = [], [], [], [], []
    base = 0
    for cls, pp, ctr, pts in zip(cls_scores, pose_preds, centernesses, points):
        s = cls.permute(1, 2, 0).reshape(-1).sigmoid()
        c = ctr.permute(1, 2, 0).reshape(-1).sigmoid()
        pp = pp.permute(1, 2, 0).reshape(-1, 3 + 6 * J)
        idx = torch.arange(s.numel())
        if nms_pre > 0 and s.numel() > nms_pre:
            idx = torch.sort(s * c, descending=True, stable=True)[1][:nms_pre]
        s, c, pp, pts = s[idx], c[idx], pp[idx], pts[idx]
        center = torch.cat([(pts - pp[:, :2]) / scale, pp[:, 2:3] * zs], 1)
        root = torch.cat([pts, pp[:, 2:3] * zs], 1)
        joints = pp[:, 3:3 + 3 * J].reshape(-1, J, 3) + root[:, None]
        joints = torch.cat([joints[..., :2] / scale, joints[..., 2:]], -1)
        C.append(center)
        P.append(joints)
        S.append(s)
        K.append(c)
        I.append(idx + base)
        base += cls.shape[-2] * cls.shape[-1]
    C, P, S, K, I = torch.cat(C), torch.cat(P), torch.cat(S), torch.cat(K), torch.cat(I)
    nms_scores = torch.stack([S * K, torch.zeros_like(S)], 1)
    vis = torch.ones(P.shape[0], J, dtype=P.dtype)
    thr = test_cfg.get('score_thr', 0.)
    if thr > 0:
        v = nms_scores[:, 0] > thr
        nms_scores, P, C, vis, I = nms_scores[v], P[v], C[v], vis[v], I[v]
    nms_post = test_cfg.get('nms_post', -1)
    if nms_post > 0 and len(nms_scores) > 0:
        area = (P[..., :2].max(1)[0] - P[..., :2].min(1)[0]).prod(-1)
        kp = torch.cat([P[..., :2], vis[..., None]], -1).reshape(len(P), -1)
        if test_cfg.get('nms_type', 'hard') == 'hard':
            keep = oks_nms(nms_scores[:, 0].numpy(), kp.numpy(), area.numpy(), test_cfg.get('nms_thr', 0.9),
                           limit=test_cfg.get('nms_post', 100) if test_cfg.get('oracle_early_stop', False) else None)
            keep = torch.from_numpy(keep[:test_cfg.get('nms_post', 100)])
        else:
            keep = torch.from_numpy(soft_oks_nms(nms_scores[:, 0].numpy(), kp.numpy(), area.numpy(),
                                                 test_cfg.get('nms_thr', 0.9), max_dets=test_cfg.get('nms_post', 100)))
        nms_scores, P, C, vis, I = nms_scores[keep], P[keep], C[keep], vis[keep], I[keep]
    if return_index:
        return nms_scores, P, vis, C, I
    return nms_scores, P, vis, C


def get_poses(cls_scores, pose_preds, centernesses, img_metas, J, strides, test_cfg, return_index=False):
    sizes = [c.shape[-2:] for c in cls_scores]
    points = get_points(sizes, strides, pose_preds[0].dtype, pose_preds[0].device)
    out = []
    for b, meta in enumerate(img_metas):
        r = decode_single([c[b].detach() for c in cls_scores], [p[b].detach() for p in pose_preds],
                          [c[b].detach() for c in centernesses], points, meta['scale_factor'], J, test_cfg,
                          return_index)
        d = dict(poses=r[1], vis=r[2], centers=r[3], image_paths=[meta.get('filename')],
                 scores=r[0][..., 0].cpu().numpy().tolist())
        if return_index:
            d['index'] = r[4]
        out.append(d)
    return out
