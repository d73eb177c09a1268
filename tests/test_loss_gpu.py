"""Targets, losses and optimizer on the GPU vs the reference fixtures / the oracle / torch.optim.SGD."""
import os

import numpy as np
import pytest
import torch

import cases

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + '.npz'))


def sd_of(z, seed):
    shapes = [[int(i) for i in row if i >= 0] for row in z['sd_shapes']]
    return cases.sd_from_manifest(z['sd_keys'], shapes, z['sd_dtypes'], seed)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)


def test_assign_targets_vs_reference_fixture(golden_dir):
    """labels bit-exact, targets / centerness to f32 rounding (fixture from the reference's get_targets)."""
    from das_amd import ops, train_ops as T
    from das_amd.losses import _pack_gt
    z = load(golden_dir, 'head_targets')
    c = cases.HEAD_CFG
    g = cases.head_gts()
    B = 2
    geom = ops.Ragged(torch.empty(sum(B * h * w for h, w in cases.HEAD_SIZES), 1, device=DEV), B, cases.HEAD_SIZES)
    rows, start = _pack_gt(g['gt_poses_3d'], DEV)
    lab, tgt, ctr = T.assign_targets(geom, c['strides'], c['regress_ranges'], rows, start, c['num_joints'])
    for l in range(2):
        s, e = geom.starts[l], geom.starts[l + 1]
        np.testing.assert_array_equal(lab[s:e].cpu().numpy(), z[f'labels{l}'])
        np.testing.assert_allclose(tgt[s:e].cpu().numpy(), z[f'targets{l}'], rtol=1e-6, atol=1e-5)
        np.testing.assert_allclose(ctr[s:e].cpu().numpy(), z[f'ctr{l}'], rtol=1e-5, atol=1e-7)


def test_assign_targets_separate_centers_and_depths_vs_oracle():
    """centers2d / depths that differ from gt_poses_3d[:, :3] (das_head.py:570-589 takes root offsets, centre box
    and depth from them, joint offsets from gt_poses_3d)."""
    from das_amd import ops, train_ops as T
    from das_amd.losses import _pack_centers, _pack_gt
    from oracle import loss as ol
    J, B = 15, 2
    rs = np.random.RandomState(9)
    gts = [cases.make_gt(rs, n, J, 832, 512, spread=60.0) for n in (4, 3)]
    poses = [g[0] for g in gts]
    c2d = [g[1] + torch.from_numpy(rs.normal(0, 6, g[1].shape).astype(np.float32)) for g in gts]
    dep = [g[2] * 1.5 + 0.1 for g in gts]
    labels = [torch.zeros(len(p), dtype=torch.long) for p in poses]
    ranges = ((-1, 80), (80, 160), (160, 320), (320, 1e8))
    pts = ol.get_points(cases.FULL_SIZES, cases.FULL_STRIDES)
    rl, rt, rc = ol.get_targets(pts, cases.FULL_STRIDES, ranges, labels, poses, c2d, dep, J)
    geom = ops.Ragged(torch.empty(sum(B * h * w for h, w in cases.FULL_SIZES), 1, device=DEV), B, cases.FULL_SIZES)
    rows, start = _pack_gt(poses, DEV)
    lab, tgt, ctr = T.assign_targets(geom, cases.FULL_STRIDES, ranges, rows, start, J,
                                     centers=_pack_centers(c2d, dep, rows.shape[0], DEV))
    for l in range(4):
        s, e = geom.starts[l], geom.starts[l + 1]
        np.testing.assert_array_equal(lab[s:e].cpu().numpy(), rl[l].numpy())
        np.testing.assert_allclose(tgt[s:e].cpu().numpy(), rt[l].numpy(), rtol=1e-6, atol=1e-5)
        np.testing.assert_allclose(ctr[s:e].cpu().numpy(), rc[l].numpy(), rtol=1e-5, atol=1e-7)
    assert sum(int((x == 0).sum()) for x in rl) > 5


def test_loss_modules_follow_the_mmdet_protocol():
    """build_loss(cfg)(pred, target, weight=, avg_factor=) — das_head.py:40-53 configs, call sites :341-344,
    :375-379, :470-471 — against the published mmdet formulas spelled in torch."""
    import torch.nn.functional as F
    from das_amd.registry import build_loss
    rs = np.random.RandomState(4)
    N = 1000
    x = torch.from_numpy(rs.standard_normal((N, 1)).astype(np.float32)).to(DEV).requires_grad_(True)
    lab = torch.from_numpy((rs.uniform(0, 1, N) > 0.1).astype(np.int64)).to(DEV)      # 0 = person, 1 = background
    w = torch.from_numpy(rs.uniform(0.5, 2, N).astype(np.float32)).to(DEV)
    fl = build_loss(dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.5))
    t = (lab == 0).float()[:, None]
    xr = x.detach().clone().requires_grad_(True)
    p = xr.sigmoid()
    ref_el = F.binary_cross_entropy_with_logits(xr, t, reduction='none') * (0.25 * t + 0.75 * (1 - t)) * \
        ((1 - p) * t + p * (1 - t)) ** 2
    for kw, ref in ((dict(avg_factor=37.0), ref_el.sum() / 37.0), (dict(), ref_el.mean()),
                    (dict(weight=w, avg_factor=5.0), (ref_el[:, 0] * w).sum() / 5.0),
                    (dict(reduction_override='sum'), ref_el.sum())):
        out = fl(x, lab, **kw)
        np.testing.assert_allclose(out.item(), 1.5 * ref.item(), rtol=2e-5)
    x.grad = None
    fl(x, lab, weight=w, avg_factor=5.0).backward()
    (1.5 * (ref_el[:, 0] * w).sum() / 5.0).backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), xr.grad.cpu().numpy(), rtol=1e-4, atol=1e-7)

    a = torch.from_numpy(rs.standard_normal(257).astype(np.float32)).to(DEV).requires_grad_(True)
    b = torch.from_numpy(rs.standard_normal(257).astype(np.float32)).to(DEV)
    wv = torch.from_numpy(rs.uniform(0.5, 2, 257).astype(np.float32)).to(DEV)
    sl = build_loss(dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0))
    ref_el = F.smooth_l1_loss(a.detach(), b, beta=1.0 / 9.0, reduction='none')
    np.testing.assert_allclose(sl(a, b, weight=wv, avg_factor=11.0).item(), ((ref_el * wv).sum() / 11.0).item(), rtol=2e-5)
    np.testing.assert_allclose(sl(a, b).item(), ref_el.mean().item(), rtol=2e-5)
    ce = build_loss(dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0))
    tgt = b.sigmoid()
    np.testing.assert_allclose(ce(a, tgt).item(), F.binary_cross_entropy_with_logits(a.detach(), tgt).item(), rtol=2e-5)
    ce(a, tgt).backward()
    ar = a.detach().clone().requires_grad_(True)
    F.binary_cross_entropy_with_logits(ar, tgt).backward()
    np.testing.assert_allclose(a.grad.cpu().numpy(), ar.grad.cpu().numpy(), rtol=1e-4, atol=1e-7)
    # RLELoss3D.forward: residual_log_likelihood_loss.py:21-37 (realnvp_rle.npz pins the oracle; here vs the oracle)
    from oracle import loss as ol
    rle = build_loss(dict(type='RLELoss3D', residual=True, loss_weight=1.0))
    nf, pred, gt = cases.randn(45, 5, 6, 3), cases.randn(46, 5, 6, 3), cases.randn(47, 5, 6, 3)
    sigma = cases.randn(48, 5, 6, 3).sigmoid() + 1e-9
    visw = (cases.randn(49, 5, 6, 1) > 0).float()
    got = rle(nf.to(DEV), pred.to(DEV), sigma.to(DEV), gt.to(DEV), visw.to(DEV), weight=2.0)
    np.testing.assert_allclose(got.item(), ol.rle_loss3d(nf, pred, sigma, gt, visw.expand(5, 6, 3), 2.0).item(), rtol=1e-5)


def test_assign_targets_full_size_vs_oracle():
    from das_amd import ops, train_ops as T
    from das_amd.losses import _pack_gt
    from oracle import loss as ol
    J, B = 15, 3
    rs = np.random.RandomState(3)
    gts = [cases.make_gt(rs, n, J, 832, 512, spread=60.0) for n in (6, 0, 2)]
    poses, c2d, dep = [g[0] for g in gts], [g[1] for g in gts], [g[2] for g in gts]
    labels = [torch.zeros(len(p), dtype=torch.long) for p in poses]
    ranges = ((-1, 80), (80, 160), (160, 320), (320, 1e8))
    pts = ol.get_points(cases.FULL_SIZES, cases.FULL_STRIDES)
    rl, rt, rc = ol.get_targets(pts, cases.FULL_STRIDES, ranges, labels, poses, c2d, dep, J)
    geom = ops.Ragged(torch.empty(sum(B * h * w for h, w in cases.FULL_SIZES), 1, device=DEV), B, cases.FULL_SIZES)
    rows, start = _pack_gt(poses, DEV)
    lab, tgt, ctr = T.assign_targets(geom, cases.FULL_STRIDES, ranges, rows, start, J)
    npos = 0
    for l in range(4):
        s, e = geom.starts[l], geom.starts[l + 1]
        np.testing.assert_array_equal(lab[s:e].cpu().numpy(), rl[l].numpy())
        np.testing.assert_allclose(tgt[s:e].cpu().numpy(), rt[l].numpy(), rtol=1e-6, atol=1e-5)
        np.testing.assert_allclose(ctr[s:e].cpu().numpy(), rc[l].numpy(), rtol=1e-5, atol=1e-7)
        npos += int((rl[l] == 0).sum())
    assert npos > 10


def test_fused_ground_truth_half_equals_the_aten_formulation():
    """losses.das_head_targets through das_assign_targets' counts + das_positive_rows (two launches) against the mask / sum /
    gather / concatenate / cumsum formulation it replaces (das_head.py:385-409), at full size with 2-D-only persons mixed in
    and more positives than one chunk of the rank kernel: counts, rows, kinds and ranks identical, the float rows bit-equal."""
    from das_amd import losses
    J, B = cases.HEAD_CFG['num_joints'], 12
    rs = np.random.RandomState(11)
    poses = []
    for b, n in enumerate((18, 0, 6, 15, 21, 9, 3, 18, 12, 24, 6, 15)):
        p = cases.make_gt(rs, n, J, 832, 512, spread=70.0)[0]
        if n:
            flat = rs.uniform(size=n) < 0.35                    # persons without depth annotation: every dz = 0
            p[torch.from_numpy(flat), 5:3 + 3 * J:3] = 0.0
        poses.append(p)
    head = build_head()
    head.strides, head.regress_ranges = cases.FULL_STRIDES, ((-1, 80), (80, 160), (160, 320), (320, 1e8))
    head.center_sample_radius = 2.5
    out = {}
    for fused in (False, True):
        losses.FUSED_TARGETS = fused
        try:
            out[fused] = losses.das_head_targets(head, B, cases.FULL_SIZES, torch.device(DEV), [p.to(DEV) for p in poses])
        finally:
            losses.FUSED_TARGETS = True
    a, b = out[False], out[True]
    assert a['npos'] == b['npos'] > 1024 and a['n3d'] == b['n3d'] and 0 < b['n3d'] < b['npos']
    assert abs(a['nvis_host'] - b['nvis_host']) <= 1e-5 * abs(a['nvis_host'])
    for k in ('labels', 'pos', 'is2d', 'slot', 'real', 'vis', 'depth_t', 'ctr_t'):
        assert a[k].shape == b[k].shape and a[k].dtype == b[k].dtype, k
        assert torch.equal(a[k], b[k]), k
    torch.testing.assert_close(b['nvis'], a['nvis'], rtol=1e-6, atol=0)
    assert 0 < int(b['is2d'].sum()) < b['npos']


def test_dense_loss_kernels_vs_torch():
    import torch.nn.functional as F
    from das_amd import train_ops as T
    from oracle.loss import sigmoid_focal_loss, smooth_l1
    n = 5000
    x = (cases.randn(1, n, 1) * 3).to(DEV).requires_grad_(True)
    lab = (torch.from_numpy(np.random.RandomState(0).uniform(size=n)) > 0.02).to(torch.int32)
    (T.FocalLossSumFn.apply(x, lab.to(DEV), 2.0, 0.25) * 0.5).backward()
    xr = x.detach().cpu().requires_grad_(True)
    (sigmoid_focal_loss(xr, lab.long(), 1).sum() * 0.5).backward()
    assert rel(x.grad.cpu().numpy(), xr.grad.numpy()) < 1e-5
    p, t = cases.randn(2, 700).to(DEV).requires_grad_(True), cases.randn(3, 700).to(DEV)
    l = T.SmoothL1SumFn.apply(p, t, 1.0 / 9)
    l.backward()
    pr = p.detach().cpu().requires_grad_(True)
    lr = smooth_l1(pr, t.cpu(), 1.0 / 9).sum()
    lr.backward()
    assert rel(l.item(), lr.item()) < 1e-5 and rel(p.grad.cpu().numpy(), pr.grad.numpy()) < 1e-6
    tt = torch.sigmoid(cases.randn(4, 700)).to(DEV)
    p.grad = None
    l = T.BCELogitsSumFn.apply(p, tt)
    l.backward()
    pr.grad = None
    lr = F.binary_cross_entropy_with_logits(pr, tt.cpu(), reduction='sum')
    lr.backward()
    assert rel(l.item(), lr.item()) < 1e-5 and rel(p.grad.cpu().numpy(), pr.grad.numpy()) < 1e-5


def build_head():
    import das_amd
    c = cases.HEAD_CFG
    J, C = c['num_joints'], c['feat_channels']
    return das_amd.DASHead(
        num_classes=1, in_channels=C, feat_channels=C, stacked_convs=2, strides=c['strides'],
        regress_ranges=c['regress_ranges'], num_joints=J, depth_factor=c['depth_factor'], z_norm=c['z_norm'],
        root_idx=c['root_idx'], cls_branch=(C,), reg_branch=((C,),) * 4, centerness_branch=(64,),
        centerness_on_reg=True, conv_bias=True, dcn_on_last_conv=True,
        recursive_update=dict(prev_loss=True, num_heads=c['num_heads'], in_channels=C, feat_channels=C,
                              num_layers=c['num_layers'], dim=3, num_joints=J),
        train_cfg=dict(code_weight=c['code_weight']), test_cfg=cases.TEST_CFG, compute_dtype=torch.float32)


def test_head_losses_vs_reference_fixture(golden_dir):
    """forward_train on the fixture's weights / features / GT: the four loss values equal the reference's
    (1e-4) and the gradients wrt the input features match within the f32 conditioning band."""
    z = load(golden_dir, 'head_train')
    head = build_head()
    head.load_state_dict(sd_of(z, 3))
    head.to(DEV).train()
    fin = [t.permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True) for t in cases.head_feats()]
    g = cases.head_gts()
    gt_poses = [p.to(DEV) for p in g['gt_poses_3d']]
    losses = head.forward_train([t.permute(0, 3, 1, 2) for t in fin], [{}, {}], None, None, gt_poses, None, None, None)
    assert set(losses) == {'loss_cls', 'loss_depth', 'loss_pose', 'loss_centerness'}
    for k, v in losses.items():
        assert rel(v.item(), float(z[k])) < 2e-4, (k, v.item(), float(z[k]))
    sum(losses.values()).backward()
    # the reference's own f32 gradients are the yardstick here (fixture); band, not equality (see test_train_gpu)
    for i, t in enumerate(fin):
        assert rel(t.grad.permute(0, 3, 1, 2).cpu().numpy(), z[f'grad_feat{i}']) < 5e-2
    pg = dict(head.named_parameters())
    for k in z.files:
        if k.startswith('pgrad:'):
            assert rel(pg[k[6:]].grad.cpu().numpy(), z[k]) < 5e-2, k


def test_head_loss_no_positives_is_zero(golden_dir):
    z = load(golden_dir, 'head_train')
    head = build_head()
    head.load_state_dict(sd_of(z, 3))
    head.to(DEV).train()
    fin = [t.to(DEV).requires_grad_(True) for t in cases.head_feats()]
    empty = [torch.zeros(0, 3 + 4 * cases.J, device=DEV)] * 2
    losses = head.forward_train(fin, [{}, {}], None, None, empty, None, None, None)
    assert all(v.item() == 0.0 for v in losses.values())
    sum(losses.values()).backward()


def test_flat_sgd_matches_torch_sgd_with_clip_and_paramwise():
    import das_amd
    from das_amd.optim import FlatSGD
    torch.manual_seed(0)
    net = das_amd.FPN([16] * 4, 24, 4, start_level=1, add_extra_convs='on_output', norm_cfg=None).to(DEV)
    ref = {n: p.detach().clone() for n, p in net.named_parameters()}
    opt = FlatSGD(net, lr=0.1, momentum=0.9, weight_decay=1e-2, bias_lr_mult=2.0, bias_decay_mult=0.0, max_grad_norm=0.5)
    rp = {n: v.clone().requires_grad_(True) for n, v in ref.items()}
    topt = torch.optim.SGD([dict(params=[v for n, v in rp.items() if not n.endswith('bias')], lr=0.1, weight_decay=1e-2),
                            dict(params=[v for n, v in rp.items() if n.endswith('bias')], lr=0.2, weight_decay=0.0)],
                           lr=0.1, momentum=0.9)
    for it in range(3):
        opt.zero_grad()
        topt.zero_grad()
        for n, p in net.named_parameters():
            gnp = cases.randn(100 + it, *p.shape).to(DEV) * (1 + it)
            p.grad.add_(gnp)
            rp[n].grad = gnp.clone()
        torch.nn.utils.clip_grad_norm_(list(rp.values()), 0.5)
        topt.step()
        opt.step(0.1)
        for n, p in net.named_parameters():
            assert rel(p.detach().cpu().numpy(), rp[n].detach().cpu().numpy()) < 1e-5, (it, n)
    assert net.lateral_convs[0].conv.weight.data_ptr() >= opt.flat_p.data_ptr()
