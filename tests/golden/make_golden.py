"""Generate tests/golden/*.npz from the REFERENCE's own source files.

Run in the authoring container only (needs /root/reference):
    python tests/golden/make_golden.py
The reference files are imported by path through tests/refstub.py; inputs/weights come
from tests/golden/cases.py seeds; only the reference's outputs are stored.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, 'tests'), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import cases  # noqa: E402
import refstub  # noqa: E402


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        out[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print('wrote', name, {k: getattr(v, 'shape', None) for k, v in out.items() if not k.startswith('sd_')})


def build_head(R, c=None, test_cfg=None):
    c = c or cases.HEAD_CFG
    J, C = c['num_joints'], c['feat_channels']
    return R.DASHead(
        num_classes=1, in_channels=C, feat_channels=C, stacked_convs=2, strides=c['strides'],
        regress_ranges=c['regress_ranges'], num_joints=J, depth_factor=c['depth_factor'], z_norm=c['z_norm'],
        root_idx=c['root_idx'], cls_branch=(C,), reg_branch=((C,),) * 4, centerness_branch=(64,),
        centerness_on_reg=True, conv_bias=True, dcn_on_last_conv=True,
        recursive_update=dict(prev_loss=True, num_heads=c['num_heads'], in_channels=C, feat_channels=C,
                              num_layers=c['num_layers'], dim=3, num_joints=J),
        train_cfg=dict(code_weight=c['code_weight']), test_cfg=test_cfg or cases.TEST_CFG)


def mupots_head_fixtures(R):
    """exp_mupots.py head topology (J=21, root 14, depth_factor 1, two recursive-update layers, four levels):
    forward train / eval, the four losses with gradients, decode (das_head.py:176-267,281-486,653-796)."""
    head = build_head(R, cases.MUPOTS_CFG, cases.FULL_TEST_CFG)
    sd = cases.det_fill(head.state_dict(), 13)
    man = cases.manifest(sd)
    feats = cases.head_feats(seed=61, sizes=cases.MUPOTS_SIZES)
    gts = cases.mupots_gts()
    gt_boxes = [torch.zeros(len(g), 4) for g in gts['gt_poses_3d']]
    head.train(True)
    fg = [f.clone().requires_grad_(True) for f in feats]
    outs = head(fg)
    losses = head.loss(*outs, gt_boxes, gts['gt_labels_3d'], gts['gt_poses_3d'], gts['gt_labels_3d'],
                       gts['centers2d'], gts['depths'], [{}, {}])
    sum(losses.values()).backward()
    arrs = dict(man)
    for name, lst in zip(('cls', 'pose', 'ctr', 'ref'), outs):
        for i, t in enumerate(lst):
            arrs[f'{name}{i}'] = t
    for k, v in losses.items():
        arrs[k] = v
    for i, f in enumerate(fg):
        arrs[f'grad_feat{i}'] = f.grad
    save('head_mupots_train', **arrs)
    head.train(False)
    with torch.no_grad():
        outs = head(feats)
        arrs = dict(man)
        for name, lst in zip(('cls', 'pose', 'ctr'), outs):
            for i, t in enumerate(lst):
                arrs[f'{name}{i}'] = t
        metas = [dict(scale_factor=np.array([1.25, 1.25, 1.25, 1.25], dtype=np.float32), filename='a'),
                 dict(scale_factor=np.array([1., 1., 1., 1.], dtype=np.float32), filename='b')]
        res = head.get_poses([o + 1.5 for o in outs[0]], [o.clone() for o in outs[1]], [o + 1.0 for o in outs[2]], metas)
        for b, r in enumerate(res):
            arrs[f'dec_poses{b}'], arrs[f'dec_centers{b}'] = r['poses'], r['centers']
            arrs[f'dec_scores{b}'] = np.array(r['scores'], dtype=np.float32)
        save('head_mupots_eval', **arrs)


def main():
    R = refstub.load()
    torch.manual_seed(0)

    # ---- MSPN2 (tiny widths in the upsample path; bottleneck widths are fixed by the class)
    # (3 stages = exp_mupots.py:17; 4 stages = BASELINE configs[2]/[3])
    for stages, train in ((1, False), (2, False), (2, True), (3, False), (4, True)):
        m = R.MSPN2(unit_channels=16, num_stages=stages, num_blocks=[1, 1, 1, 1], norm_cfg=dict(type='BN'))
        sd = cases.det_fill(m.state_dict(), 1)
        man = cases.manifest(sd)
        m.train(train)
        x = cases.randn(7, 2, 3, 64, 96).requires_grad_(True)
        outs = m(x)
        sum((o * o).sum() for o in outs).backward()
        extra = {}
        if train:
            extra = dict(rm_top=m.state_dict()['top.top.0.bn.running_mean'],
                         rv_last=m.state_dict()[f'multi_stage_mspn.{stages - 1}.upsample.up4.in_skip.bn.running_var'])
        save(f'mspn_s{stages}_{"train" if train else "eval"}', **man, **{f'out{i}': o for i, o in enumerate(outs)},
             grad_x=x.grad, **extra)

    # ---- DASHead forward (train / eval), loss (+grads), targets, decode
    head = build_head(R)
    sd = cases.det_fill(head.state_dict(), 3)
    man = cases.manifest(sd)
    feats = cases.head_feats()
    gts = cases.head_gts()
    gt_boxes = [torch.zeros(len(g), 4) for g in gts['gt_poses_3d']]
    head.train(True)
    fg = [f.clone().requires_grad_(True) for f in feats]
    outs = head(fg)
    losses = head.loss(*outs, gt_boxes, gts['gt_labels_3d'], gts['gt_poses_3d'], gts['gt_labels_3d'],
                       gts['centers2d'], gts['depths'], [{}, {}])
    sum(losses.values()).backward()
    arrs = dict(man)
    for name, lst in zip(('cls', 'pose', 'ctr', 'ref'), outs):
        for i, t in enumerate(lst):
            arrs[f'{name}{i}'] = t
    for k, v in losses.items():
        arrs[k] = v
    arrs['grad_feat0'], arrs['grad_feat1'] = fg[0].grad, fg[1].grad
    pg = dict(head.named_parameters())
    for k in ('cls_convs.1.conv.weight', 'pose_convs.1.conv.conv_offset.weight', 'scales.0.2.scale',
              'recursive_update_branch.layer_1.next_level_offset.sampling_offset.weight', 'flow3d.s.0.0.weight'):
        arrs['pgrad:' + k] = pg[k].grad
    save('head_train', **arrs)

    pts = head.get_points(cases.HEAD_SIZES, torch.float32, 'cpu')
    lab, tgt, ctr = head.get_targets(pts, gt_boxes, gts['gt_labels_3d'], gts['gt_poses_3d'], gts['gt_labels_3d'],
                                     gts['centers2d'], gts['depths'])
    save('head_targets', points0=pts[0], points1=pts[1], labels0=lab[0], labels1=lab[1], targets0=tgt[0],
         targets1=tgt[1], ctr0=ctr[0], ctr1=ctr[1])

    head.train(False)
    with torch.no_grad():
        outs = head(feats)
        arrs = dict(man)
        for name, lst in zip(('cls', 'pose', 'ctr'), outs):
            for i, t in enumerate(lst):
                arrs[f'{name}{i}'] = t
        save('head_eval', **arrs)

        # decode at tiny sizes on shifted logits so that many candidates pass the threshold
        metas = [dict(scale_factor=np.array([1.3, 1.1, 1.3, 1.1], dtype=np.float32), filename='a'),
                 dict(scale_factor=np.array([1., 1., 1., 1.], dtype=np.float32), filename='b')]
        cls = [o + 1.0 for o in outs[0]]
        ctrs = [o + 1.0 for o in outs[2]]
        res = head.get_poses([o.clone() for o in cls], [o.clone() for o in outs[1]], [o.clone() for o in ctrs], metas)
        arrs = {}
        for b, r in enumerate(res):
            arrs[f'poses{b}'], arrs[f'centers{b}'], arrs[f'vis{b}'] = r['poses'], r['centers'], r['vis']
            arrs[f'scores{b}'] = np.array(r['scores'], dtype=np.float32)
        save('decode_tiny', **arrs)

    mupots_head_fixtures(R)

    # ---- decode at the full 512x832 level sizes, J=15 (topk(1000) on levels 0 and 1)
    full = build_full_head(R)
    cls, pose, ctr = cases.full_decode_inputs()
    metas = [dict(scale_factor=np.array([1.3, 1.3, 1.3, 1.3], dtype=np.float32), filename='a'),
             dict(scale_factor=np.array([1., 1., 1., 1.], dtype=np.float32), filename='b')]
    full.train(False)
    with torch.no_grad():
        res = full.get_poses([o.clone() for o in cls], [o.clone() for o in pose], [o.clone() for o in ctr], metas)
    arrs = {}
    for b, r in enumerate(res):
        arrs[f'poses{b}'], arrs[f'centers{b}'], arrs[f'vis{b}'] = r['poses'], r['centers'], r['vis']
        arrs[f'scores{b}'] = np.array(r['scores'], dtype=np.float32)
    save('decode_full', **arrs)

    # ---- offset_sample alone
    B, Jn, heads, h, w = 2, 3, 4, 10, 14
    uvd = cases.randn(31, B, Jn * 3, h, w) * 2
    so = cases.randn(32, B, Jn * heads * 2, h, w) * 1.5
    conf = cases.randn(33, B, Jn * 3, h, w)
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing='ij')
    new, _ = R.offset_sample(uvd, so, conf, (B, Jn, heads, 3), torch.stack((xs, ys), 0) + 0.5)
    save('offset_sample', out=new.reshape(B, Jn * 3, h, w))

    # ---- RealNVP log_prob + RLE loss
    f3, f2 = R.RealNVP(), R.RealNVP2D()
    cases.det_fill(f3.state_dict(), 41)
    cases.det_fill(f2.state_dict(), 42)
    x3, x2 = cases.randn(43, 37, 3), cases.randn(44, 29, 2)
    with torch.no_grad():
        lp3, lp2 = f3.log_prob(x3), f2.log_prob(x2)
    rle = R.RLELoss3D(residual=True)
    nf = cases.randn(45, 5, 6, 3)
    pred, gt = cases.randn(46, 5, 6, 3), cases.randn(47, 5, 6, 3)
    sigma = cases.randn(48, 5, 6, 3).sigmoid() + 1e-9
    visw = (cases.randn(49, 5, 6, 1) > 0).float().expand(5, 6, 3)
    man3, man2 = cases.manifest(f3.state_dict()), cases.manifest(f2.state_dict())
    save('realnvp_rle', lp3=lp3, lp2=lp2, rle=rle(nf, pred, sigma, gt, visw, weight=2.0),
         **{'f3_' + k: v for k, v in man3.items()}, **{'f2_' + k: v for k, v in man2.items()})


def build_full_head(R):
    J = cases.FULL_J
    return R.DASHead(
        num_classes=1, in_channels=32, feat_channels=32, stacked_convs=2, strides=cases.FULL_STRIDES,
        regress_ranges=((-1, 80), (80, 160), (160, 320), (320, 1e8)), num_joints=J, depth_factor=20, z_norm=50,
        root_idx=2, cls_branch=(32,), reg_branch=((32,),) * 4, centerness_branch=(64,), centerness_on_reg=True,
        conv_bias=True, dcn_on_last_conv=True,
        recursive_update=dict(prev_loss=True, num_heads=4, in_channels=32, feat_channels=32, num_layers=1, dim=3,
                              num_joints=J),
        train_cfg=dict(code_weight=[1.0, 1.0, 1] + [2] * J * 6), test_cfg=cases.FULL_TEST_CFG)


if __name__ == '__main__':
    main()
