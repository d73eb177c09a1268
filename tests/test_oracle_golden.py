"""Oracle (CPU restatement) vs the committed fixtures captured from the reference's own
source files (tests/golden/make_golden.py). Runs everywhere, no GPU, no /root/reference."""
import os

import numpy as np
import pytest
import torch

import cases
from oracle import backbone as ob
from oracle import decode as od
from oracle import head as oh
from oracle import loss as ol

TOL = dict(rtol=2e-4, atol=2e-5)


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + '.npz'))


def sd_of(z, seed, prefix=''):
    shapes = [[int(i) for i in row if i >= 0] for row in z[prefix + 'sd_shapes']]
    return cases.sd_from_manifest(z[prefix + 'sd_keys'], shapes, z[prefix + 'sd_dtypes'], seed)


@pytest.mark.parametrize('stages,train', [(1, False), (2, False), (2, True), (3, False), (4, True)])
def test_mspn2_forward_backward(golden_dir, stages, train):
    z = load(golden_dir, f'mspn_s{stages}_{"train" if train else "eval"}')
    sd = sd_of(z, 1)
    x = cases.randn(7, 2, 3, 64, 96).requires_grad_(True)
    outs = ob.mspn2_forward(sd, x, stages, (1, 1, 1, 1), train=train)
    for i, o in enumerate(outs):
        np.testing.assert_allclose(o.detach().numpy(), z[f'out{i}'], **TOL)
    sum((o * o).sum() for o in outs).backward()
    g = z['grad_x']
    np.testing.assert_allclose(x.grad.numpy(), g, rtol=2e-3, atol=2e-4 * np.abs(g).max())
    if train:
        np.testing.assert_allclose(sd['top.top.0.bn.running_mean'].numpy(), z['rm_top'], **TOL)
        np.testing.assert_allclose(
            sd[f'multi_stage_mspn.{stages - 1}.upsample.up4.in_skip.bn.running_var'].numpy(), z['rv_last'], **TOL)


def test_head_eval(golden_dir):
    z = load(golden_dir, 'head_eval')
    sd = sd_of(z, 3)
    with torch.no_grad():
        outs = oh.head_forward(sd, cases.head_feats(), cases.HEAD_CFG, '', train=False)
    for name, lst in zip(('cls', 'pose', 'ctr'), outs):
        for i, t in enumerate(lst):
            np.testing.assert_allclose(t.numpy(), z[f'{name}{i}'], **TOL)


def test_head_train_loss_and_grads(golden_dir):
    z = load(golden_dir, 'head_train')
    sd = {k: (v.requires_grad_(True) if v.is_floating_point() and not k.endswith('.mask') else v)
          for k, v in sd_of(z, 3).items()}
    feats = [f.requires_grad_(True) for f in cases.head_feats()]
    outs = oh.head_forward(sd, feats, cases.HEAD_CFG, '', train=True)
    for name, lst in zip(('cls', 'pose', 'ctr', 'ref'), outs):
        for i, t in enumerate(lst):
            np.testing.assert_allclose(t.detach().numpy(), z[f'{name}{i}'], **TOL)
    losses = ol.head_loss(sd, '', *outs, cases.head_gts(), cases.HEAD_CFG)
    for k, v in losses.items():
        np.testing.assert_allclose(v.item(), z[k], rtol=1e-4)
    sum(losses.values()).backward()

    def close(a, b):
        np.testing.assert_allclose(a, b, rtol=2e-3, atol=2e-5 * max(1.0, np.abs(b).max()))
    close(feats[0].grad.numpy(), z['grad_feat0'])
    close(feats[1].grad.numpy(), z['grad_feat1'])
    for k in z.files:
        if k.startswith('pgrad:'):
            close(sd[k[6:]].grad.numpy(), z[k])


def _mupots_metas():
    return [dict(scale_factor=np.array([1.25, 1.25, 1.25, 1.25], dtype=np.float32), filename='a'),
            dict(scale_factor=np.array([1., 1., 1., 1.], dtype=np.float32), filename='b')]


def test_head_mupots_topology(golden_dir):
    """exp_mupots.py head (J=21, root 14, depth_factor 1, two recursive-update layers, 4 levels): forward eval +
    decode, forward train + the four losses + feature gradients, against the reference fixtures."""
    c = cases.MUPOTS_CFG
    z = load(golden_dir, 'head_mupots_eval')
    feats = cases.head_feats(seed=61, sizes=cases.MUPOTS_SIZES)
    with torch.no_grad():
        outs = oh.head_forward(sd_of(z, 13), feats, c, '', train=False)
    for name, lst in zip(('cls', 'pose', 'ctr'), outs):
        for i, t in enumerate(lst):
            np.testing.assert_allclose(t.numpy(), z[f'{name}{i}'], **TOL)
    res = od.get_poses([o + 1.5 for o in outs[0]], outs[1], [o + 1.0 for o in outs[2]], _mupots_metas(),
                       cases.MUPOTS_J, c['strides'], cases.FULL_TEST_CFG)
    for b, r in enumerate(res):
        assert r['poses'].shape == z[f'dec_poses{b}'].shape
        np.testing.assert_allclose(np.array(r['scores'], dtype=np.float32), z[f'dec_scores{b}'], rtol=1e-5)
        np.testing.assert_allclose(r['poses'].numpy(), z[f'dec_poses{b}'], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(r['centers'].numpy(), z[f'dec_centers{b}'], rtol=1e-4, atol=1e-4)

    z = load(golden_dir, 'head_mupots_train')
    sd = {k: (v.requires_grad_(True) if v.is_floating_point() and not k.endswith('.mask') else v)
          for k, v in sd_of(z, 13).items()}
    fg = [f.requires_grad_(True) for f in feats]
    outs = oh.head_forward(sd, fg, c, '', train=True)
    for name, lst in zip(('cls', 'pose', 'ctr', 'ref'), outs):
        for i, t in enumerate(lst):
            np.testing.assert_allclose(t.detach().numpy(), z[f'{name}{i}'], **TOL)
    losses = ol.head_loss(sd, '', *outs, cases.mupots_gts(), c)
    for k, v in losses.items():
        np.testing.assert_allclose(v.item(), z[k], rtol=1e-4)
    sum(losses.values()).backward()
    for i, f in enumerate(fg):
        g = z[f'grad_feat{i}']
        np.testing.assert_allclose(f.grad.numpy(), g, rtol=2e-3, atol=2e-5 * max(1.0, np.abs(g).max()))


def test_points_and_targets(golden_dir):
    z = load(golden_dir, 'head_targets')
    c = cases.HEAD_CFG
    pts = ol.get_points(cases.HEAD_SIZES, c['strides'])
    g = cases.head_gts()
    lab, tgt, ctr = ol.get_targets(pts, c['strides'], c['regress_ranges'], g['gt_labels_3d'], g['gt_poses_3d'],
                                   g['centers2d'], g['depths'], c['num_joints'])
    for i in range(2):
        np.testing.assert_array_equal(pts[i].numpy(), z[f'points{i}'])
        np.testing.assert_array_equal(lab[i].numpy(), z[f'labels{i}'])
        np.testing.assert_allclose(tgt[i].numpy(), z[f'targets{i}'], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(ctr[i].numpy(), z[f'ctr{i}'], rtol=1e-5, atol=1e-7)
    assert (z['labels0'] == 0).sum() > 0  # fixture has positives
    assert (z['labels0'][384:] == 1).all()  # image 1 has no GT: all background


def _check_decode(z, res):
    for b, r in enumerate(res):
        assert r['poses'].shape == z[f'poses{b}'].shape
        np.testing.assert_allclose(r['poses'].numpy(), z[f'poses{b}'], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(r['centers'].numpy(), z[f'centers{b}'], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(np.array(r['scores'], dtype=np.float32), z[f'scores{b}'], rtol=1e-5)
        np.testing.assert_array_equal(r['vis'].numpy(), z[f'vis{b}'])
        assert len(set(r['index'].tolist())) == len(r['index'])


def test_decode_tiny(golden_dir):
    z = load(golden_dir, 'decode_tiny')
    ze = load(golden_dir, 'head_eval')
    cls = [torch.from_numpy(ze[f'cls{i}']) + 1.0 for i in range(2)]
    ctr = [torch.from_numpy(ze[f'ctr{i}']) + 1.0 for i in range(2)]
    pose = [torch.from_numpy(ze[f'pose{i}']) for i in range(2)]
    metas = [dict(scale_factor=np.array([1.3, 1.1, 1.3, 1.1], dtype=np.float32), filename='a'),
             dict(scale_factor=np.array([1., 1., 1., 1.], dtype=np.float32), filename='b')]
    res = od.get_poses(cls, pose, ctr, metas, cases.J, cases.HEAD_CFG['strides'], cases.TEST_CFG, return_index=True)
    _check_decode(z, res)


def test_decode_full_size(golden_dir):
    z = load(golden_dir, 'decode_full')
    cls, pose, ctr = cases.full_decode_inputs()
    metas = [dict(scale_factor=np.array([1.3, 1.3, 1.3, 1.3], dtype=np.float32), filename='a'),
             dict(scale_factor=np.array([1., 1., 1., 1.], dtype=np.float32), filename='b')]
    res = od.get_poses(cls, pose, ctr, metas, cases.FULL_J, cases.FULL_STRIDES, cases.FULL_TEST_CFG, return_index=True)
    _check_decode(z, res)


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_decode_soft_nms(golden_dir, tag):
    """nms_type='soft' (das_head.py:784-790, pose_nms.py:128-194) against the reference's own output."""
    z = load(golden_dir, 'decode_soft')
    cls, pose, ctr = cases.full_decode_inputs(**cases.SOFT_DECODE_INPUTS[tag])
    metas = [dict(scale_factor=np.array([1.3, 1.3, 1.3, 1.3], dtype=np.float32), filename='a'),
             dict(scale_factor=np.array([1., 1., 1., 1.], dtype=np.float32), filename='b')]
    res = od.get_poses(cls, pose, ctr, metas, cases.FULL_J, cases.FULL_STRIDES, cases.SOFT_DECODE_CFGS[tag])
    for b, r in enumerate(res):
        assert r['poses'].shape[0] == cases.SOFT_DECODE_CFGS[tag]['nms_post']
        np.testing.assert_array_equal(np.array(r['scores'], dtype=np.float32), z[f'{tag}_scores{b}'])
        np.testing.assert_allclose(r['poses'].numpy(), z[f'{tag}_poses{b}'], **TOL)
        np.testing.assert_allclose(r['centers'].numpy(), z[f'{tag}_centers{b}'], **TOL)


def test_decode_empty():
    cls, pose, ctr = cases.full_decode_inputs(bias=-20.0)
    metas = [dict(scale_factor=np.ones(4, dtype=np.float32), filename='a')] * 2
    res = od.get_poses(cls, pose, ctr, metas, cases.FULL_J, cases.FULL_STRIDES, cases.FULL_TEST_CFG)
    assert res[0]['poses'].shape == (0, cases.FULL_J, 3) and res[0]['scores'] == []


def test_offset_sample(golden_dir):
    z = load(golden_dir, 'offset_sample')
    B, Jn, heads, h, w = 2, 3, 4, 10, 14
    out = oh.offset_sample(cases.randn(31, B, Jn * 3, h, w) * 2, cases.randn(32, B, Jn * heads * 2, h, w) * 1.5,
                           cases.randn(33, B, Jn * 3, h, w), Jn, heads)
    np.testing.assert_allclose(out.numpy(), z['out'], **TOL)


def test_realnvp_and_rle(golden_dir):
    z = load(golden_dir, 'realnvp_rle')
    s3 = {'f.' + k: v for k, v in sd_of(z, 41, 'f3_').items()}
    s2 = {'f.' + k: v for k, v in sd_of(z, 42, 'f2_').items()}
    np.testing.assert_allclose(ol.realnvp_log_prob(s3, 'f', cases.randn(43, 37, 3)).numpy(), z['lp3'], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(ol.realnvp_log_prob(s2, 'f', cases.randn(44, 29, 2)).numpy(), z['lp2'], rtol=1e-4, atol=1e-4)
    sigma = cases.randn(48, 5, 6, 3).sigmoid() + 1e-9
    visw = (cases.randn(49, 5, 6, 1) > 0).float().expand(5, 6, 3)
    v = ol.rle_loss3d(cases.randn(45, 5, 6, 3), cases.randn(46, 5, 6, 3), sigma, cases.randn(47, 5, 6, 3), visw, 2.0)
    np.testing.assert_allclose(v.item(), z['rle'], rtol=1e-5)
    # Sum(vis) < 1 early-out returns Sum(vis) itself (residual_log_likelihood_loss.py:24-25)
    assert ol.rle_loss3d(torch.zeros(1, 2, 3), torch.zeros(1, 2, 3), torch.ones(1, 2, 3), torch.zeros(1, 2, 3),
                         torch.zeros(1, 2, 3), 2.0).item() == 0.0


def test_loss_no_positives_is_zero():
    g = cases.head_gts(counts=(0, 0))
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'head_train.npz'))
    sd = sd_of(z, 3)
    with torch.no_grad():
        outs = oh.head_forward(sd, cases.head_feats(), cases.HEAD_CFG, '', train=True)
        losses = ol.head_loss(sd, '', *outs, g, cases.HEAD_CFG)
    assert all(v.item() == 0.0 for v in losses.values())


def test_dcn_zero_offset_equals_masked_conv():
    """Published DCNv2 semantics: zero offsets + mask m == m * plain conv."""
    from oracle.nn_ops import modulated_deform_conv2d
    x, w, b = cases.randn(1, 2, 8, 9, 11), cases.randn(2, 6, 8, 3, 3), cases.randn(3, 6)
    off = torch.zeros(2, 18, 9, 11)
    mask = torch.full((2, 9, 9, 11), 0.5)
    y = modulated_deform_conv2d(x, off, mask, w, b)
    ref = 0.5 * torch.nn.functional.conv2d(x, w, None, 1, 1) + b[None, :, None, None]
    np.testing.assert_allclose(y.numpy(), ref.numpy(), rtol=1e-4, atol=1e-5)


def test_dcn_against_scalar_loops():
    """DCNv2 restatement vs a scalar-loop transcription of the published im2col rule."""
    from oracle.nn_ops import modulated_deform_conv2d
    B, C, O, H, W = 1, 2, 2, 5, 6
    x, w = cases.randn(4, B, C, H, W), cases.randn(5, O, C, 3, 3)
    off, mask = cases.randn(6, B, 18, H, W) * 1.5, cases.randn(7, B, 9, H, W).sigmoid()
    y = modulated_deform_conv2d(x, off, mask, w, None).numpy()
    xn, wn, on, mn = x.numpy(), w.numpy(), off.numpy(), mask.numpy()
    ref = np.zeros((B, O, H, W), np.float64)
    for oy in range(H):
        for ox in range(W):
            for k in range(9):
                i, j = divmod(k, 3)
                py, px = oy - 1 + i + on[0, 2 * k, oy, ox], ox - 1 + j + on[0, 2 * k + 1, oy, ox]
                if not (py > -1 and px > -1 and py < H and px < W):
                    continue
                y0, x0 = int(np.floor(py)), int(np.floor(px))
                ly, lx = py - y0, px - x0
                for c in range(C):
                    v = 0.0
                    for yy, xx, wt in ((y0, x0, (1 - ly) * (1 - lx)), (y0, x0 + 1, (1 - ly) * lx),
                                       (y0 + 1, x0, ly * (1 - lx)), (y0 + 1, x0 + 1, ly * lx)):
                        if 0 <= yy <= H - 1 and 0 <= xx <= W - 1:
                            v += wt * xn[0, c, yy, xx]
                    ref[0, :, oy, ox] += wn[:, c, i, j] * v * mn[0, k, oy, ox]
    np.testing.assert_allclose(y, ref, rtol=1e-4, atol=1e-5)


def test_oracle_nms_early_stop_is_the_truncated_full_run():
    """oracle/decode.py oks_nms(limit=k) — used by the 1080p decode tests, where the greedy loop run to the end would take
    minutes — returns exactly the first k entries of the full run (the reference's `keep[:nms_post]`, das_head.py:783-788)."""
    import numpy as np
    from oracle import decode as od
    rs = np.random.RandomState(4)
    n, J = 400, 15
    kp = rs.normal(0, 40, (n, J, 3)).astype(np.float32)
    kp[: n // 2] = kp[n // 2:] + rs.normal(0, 3, (n // 2, J, 3)).astype(np.float32)   # near-duplicates: suppression happens
    kp[..., 2] = 1
    sc = rs.rand(n).astype(np.float32)
    area = (kp[..., 0].max(1) - kp[..., 0].min(1)) * (kp[..., 1].max(1) - kp[..., 1].min(1))
    full = od.oks_nms(sc, kp.reshape(n, -1), area, 0.5)
    assert len(full) < n
    for k in (1, 7, 50, len(full), len(full) + 10):
        np.testing.assert_array_equal(od.oks_nms(sc, kp.reshape(n, -1), area, 0.5, limit=k), full[:k])
