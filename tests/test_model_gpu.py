"""Module-level parity on the GPU: das_amd modules (HIP path) vs the committed reference
fixtures and vs the CPU oracle, with weights rebuilt from the fixture manifests."""
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


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)


def nchw_np(t):
    return t.float().contiguous().cpu().numpy()


@pytest.mark.parametrize('stages,train', [(1, False), (2, False), (2, True)])
def test_mspn2_f32_vs_reference_fixture(golden_dir, stages, train):
    import das_amd
    z = load(golden_dir, f'mspn_s{stages}_{"train" if train else "eval"}')
    m = das_amd.MSPN2(unit_channels=16, num_stages=stages, num_blocks=[1, 1, 1, 1], compute_dtype='f32')
    missing = m.load_state_dict(sd_of(z, 1), strict=True)
    m.to(DEV).train(train)
    with torch.no_grad():
        outs = m(cases.randn(7, 2, 3, 64, 96).to(DEV))
    assert [tuple(o.shape) for o in outs] == [(2, 16, 16, 24), (2, 16, 8, 12), (2, 16, 4, 6), (2, 16, 2, 3)]
    for i, o in enumerate(outs):
        assert rel_err(nchw_np(o), z[f'out{i}']) < 1e-4, i
    if train:
        sd = m.state_dict()
        np.testing.assert_allclose(sd['top.top.0.bn.running_mean'].cpu().numpy(), z['rm_top'], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(sd[f'multi_stage_mspn.{stages - 1}.upsample.up4.in_skip.bn.running_var'].cpu().numpy(),
                                   z['rv_last'], rtol=1e-4, atol=1e-6)
        assert int(sd['top.top.0.bn.num_batches_tracked']) == 1


def test_mspn2_bf16_close_to_oracle(golden_dir):
    import das_amd
    from oracle import backbone as ob
    z = load(golden_dir, 'mspn_s2_eval')
    sd = sd_of(z, 1)
    m = das_amd.MSPN2(unit_channels=16, num_stages=2, num_blocks=[1, 1, 1, 1], compute_dtype='bf16')
    m.load_state_dict(sd)
    m.to(DEV).eval()
    x = cases.randn(7, 2, 3, 64, 96)
    with torch.no_grad():
        outs = m(x.to(DEV))
        ref = ob.mspn2_forward(sd, x, 2, (1, 1, 1, 1))
    for o, r in zip(outs, ref):
        assert o.dtype == torch.bfloat16
        # bf16 storage between ~30 layers: a few 1e-2 of the map's range
        assert rel_err(nchw_np(o), r.numpy()) < 6e-2


def test_fpn_f32_vs_oracle():
    import das_amd
    from oracle import backbone as ob
    f = das_amd.FPN([16] * 4, 24, 4, start_level=1, add_extra_convs='on_output', relu_before_extra_convs=True,
                    norm_cfg=dict(type='BN'))
    cases.det_fill(f.state_dict(), 2)
    sd = {k: v.clone() for k, v in f.state_dict().items()}
    f.to(DEV).eval()
    feats = [cases.randn(i, 2, 16, 32 >> i, 48 >> i) for i in range(4)]
    with torch.no_grad():
        outs = f([t.to(DEV) for t in feats])
        ref = ob.fpn_forward(sd, feats)
    assert [tuple(o.shape[-2:]) for o in outs] == [(16, 24), (8, 12), (4, 6), (2, 3)]
    for o, r in zip(outs, ref):
        assert rel_err(nchw_np(o), r.numpy()) < 1e-4


def build_head(test_cfg=None):
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
        train_cfg=dict(code_weight=c['code_weight']), test_cfg=test_cfg or cases.TEST_CFG, compute_dtype=torch.float32)


@pytest.mark.parametrize('train', [False, True])
def test_dashead_f32_vs_reference_fixture(golden_dir, train):
    z = load(golden_dir, 'head_train' if train else 'head_eval')
    head = build_head()
    head.load_state_dict(sd_of(z, 3), strict=True)
    head.to(DEV).train(train)
    with torch.no_grad():
        outs = head([f.to(DEV) for f in cases.head_feats()])
    names = ('cls', 'pose', 'ctr', 'ref') if train else ('cls', 'pose', 'ctr')
    assert len(outs) == len(names)
    for name, lst in zip(names, outs):
        for i, t in enumerate(lst):
            ref = z[f'{name}{i}']
            assert tuple(t.shape) == ref.shape
            assert rel_err(nchw_np(t), ref) < 2e-4, (name, i)


def test_dashead_decode_vs_reference_fixture(golden_dir):
    """get_poses on the reference's own eval outputs: kept poses/scores/order equal the reference's."""
    z, ze = load(golden_dir, 'decode_tiny'), load(golden_dir, 'head_eval')
    head = build_head().to(DEV).eval()
    cls = [torch.from_numpy(ze[f'cls{i}']).to(DEV) + 1.0 for i in range(2)]
    ctr = [torch.from_numpy(ze[f'ctr{i}']).to(DEV) + 1.0 for i in range(2)]
    pose = [torch.from_numpy(ze[f'pose{i}']).to(DEV) for i in range(2)]
    metas = [dict(scale_factor=np.array([1.3, 1.1, 1.3, 1.1], dtype=np.float32), filename='a'),
             dict(scale_factor=np.array([1., 1., 1., 1.], dtype=np.float32), filename='b')]
    res = head.get_poses(cls, pose, ctr, metas)
    for b, r in enumerate(res):
        assert r['poses'].shape == z[f'poses{b}'].shape
        np.testing.assert_allclose(r['poses'].cpu().numpy(), z[f'poses{b}'], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(r['centers'].cpu().numpy(), z[f'centers{b}'], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(np.array(r['scores'], dtype=np.float32), z[f'scores{b}'], rtol=1e-5)
        np.testing.assert_array_equal(r['vis'].cpu().numpy(), z[f'vis{b}'])
        assert r['image_paths'] == [metas[b]['filename']]


def test_dashead_decode_soft_nms_cfg(golden_dir):
    """`test_cfg.nms_type='soft'` through the head's own `get_poses` (das_head.py:784-790) against the oracle."""
    from oracle import decode as od
    ze = load(golden_dir, 'head_eval')
    head = build_head().to(DEV).eval()
    cfg = dict(head.test_cfg, nms_type='soft', nms_thr=0.3, nms_post=12)
    cls = [torch.from_numpy(ze[f'cls{i}']) + 1.0 for i in range(2)]
    ctr = [torch.from_numpy(ze[f'ctr{i}']) + 1.0 for i in range(2)]
    pose = [torch.from_numpy(ze[f'pose{i}']) for i in range(2)]
    metas = [dict(scale_factor=np.array([1.3, 1.1, 1.3, 1.1], dtype=np.float32), filename='a'),
             dict(scale_factor=np.array([1., 1., 1., 1.], dtype=np.float32), filename='b')]
    ref = od.get_poses(cls, pose, ctr, metas, head.num_joints, head.strides, cfg)
    res = head.get_poses([t.to(DEV) for t in cls], [t.to(DEV) for t in pose], [t.to(DEV) for t in ctr], metas, cfg=cfg)
    hard = head.get_poses([t.to(DEV) for t in cls], [t.to(DEV) for t in pose], [t.to(DEV) for t in ctr], metas,
                          cfg=dict(cfg, nms_type='hard'))
    for r, m, h in zip(ref, res, hard):
        assert m['poses'].shape == r['poses'].shape and m['poses'].shape[0] > 0
        np.testing.assert_allclose(np.array(m['scores'], dtype=np.float32), np.array(r['scores'], dtype=np.float32), rtol=1e-6)
        np.testing.assert_allclose(m['poses'].cpu().numpy(), r['poses'].numpy(), rtol=1e-4, atol=1e-4)
        assert m['scores'][0] == h['scores'][0]   # both start from the best candidate


def tiny_detector_cfg(J=15):
    return dict(
        type='DAS', pretrained=None,
        backbone=dict(type='MSPN2', unit_channels=32, num_stages=1, num_units=4, num_blocks=[1, 1, 1, 1],
                      norm_cfg=dict(type='BN'), compute_dtype='f32'),
        neck=dict(type='FPN', in_channels=[32] * 4, out_channels=32, start_level=1, add_extra_convs='on_output',
                  num_outs=4, relu_before_extra_convs=True, norm_cfg=dict(type='BN')),
        bbox_head=dict(type='DASHead', num_classes=1, in_channels=32, feat_channels=32, stacked_convs=2,
                       strides=[8, 16, 32, 64], regress_ranges=((-1, 80), (80, 160), (160, 320), (320, 1e8)),
                       num_joints=J, depth_factor=20, z_norm=50, root_idx=2, cls_branch=(32,), reg_branch=((32,),) * 4,
                       centerness_on_reg=True, conv_bias=True, dcn_on_last_conv=True,
                       recursive_update=dict(prev_loss=True, num_heads=4, in_channels=32, feat_channels=32,
                                             num_layers=1, dim=3, num_joints=J)),
        train_cfg=dict(code_weight=[1.0, 1.0, 1] + [2] * J * 6),
        test_cfg=dict(nms_across_levels=False, nms_pre=1000, nms_post=100, nms_thr=0.9, score_thr=0.07))


def test_detector_end_to_end_f32_vs_oracle():
    """img -> MSPN2 -> FPN -> DASHead -> decode, f32 path, against the oracle chained the same way."""
    import das_amd
    from oracle import backbone as ob
    from oracle import decode as od
    from oracle import head as oh
    cfg = tiny_detector_cfg()
    model = das_amd.build_model(cfg)
    sd = cases.det_fill(model.state_dict(), 12)
    # bias the class logits so that a few hundred locations pass the 0.07 threshold
    sd['bbox_head.conv_cls.bias'].fill_(0.5)
    sd['bbox_head.conv_centerness.bias'].fill_(1.0)
    sd = {k: v.clone() for k, v in sd.items()}
    model.to(DEV).eval()
    img = cases.randn(99, 2, 3, 128, 192)
    metas = [dict(scale_factor=np.array([1.2, 1.2, 1.2, 1.2], dtype=np.float32), filename='a'),
             dict(scale_factor=np.ones(4, dtype=np.float32), filename='b')]
    res = model(img.to(DEV), metas, return_loss=False, rescale=True)

    bsd = {k[len('backbone.'):]: v for k, v in sd.items() if k.startswith('backbone.')}
    nsd = {k[len('neck.'):]: v for k, v in sd.items() if k.startswith('neck.')}
    hsd = {k[len('bbox_head.'):]: v for k, v in sd.items() if k.startswith('bbox_head.')}
    hcfg = dict(num_joints=15, root_idx=2, depth_factor=20, z_norm=50, strides=[8, 16, 32, 64], stacked_convs=2,
                num_heads=4, num_layers=1)
    with torch.no_grad():
        feats = ob.fpn_forward(nsd, ob.mspn2_forward(bsd, img, 1, (1, 1, 1, 1)))
        o_cls, o_pose, o_ctr = oh.head_forward(hsd, feats, hcfg, '', False)
    ref = od.get_poses(o_cls, o_pose, o_ctr, metas, 15, hcfg['strides'], cfg['test_cfg'], return_index=True)

    # head maps first (f32 path within 1e-4 of the map range)
    with torch.no_grad():
        outs = model.bbox_head(model.extract_feat(img.to(DEV)))
    for lst, rl in zip(outs, (o_cls, o_pose, o_ctr)):
        for t, r in zip(lst, rl):
            assert rel_err(nchw_np(t), r.numpy()) < 2e-4
    for r, o in zip(res, ref):
        assert len(o['scores']) > 5
        assert r['poses'].shape == o['poses'].shape
        np.testing.assert_allclose(np.array(r['scores']), np.array(o['scores']), rtol=2e-4)
        np.testing.assert_allclose(r['poses'].cpu().numpy(), o['poses'].numpy(), rtol=1e-4, atol=2e-3)
        np.testing.assert_allclose(r['centers'].cpu().numpy(), o['centers'].numpy(), rtol=1e-4, atol=2e-3)


def test_unconsumed_finest_map_is_skipped_without_changing_anything_else():
    """The neck starts at level 1 (every DAS config), so the last MSPN stage's finest map has no consumer; the reference
    computes it anyway (mspn_mmpose.py:381-404). Here the detector does not (eval) / reduces it to the BatchNorm
    statistics its two conv layers must keep advancing (train): neck outputs, losses, gradients and EVERY buffer of the
    state dict must equal the run that computes the map in full."""
    import das_amd
    from das_amd.datasets import SyntheticPoseDataset, collate
    cfg = tiny_detector_cfg()
    cfg['backbone'].update(compute_dtype='f32', num_stages=2)
    ds = SyntheticPoseDataset(num_joints=15, img_shape=(128, 192), length=4, seed=3, max_persons=3)
    data = collate([ds[i] for i in range(2)], device=DEV)
    res = {}
    for skip in (True, False):
        torch.manual_seed(0)
        model = das_amd.build_model(cfg)
        model.init_weights()
        assert model.backbone.skip_unused_finest      # (set by the detector: neck.start_level == 1)
        model.backbone.skip_unused_finest = skip
        model.to(DEV).eval()
        with torch.no_grad():
            feats_eval = [f.float().clone() for f in model.extract_feat(data['img'])]
        model.train()
        out = model.train_step(data, None)
        out['loss'].backward()
        torch.cuda.synchronize()
        res[skip] = (feats_eval, float(out['loss']), {n: b.detach().clone() for n, b in model.named_buffers()},
                     {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    (fa, la, ba, ga), (fb, lb, bb, gb) = res[True], res[False]
    for a, b in zip(fa, fb):
        assert torch.equal(a, b)
    assert abs(la - lb) <= 1e-4 * abs(lb)
    assert set(ba) == set(bb) and set(ga) == set(gb)
    moved = 0
    for n in bb:
        # (two independent runs: float atomics in the statistics reorder sums; the deepest layers of this tiny net see 48 rows)
        torch.testing.assert_close(ba[n].float(), bb[n].float(), rtol=2e-3, atol=2e-4, msg=n)
        if 'up4' in n and 'multi_stage_mspn.1' in n and 'running_var' in n:
            moved += int(not torch.allclose(bb[n], torch.ones_like(bb[n])))
    assert moved >= 2       # the skipped unit's BatchNorm layers did advance their running statistics
    # (gradient VALUES of two independent runs of this tiny train-mode net differ by 1e-2 ... 2e-1 of their scale — ReLU masks
    # flip with the summation order of the statistics, DESIGN section 3 — so only the set of parameters that received a
    # gradient is compared: the skipped unit's parameters get none in either run, as in the reference)
