"""The BASELINE.json configurations that round 1 never built under `-m gpu`:
  configs[2]/[3]  MSPN-50 4-stage: tiny-width forward / backward against the reference fixture
                  (mspn_mmpose.py:603-667) and the f64 oracle, and ONE full-width 4-stage train step with
                  property checks;
  configs[4]      exp_mupots.py topology: 3-stage backbone forward, J=21 / root 14 / depth_factor 1 /
                  two recursive-update layers head — forward (train, eval), decode, losses and gradients on the HIP
                  path against fixtures captured from the reference (das_head.py:176-267,281-486,653-796;
                  recursive_update.py:238-255).
"""
import os

import numpy as np
import pytest
import torch

import cases
from test_train_gpu import band_check, grad_sd, param_errors, rel

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + '.npz'))


def sd_of(z, seed):
    shapes = [[int(i) for i in row if i >= 0] for row in z['sd_shapes']]
    return cases.sd_from_manifest(z['sd_keys'], shapes, z['sd_dtypes'], seed)


def test_mspn2_three_stage_eval_vs_reference_fixture(golden_dir):
    import das_amd
    z = load(golden_dir, 'mspn_s3_eval')
    m = das_amd.MSPN2(unit_channels=16, num_stages=3, num_blocks=[1, 1, 1, 1], compute_dtype='f32')
    m.load_state_dict(sd_of(z, 1), strict=True)
    m.to(DEV).eval()
    with torch.no_grad():
        outs = m(cases.randn(7, 2, 3, 64, 96).to(DEV))
    for i, o in enumerate(outs):
        assert rel(o.float().cpu().numpy(), z[f'out{i}']) < 1e-4, i


def test_mspn2_four_stage_train_forward_backward(golden_dir):
    """4-stage (cross-stage skips through three seams), train-mode BN: forward vs the reference fixture, parameter
    gradients vs the f64 oracle in the oracle-f32 error band (see test_train_gpu.py for the yardstick)."""
    import das_amd
    from oracle import backbone as ob
    z = load(golden_dir, 'mspn_s4_train')
    sd = sd_of(z, 1)
    m = das_amd.MSPN2(unit_channels=16, num_stages=4, num_blocks=[1, 1, 1, 1], compute_dtype='f32')
    m.load_state_dict(sd, strict=True)
    m.to(DEV).train()
    x = cases.randn(7, 2, 3, 64, 96)
    gs = [cases.randn(80 + i, 2, 16, 16 >> i, 24 >> i) for i in range(4)]
    outs = m(x.to(DEV))
    for i, o in enumerate(outs):
        # 4 stages of train-mode BN in f32: the ORACLE itself is held to 2e-4 of the map range against this fixture
        # (tests/test_oracle_golden.py TOL); the HIP result moves by a few 1e-5 from run to run (atomic order of the
        # statistics) and sat at 1.3...2.02e-4 over repeated runs, hence 3e-4. The 1- and 2-stage fixtures hold 1e-4
        # (test_model_gpu.py)
        assert rel(o.detach().float().cpu().numpy(), z[f'out{i}']) < 3e-4, i
    msd = m.state_dict()
    np.testing.assert_allclose(msd['top.top.0.bn.running_mean'].cpu().numpy(), z['rm_top'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(msd['multi_stage_mspn.3.upsample.up4.in_skip.bn.running_var'].cpu().numpy(), z['rv_last'],
                               rtol=1e-4, atol=1e-6)
    sum((o.float() * g.to(DEV)).sum() for o, g in zip(outs, gs)).backward()
    refs = {}
    for dt in (torch.float64, torch.float32):
        osd = grad_sd(sd, dt)
        oo = ob.mspn2_forward(osd, x.to(dt), 4, (1, 1, 1, 1), train=True)
        sum((o * g.to(dt)).sum() for o, g in zip(oo, gs)).backward()
        refs[dt] = osd
    e_hip, e_o32 = param_errors(m, refs[torch.float64], refs[torch.float32])
    assert len(e_hip) > 300
    band_check(e_hip, e_o32, 'mspn2 4-stage params', slack=1.5)


def test_full_width_four_stage_train_step_properties():
    """BASELINE configs[2] at B=2: the real MSPN-50 4-stage + FPN + DASHead (J=15), bf16, full 512x832 frames — the
    dispatch the benchmark runs. Finite, bounded losses; gradients on the set the reference trains; weights move."""
    import bench
    from das_amd.datasets import SyntheticPoseDataset, collate
    from das_amd.optim import FlatSGD, train_iteration
    model = bench.build_model(DEV, seed=0, dtype='bf16', num_stages=4, train=True)
    ds = SyntheticPoseDataset(num_joints=15, img_shape=(512, 832), length=2, seed=0)
    data = collate([ds[i] for i in range(2)], device=DEV)
    opt = FlatSGD(model, lr=2e-3, momentum=0.9, weight_decay=1e-4, bias_lr_mult=2.0, bias_decay_mult=0.0,
                  max_grad_norm=35.0)
    losses = []
    p0 = opt.flat_p.clone()
    for it in range(8):
        out = train_iteration(model, opt, data, 2e-3)
        assert all(np.isfinite(v) for v in out['log_vars'].values()), out['log_vars']
        losses.append(out['log_vars']['loss'])
        if it == 0:
            g = opt.flat_g
            assert bool(torch.isfinite(g).all())
            dead = [n for n, p in model.named_parameters() if float(p.grad.abs().max()) == 0.0]
            allowed = ('multi_stage_mspn.3.upsample.up4', 'flow2d', 'flow3d', 'conv_reg_prevs.0.', 'conv_regs.0.')
            odd = [n for n in dead if not any(a in n for a in allowed) and not n.startswith('bbox_head.scales')]
            assert not odd, odd[:12]
    # B = 2 with train-mode BN at the reference's lr is too noisy for a monotone loss over a few steps (the tiny-width
    # test_train_step_gpu.py checks the decrease): here the step must stay bounded and move every trained weight
    assert max(losses) < 1.5 * losses[0], losses
    moved = (opt.flat_p - p0).abs()
    assert float(moved.max()) > 0 and bool(torch.isfinite(opt.flat_p).all())
    assert len(model.backbone.multi_stage_mspn) == 4


def build_mupots_head(dtype=torch.float32):
    import das_amd
    c = cases.MUPOTS_CFG
    J, C = c['num_joints'], c['feat_channels']
    return das_amd.DASHead(
        num_classes=1, in_channels=C, feat_channels=C, stacked_convs=2, strides=c['strides'],
        regress_ranges=c['regress_ranges'], num_joints=J, depth_factor=c['depth_factor'], z_norm=c['z_norm'],
        root_idx=c['root_idx'], cls_branch=(C,), reg_branch=((C,),) * 4, centerness_branch=(64,),
        centerness_on_reg=True, conv_bias=True, dcn_on_last_conv=True,
        recursive_update=dict(prev_loss=True, num_heads=c['num_heads'], in_channels=C, feat_channels=C,
                              num_layers=c['num_layers'], dim=3, num_joints=J),
        train_cfg=dict(code_weight=c['code_weight']), test_cfg=cases.FULL_TEST_CFG, compute_dtype=dtype)


def test_mupots_head_eval_and_decode_vs_reference_fixture(golden_dir):
    z = load(golden_dir, 'head_mupots_eval')
    head = build_mupots_head()
    head.load_state_dict(sd_of(z, 13), strict=True)
    head.to(DEV).eval()
    feats = [f.to(DEV) for f in cases.head_feats(seed=61, sizes=cases.MUPOTS_SIZES)]
    with torch.no_grad():
        outs = head(feats)
    for name, lst in zip(('cls', 'pose', 'ctr'), outs):
        for i, t in enumerate(lst):
            assert tuple(t.shape) == z[f'{name}{i}'].shape
            assert rel(t.float().cpu().numpy(), z[f'{name}{i}']) < 2e-4, (name, i)
    # decode on the REFERENCE's maps (so that the kept set cannot differ through forward rounding)
    metas = [dict(scale_factor=np.array([1.25, 1.25, 1.25, 1.25], dtype=np.float32), filename='a'),
             dict(scale_factor=np.array([1., 1., 1., 1.], dtype=np.float32), filename='b')]
    cls = [torch.from_numpy(z[f'cls{i}']).to(DEV) + 1.5 for i in range(4)]
    ctr = [torch.from_numpy(z[f'ctr{i}']).to(DEV) + 1.0 for i in range(4)]
    pose = [torch.from_numpy(z[f'pose{i}']).to(DEV) for i in range(4)]
    res = head.get_poses(cls, pose, ctr, metas)
    for b, r in enumerate(res):
        assert r['poses'].shape == z[f'dec_poses{b}'].shape == (100, 21, 3)
        np.testing.assert_allclose(np.array(r['scores'], dtype=np.float32), z[f'dec_scores{b}'], rtol=1e-5)
        np.testing.assert_allclose(r['poses'].cpu().numpy(), z[f'dec_poses{b}'], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(r['centers'].cpu().numpy(), z[f'dec_centers{b}'], rtol=1e-4, atol=1e-4)


def test_mupots_head_train_losses_vs_reference_fixture(golden_dir):
    z = load(golden_dir, 'head_mupots_train')
    head = build_mupots_head()
    head.load_state_dict(sd_of(z, 13), strict=True)
    head.to(DEV).train()
    feats = cases.head_feats(seed=61, sizes=cases.MUPOTS_SIZES)
    fin = [t.permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True) for t in feats]
    with torch.no_grad():
        outs = head([t.permute(0, 3, 1, 2) for t in fin])
    for name, lst in zip(('cls', 'pose', 'ctr', 'ref'), outs):
        for i, t in enumerate(lst):
            assert rel(t.float().cpu().numpy(), z[f'{name}{i}']) < 2e-4, (name, i)
    g = cases.mupots_gts()
    losses = head.forward_train([t.permute(0, 3, 1, 2) for t in fin], [{}, {}], None, None,
                                [p.to(DEV) for p in g['gt_poses_3d']], None, None, None)
    for k, v in losses.items():
        assert rel(v.item(), float(z[k])) < 2e-4, (k, v.item(), float(z[k]))
    sum(losses.values()).backward()
    for i, t in enumerate(fin):   # (f32 conditioning band, see test_loss_gpu.py)
        assert rel(t.grad.permute(0, 3, 1, 2).cpu().numpy(), z[f'grad_feat{i}']) < 5e-2, i


def test_mupots_full_size_inference_bf16():
    """configs[4] end to end on the HIP path at 768x1024, B=3: 3-stage MSPN-50 + FPN + the J=21 head + decode;
    checks geometry (16 320 locations), finite outputs and that decode returns poses."""
    import das_amd
    import bench
    J = 21
    cfg = bench.model_cfg(3, 'bf16')
    cfg['bbox_head'].update(num_joints=J, root_idx=14, depth_factor=1)
    cfg['bbox_head']['recursive_update'].update(num_layers=2, num_joints=J)
    cfg['train_cfg'] = dict(code_weight=[1.0, 1.0, 1] + [2] * J * 6)
    torch.manual_seed(0)
    model = das_amd.build_model(cfg)
    model.init_weights()
    model.to(DEV).eval()
    img = torch.randn(3, 3, 768, 1024, device=DEV)
    metas = [dict(scale_factor=np.ones(4, dtype=np.float32), filename=str(i)) for i in range(3)]
    with torch.no_grad():
        cls, pose, ctr = model.bbox_head(model.extract_feat(img))
    assert sum(c.shape[-2] * c.shape[-1] for c in cls) == 16320
    assert pose[0].shape[1] == 3 + 6 * J
    assert all(bool(torch.isfinite(t.float()).all()) for t in list(cls) + list(pose) + list(ctr))
    bench.calibrate_scores(model, img, metas, target=120)
    res = model(img, metas, return_loss=False, rescale=True)
    assert len(res) == 3 and all(r['poses'].shape[1:] == (J, 3) for r in res)
    assert sum(len(r['scores']) for r in res) > 0
