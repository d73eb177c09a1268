"""CPU-side checks: the C-ABI library loads and exports every symbol include/das_hip.h declares,
the host mirror keeps the reference's registry / config / state-dict contract, and the product path
refuses to run without the GPU (no CPU fallback). No compute calls are made here."""
import os
import re

import numpy as np
import pytest
import torch

import cases
import refstub

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from das_amd import _lib
    hdr = open(os.path.join(ROOT, 'include', 'das_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(das_[a-z0-9_]+)\s*\(', hdr))
    assert declared, 'no declarations parsed'
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.das_abi_version() == 4 and lib.das_target_arch() == b'gfx950'


def test_ops_refuse_cpu_tensors():
    from das_amd import _lib, ops
    with pytest.raises(_lib.DasHipError):
        ops.conv2d(torch.zeros(1, 4, 4, 8), torch.zeros(8, 1, 1, 8), 1, 1)
    with pytest.raises(_lib.DasHipError):
        ops.maxpool3x3s2(torch.zeros(1, 4, 4, 8))


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'das_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', src, flags=re.M), f


def _manifest_shapes(z):
    return {str(k): tuple(int(i) for i in row if i >= 0) for k, row in zip(z['sd_keys'], z['sd_shapes'])}


def test_state_dict_keys_match_reference_manifests(golden_dir):
    """Fixture manifests were captured from the reference modules' own state_dict()."""
    import das_amd
    z = np.load(os.path.join(golden_dir, 'mspn_s2_eval.npz'))
    m = das_amd.MSPN2(unit_channels=16, num_stages=2, num_blocks=[1, 1, 1, 1])
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == _manifest_shapes(z)
    z = np.load(os.path.join(golden_dir, 'head_eval.npz'))
    c = cases.HEAD_CFG
    C = c['feat_channels']
    h = das_amd.DASHead(num_classes=1, in_channels=C, feat_channels=C, stacked_convs=2, strides=c['strides'],
                        regress_ranges=c['regress_ranges'], num_joints=3, depth_factor=20, z_norm=50, root_idx=1,
                        cls_branch=(C,), reg_branch=((C,),) * 4, conv_bias=True, dcn_on_last_conv=True,
                        recursive_update=dict(prev_loss=True, num_heads=4, in_channels=C, feat_channels=C, num_layers=2,
                                              dim=3, num_joints=3))
    assert {k: tuple(v.shape) for k, v in h.state_dict().items()} == _manifest_shapes(z)


def test_full_size_key_and_param_counts():
    """768 backbone keys / 55.96 M params (2-stage), 390 head keys / 7.86 M params (J=15)."""
    import das_amd
    cfg = das_amd.Config.fromfile(os.path.join(ROOT, 'configs/das/exp_panoptic.py'))
    cfg.model.pretrained = None
    m = das_amd.build_model(cfg.model)
    bsd = m.backbone.state_dict()
    assert len(bsd) == 768 and abs(sum(p.numel() for p in m.backbone.parameters()) / 1e6 - 55.96) < 0.01
    assert len(m.bbox_head.state_dict()) == 390
    assert abs(sum(p.numel() for p in m.bbox_head.parameters()) / 1e6 - 7.857) < 0.01
    for k in ('top.top.0.conv.weight', 'multi_stage_mspn.1.downsample.layer4.0.downsample.bn.running_var',
              'multi_stage_mspn.0.upsample.up4.cross_conv.conv.weight', 'multi_stage_mspn.0.upsample.up1.out_skip1.bn.bias'):
        assert k in bsd, k
    assert 'multi_stage_mspn.1.upsample.up4.cross_conv.conv.weight' not in bsd  # last stage: no cross conv
    hsd = m.bbox_head.state_dict()
    assert tuple(hsd['cls_convs.1.conv.conv_offset.weight'].shape) == (27, 256, 3, 3)
    assert tuple(hsd['recursive_update_branch.layer_0.next_level_offset.sampling_offset.weight'].shape) == (120, 256, 1, 1)
    assert tuple(hsd['flow3d.mask'].shape) == (6, 3) and 'scales.3.3.scale' in hsd
    assert m.neck.start_level == 1 and len(m.neck.fpn_convs) == 4 and m.bbox_head.strides == [8, 16, 32, 64]


def test_config_base_delete_and_cfg_options(tmp_path):
    import das_amd
    (tmp_path / 'base.py').write_text("model = dict(backbone=dict(type='ResNet', depth=50), neck=dict(a=1, b=2))\nlr = 1\n")
    (tmp_path / 'exp.py').write_text(
        "_base_ = ['./base.py']\nmodel = dict(backbone=dict(_delete_=True, type='MSPN2', num_stages=2), neck=dict(b=3))\n")
    cfg = das_amd.Config.fromfile(str(tmp_path / 'exp.py'))
    assert cfg.model.backbone == dict(type='MSPN2', num_stages=2)
    assert cfg.model.neck == dict(a=1, b=3) and cfg.lr == 1
    cfg.merge_from_dict(das_amd.config.parse_cfg_options(['model.neck.a=5', 'model.backbone.num_stages=4', 'name=abc']))
    assert cfg.model.neck.a == 5 and cfg.model.backbone.num_stages == 4 and cfg.name == 'abc'
    with pytest.raises(AttributeError):
        cfg.model.nope


def test_registry_protocol():
    import das_amd
    assert {'MSPN2'} <= set(das_amd.BACKBONES.module_dict) and 'DASHead' in das_amd.HEADS and 'DAS' in das_amd.DETECTORS
    assert {'FocalLoss', 'SmoothL1Loss', 'CrossEntropyLoss', 'RLELoss3D'} <= set(das_amd.LOSSES.module_dict)
    with pytest.raises(KeyError):
        das_amd.build_backbone(dict(type='NoSuchBackbone'))
    with pytest.raises(KeyError):
        das_amd.BACKBONES.register_module()(das_amd.MSPN2)  # duplicate without force


@pytest.mark.skipif(not refstub.available(), reason='/root/reference not mounted')
def test_reference_config_file_loads_unchanged_and_keys_match_reference_modules():
    import das_amd
    cfg = das_amd.Config.fromfile('/root/reference/configs/das/exp_panoptic.py')
    cfg.model.pretrained = None
    model = das_amd.build_model(cfg.model)
    assert cfg.model.neck.start_level == 1 and cfg.model.bbox_head.stacked_convs == 2  # inherited from _base_
    R = refstub.load()
    bb = dict(cfg.model.backbone)
    bb.pop('type')
    ref_b = R.MSPN2(**bb)
    assert {k: tuple(v.shape) for k, v in ref_b.state_dict().items()} == \
           {k: tuple(v.shape) for k, v in model.backbone.state_dict().items()}
    hd = dict(cfg.model.bbox_head)
    hd.pop('type')
    ref_h = R.DASHead(**hd, train_cfg=cfg.model.train_cfg, test_cfg=cfg.model.test_cfg)
    assert {k: tuple(v.shape) for k, v in ref_h.state_dict().items()} == \
           {k: tuple(v.shape) for k, v in model.bbox_head.state_dict().items()}


def test_synthetic_dataset_layout():
    from das_amd.datasets import SyntheticPoseDataset, collate
    ds = SyntheticPoseDataset(num_joints=15, img_shape=(64, 96), length=4, seed=3)
    s = ds[1]
    G = s['gt_poses_3d'].shape[0]
    assert s['gt_poses_3d'].shape == (G, 3 + 4 * 15) and s['centers2d'].shape == (G, 2) and s['depths'].shape == (G,)
    assert torch.equal(ds[1]['img'], s['img'])  # deterministic
    root_dz = s['gt_poses_3d'][:, 3 + 3 * 2 + 2]
    assert float(root_dz.abs().max()) == 0.0
    b = collate([ds[0], ds[1]])
    assert b['img'].shape == (2, 3, 64, 96) and len(b['gt_poses_3d']) == 2


def _resnet50_like_checkpoint(seed=0):
    """A ResNet-50 classification checkpoint in the torchvision / MMPose key layout (conv1, bn1, layerL.b.convK,
    layerL.0.downsample.{0,1}) with random values."""
    g = torch.Generator().manual_seed(seed)
    sd = {'conv1.weight': torch.randn(64, 3, 7, 7, generator=g)}
    for n in ('weight', 'bias', 'running_mean', 'running_var'):
        sd[f'bn1.{n}'] = torch.rand(64, generator=g) + 0.5
    inp = 64
    for L, (blocks, mid) in enumerate(zip((3, 4, 6, 3), (64, 128, 256, 512)), 1):
        for b in range(blocks):
            cin = inp if b == 0 else mid * 4
            for name, shape in (('conv1', (mid, cin, 1, 1)), ('conv2', (mid, mid, 3, 3)), ('conv3', (mid * 4, mid, 1, 1))):
                sd[f'layer{L}.{b}.{name}.weight'] = torch.randn(*shape, generator=g) * 0.05
            for i, c in ((1, mid), (2, mid), (3, mid * 4)):
                for n in ('weight', 'bias', 'running_mean', 'running_var'):
                    sd[f'layer{L}.{b}.bn{i}.{n}'] = torch.rand(c, generator=g) + 0.5
            if b == 0:
                sd[f'layer{L}.0.downsample.0.weight'] = torch.randn(mid * 4, cin, 1, 1, generator=g) * 0.05
                for n in ('weight', 'bias', 'running_mean', 'running_var'):
                    sd[f'layer{L}.0.downsample.1.{n}'] = torch.rand(mid * 4, generator=g) + 0.5
        inp = mid * 4
    sd['fc.weight'] = torch.randn(1000, 2048, generator=g)
    return sd


def test_mspn2_init_weights_resnet_remap(tmp_path):
    """mspn_mmpose.py:694-721: a ResNet-50 checkpoint lands on the stem and on EVERY stage's downsample module
    (downsample.0 -> downsample.conv, downsample.1 -> downsample.bn, conv1 -> top.0.conv, bn1 -> top.0.bn), with
    `module.` / `backbone.` prefixes stripped; the upsample modules keep their kaiming initialisation."""
    import das_amd
    sd = _resnet50_like_checkpoint()
    path = str(tmp_path / 'resnet50.pth')
    torch.save({'state_dict': {'module.backbone.' + k: v for k, v in sd.items()}}, path)
    m = das_amd.MSPN2(unit_channels=256, num_stages=2, num_blocks=[3, 4, 6, 3], pretrained=path)
    report = m.init_weights()
    assert all(not r.missing_keys for r in report[1:]), report[1].missing_keys[:5]    # every downsample key was covered
    msd = m.state_dict()
    assert torch.equal(msd['top.top.0.conv.weight'], sd['conv1.weight'])
    assert torch.equal(msd['top.top.0.bn.running_var'], sd['bn1.running_var'])
    for s in range(2):
        pre = f'multi_stage_mspn.{s}.downsample.'
        assert torch.equal(msd[pre + 'layer1.0.conv1.weight'], sd['layer1.0.conv1.weight'])
        assert torch.equal(msd[pre + 'layer3.5.conv2.weight'], sd['layer3.5.conv2.weight'])
        assert torch.equal(msd[pre + 'layer4.0.downsample.conv.weight'], sd['layer4.0.downsample.0.weight'])
        assert torch.equal(msd[pre + 'layer2.0.downsample.bn.running_mean'], sd['layer2.0.downsample.1.running_mean'])
    up = msd['multi_stage_mspn.0.upsample.up1.in_skip.conv.weight']
    assert abs(float(up.std()) - (2.0 / 256) ** 0.5) < 0.02                  # kaiming fan_out of a 1x1 2048 -> 256 conv
    with pytest.raises(FileNotFoundError):
        das_amd.MSPN2(num_stages=1, num_blocks=[1, 1, 1, 1], pretrained='nowhere.pth').init_weights()


def test_mspn2_init_weights_detector_checkpoint(tmp_path, monkeypatch):
    """mspn_mmpose.py:672-680: a `weights/...` path is a detector / MSPN checkpoint whose `backbone.` keys load by name."""
    import das_amd
    src = das_amd.MSPN2(unit_channels=16, num_stages=2, num_blocks=[1, 1, 1, 1])
    src.init_weights()
    os.makedirs(tmp_path / 'weights')
    torch.save({'state_dict': {'backbone.' + k: v for k, v in src.state_dict().items()} | {'keypoint_head.x': torch.zeros(1)}},
               str(tmp_path / 'weights' / 'mspn.pth'))
    monkeypatch.chdir(tmp_path)
    dst = das_amd.MSPN2(unit_channels=16, num_stages=2, num_blocks=[1, 1, 1, 1], pretrained='weights/mspn.pth')
    rep = dst.init_weights()
    assert not rep.missing_keys and not rep.unexpected_keys
    for k, v in src.state_dict().items():
        assert torch.equal(v, dst.state_dict()[k]), k


def test_lazy_log_vars_equal_the_eager_ones():
    """`_parse_losses(lazy=True)` (what train_iteration uses: the read-back happens after the whole step has been queued)
    resolves to the same loss and the same log variables as the eager form the reference's `_parse_losses` has
    (mmdet BaseDetector._parse_losses: sum of the entries whose key contains 'loss', lists summed entry by entry)."""
    import torch
    from das_amd.detectors import DAS, LazyLogVars
    losses = {'loss_cls': torch.tensor([0.25, 0.75]), 'loss_pose': [torch.tensor(1.5), torch.tensor([2.0, 4.0])],
              'acc': torch.tensor(0.5), 'loss_depth': torch.tensor(3.0)}
    loss_e, lv_e = DAS._parse_losses(losses)
    loss_l, lv_l = DAS._parse_losses(losses, lazy=True)
    assert isinstance(lv_l, LazyLogVars) and torch.equal(loss_e, loss_l)
    assert float(loss_e) == 0.5 + (1.5 + 3.0) + 3.0
    got = lv_l.resolve()
    assert list(got) == list(lv_e) == ['loss_cls', 'loss_pose', 'acc', 'loss_depth', 'loss'] and got == lv_e
    # the lazy form IS a mapping (what train_iteration hands back): reading it resolves it
    assert list(lv_l) == list(lv_e) and dict(lv_l.items()) == lv_e and lv_l['loss'] == lv_e['loss'] and len(lv_l) == 5
    assert lv_l == lv_e and 'acc' in lv_l


@pytest.mark.parametrize('n,world', [(10, 4), (8, 4), (3, 4), (7, 2), (1, 3), (0, 2)])
def test_multi_rank_result_collection_keeps_the_tail(n, world):
    """tools/test.py: rank r evaluates samples r, r + world, ...; the collected list is the dataset order and keeps
    every result also when len(dataset) % world != 0 (the reference pads the sampler and trims: tools/test.py:205-206)."""
    from das_amd.datasets import collect_results
    parts = [[dict(idx=i) for i in range(r, n, world)] for r in range(world)]
    got = collect_results(parts, n)
    assert [r['idx'] for r in got] == list(range(n))
    # a sampler padded the reference's way (indices repeated up to a multiple of world) is trimmed to the dataset size
    padded = [[dict(idx=i % max(n, 1)) for i in range(r, -(-n // world) * world, world)] for r in range(world)] if n else parts
    assert [r['idx'] for r in collect_results(padded, n)] == list(range(n))


@pytest.mark.parametrize('H,Ho', [(8, 16), (13, 26), (5, 9), (1, 4), (7, 7), (6, 15), (26, 52)])
def test_upsample_adjoint_tables_match_torch_interpolate(H, Ho):
    """ops._upsample_tables (host side of das_upsample_stats_lowres / das_upmerge_backward_lowres): the three diagonals of
    U^T U and U^T 1 for U = bilinear upsampling with align_corners along one axis, against the matrix torch's interpolation
    defines (F.interpolate of the identity, f64): U^T U is tridiagonal, and with both axes the tables reproduce
    sum upsample(z) and sum upsample(z)^2 from z alone."""
    import torch.nn.functional as F
    from das_amd import ops
    a, w = ops._upsample_tables(H, Ho, 'cpu')
    U = F.interpolate(torch.eye(H, dtype=torch.float64)[None], size=Ho, mode='linear', align_corners=True)[0].T   # (Ho, H)
    A = U.T @ U
    assert float((A - torch.triu(torch.tril(A, 1), -1)).abs().max()) < 1e-12
    for d in (-1, 0, 1):
        idx = torch.arange(max(0, -d), min(H, H - d))
        assert torch.allclose(a[idx, d + 1].double(), A[idx, idx + d], atol=2e-6), d
    assert torch.allclose(w.double(), U.sum(0), atol=2e-6)
    # two axes: statistics of the upsampled tensor from the low resolution
    W, Wo = 5, 11
    aw, ww = ops._upsample_tables(W, Wo, 'cpu')
    z = cases.randn(9, H, W).double()
    up = F.interpolate(z[None, None], size=(Ho, Wo), mode='bilinear', align_corners=True)[0, 0]
    zp = F.pad(z, (1, 1, 1, 1))
    G = sum(a[:, dh + 1].double()[:, None] * aw[:, dw + 1].double()[None, :] * zp[1 + dh:1 + dh + H, 1 + dw:1 + dw + W]
            for dh in (-1, 0, 1) for dw in (-1, 0, 1))
    assert abs(float((w.double()[:, None] * ww.double()[None, :] * z).sum() - up.sum())) < 1e-5 * max(1.0, float(up.abs().sum()))
    assert abs(float((z * G).sum() - (up * up).sum())) < 1e-5 * float((up * up).sum())


def test_flat_sgd_pads_odd_channel_counts_inside_its_storage_and_nowhere_else():
    """FlatSGD stores a conv weight whose output channels are not a multiple of 8 with zero rows up to the next multiple, a
    1-D parameter with zeros up to a multiple of 8 elements (the padded block is the operand the kernels read): the
    parameters keep their own shapes and values, state dict and dense momentum see no padding, the padding is zero and
    the buckets cover it."""
    import torch.nn as nn
    from das_amd.optim import FlatSGD
    torch.manual_seed(0)
    net = nn.Sequential(nn.Conv2d(16, 45, 1), nn.Conv2d(48, 27, 3, padding=1), nn.Conv2d(3, 8, 3), nn.Conv2d(8, 16, 1, bias=False))
    want = {k: v.clone() for k, v in net.state_dict().items()}
    opt = FlatSGD(net, lr=0.1, momentum=0.9, bucket_mb=0)
    for k, v in net.state_dict().items():
        assert v.shape == want[k].shape and torch.equal(v, want[k]), k
    by_name = dict(zip(opt._names, opt.slots))
    w45, b45, w27, w3, w16 = (by_name[n] for n in ('0.weight', '0.bias', '1.weight', '2.weight', '3.weight'))
    assert (w45.o_pad, w45.span, w45.numel, w45.packable) == (48, 48 * 16, 45 * 16, True)
    assert (w27.o_pad, w27.span, w27.packable) == (32, 32 * 9 * 48, True)
    assert (b45.span, b45.numel) == (48, 45) and b45.padded().numel() == 48
    assert (w3.o_pad, w3.span, w3.packable) == (8, 8 * 9 * 3, False)        # (3 input channels: the per-layer path, no padding)
    assert (w16.o_pad, w16.span, w16.packable) == (16, 16 * 8, True)
    assert w45.direct(16, 48) and w45.direct(16, 45) and not w45.direct(16, 56) and not w45.direct(8, 48)
    for sl in opt.slots:
        assert sl.off % 8 == 0
        for buf in (opt.flat_p, opt.flat_g, opt.flat_m):
            assert float(buf[sl.off + sl.numel:sl.off + sl.span].abs().sum()) == 0.0
    ends = sorted(sl.off + sl.span for sl in opt.slots)
    assert opt.buckets[-1][1] == opt.flat_p.numel() >= ends[-1]
    dense = opt._dense_momentum()
    assert {k: tuple(v.shape) for k, v in dense.items()} == {k: tuple(v.shape) for k, v in want.items()}


def test_zero_arena_never_refills_a_slice_that_may_still_be_in_use():
    """nn._ZeroArena hands out zeroed slices of two alternating buffers. A layer pair takes its second slice BEFORE the
    first one's consumer is launched (the two convs of an upsample-unit merge, a dual apply, a held finalize pair): the take
    that switches buffers must leave the slices of the buffer it leaves untouched (one buffer refilled in place zeroed the
    first layer's statistics whenever the refill fell between the two takes — rounds 3-4), and everything it hands out
    must be zero again however the previous holders left it."""
    from das_amd.nn import _ZeroArena
    dev = torch.device('cpu')
    a = _ZeroArena(cap=1024)
    first = a.take(960, dev)
    first.fill_(3.0)                      # (a conv epilogue accumulated its statistics)
    second = a.take(128, dev)             # does not fit: the other buffer takes over
    assert float(first.min()) == 3.0, 'the pending slice was refilled under its consumer'
    assert float(second.abs().max()) == 0.0 and second.data_ptr() != first.data_ptr()
    second.fill_(5.0)
    for _ in range(40):                   # many switches later every slice still comes out zeroed
        s = a.take(200, dev)
        assert s.numel() == 256 and float(s.abs().max()) == 0.0
        s.fill_(7.0)
    a.reset()
    assert float(a.take(64, dev).abs().max()) == 0.0


def test_gc_park_policy_of_the_training_loop():
    """optim.GcPark (the ONE garbage-collector policy of tools/train.py and bench.py): automatic collection stays on
    during warm-up, is parked (everything alive frozen) from step `warmup` on, collect() runs a collection at the loop's quiet
    points, release() restores the interpreter's state."""
    import gc
    from das_amd.optim import GcPark
    assert gc.isenabled()
    g = GcPark(warmup=3)
    try:
        g.step(); g.step()
        assert gc.isenabled() and not g.parked
        g.collect()                       # (a no-op before parking)
        g.step()
        assert g.parked and not gc.isenabled() and gc.get_freeze_count() > 0

        class Node:
            pass
        a, b = Node(), Node()
        a.other, b.other = b, a           # a reference cycle: only the cyclic collector frees it
        import weakref
        w = weakref.ref(a)
        del a, b
        assert w() is not None
        g.collect()
        assert w() is None
    finally:
        g.release()
    assert gc.isenabled() and gc.get_freeze_count() == 0 and not g.parked
