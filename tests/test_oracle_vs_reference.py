"""Pin the oracle against the reference's own source files, imported by path through
tests/refstub.py. Only runs where /root/reference is mounted (the authoring container);
skipped on the GPU box."""
import numpy as np
import pytest
import torch

import cases
import refstub
from oracle import backbone as ob
from oracle import decode as od
from oracle import head as oh
from oracle import loss as ol

pytestmark = pytest.mark.skipif(not refstub.available(), reason='/root/reference not mounted')


@pytest.fixture(scope='module')
def R():
    return refstub.load()


@pytest.mark.parametrize('stages', [1, 3])
@pytest.mark.parametrize('train', [False, True])
def test_mspn2_live(R, stages, train):
    m = R.MSPN2(unit_channels=16, num_stages=stages, num_blocks=[1, 2, 1, 1], norm_cfg=dict(type='BN'))
    cases.det_fill(m.state_dict(), 9)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m.train(train)
    x = cases.randn(3, 2, 3, 64, 64)
    with torch.no_grad():
        ref = m(x)
        mine = ob.mspn2_forward(sd, x, stages, (1, 2, 1, 1), train=train)
    for a, b in zip(ref, mine):
        assert torch.equal(a, b)
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k]), k


def test_fpn_live(R):
    f = refstub.RefFPN([16] * 4, 24, 4)
    cases.det_fill(f.state_dict(), 2)
    feats = [cases.randn(i, 2, 16, 32 >> i, 48 >> i) for i in range(4)]
    f.eval()
    with torch.no_grad():
        ref = f(feats)
        mine = ob.fpn_forward(f.state_dict(), feats)
    assert [tuple(t.shape[-2:]) for t in mine] == [(16, 24), (8, 12), (4, 6), (2, 3)]
    for a, b in zip(ref, mine):
        assert torch.equal(a, b)


def test_head_full_topology_live(R):
    """exp_mupots-like topology (J=21, root 14, 2 RU layers) at reduced width."""
    J, C = 21, 32
    cfg = dict(num_joints=J, root_idx=14, depth_factor=1, z_norm=50, strides=[8, 16, 32, 64], stacked_convs=2,
               num_heads=4, num_layers=2)
    head = R.DASHead(
        num_classes=1, in_channels=C, feat_channels=C, stacked_convs=2, strides=cfg['strides'],
        regress_ranges=((-1, 80), (80, 160), (160, 320), (320, 1e8)), num_joints=J, depth_factor=1, z_norm=50,
        root_idx=14, cls_branch=(C,), reg_branch=((C,),) * 4, centerness_on_reg=True, conv_bias=True,
        dcn_on_last_conv=True,
        recursive_update=dict(prev_loss=True, num_heads=4, in_channels=C, feat_channels=C, num_layers=2, dim=3,
                              num_joints=J),
        train_cfg=dict(code_weight=[1.0] * (3 + 6 * J)), test_cfg=cases.FULL_TEST_CFG)
    sd = cases.det_fill(head.state_dict(), 4)
    feats = [cases.randn(60 + i, 1, C, 16 >> i, 24 >> i) for i in range(4)]
    for train in (True, False):
        head.train(train)
        with torch.no_grad():
            ref = head(feats)
            mine = oh.head_forward(sd, feats, cfg, '', train)
        for r, m in zip(ref, mine):
            for a, b in zip(r, m):
                assert torch.equal(a, b)


def test_oks_nms_live(R):
    rs = np.random.RandomState(0)
    n, J = 60, 15
    kp = rs.uniform(0, 200, (n, J, 3)).astype(np.float32)
    kp[n // 2:] = kp[:n // 2] + rs.normal(0, 2.0, (n - n // 2, J, 3)).astype(np.float32)  # near-duplicates
    kp[..., 2] = 1
    scores = rs.uniform(0.1, 1, n).astype(np.float32)
    areas = (kp[..., 0].max(1) - kp[..., 0].min(1)) * (kp[..., 1].max(1) - kp[..., 1].min(1))
    db = [dict(score=np.array(scores[i]), keypoints=kp[i], area=np.array(areas[i])) for i in range(n)]
    for thr in (0.9, 0.5, 0.1):
        ref = R.oks_nms(db, thr)
        mine = od.oks_nms(scores, kp.reshape(n, -1), areas, thr)
        assert ref.tolist() == mine.tolist()
        assert 0 < len(mine) <= n


def test_soft_oks_nms_live(R):
    """pose_nms.py:128-194 against the restatement: same kept indices in the same order, several thresholds / caps."""
    from mmdet3d.core.post_processing import pose_nms as ref_nms
    rs = np.random.RandomState(1)
    for n, J in ((60, 15), (90, 17), (40, 21)):
        kp = rs.uniform(0, 200, (n, J, 3)).astype(np.float32)
        kp[n // 2:] = kp[:n // 2] + rs.normal(0, 3.0, (n - n // 2, J, 3)).astype(np.float32)  # near-duplicates
        kp[..., 2] = 1
        scores = rs.uniform(0.1, 1, n).astype(np.float32)
        areas = (kp[..., 0].max(1) - kp[..., 0].min(1)) * (kp[..., 1].max(1) - kp[..., 1].min(1))
        db = [dict(score=np.array(scores[i]), keypoints=kp[i], area=np.array(areas[i])) for i in range(n)]
        for thr, cap in ((0.9, 20), (0.5, 100), (0.1, 7)):
            ref = ref_nms.soft_oks_nms(db, thr, max_dets=cap)
            mine = od.soft_oks_nms(scores, kp.reshape(n, -1), areas, thr, max_dets=cap)
            assert ref.tolist() == mine.tolist()
            assert len(mine) == min(n, cap)
