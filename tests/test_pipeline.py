"""Pose data pipeline, host half: the annotation transforms of das_amd/pipelines.py against
fixtures captured from the reference's own transforms (tests/golden/make_golden_pipeline.py ->
mmdet3d/datasets/pipelines/transforms_3d.py:32-56,293-318,864-898,988-1058), the random-number call order, and the
numpy oracle of the image ops (oracle/pipeline.py) on cases with a known answer."""
import copy
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import pipeline_cases as PC  # noqa: E402


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, f'pipeline_{name}.npz'))


@pytest.mark.parametrize('case', PC.CASES, ids=[c[0] for c in PC.CASES])
def test_annotation_transforms_match_the_reference_bit_for_bit(golden_dir, case):
    from das_amd import pipelines as P
    name, seed, sf, (scale_depth, abs_dz), (rot, scale, trans), ubc = case
    z = load(golden_dir, name)
    res = PC.annotations(seed)
    res['scale_factor'] = np.array([sf[0], sf[1], sf[0], sf[1]], dtype=np.float32)
    r1 = copy.deepcopy(res)
    P.resize_pose(r1, scale_depth, abs_dz)
    for k in ('gt_poses_3d', 'centers2d', 'depths'):
        assert r1[k].dtype == z['resize_' + k].dtype and np.array_equal(r1[k], z['resize_' + k]), k
    r2 = copy.deepcopy(r1)
    r2['gt_poses_3d'] = r2['gt_poses_3d'].astype(np.float32)
    P.flip_pose(r2, PC.J, PC.FLIP_PAIRS)
    for k in ('gt_poses_3d', 'centers2d'):
        assert np.array_equal(r2[k], z['flip_' + k]), k
    g = P.GlobalRotScaleTransPose(rot_range=[0, 0], scale_ratio_range=[1, 1], translation_std=[0, 0], num_joints=PC.J,
                                  scale_depth=scale_depth, abs_dz=abs_dz, img_norm_cfg=PC.IMG_NORM, use_bbox_center=ubc)
    r3 = copy.deepcopy(res)
    r3['pcd_rot'], r3['pcd_scale_factor'], r3['pcd_trans'] = rot, scale, np.array(trans)
    M = g.matrix(r3)
    out = P.warp_annotations(r3, M, scale, PC.J, scale_depth, abs_dz, ubc)
    assert (out is None) == bool(z['warp_dropped'])
    if out is not None:
        assert np.array_equal(M, z['warp_transform_mat'])
        for k in ('gt_poses_3d', 'centers2d', 'depths', 'gt_bboxes', 'gt_labels'):
            assert np.array_equal(np.asarray(out[k]), z['warp_' + k]), k
        assert g.img_mean == PC.IMG_NORM['mean'][::-1]


def test_random_draws_follow_the_reference_call_order():
    """Same numpy seed -> the draws mmdet's Resize.random_sample / RandomFlip / PhotoMetricDistortion and the reference's
    GlobalRotScaleTransPose (transforms_3d.py:1060-1091,1112-1116) make, in their order."""
    from das_amd import pipelines as P
    np.random.seed(5)
    r = {}
    P.ResizePose(img_scale=[(1333, 512), (1333, 640)], multiscale_mode='range', keep_ratio=True)._random_scale(r)
    np.random.seed(5)
    long_edge = np.random.randint(1333, 1334)
    short_edge = np.random.randint(512, 641)
    assert r['scale'] == (long_edge, short_edge)
    pm = P.PhotoMetricDistortion(32, (0.7, 1.3), (0.7, 1.3), 18)
    np.random.seed(9)
    p = pm.draw()
    np.random.seed(9)
    q = dict(brightness=None, contrast=None, saturation=None, hue=None, perm=None)
    if np.random.randint(2):
        q['brightness'] = np.random.uniform(-32, 32)
    mode = np.random.randint(2)
    if mode == 1 and np.random.randint(2):
        q['contrast'] = np.random.uniform(0.7, 1.3)
    if np.random.randint(2):
        q['saturation'] = np.random.uniform(0.7, 1.3)
    if np.random.randint(2):
        q['hue'] = np.random.uniform(-18, 18)
    if mode == 0 and np.random.randint(2):
        q['contrast'] = np.random.uniform(0.7, 1.3)
    if np.random.randint(2):
        q['perm'] = np.random.permutation(3)
    for k in ('brightness', 'contrast', 'saturation', 'hue'):
        assert p[k] == q[k], k
    assert (p['perm'] is None) == (q['perm'] is None) and (p['perm'] is None or list(p['perm']) == list(q['perm']))
    g = P.GlobalRotScaleTransPose(rot_range=[-0.15, 0.15], scale_ratio_range=[0.8, 1.2], translation_std=[0.15, 0.15],
                                  num_joints=15)
    np.random.seed(3)
    r = {}
    g.draw(r)
    np.random.seed(3)
    rot = np.random.uniform(-0.15, 0.15)
    sc = np.random.uniform(0.8, 1.2)
    tr = np.random.normal(scale=np.array([0.15, 0.15], dtype=np.float32), size=2).T
    assert r['pcd_rot'] == rot / np.pi * 180 and r['pcd_scale_factor'] == sc and np.array_equal(r['pcd_trans'], tr)


def test_oracle_image_ops_known_answers():
    from oracle import pipeline as O
    rng = np.random.RandomState(0)
    img = rng.uniform(0, 255, (23, 31, 3)).astype(np.float32)
    assert np.array_equal(O.resize_bilinear(img, (31, 23)), img)                 # same size: identity
    up = O.resize_bilinear(img, (62, 46))
    assert up.shape == (46, 62, 3) and abs(float(up.mean()) - float(img.mean())) < 1.0
    eye = np.array([[1.0, 0, 0], [0, 1.0, 0]])
    assert np.array_equal(O.warp_affine(img, eye, (31, 23), [1, 2, 3]), img)     # identity map
    sh = O.warp_affine(img, np.array([[1.0, 0, 2], [0, 1.0, 0]]), (31, 23), [1, 2, 3])
    assert np.array_equal(sh[:, 2:], img[:, :-2]) and np.array_equal(sh[:, 0], np.tile(np.float32([1, 2, 3]), (23, 1)))
    back = O.hsv2bgr(O.bgr2hsv(img))
    np.testing.assert_allclose(back, img, rtol=0, atol=2e-3)
    assert np.array_equal(O.photometric(img, dict(contrast_first=True)), back)
    n = O.normalize(img, [123.675, 116.28, 103.53], [58.395, 57.12, 57.375], True)
    ref = ((img[..., ::-1].astype(np.float64) - [123.675, 116.28, 103.53]).astype(np.float32).astype(np.float64)
           * (1 / np.float64([58.395, 57.12, 57.375]))).astype(np.float32)
    assert np.array_equal(n, ref)
    p = O.pad_to_multiple(n, 32)
    assert p.shape == (32, 32, 3) and np.array_equal(p[:23, :31], n) and not p[23:].any()
    assert O.rescale_size(960, 540, (1333, 512)) == (910, 512)
