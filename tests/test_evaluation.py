"""das_amd's dataset classes / evaluators against fixtures produced by the REFERENCE's own
dataset classes on the same seeded synthetic annotations and detections (tests/golden/make_golden_eval.py).
CPU only."""
import json
import os

import numpy as np
import pytest

import eval_cases as EC
from das_amd import evaluation as E
from das_amd.datasets import build_dataset
from das_amd.pose_datasets import CMUPanopticDataset, MuPots3DHP


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + '.npz'))


@pytest.mark.parametrize('tag,kw', [('abs', dict(norm_depth=True, abs_dz=True, depth_factor=1)),
                                    ('df20', dict(norm_depth=True, abs_dz=False, depth_factor=20))])
def test_panoptic_parse_and_mpjpe(golden_dir, tmp_path, tag, kw):
    z = load(golden_dir, f'eval_panoptic_{tag}')
    ann = EC.panoptic_annotation()
    ds = CMUPanopticDataset(ann_file=ann, data_root='/data/panoptic', test_mode=True, **kw)
    parsed = [ds.get_ann_info(i) for i in range(len(ds))]
    for i, p in enumerate(parsed):          # _parse_ann_info: targets bit-identical to the reference's
        np.testing.assert_array_equal(p['gt_poses_3d'], z[f'gt{i}'])
        np.testing.assert_array_equal(p['centers2d'], z[f'c2d{i}'])
        np.testing.assert_array_equal(p['bboxes_ignore'], z[f'ign{i}'])
    assert parsed[0]['bboxes_ignore'].shape[0] == 1      # the person with an invisible root
    outs = EC.panoptic_outputs(ann, [p['gt_poses_3d'] for p in parsed], depth_factor=kw['depth_factor'])
    res = ds.evaluate(outs, res_folder=str(tmp_path))
    assert list(res) == ['MPJPE:'] and res['MPJPE:'].endswith('mm')
    assert abs(float(res['MPJPE:'][:-2]) - float(z['mpjpe'])) <= 0.011      # (the reference prints two decimals)
    with open(tmp_path / 'result_keypoints.json') as f:
        rec = json.load(f)
    assert len(rec) == int(z['n_records'])
    np.testing.assert_allclose(np.array([r['keypoints'] for r in rec]), z['rec_kpts'], rtol=1e-6)
    np.testing.assert_allclose(np.array([r['bbox'] for r in rec]), z['rec_bbox'], rtol=1e-6)
    np.testing.assert_array_equal(np.array([r['image_id'] for r in rec]), z['rec_img'])
    assert set(rec[0]) == {'image_id', 'category_id', 'keypoints', 'score', 'bbox'}
    # evaluation from the written file equals evaluation from memory (do_python_keypoint_eval accepts a folder)
    assert ds.do_python_keypoint_eval(str(tmp_path))[0][1] == res['MPJPE:']
    with pytest.raises(KeyError):
        ds.evaluate(outs, res_folder=str(tmp_path), metric='pck')


def test_mupots_parse_and_pck(golden_dir, tmp_path):
    z = load(golden_dir, 'eval_mupots')
    ann, mats = EC.mupots_annotation()
    root = str(tmp_path / 'mupots')
    EC.write_mupots_mats(root, mats)
    ds = build_dataset(dict(type='MuPots3DHP', ann_file=ann, data_root=root, test_mode=True))
    assert isinstance(ds, MuPots3DHP) and ds.num_joints == 17 and ds.ROOT_IDX == 14
    np.testing.assert_array_equal(ds.get_ann_info(0)['gt_poses_3d'], z['gt0'])
    np.testing.assert_array_equal(ds.get_ann_info(5)['gt_poses_3d'], z['gt5'])
    outs = EC.mupots_outputs(ann, root)
    for mode in ('all', 'matched'):
        res = ds.evaluate(outs, res_folder=str(tmp_path / mode), eval_mode=mode)
        ref = z[f'pck_{mode}']
        assert abs(float(res['PCK_MEAN:']) - ref[0]) <= 0.011 and abs(float(res['PCK_MEAN_ABS:']) - ref[1]) <= 0.011, (res, ref)


def test_mupots_helpers_vs_reference(golden_dir):
    z = load(golden_dir, 'eval_mupots')
    rs = np.random.RandomState(9)
    a, b = rs.normal(0, 200, (3, 17)), rs.normal(0, 200, (3, 17))
    np.testing.assert_allclose(E.procrustes(a.copy(), b.copy()), z['procrustes'], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(E.norm_by_bone_length(a.copy(), b.copy(), E.MPII_O1, E.SAFE_TRAVERSAL[1:]), z['bone'],
                               rtol=1e-12)
    # camera helpers invert each other (mytools/vis_3d.py)
    K = np.array([[1400.0, 0, 960.0], [0, 1390.0, 540.0], [0, 0, 1]])
    R, t = np.eye(3), np.array([[10.0], [-5.0], [300.0]])
    X = rs.normal(0, 50, (3, 9))
    back = E.pixel2world(E.world2pixel(X.copy(), K, R, t), K, R, t)[-1]
    np.testing.assert_allclose(back, X, atol=2e-2)       # (world2pixel divides by z + 1e-5)


def test_match_people_keeps_index_0_for_nan_distances():
    """`match` (mupots_3dhp.py:556-565) tests `diffs.min() > threshold`: False for NaN, so the all-zero stand-in of a frame
    without predictions is matched (index 0), not dropped (-1) — matched-mode PCK counts those persons as misses."""
    rs = np.random.RandomState(5)
    gt = [rs.normal(0, 300, (3, 17)) + np.array([[0.0], [0.0], [3000.0]]) for _ in range(3)]
    with np.errstate(all='ignore'):
        rel, ab = E.match_people(gt, np.zeros((1, 3, 17)), E.MPII_O1, E.SAFE_TRAVERSAL[1:])
    assert rel == [0, 0, 0] and ab == [0, 0, 0]


def test_mupots_helpers_live_against_the_reference_over_random_frames():
    """Authoring container only: the batched helpers against the reference's one-pair-at-a-time functions imported
    from /root/reference (mupots_3dhp.py:480-566) on 40 random frames, incl. frames whose closest prediction is too far."""
    import refstub
    if not os.path.isdir(refstub.REF):
        pytest.skip('reference tree absent (GPU box)')
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'golden'))
    from make_golden_eval import load_ref_datasets
    _, mup = load_ref_datasets()
    o1 = mup.mpii_get_joints('relavant')[1]
    trav = E.SAFE_TRAVERSAL[1:]
    assert o1 == E.MPII_O1
    rs = np.random.RandomState(3)
    for it in range(40):
        G, P = rs.randint(1, 5), rs.randint(1, 7)
        gt = [rs.normal(0, 300, (3, 17)) + np.array([[0.0], [0.0], [3000.0]]) for _ in range(G)]
        pred = np.stack([gt[rs.randint(G)] + rs.normal(0, 40 if rs.rand() < 0.7 else 600, (3, 17)) for _ in range(P)])
        ref = mup.match(gt, pred.copy(), o1, trav)
        got = E.match_people(gt, pred.copy(), o1, trav)
        assert [int(i) for i in ref[0]] == got[0] and [int(i) for i in ref[1]] == got[1], it
        a, b = pred[0], gt[0]
        np.testing.assert_allclose(E.procrustes(a.copy(), b.copy()), mup.procrustes(a.copy(), b.copy()), rtol=1e-9, atol=1e-8)
        np.testing.assert_allclose(E.norm_by_bone_length(a.copy(), b.copy(), o1, trav),
                                   mup.norm_by_bone_length(a.copy(), b.copy(), o1, trav), rtol=1e-12)
    # a frame with no valid prediction: the caller's stand-in is one all-zero pose (root depth 0 -> NaN distances);
    # the reference's `diffs.min() > threshold` is False for NaN, so every GT person is matched to index 0
    gt = [rs.normal(0, 300, (3, 17)) + np.array([[0.0], [0.0], [3000.0]]) for _ in range(2)]
    zero = np.zeros((1, 3, 17))
    with np.errstate(all='ignore'):
        ref = mup.match(gt, zero.copy(), o1, trav)
        got = E.match_people(gt, zero.copy(), o1, trav)
    assert [int(i) for i in ref[0]] == got[0] == [0, 0] and [int(i) for i in ref[1]] == got[1] == [0, 0]
    # batched alignment == pair by pair
    A, B = rs.normal(0, 100, (5, 3, 17)), rs.normal(0, 100, (5, 3, 17))
    np.testing.assert_allclose(E.procrustes(A, B), np.stack([E.procrustes(x, y) for x, y in zip(A, B)]), rtol=1e-12)
