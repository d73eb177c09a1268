"""Training-dataset classes and raw-data converters against fixtures produced by the reference's
own code on the same seeded synthetic inputs (tests/golden/make_golden_datasets.py):
  MuCo3DHPDataset / COCOKeypointsDataset._parse_ann_info  (mmdet3d/datasets/muco_3dhp.py:124-246,
                                                           coco_keypoints_dataset.py:133-287 incl. the joint remap)
  tools/convert_panoptic.py, tools/convert_muco.py         (mytools/panoptic2coco.py, mytools/muco2coco.py)"""
import gzip
import importlib.util
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, 'golden'))
import dataset_cases as DC  # noqa: E402


def tool(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, 'tools', name + '.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def check_parsed(ds, z):
    none = z['none']
    assert len(ds) == len(none)
    for i in range(len(ds)):
        p = ds.get_ann_info(i)
        assert (p is None) == bool(none[i]), i
        if p is not None:
            for k in ('bboxes', 'labels', 'gt_poses_3d', 'centers2d', 'depths', 'bboxes_ignore'):
                ref = z[f'{k}{i}']
                assert p[k].dtype == ref.dtype and np.array_equal(p[k], ref), (i, k)


@pytest.mark.parametrize('tag,kw', [('plain', dict(norm_depth=False, abs_dz=False, depth_factor=1, use_bbox_center=False)),
                                    ('abs', dict(norm_depth=True, abs_dz=True, depth_factor=20, use_bbox_center=True))])
def test_muco_annotation_parsing_matches_the_reference(golden_dir, tag, kw):
    from das_amd.pose_datasets import MuCo3DHPDataset
    ds = MuCo3DHPDataset(DC.muco_annotation(), **kw)
    check_parsed(ds, np.load(os.path.join(golden_dir, f'datasets_muco_{tag}.npz')))
    assert ds.get_ann_info(0)['cam']['K'].shape == (2, 3)


@pytest.mark.parametrize('tag,kw', [('panoptic', dict(convert_ids='panoptic', use_bbox_center=False)),
                                    ('muco', dict(convert_ids='muco', use_bbox_center=True)),
                                    ('raw', dict(convert_ids=None, use_bbox_center=False))])
def test_coco_keypoints_parsing_and_joint_remap_match_the_reference(golden_dir, tag, kw):
    from das_amd.pose_datasets import COCOKeypointsDataset
    ds = COCOKeypointsDataset(ann_file=DC.coco_annotation(), **kw)
    z = np.load(os.path.join(golden_dir, f'datasets_coco_{tag}.npz'))
    check_parsed(ds, z)
    width = {'panoptic': 15, 'muco': 21, 'raw': 17}[tag]
    first = next(i for i in range(len(ds)) if not z['none'][i])
    assert ds.get_ann_info(first)['gt_poses_3d'].shape[1] == 3 + 4 * width


def test_dataset_types_resolve_through_the_registry():
    from das_amd.datasets import build_dataset
    ds = build_dataset(dict(type='COCOKeypointsDataset', ann_file=DC.coco_annotation(), convert_ids='panoptic'))
    assert type(ds).__name__ == 'COCOKeypointsDataset' and len(ds) == 6
    ds = build_dataset(dict(type='MuCo3DHPDataset', ann_file=DC.muco_annotation(), pipeline=None))
    assert type(ds).__name__ == 'MuCo3DHPDataset'
    both = build_dataset([dict(type='COCOKeypointsDataset', ann_file=DC.coco_annotation(), convert_ids='panoptic', classes=('person',)),
                          dict(type='MuCo3DHPDataset', ann_file=DC.muco_annotation())])
    assert len(both) == 12 and both.datasets[1] is not None and both.cumulative_sizes == [6, 12]


def same(a, b, path=''):
    """JSON equality with float tolerance 0: the converters do the reference's arithmetic in the same order."""
    if isinstance(a, dict):
        assert isinstance(b, dict) and set(a) == set(b), (path, set(a) ^ set(b))
        for k in a:
            same(a[k], b[k], f'{path}/{k}')
    elif isinstance(a, list):
        assert isinstance(b, list) and len(a) == len(b), (path, len(a), len(b))
        for i, (x, y) in enumerate(zip(a, b)):
            same(x, y, f'{path}[{i}]')
    else:
        assert a == b, (path, a, b)


def test_panoptic_converter_matches_the_reference(golden_dir, tmp_path):
    conv = tool('convert_panoptic')
    DC.write_panoptic_tree(str(tmp_path))
    quiet = lambda *a: None
    train = conv.convert_split(str(tmp_path), **conv.SPLITS['train'], log=quiet)
    val = conv.convert_split(str(tmp_path), ['160226_haggling1', '160422_haggling1'], interval=None, total=80, strict=False,
                             log=quiet)
    for name, db in (('train', train), ('val', val)):
        with gzip.open(os.path.join(golden_dir, f'convert_panoptic_{name}.json.gz'), 'rt') as f:
            ref = json.load(f)
        assert len(db['images']) == len(ref['images']) > 10
        same(json.loads(json.dumps(db)), ref)
    # the files the tool writes load straight into the dataset class
    from das_amd.pose_datasets import CMUPanopticDataset
    ds = CMUPanopticDataset(ann_file=json.loads(json.dumps(train)), data_root=str(tmp_path))
    assert any(ds.get_ann_info(i) is not None for i in range(len(ds)))


def test_muco_converter_matches_the_reference(golden_dir, tmp_path):
    conv = tool('convert_muco')
    DC.write_muco_tree(str(tmp_path))
    with open(tmp_path / 'annotations/MuCo-3DHP.json') as f:
        db = json.load(f)
    got = {name + '.json': sub for name, sub in conv.subsets(db)}
    with gzip.open(os.path.join(golden_dir, 'convert_muco.json.gz'), 'rt') as f:
        ref = json.load(f)
    same(json.loads(json.dumps(got)), ref)
