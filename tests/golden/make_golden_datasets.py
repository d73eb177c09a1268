"""Generate the fixtures of the training-dataset classes and the converters from the REFERENCE (authoring container only):
  datasets_muco_*.npz / datasets_coco_*.npz   MuCo3DHPDataset / COCOKeypointsDataset._parse_ann_info
                                               (mmdet3d/datasets/muco_3dhp.py:124-246, coco_keypoints_dataset.py:133-287)
  convert_panoptic_{train,val}.json.gz         mytools/panoptic2coco.py on the synthetic raw tree of dataset_cases.py
  convert_muco.json.gz                         mytools/muco2coco.py
The dataset classes derive from mmdet's CocoDataset (absent): instances are made with object.__new__ (as in
make_golden_eval.py); cv2 / xtcocotools / mmcv are imported by those files but unused on this path: empty stand-ins.
    python tests/golden/make_golden_datasets.py
"""
import gzip
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, 'tests'), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import dataset_cases as DC  # noqa: E402
import refstub  # noqa: E402
from das_amd.pose_datasets import CocoLite  # noqa: E402
from make_golden_eval import fake  # noqa: E402

REF = refstub.REF


def load_ref():
    if not hasattr(np, 'float'):
        np.float = float
    if not hasattr(np, 'long'):
        np.long = np.int64
    reg = refstub.Registry('dataset')

    class CocoDataset:
        pass
    refstub._pkg('mmcv')
    refstub._pkg('mmdet')
    refstub._mod('mmdet.datasets', DATASETS=reg, CocoDataset=CocoDataset)
    refstub._mod('cv2')
    refstub._pkg('xtcocotools')
    refstub._mod('xtcocotools.cocoeval', COCOeval=None)
    refstub._mod('xtcocotools.coco', COCO=None)
    refstub._pkg('mmdet3d', os.path.join(REF, 'mmdet3d'))
    refstub._pkg('mmdet3d.core', os.path.join(REF, 'mmdet3d/core'))
    refstub._mod('mmdet3d.core.post_processing', oks_nms=None, soft_oks_nms=None)
    refstub._pkg('mmdet3d.utils')
    refstub._mod('mmdet3d.utils.tsv_file', TSVFile=None, CompositeTSVFile=None)
    refstub._mod('mmdet3d.utils.tsv_file_ops', load_linelist_file=None, load_from_yaml_file=None, find_file_path_in_yaml=None)
    refstub._pkg('mmdet3d.datasets', os.path.join(REF, 'mmdet3d/datasets'))
    muco = refstub._load('mmdet3d.datasets.muco_3dhp', 'mmdet3d/datasets/muco_3dhp.py')
    coco = refstub._load('mmdet3d.datasets.coco_keypoints_dataset', 'mmdet3d/datasets/coco_keypoints_dataset.py')
    return muco, coco


def parse_all(ds, coco):
    return [ds._parse_ann_info(coco.load_imgs([i])[0], coco.load_anns(coco.get_ann_ids(img_ids=[i]))) for i in ds.img_ids]


def pack(parsed):
    out = {'none': np.array([p is None for p in parsed])}
    for i, p in enumerate(parsed):
        if p is not None:
            for k in ('bboxes', 'labels', 'gt_poses_3d', 'centers2d', 'depths', 'bboxes_ignore'):
                out[f'{k}{i}'] = np.asarray(p[k])
    return out


def main():
    muco, cocok = load_ref()
    ann = DC.muco_annotation()
    coco = CocoLite(ann)
    for tag, kw in (('plain', dict(norm_depth=False, abs_dz=False, depth_factor=1, use_bbox_center=False)),
                    ('abs', dict(norm_depth=True, abs_dz=True, depth_factor=20, use_bbox_center=True))):
        ds = fake(muco.MuCo3DHPDataset, coco, num_joints=21, **kw)
        ds.test_mode = False
        np.savez_compressed(os.path.join(HERE, f'datasets_muco_{tag}.npz'), **pack(parse_all(ds, coco)))
    ann = DC.coco_annotation()
    coco = CocoLite(ann)
    for tag, kw in (('panoptic', dict(convert_ids='panoptic', use_bbox_center=False)),
                    ('muco', dict(convert_ids='muco', use_bbox_center=True)), ('raw', dict(convert_ids=None, use_bbox_center=False))):
        ds = fake(cocok.COCOKeypointsDataset, coco, num_joints=17, **kw)
        ds.test_mode = False
        np.savez_compressed(os.path.join(HERE, f'datasets_coco_{tag}.npz'), **pack(parse_all(ds, coco)))
    # ---- converters
    import importlib.util
    spec = importlib.util.spec_from_file_location('ref_panoptic2coco', os.path.join(REF, 'mytools/panoptic2coco.py'))
    p2c = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(p2c)
    with tempfile.TemporaryDirectory() as root:
        DC.write_panoptic_tree(root)
        p2c.DATA_ROOT = root
        p2c.tqdm = lambda x: x
        p2c.Panoptic('train')
        with open(os.path.join(root, 'annotations/train.json')) as f:
            train = json.load(f)
        val = object.__new__(p2c.Panoptic)        # a validation split with a smaller frame budget than the hard-coded 2400
        val.joints_def, val.root_id, val.limbs, val.num_joints = p2c.JOINTS_DEF, 2, p2c.LIMBS, 15
        val.dataset_root, val.cam_list = root, [(0, 16), (0, 30)]
        val.sequence_list, val._interval, val._total = ['160226_haggling1', '160422_haggling1'], None, 80
        vdb = json.loads(json.dumps(val._get_db()))
    for name, db in (('train', train), ('val', vdb)):
        with gzip.open(os.path.join(HERE, f'convert_panoptic_{name}.json.gz'), 'wt') as f:
            json.dump(db, f)
        print('panoptic', name, len(db['images']), len(db['annotations']))
    spec = importlib.util.spec_from_file_location('ref_muco2coco', os.path.join(REF, 'mytools/muco2coco.py'))
    m2c = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m2c)
    with tempfile.TemporaryDirectory() as root:
        DC.write_muco_tree(root)
        m2c.main(type('A', (), dict(root=root))())
        outs = {}
        for fn in sorted(os.listdir(os.path.join(root, 'annotations'))):
            if fn.startswith('train'):
                with open(os.path.join(root, 'annotations', fn)) as f:
                    outs[fn] = json.load(f)
    with gzip.open(os.path.join(HERE, 'convert_muco.json.gz'), 'wt') as f:
        json.dump(outs, f)
    print('muco', {k: len(v['images']) for k, v in outs.items()})


if __name__ == '__main__':
    main()
