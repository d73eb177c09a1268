"""Generate tests/golden/eval_{panoptic,mupots}.npz from the REFERENCE's dataset classes
(mmdet3d/datasets/cmupanoptic_mono_dataset.py, mupots_3dhp.py) — authoring container only.

The classes derive from mmdet's CocoDataset (not installed): instances are created with object.__new__ and given the
attributes their evaluation / annotation-parsing methods read; the annotation index is das_amd's CocoLite over the
synthetic COCO-style dicts of eval_cases.py (the reference only calls coco.anns / imgs / get_ann_ids / load_anns /
load_imgs on it). cv2 and mmcv are imported by those files but not used on this path: empty stand-ins.
    python tests/golden/make_golden_eval.py
"""
import json
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, 'tests'), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import eval_cases as EC  # noqa: E402
import refstub  # noqa: E402
from das_amd.pose_datasets import CocoLite  # noqa: E402

REF = refstub.REF


def load_ref_datasets():
    if not hasattr(np, 'float'):
        np.float = float
    reg = refstub.Registry('dataset')

    class CocoDataset:      # stand-in base: never constructed
        pass
    refstub._pkg('mmcv')
    refstub._pkg('mmdet')
    refstub._mod('mmdet.datasets', DATASETS=reg, CocoDataset=CocoDataset)
    refstub._mod('mmdet.datasets.builder', DATASETS=reg)
    refstub._mod('mmdet.datasets.coco', CocoDataset=CocoDataset)
    refstub._mod('cv2')
    refstub._pkg('mytools', os.path.join(REF, 'mytools'))
    refstub._load('mytools.vis_3d', 'mytools/vis_3d.py')
    refstub._pkg('mmdet3d', os.path.join(REF, 'mmdet3d'))
    refstub._pkg('mmdet3d.datasets', os.path.join(REF, 'mmdet3d/datasets'))
    pan = refstub._load('mmdet3d.datasets.cmupanoptic_mono_dataset', 'mmdet3d/datasets/cmupanoptic_mono_dataset.py')
    mup = refstub._load('mmdet3d.datasets.mupots_3dhp', 'mmdet3d/datasets/mupots_3dhp.py')
    return pan, mup


def fake(cls, coco, **attrs):
    o = object.__new__(cls)
    o.coco = coco
    o.cat_ids = [1]
    o.cat2label = {1: 0}
    o.img_ids = list(coco.imgs.keys())
    o.test_mode = True
    for k, v in attrs.items():
        setattr(o, k, v)
    return o


def main():
    pan, mup = load_ref_datasets()
    # ---- CMU Panoptic: annotation parsing + MPJPE
    for tag, kw in (('abs', dict(norm_depth=True, abs_dz=True, depth_factor=1)),
                    ('df20', dict(norm_depth=True, abs_dz=False, depth_factor=20))):
        ann = EC.panoptic_annotation()
        coco = CocoLite(ann)
        ds = fake(pan.CMUPanopticDataset, coco, num_joints=15, use_bbox_center=False, data_root='/data/panoptic', **kw)
        ds.name2id = {os.path.basename(i['file_name']): i['id'] for i in ann['images']}
        parsed = [ds._parse_ann_info(coco.load_imgs([i])[0], coco.load_anns(coco.get_ann_ids(img_ids=[i]))) for i in ds.img_ids]
        outs = EC.panoptic_outputs(ann, [p['gt_poses_3d'] for p in parsed], depth_factor=kw['depth_factor'])
        with tempfile.TemporaryDirectory() as td:
            res = ds.evaluate(outs, res_folder=td)
            with open(os.path.join(td, 'result_keypoints.json')) as f:
                records = json.load(f)
        arrs = {f'gt{i}': p['gt_poses_3d'] for i, p in enumerate(parsed)}
        arrs.update({f'c2d{i}': p['centers2d'] for i, p in enumerate(parsed)})
        arrs.update({f'ign{i}': p['bboxes_ignore'] for i, p in enumerate(parsed)})
        np.savez_compressed(os.path.join(HERE, f'eval_panoptic_{tag}.npz'), mpjpe=np.array(float(res['MPJPE:'][:-2])),
                            n_records=np.array(len(records)), rec_kpts=np.array([r['keypoints'] for r in records]),
                            rec_bbox=np.array([r['bbox'] for r in records]), rec_img=np.array([r['image_id'] for r in records]),
                            **arrs)
        print('wrote eval_panoptic', tag, res, len(records))

    # ---- MuPoTS-3D: annotation parsing + 3DPCK (relative / absolute), both evaluation modes
    ann, mats = EC.mupots_annotation()
    with tempfile.TemporaryDirectory() as root:
        EC.write_mupots_mats(root, mats)
        coco = CocoLite(ann)
        ds = fake(mup.MuPots3DHP, coco, num_joints=17, use_bbox_center=False, norm_depth=False, abs_dz=False, depth_factor=1,
                  data_root=root)
        ds.name2id = {i['file_name']: i['id'] for i in ann['images']}
        parsed = [ds._parse_ann_info(coco.load_imgs([i])[0], coco.load_anns(coco.get_ann_ids(img_ids=[i]))) for i in ds.img_ids]
        outs = EC.mupots_outputs(ann, root)
        out = {}
        for mode in ('all', 'matched'):
            with tempfile.TemporaryDirectory() as td:
                res = ds.evaluate(outs, res_folder=td, eval_mode=mode)
            out[mode] = (float(res['PCK_MEAN:']), float(res['PCK_MEAN_ABS:']))
            print('mupots', mode, res)
        # the pure helpers on a fixed pair of poses
        rs = np.random.RandomState(9)
        a, b = rs.normal(0, 200, (3, 17)), rs.normal(0, 200, (3, 17))
        np.savez_compressed(os.path.join(HERE, 'eval_mupots.npz'), pck_all=np.array(out['all']),
                            pck_matched=np.array(out['matched']), gt0=parsed[0]['gt_poses_3d'], gt5=parsed[5]['gt_poses_3d'],
                            procrustes=mup.procrustes(a.copy(), b.copy()),
                            bone=mup.norm_by_bone_length(a.copy(), b.copy(), mup.mpii_get_joints('relavant')[1],
                                                         [i - 1 for i in [16, 2, 1, 17, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14]]))


if __name__ == '__main__':
    main()
