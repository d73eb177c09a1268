"""Generate tests/golden/pipeline_<case>.npz from the REFERENCE's transforms (mmdet3d/datasets/pipelines/transforms_3d.py)
— authoring container only. The file imports mmcv / mmdet / cv2 (absent here): empty stand-ins, plus
  * `Resize` / `RandomFlip` base classes that do nothing (the subclasses' own annotation methods are called directly),
  * `cv2.getAffineTransform` = the three-point solve, `cv2.warpAffine` = identity (the image half is not pinned, see
    oracle/pipeline.py).
What is captured is the reference's OWN arithmetic: ResizePose._resize_pose, RandomFlipPose3D.random_flip_data_3d,
get_affine_transform, GlobalRotScaleTransPose._transform (annotation half).
    python tests/golden/make_golden_pipeline.py
"""
import copy
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, 'tests'), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import pipeline_cases as PC  # noqa: E402
import refstub  # noqa: E402

REF = refstub.REF


def three_point(src, dst):
    A = np.zeros((6, 6))
    b = np.zeros(6)
    for i in range(3):
        A[i, :3] = [src[i][0], src[i][1], 1]
        A[i + 3, 3:] = [src[i][0], src[i][1], 1]
        b[i], b[i + 3] = dst[i][0], dst[i][1]
    return np.linalg.solve(A, b).reshape(2, 3)


def load_ref():
    if not hasattr(np, 'float'):
        np.float = float
    reg = refstub.Registry('pipeline')

    class Base:
        def __init__(self, *a, **k):
            pass

        def __call__(self, results):
            return results
    refstub._pkg('mmcv')
    sys.modules['mmcv'].is_tuple_of = lambda *a, **k: True
    refstub._mod('mmcv.utils', build_from_cfg=None)
    refstub._pkg('mmdet')
    refstub._pkg('mmdet.datasets')
    refstub._mod('mmdet.datasets.builder', PIPELINES=reg)
    refstub._mod('mmdet.datasets.pipelines', RandomFlip=Base, Resize=Base)
    refstub._mod('cv2', getAffineTransform=three_point, warpAffine=lambda img, *a, **k: img, INTER_LINEAR=1)
    refstub._pkg('mmdet3d', os.path.join(REF, 'mmdet3d'))
    refstub._mod('mmdet3d.core', VoxelGenerator=None)
    refstub._mod('mmdet3d.core.bbox', CameraInstance3DBoxes=None, DepthInstance3DBoxes=None, LiDARInstance3DBoxes=None,
                 box_np_ops=None)
    refstub._pkg('mmdet3d.datasets', os.path.join(REF, 'mmdet3d/datasets'))
    refstub._mod('mmdet3d.datasets.builder', OBJECTSAMPLERS=refstub.Registry('sampler'))
    refstub._pkg('mmdet3d.datasets.pipelines', os.path.join(REF, 'mmdet3d/datasets/pipelines'))
    refstub._mod('mmdet3d.datasets.pipelines.data_augment_utils', noise_per_object_v3_=None)
    return refstub._load('mmdet3d.datasets.pipelines.transforms_3d', 'mmdet3d/datasets/pipelines/transforms_3d.py')


def main():
    T = load_ref()
    for name, seed, sf, (scale_depth, abs_dz), (rot, scale, trans), ubc in PC.CASES:
        out = {}
        res = PC.annotations(seed)
        res['scale_factor'] = np.array([sf[0], sf[1], sf[0], sf[1]], dtype=np.float32)
        rp = T.ResizePose(scale_depth=scale_depth, abs_dz=abs_dz)
        r1 = copy.deepcopy(res)
        rp._resize_pose(r1)
        for k in ('gt_poses_3d', 'centers2d', 'depths'):
            out['resize_' + k] = r1[k]
        # flip on the resized annotations
        fl = T.RandomFlipPose3D(flip_ratio_bev_horizontal=0.5, num_joints=PC.J, flip_pairs=PC.FLIP_PAIRS)
        r2 = copy.deepcopy(r1)
        r2['gt_poses_3d'] = r2['gt_poses_3d'].astype(np.float32)
        fl.random_flip_data_3d(r2, 'horizontal')
        for k in ('gt_poses_3d', 'centers2d'):
            out['flip_' + k] = r2[k]
        # rotation / scale / translation on the original annotations
        g = T.GlobalRotScaleTransPose(rot_range=[0, 0], scale_ratio_range=[1, 1], translation_std=[0, 0], num_joints=PC.J,
                                      scale_depth=scale_depth, abs_dz=abs_dz, img_norm_cfg=PC.IMG_NORM, use_bbox_center=ubc)
        r3 = copy.deepcopy(res)
        r3['img'] = np.zeros(res['img_shape'], dtype=np.float32)
        r3['pcd_rot'], r3['pcd_scale_factor'], r3['pcd_trans'] = rot, scale, np.array(trans)
        r3 = g._transform(r3)
        out['warp_dropped'] = np.array(r3 is None)
        if r3 is not None:
            for k in ('gt_poses_3d', 'centers2d', 'depths', 'gt_bboxes', 'gt_labels', 'transform_mat'):
                out['warp_' + k] = np.asarray(r3[k])
        np.savez_compressed(os.path.join(HERE, f'pipeline_{name}.npz'), **out)
        print(name, {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
