"""Pose data pipeline, image half on the GPU (das_amd/csrc/augment.hip through das_amd.image_ops): every kernel against
the numpy oracle (oracle/pipeline.py) BIT-EXACTLY — they are float32 gathers / elementwise formulas with a fixed order
of operations — and one full `Compose` chain of the reference's train pipeline (configs/das/exp_panoptic.py:59-98)
against the oracle chain driven by the same seed."""
import copy
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import pipeline_cases as PC  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def image(seed, h, w):
    r = np.random.RandomState(seed)
    img = r.uniform(0, 255, (h, w, 3)).astype(np.float32)
    img[: h // 3, : w // 4] = r.randint(0, 256, 3).astype(np.float32)      # a flat patch (s = 0 / ties in max)
    img[h // 2, :, :] = np.float32(200)                                     # grey row
    return img


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize('hw,size', [((37, 53), (41, 29)), ((540, 960), (910, 512)), ((64, 48), (48, 64)), ((33, 33), (99, 66))])
def test_resize_bilinear_bit_exact(hw, size):
    from das_amd import image_ops as I
    from oracle import pipeline as O
    img = image(1, *hw)
    assert np.array_equal(I.resize_bilinear(dev(img), size).cpu().numpy(), O.resize_bilinear(img, size))


def test_flip_bit_exact():
    from das_amd import image_ops as I
    img = image(2, 31, 45)
    assert np.array_equal(I.flip_horizontal(dev(img)).cpu().numpy(), img[:, ::-1])


PHOTO = [
    dict(brightness=17.3, contrast=1.21, contrast_first=True, saturation=0.8, hue=11.0, perm=[2, 0, 1]),
    dict(brightness=-30.5, contrast=0.74, contrast_first=False, saturation=1.27, hue=-17.5, perm=None),
    dict(brightness=None, contrast=None, contrast_first=True, saturation=None, hue=None, perm=[1, 2, 0]),
    dict(brightness=None, contrast=1.3, contrast_first=False, saturation=None, hue=17.9, perm=None),
]


@pytest.mark.parametrize('p', PHOTO)
def test_photometric_bit_exact(p):
    from das_amd import image_ops as I
    from oracle import pipeline as O
    img = image(3, 45, 61)
    got = I.photometric_(dev(img), **p).cpu().numpy()
    ref = O.photometric(img, p)
    assert np.array_equal(got, ref), float(np.abs(got - ref).max())


WARPS = [
    (np.array([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]), (53, 37)),
    (np.array([[1.2048, 0.0, -115.2], [0.0, 1.2048, 31.7]]), (53, 37)),
    (np.array([[0.7071, -0.5, 12.3], [0.5, 0.7071, -4.4]]), (53, 37)),
    (np.array([[0.61, 0.02, 200.5], [-0.02, 0.61, 150.25]]), (960, 540)),
]


@pytest.mark.parametrize('M,size', WARPS)
def test_warp_affine_bit_exact(M, size):
    from das_amd import image_ops as I
    from oracle import pipeline as O
    img = image(4, size[1], size[0])
    border = [103.53, 116.28, 123.675]
    got = I.warp_affine(dev(img), M, size, border).cpu().numpy()
    ref = O.warp_affine(img, M, size, border)
    assert np.array_equal(got, ref), float(np.abs(got - ref).max())


@pytest.mark.parametrize('mean,std,to_rgb', [([123.675, 116.28, 103.53], [58.395, 57.12, 57.375], True),
                                             ([127.0, 127.0, 127.0], [1.0, 1.0, 1.0], False)])
def test_normalize_pad_bit_exact(mean, std, to_rgb):
    from das_amd import image_ops as I
    from oracle import pipeline as O
    img = image(5, 45, 70)
    got = I.normalize_pad_chw(dev(img), mean, std, to_rgb, (64, 96)).cpu().numpy()
    ref = O.pad_to_multiple(O.normalize(img, mean, std, to_rgb), 32).transpose(2, 0, 1)
    assert got.shape == (3, 64, 96) and np.array_equal(got, ref)


def test_train_pipeline_chain_against_the_oracle_chain(tmp_path):
    """The reference's train pipeline through das_amd.pipelines.Compose on the GPU; the oracle chain re-applies the same
    drawn parameters with numpy. Image bit-exact, annotations identical (they are the same host code)."""
    from das_amd import pipelines as P
    from oracle import pipeline as O
    h, w = 270, 480
    img = image(7, h, w)
    np.save(tmp_path / 'frame.npy', img)
    ann = PC.annotations(21, n=5, h=h, w=w)
    cfg = [
        dict(type='LoadImageFromFile', to_float32=True),
        dict(type='LoadAnnotationsPose3D', with_bbox=True, with_label=True),
        dict(type='ResizePose', scale_depth=True, abs_dz=False, img_scale=[(667, 256), (667, 320)], multiscale_mode='range',
             keep_ratio=True),
        dict(type='RandomFlipPose3D', flip_ratio_bev_horizontal=1.0, flip_pairs=PC.FLIP_PAIRS, num_joints=PC.J),
        dict(type='PhotoMetricDistortion', brightness_delta=32, contrast_range=(0.7, 1.3), saturation_range=(0.7, 1.3),
             hue_delta=18),
        dict(type='GlobalRotScaleTransPose', scale_depth=True, abs_dz=False, rot_range=[-0.1, 0.1],
             scale_ratio_range=[0.9, 1.1], translation_std=[0.02, 0.02], num_joints=PC.J, img_norm_cfg=PC.IMG_NORM,
             use_bbox_center=False),
        dict(type='Normalize', **PC.IMG_NORM),
        dict(type='Pad', size_divisor=32),
        dict(type='DefaultFormatBundlePose3D', class_names=['person']),
        dict(type='Collect3D', keys=['img', 'gt_bboxes', 'gt_labels', 'gt_poses_3d', 'gt_labels_3d', 'centers2d', 'depths']),
    ]
    pipe = P.Compose(cfg)
    src = dict(img_info=dict(filename=str(tmp_path / 'frame.npy')), img_prefix=None,
               ann_info=dict(bboxes=ann['gt_bboxes'], labels=ann['gt_labels'], centers2d=ann['centers2d'],
                             depths=ann['depths'], gt_poses_3d=ann['gt_poses_3d'], gt_labels_3d=ann['gt_labels_3d']))
    np.random.seed(123)
    out = pipe(copy.deepcopy(src))
    assert out is not None
    meta = out['img_metas']
    # ---- oracle chain with the parameters the pipeline drew
    sf = meta['scale_factor']
    new_h, new_w = meta['img_shape'][:2]
    ref = O.resize_bilinear(img, (new_w, new_h))
    ref = O.flip_horizontal(ref)
    np.random.seed(123)
    np.random.randint(667, 668); np.random.randint(256, 321)                      # ResizePose
    np.random.choice(['horizontal', None], p=[1.0, 0.0])                          # RandomFlip
    p = P.PhotoMetricDistortion(32, (0.7, 1.3), (0.7, 1.3), 18).draw()
    ref = O.photometric(ref, p)
    ref = O.warp_affine(ref, meta['transform_mat'], (new_w, new_h), PC.IMG_NORM['mean'][::-1])
    # (mmdet's Normalize keeps mean / std as float32 arrays; mmcv.imnormalize widens THOSE to f64)
    ref = O.normalize(ref, np.float32(PC.IMG_NORM['mean']), np.float32(PC.IMG_NORM['std']), True)
    ref = O.pad_to_multiple(ref, 32).transpose(2, 0, 1)
    got = out['img'].cpu().numpy()
    assert got.shape == ref.shape and got.shape[1] % 32 == 0 and got.shape[2] % 32 == 0
    assert np.array_equal(got, ref), float(np.abs(got - ref).max())
    assert out['gt_poses_3d'].shape[1] == 3 + 4 * PC.J and out['gt_poses_3d'].dtype == torch.float32
    assert len(out['gt_bboxes']) == len(out['gt_poses_3d']) == len(out['gt_labels']) == len(out['depths'])
    assert meta['flip'] and tuple(meta['pad_shape'][:2]) == got.shape[1:]
    assert abs(float(sf[0]) - new_w / w) < 1e-6


def test_pipeline_output_feeds_a_train_step(tmp_path):
    """Two frames through the GPU pipeline, collated, into `model.train_step` (tiny-width DAS): finite losses."""
    import bench
    from das_amd import pipelines as P
    from das_amd.datasets import collate
    h, w = 256, 384
    samples = []
    pipe = P.Compose([
        dict(type='LoadImageFromFile', to_float32=True),
        dict(type='LoadAnnotationsPose3D', with_bbox=True, with_label=True),
        dict(type='ResizePose', scale_depth=True, img_scale=[(384, 256)], keep_ratio=True),
        dict(type='RandomFlipPose3D', flip_ratio_bev_horizontal=0.5, flip_pairs=PC.FLIP_PAIRS, num_joints=PC.J),
        dict(type='PhotoMetricDistortion'),
        dict(type='GlobalRotScaleTransPose', scale_depth=True, rot_range=[-0.05, 0.05], scale_ratio_range=[0.95, 1.05],
             translation_std=[0.01, 0.01], num_joints=PC.J, img_norm_cfg=PC.IMG_NORM),
        dict(type='Normalize', **PC.IMG_NORM),
        dict(type='Pad', size_divisor=32),
        dict(type='DefaultFormatBundlePose3D', class_names=['person']),
        dict(type='Collect3D', keys=['img', 'gt_bboxes', 'gt_labels', 'gt_poses_3d', 'gt_labels_3d', 'centers2d', 'depths']),
    ])
    np.random.seed(1)
    for i in range(2):
        np.save(tmp_path / f'f{i}.npy', image(30 + i, h, w))
        ann = PC.annotations(40 + i, n=3, h=h, w=w)
        out = pipe(dict(img_info=dict(filename=str(tmp_path / f'f{i}.npy')), img_prefix=None,
                        ann_info=dict(bboxes=ann['gt_bboxes'], labels=ann['gt_labels'], centers2d=ann['centers2d'],
                                      depths=ann['depths'], gt_poses_3d=ann['gt_poses_3d'], gt_labels_3d=ann['gt_labels_3d'])))
        assert out is not None
        samples.append(out)
    data = collate(samples, device=DEV)
    assert data['img'].shape == (2, 3, 256, 384) and data['img'].is_cuda
    model = bench.build_model(torch.device(DEV), seed=0, dtype='f32', num_stages=1, train=True)
    res = model.train_step(data, None)
    assert np.isfinite(float(res['loss'].detach())) and res['num_samples'] == 2


@pytest.mark.parametrize('hw,size', [((37, 53), (41, 29)), ((540, 960), (1138, 640)), ((64, 48), (48, 64))])
def test_resize_uint8_bit_exact(hw, size):
    """the test pipeline resizes the decoded 8-bit frame: OpenCV's fixed-point INTER_LINEAR"""
    from das_amd import image_ops as I
    from oracle import pipeline as O
    img = np.random.RandomState(11).randint(0, 256, hw + (3,)).astype(np.uint8)
    got = I.resize_bilinear(dev(img), size).cpu().numpy()
    assert got.dtype == np.uint8 and np.array_equal(got, O.resize_bilinear_u8(img, size))


def test_recorded_image_ops_replay_bit_identically(tmp_path):
    """The worker-process path of das_amd.loader.ProcessLoader: with `pipelines.DEFER_IMAGE_OPS` the stages record their
    image ops on a FramePlan (no GPU call) and draw / move the annotations as ever; the replay on the GPU must give the
    immediate path's image bit for bit, into a batch slot as well, for both frame kinds (.npy BGR, PIL-decoded RGB) —
    and the plan must survive pickling (it crosses a process boundary)."""
    import pickle
    from PIL import Image
    from das_amd import pipelines as P
    from das_amd.image_ops import FramePlan
    h, w = 270, 480
    frame = np.random.RandomState(5).randint(0, 256, (h, w, 3)).astype(np.uint8)
    np.save(tmp_path / 'frame.npy', frame)
    Image.fromarray(frame).save(tmp_path / 'frame.png')
    ann = PC.annotations(21, n=5, h=h, w=w)
    cfg = [
        dict(type='LoadImageFromFile', to_float32=True),
        dict(type='LoadAnnotationsPose3D', with_bbox=True, with_label=True),
        dict(type='ResizePose', scale_depth=True, abs_dz=False, img_scale=[(667, 256), (667, 320)], multiscale_mode='range',
             keep_ratio=True),
        dict(type='RandomFlipPose3D', flip_ratio_bev_horizontal=0.5, flip_pairs=PC.FLIP_PAIRS, num_joints=PC.J),
        dict(type='PhotoMetricDistortion', brightness_delta=32, contrast_range=(0.7, 1.3), saturation_range=(0.7, 1.3),
             hue_delta=18),
        dict(type='GlobalRotScaleTransPose', scale_depth=True, abs_dz=False, rot_range=[-0.1, 0.1],
             scale_ratio_range=[0.9, 1.1], translation_std=[0.02, 0.02], num_joints=PC.J, img_norm_cfg=PC.IMG_NORM,
             use_bbox_center=False),
        dict(type='Normalize', **PC.IMG_NORM),
        dict(type='Pad', size_divisor=32),
        dict(type='DefaultFormatBundlePose3D', class_names=['person']),
        dict(type='Collect3D', keys=['img', 'gt_bboxes', 'gt_labels', 'gt_poses_3d', 'gt_labels_3d', 'centers2d', 'depths']),
    ]
    pipe = P.Compose(cfg)
    for name in ('frame.npy', 'frame.png'):
        src = dict(img_info=dict(filename=str(tmp_path / name)), img_prefix=None,
                   ann_info=dict(bboxes=ann['gt_bboxes'], labels=ann['gt_labels'], centers2d=ann['centers2d'],
                                 depths=ann['depths'], gt_poses_3d=ann['gt_poses_3d'], gt_labels_3d=ann['gt_labels_3d']))
        for seed in (1, 2, 3):
            np.random.seed(seed)
            now = pipe(copy.deepcopy(src))
            P.DEFER_IMAGE_OPS = True
            try:
                np.random.seed(seed)
                later = pipe(copy.deepcopy(src))
            finally:
                P.DEFER_IMAGE_OPS = False
            if now is None:
                assert later is None
                continue
            plan = later['img']
            assert isinstance(plan, FramePlan) and tuple(plan.shape) == tuple(now['img'].shape)
            plan = pickle.loads(pickle.dumps(plan))
            assert torch.equal(plan.run('cuda'), now['img'])
            batch = torch.full((2,) + tuple(plan.shape), 7.0, device='cuda')
            plan.run('cuda', out=batch[1])
            assert torch.equal(batch[1], now['img']) and float(batch[0].min()) == 7.0
            for k in ('gt_bboxes', 'gt_labels', 'gt_poses_3d', 'gt_labels_3d', 'centers2d', 'depths'):
                assert torch.equal(later[k], now[k]), k
            assert later['img_metas']['flip'] == now['img_metas']['flip']


def test_test_pipeline_chain(tmp_path):
    """configs/das/exp_panoptic.py:138-155: MultiScaleFlipAug(img_scale, flip=False)[Resize, RandomFlipPose3D(0),
    Normalize, Pad, DefaultFormatBundlePose3D, Collect3D] on an 8-bit frame."""
    from das_amd import pipelines as P
    from oracle import pipeline as O
    h, w = 270, 480
    img = np.random.RandomState(12).randint(0, 256, (h, w, 3)).astype(np.uint8)
    np.save(tmp_path / 'frame.npy', img)
    ann = PC.annotations(22, n=3, h=h, w=w)
    pipe = P.Compose([
        dict(type='LoadImageFromFile'),
        dict(type='LoadAnnotationsPose3D', with_pose_3d=True, with_label_3d=False),
        dict(type='MultiScaleFlipAug', img_scale=(667, 320), flip=False, transforms=[
            dict(type='Resize', keep_ratio=True),
            dict(type='RandomFlipPose3D', flip_ratio_bev_horizontal=0.0, flip_pairs=PC.FLIP_PAIRS, num_joints=PC.J),
            dict(type='Normalize', **PC.IMG_NORM),
            dict(type='Pad', size_divisor=32),
            dict(type='DefaultFormatBundlePose3D', class_names=['person'], with_label=False),
            dict(type='Collect3D', keys=['img', 'gt_poses_3d', 'depths']),
        ])])
    out = pipe(dict(img_info=dict(filename=str(tmp_path / 'frame.npy')), img_prefix=None,
                    ann_info=dict(centers2d=ann['centers2d'], depths=ann['depths'], gt_poses_3d=ann['gt_poses_3d'],
                                  cam=dict(K=np.eye(3)))))
    assert isinstance(out['img'], list) and len(out['img']) == 1
    meta = out['img_metas'][0]
    nw, nh = O.rescale_size(w, h, (667, 320))
    assert tuple(meta['img_shape'][:2]) == (nh, nw) and not meta['flip'] and 'cam' in meta
    ref = O.resize_bilinear_u8(img, (nw, nh)).astype(np.float32)
    ref = O.pad_to_multiple(O.normalize(ref, np.float32(PC.IMG_NORM['mean']), np.float32(PC.IMG_NORM['std']), True), 32)
    got = out['img'][0].cpu().numpy()
    assert np.array_equal(got, ref.transpose(2, 0, 1))
    assert np.array_equal(out['gt_poses_3d'][0].numpy(), ann['gt_poses_3d'])      # plain Resize leaves the poses alone
