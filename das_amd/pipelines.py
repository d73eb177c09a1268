"""Pose data pipeline as a GPU-side augmentation stage.

The reference's train pipeline (configs/das/exp_panoptic.py:59-98) is a CPU chain of numpy / OpenCV transforms:
LoadImageFromFile, LoadAnnotationsPose3D (loading.py:671-736), ResizePose (transforms_3d.py:19-61), RandomFlipPose3D
(:235-356), PhotoMetricDistortion (mmdet), GlobalRotScaleTransPose (:901-1129), Normalize, Pad (mmdet),
DefaultFormatBundlePose3D (formating.py:383-442), Collect3D (:83-...). The classes here keep those registry names,
constructor keywords, `results` keys and random-number call order, with the work split in two:

  * annotations (a few persons x (3 + 4 J) floats): numpy on the host — the reference's own arithmetic, restated
    line by line and pinned bit-exactly against the imported reference (tests/golden/pipeline_*.npz);
  * the image: a float32 HWC (BGR) CUDA tensor from the decoded frame on, transformed by the HIP kernels of
    das_amd/csrc/augment.hip (das_amd.image_ops), which restate the OpenCV / mmcv ops the reference calls.
    `Normalize` and `Pad` only record their parameters; `DefaultFormatBundlePose3D` runs them with the HWC -> CHW
    transpose as ONE pass (das_img_normalize_pad_chw), optionally straight into a slot of the batch tensor.

Random draws use numpy's global generator in the reference's order, so a seeded run draws the same augmentation
parameters as the reference pipeline would.
"""
import math
import os
import threading

import numpy as np
import torch

from .datasets import PIPELINES
from .registry import build_from_cfg


class Compose:
    """Chain of transforms; a transform returning None drops the sample (mmdet's Compose semantics)."""

    def __init__(self, transforms):
        self.transforms = [build_from_cfg(t, PIPELINES) if isinstance(t, dict) else t for t in transforms]

    def __call__(self, data):
        for t in self.transforms:
            data = t(data)
            if data is None:
                return None
        return data


def _img_ops():
    from . import image_ops
    return image_ops


# True in the worker processes of das_amd.loader.ProcessLoader (which never touch the GPU): LoadImageFromFile hands the
# decoded frame on as an image_ops.FramePlan, every image op of the later stages is recorded on it instead of run, and
# the trainer replays the record on the GPU. The annotation arithmetic and the random draws are the same code either way.
DEFER_IMAGE_OPS = False


def _is_dev(img):
    return isinstance(img, torch.Tensor) and img.is_cuda


# ---------------------------------------------------------------------------------------------- loading
class _Staging(threading.local):
    """Per-thread page-locked staging buffers for the frame upload, allocated once per frame size and reused (two per
    size, alternating: the copy out of one may still be in flight while the next frame is decoded into the other).
    Why: a host-to-device copy from ordinary (pageable) memory makes the HIP runtime page-lock and unlock the 6 MB
    source around every copy — driver work that stalls the other streams of the GPU: 16 such uploads per step beside
    the trainer cost it a quarter of its throughput, although the data path's own GPU time is 0.2 ms per frame
    (tools/dev/scripts/loader_prof.sh). Allocating page-locked memory per frame is worse still (hipHostMalloc
    synchronises the device); a persistent buffer costs one host memcpy per frame."""
    CAP = 256 << 20      # bytes of page-locked memory per thread; frames beyond that go up from pageable memory

    def __init__(self):
        self.rings, self.bytes = {}, 0

    def upload(self, img, device):
        n = img.nbytes
        ring = self.rings.get(n)
        if ring is None:
            if self.bytes + 2 * n > self.CAP:
                return torch.from_numpy(img).to(device)
            ring = self.rings[n] = [[torch.empty(n, dtype=torch.uint8).pin_memory(), None] for _ in range(2)] + [0]
            self.bytes += 2 * n
        slot = ring[ring[2]]
        ring[2] ^= 1
        if slot[1] is not None:
            slot[1].synchronize()        # the copy that last read this buffer is done (two uploads ago: no wait in practice)
        np.copyto(slot[0].numpy().reshape(img.shape), img)
        t = slot[0].view(img.shape).to(device, non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record()
        return t


_STAGING = _Staging()


def _to_device(img, device):
    """HWC uint8 frame -> device tensor; through this thread's page-locked staging buffers when `device` is a GPU."""
    if torch.device(device).type != 'cuda' or img.dtype != np.uint8:
        return torch.from_numpy(img).to(device)
    return _STAGING.upload(np.ascontiguousarray(img), device)


@PIPELINES.register_module()
class LoadImageFromFile:
    """mmdet LoadImageFromFile(to_float32=True): BGR HWC. The decoded frame goes to the device at once (`device`,
    default 'cuda'); `.npy` files hold an HWC BGR array, anything else is decoded with PIL (OpenCV is not in this image)."""

    def __init__(self, to_float32=False, color_type='color', file_client_args=None, device='cuda'):
        self.to_float32, self.device = to_float32, device

    def __call__(self, results):
        if results.get('img_prefix') is not None:
            filename = os.path.join(results['img_prefix'], results['img_info']['filename'])
        else:
            filename = results['img_info']['filename']
        if DEFER_IMAGE_OPS:
            rgb = not filename.endswith('.npy')
            if rgb:
                from PIL import Image
                pil = Image.open(filename)
                img = np.array(pil if pil.mode == 'RGB' else pil.convert('RGB'))
            else:
                img = np.ascontiguousarray(np.load(filename))
            t = _img_ops().FramePlan(img, rgb=rgb, to_float=self.to_float32)
        elif filename.endswith('.npy'):
            img = np.ascontiguousarray(np.load(filename))
            t = _to_device(img, self.device)
        else:
            # RGB -> BGR on the device: reversing the channel axis of a 1920 x 1080 frame on the host is a strided
            # byte copy of 11 ms that holds the GIL (the JPEG decode itself, 10 ms, releases it) — with it, eight
            # loader threads beside the trainer delivered 108 img/s; the frame goes up as decoded and is flipped there
            from PIL import Image
            pil = Image.open(filename)
            if pil.mode != 'RGB':           # (convert() copies the frame even when there is nothing to convert: 1.2 ms)
                pil = pil.convert('RGB')
            img = np.array(pil)
            t = _to_device(img, self.device).flip(-1)
        if not DEFER_IMAGE_OPS:
            t = (t.float() if self.to_float32 else t).contiguous()
        results['filename'] = filename
        results['ori_filename'] = results['img_info']['filename']
        results['img'] = t
        results['img_shape'] = tuple(img.shape)
        results['ori_shape'] = tuple(img.shape)
        results['img_fields'] = ['img']
        return results


@PIPELINES.register_module()
class LoadAnnotationsPose3D:
    """loading.py:671-736 (+ mmdet LoadAnnotations for bboxes / labels)."""

    def __init__(self, with_pose_3d=True, with_label_3d=True, with_bbox=False, with_label=False, poly2mask=True,
                 file_client_args=None):
        self.with_pose_3d, self.with_label_3d, self.with_bbox, self.with_label = with_pose_3d, with_label_3d, with_bbox, with_label

    def __call__(self, results):
        ann = results['ann_info']
        if self.with_bbox:
            results['gt_bboxes'] = ann['bboxes'].copy()
            if ann.get('bboxes_ignore') is not None:
                results['gt_bboxes_ignore'] = ann['bboxes_ignore'].copy()
                results.setdefault('bbox_fields', []).append('gt_bboxes_ignore')
            results.setdefault('bbox_fields', []).append('gt_bboxes')
        if self.with_label:
            results['gt_labels'] = ann['labels'].copy()
        if self.with_pose_3d:
            results['centers2d'] = ann['centers2d']
            results['depths'] = ann['depths']
            results['gt_poses_3d'] = ann['gt_poses_3d']
        if self.with_label_3d:
            results['gt_labels_3d'] = ann['gt_labels_3d']
        if 'cam' in ann:
            results['cam'] = ann['cam']
        return results


# ---------------------------------------------------------------------------------------------- resize
class PoseTable:
    """Column groups of a `gt_poses_3d` matrix, one row per person:
    [root u, root v, root depth | J x (u, v, dz) | J x visibility]  (4 J + 3 columns, cmupanoptic_mono_dataset.py:171-207).
    `joints` and `vis` are private copies the transforms edit; `pack` writes a matrix in the same layout."""

    def __init__(self, rows, num_joints=None):
        self.n = len(rows)
        self.J = (rows.shape[-1] - 3) // 4 if num_joints is None else num_joints
        split = 3 + 3 * self.J
        self.root, self.depth = rows[:, :2], rows[:, 2]
        self.joints = np.array(rows[:, 3:split]).reshape(self.n, self.J, 3)
        self.vis = np.array(rows[:, split:]).reshape(self.n, self.J)

    def pack(self, root, depth):
        return np.concatenate([root, depth.reshape(-1, 1), self.joints.reshape(self.n, -1), self.vis], axis=-1)


def resize_pose(results, scale_depth, abs_dz):
    """Annotation half of ResizePose (what transforms_3d.py:32-56 computes), in place on `results`: image coordinates
    follow the (w, h) scale factors; with `scale_depth` the root depth shrinks by the geometric mean of the two
    (a resized image looks like the same scene seen from further away), and so do the joints' relative depths
    unless they are absolute (`abs_dz`)."""
    wh = results['scale_factor'][:2]
    table = PoseTable(results['gt_poses_3d'])
    assert np.array_equal(results['centers2d'], table.root) and np.array_equal(results['depths'], table.depth)
    root, depth = results['centers2d'] * wh, results['depths']
    table.joints[..., :2] *= wh
    if scale_depth:
        shrink = np.sqrt(wh.prod())
        depth = depth / shrink
        if not abs_dz:
            table.joints[..., 2] /= shrink
    results['centers2d'], results['depths'] = root, depth
    results['gt_poses_3d'] = table.pack(root, depth)


@PIPELINES.register_module()
class ResizePose:
    """transforms_3d.py:19-61 over mmdet's Resize (multiscale_mode 'range' / 'value', keep_ratio)."""

    def __init__(self, scale_depth=False, abs_dz=False, img_scale=None, multiscale_mode='range', ratio_range=None,
                 keep_ratio=True, bbox_clip_border=True, backend='cv2', override=False):
        self.scale_depth, self.abs_dz = scale_depth, abs_dz
        if abs_dz:
            assert scale_depth
        self.img_scale = None if img_scale is None else (img_scale if isinstance(img_scale, list) else [img_scale])
        assert multiscale_mode in ('value', 'range') and ratio_range is None, 'as the DAS configs use it'
        self.multiscale_mode, self.keep_ratio, self.bbox_clip_border = multiscale_mode, keep_ratio, bbox_clip_border

    def _random_scale(self, results):
        if len(self.img_scale) == 1:
            scale = tuple(self.img_scale[0])
        elif self.multiscale_mode == 'range':      # mmdet Resize.random_sample
            longs, shorts = [max(s) for s in self.img_scale], [min(s) for s in self.img_scale]
            long_edge = np.random.randint(min(longs), max(longs) + 1)
            short_edge = np.random.randint(min(shorts), max(shorts) + 1)
            scale = (long_edge, short_edge)
        else:                                      # random_select
            scale = tuple(self.img_scale[np.random.randint(len(self.img_scale))])
        results['scale'] = scale

    def __call__(self, results):
        if 'scale' not in results:
            self._random_scale(results)
        img = results['img']
        h, w = results['img_shape'][:2]
        if self.keep_ratio:                        # mmcv.imrescale
            s = results['scale']
            f = min(max(s) / max(h, w), min(s) / min(h, w))
            new_w, new_h = int(w * float(f) + 0.5), int(h * float(f) + 0.5)
        else:
            new_w, new_h = results['scale']
        results['img'] = _img_ops().resize_bilinear(img, (new_w, new_h))
        w_scale, h_scale = new_w / w, new_h / h
        results['scale_factor'] = np.array([w_scale, h_scale, w_scale, h_scale], dtype=np.float32)
        results['img_shape'] = (new_h, new_w, 3)
        results['pad_shape'] = (new_h, new_w, 3)
        results['keep_ratio'] = self.keep_ratio
        for key in results.get('bbox_fields', []):  # mmdet Resize._resize_bboxes
            b = results[key] * results['scale_factor']
            if self.bbox_clip_border:
                b[:, 0::2] = np.clip(b[:, 0::2], 0, new_w)
                b[:, 1::2] = np.clip(b[:, 1::2], 0, new_h)
            results[key] = b
        resize_pose(results, self.scale_depth, self.abs_dz)
        return results


@PIPELINES.register_module()
class Resize(ResizePose):
    """mmdet Resize (the test pipeline's `dict(type='Resize', keep_ratio=True)` inside MultiScaleFlipAug): image and
    boxes only."""

    def __init__(self, **kwargs):
        super().__init__(scale_depth=False, abs_dz=False, **kwargs)

    def __call__(self, results):
        pose = {k: results.pop(k) for k in ('gt_poses_3d', 'centers2d', 'depths') if k in results}
        try:
            # (ResizePose's image / box half; the pose half is skipped: plain Resize leaves those keys alone)
            if 'scale' not in results:
                self._random_scale(results)
            h, w = results['img_shape'][:2]
            if self.keep_ratio:
                sc = results['scale']
                f = min(max(sc) / max(h, w), min(sc) / min(h, w))
                new_w, new_h = int(w * float(f) + 0.5), int(h * float(f) + 0.5)
            else:
                new_w, new_h = results['scale']
            results['img'] = _img_ops().resize_bilinear(results['img'], (new_w, new_h))
            w_scale, h_scale = new_w / w, new_h / h
            results['scale_factor'] = np.array([w_scale, h_scale, w_scale, h_scale], dtype=np.float32)
            results['img_shape'] = results['pad_shape'] = (new_h, new_w, 3)
            results['keep_ratio'] = self.keep_ratio
            for key in results.get('bbox_fields', []):
                b = results[key] * results['scale_factor']
                if self.bbox_clip_border:
                    b[:, 0::2] = np.clip(b[:, 0::2], 0, new_w)
                    b[:, 1::2] = np.clip(b[:, 1::2], 0, new_h)
                results[key] = b
        finally:
            results.update(pose)
        return results


@PIPELINES.register_module()
class MultiScaleFlipAug:
    """mmdet MultiScaleFlipAug: the wrapped transforms once per (scale, flip) combination; the outputs' values are
    gathered into lists per key (one entry per augmentation: the DAS configs use a single scale, no flip)."""

    def __init__(self, transforms, img_scale=None, scale_factor=None, flip=False, flip_direction='horizontal'):
        assert (img_scale is None) != (scale_factor is None)
        self.transforms = Compose(transforms)
        self.img_scale = img_scale if isinstance(img_scale, list) else [img_scale] if img_scale is not None else \
            (scale_factor if isinstance(scale_factor, list) else [scale_factor])
        self.scale_key = 'scale' if img_scale is not None else 'scale_factor'
        self.flip = flip
        self.flip_direction = flip_direction if isinstance(flip_direction, list) else [flip_direction]

    def __call__(self, results):
        import copy
        aug = []
        flips = [(False, None)] + ([(True, d) for d in self.flip_direction] if self.flip else [])
        for scale in self.img_scale:
            for flip, direction in flips:
                r = copy.copy(results)
                r[self.scale_key] = scale
                r['flip'], r['flip_direction'] = flip, direction
                aug.append(self.transforms(r))
        return {k: [a[k] for a in aug] for k in aug[0]}


# ---------------------------------------------------------------------------------------------- flip
def flip_pose(results, num_joints, flip_pairs):
    """Annotation half of a horizontal flip (the 'gt_poses_3d' branch of transforms_3d.py:293-318), in place: columns
    mirror about the image (x -> w - x - 1) and left / right joints trade places."""
    width = results['img_shape'][1]
    table = PoseTable(results['gt_poses_3d'], num_joints)
    order = np.arange(num_joints)            # joint j of the flipped person is joint order[j] of the original
    for left, right in flip_pairs:
        order[[left, right]] = order[[right, left]]
    table.joints, table.vis = table.joints[:, order], table.vis[:, order]
    root = results['centers2d']
    for x in (root[..., 0], table.joints[..., 0]):
        x[...] = width - x - 1
    results['gt_poses_3d'] = table.pack(root, results['depths'])


@PIPELINES.register_module()
class RandomFlipPose3D:
    """transforms_3d.py:235-356 over mmdet's RandomFlip (horizontal)."""

    def __init__(self, sync_2d=True, flip_ratio_bev_horizontal=0.0, flip_ratio_bev_vertical=0.0, num_joints=15,
                 flip_pairs=None, direction='horizontal'):
        assert sync_2d and flip_ratio_bev_vertical == 0 and direction == 'horizontal'
        self.flip_ratio, self.num_joints, self.flip_pairs = flip_ratio_bev_horizontal, num_joints, flip_pairs or []

    def __call__(self, results):
        if 'flip' not in results:                  # mmdet RandomFlip.__call__
            cur_dir = np.random.choice(['horizontal', None], p=[self.flip_ratio, 1 - self.flip_ratio])
            results['flip'] = cur_dir is not None
        results.setdefault('flip_direction', 'horizontal' if results['flip'] else None)
        if results['flip']:
            results['img'] = _img_ops().flip_horizontal(results['img'])
            w = results['img_shape'][1]
            for key in results.get('bbox_fields', []):   # mmdet bbox_flip
                b = results[key]
                flipped = b.copy()
                flipped[..., 0::4] = w - b[..., 2::4]
                flipped[..., 2::4] = w - b[..., 0::4]
                results[key] = flipped
        results['pcd_horizontal_flip'] = results['flip']
        results['pcd_vertical_flip'] = False
        results.setdefault('transformation_3d_flow', [])
        if results['pcd_horizontal_flip']:
            if 'gt_poses_3d' in results:
                flip_pose(results, self.num_joints, self.flip_pairs)
            results['transformation_3d_flow'].extend(['HF'])
        return results


# ---------------------------------------------------------------------------------------------- colour
@PIPELINES.register_module()
class PhotoMetricDistortion:
    """mmdet PhotoMetricDistortion: the draws (numpy.random, this order) on the host, the pixels in one kernel pass."""

    def __init__(self, brightness_delta=32, contrast_range=(0.5, 1.5), saturation_range=(0.5, 1.5), hue_delta=18):
        self.brightness_delta = brightness_delta
        self.contrast_lower, self.contrast_upper = contrast_range
        self.saturation_lower, self.saturation_upper = saturation_range
        self.hue_delta = hue_delta

    def draw(self):
        r = np.random
        p = dict(brightness=None, contrast=None, saturation=None, hue=None, perm=None)
        if r.randint(2):
            p['brightness'] = r.uniform(-self.brightness_delta, self.brightness_delta)
        mode = r.randint(2)
        p['contrast_first'] = mode == 1
        if mode == 1 and r.randint(2):
            p['contrast'] = r.uniform(self.contrast_lower, self.contrast_upper)
        if r.randint(2):
            p['saturation'] = r.uniform(self.saturation_lower, self.saturation_upper)
        if r.randint(2):
            p['hue'] = r.uniform(-self.hue_delta, self.hue_delta)
        if mode == 0 and r.randint(2):
            p['contrast'] = r.uniform(self.contrast_lower, self.contrast_upper)
        if r.randint(2):
            p['perm'] = r.permutation(3)
        return p

    def __call__(self, results):
        p = results['photometric'] = self.draw()
        results['img'] = _img_ops().photometric_(results['img'], **p)
        return results


# ---------------------------------------------------------------------------------------------- rot / scale / trans
def affine_from_points(src, dst):
    """cv2.getAffineTransform: the 2x3 map taking three points to three points (f64 solve)."""
    A = np.zeros((6, 6))
    b = np.zeros(6)
    for i in range(3):
        A[i, :3] = [src[i][0], src[i][1], 1]
        A[i + 3, 3:] = [src[i][0], src[i][1], 1]
        b[i], b[i + 3] = dst[i][0], dst[i][1]
    return np.linalg.solve(A, b).reshape(2, 3)


def _corner_triangle(origin, spoke):
    """Three f32 points that pin a similarity: `origin`, `origin + spoke`, and the spoke turned a quarter turn about its
    tip (what the reference's point construction yields, transforms_3d.py:864-898: its third point is
    tip + perp(origin - tip))."""
    tri = np.zeros((3, 2), dtype=np.float32)
    tri[0] = origin
    tri[1] = origin + spoke
    back = tri[0] - tri[1]
    tri[2] = tri[1] + np.array([-back[1], back[0]], dtype=np.float32)
    return tri


def window_to_frame_affine(center, extent, rot_deg, frame_wh, inverse=False):
    """2x3 affine that shows the source window (centre `center`, width `extent[0]`, turned by `rot_deg` degrees) in an
    output frame of `frame_wh` pixels. As in the reference (transforms_3d.py:864-898) the map is pinned by three point
    pairs rounded to f32 and solved the way cv2.getAffineTransform does, so the matrix — and every annotation moved
    with it — is bit-identical to the reference's."""
    extent = np.asarray(extent, dtype=float) * np.ones(2)
    fw, fh = frame_wh[0], frame_wh[1]
    turn = np.pi * rot_deg / 180
    reach = extent[0] * -0.5                          # half a window width, "up" in image coordinates
    window = _corner_triangle(np.asarray(center, dtype=float),
                              np.array([-(reach * np.sin(turn)), reach * np.cos(turn)]))
    frame = _corner_triangle(np.array([fw * 0.5, fh * 0.5]), np.array([0, fw * -0.5], dtype=np.float32))
    return affine_from_points(frame, window) if inverse else affine_from_points(window, frame)


def _move_points(xy, affine):
    """(N, 2) points through a 2x3 affine, in f64 as one (N, 3) x (3, 2) product."""
    homog = np.concatenate([xy, np.ones((len(xy), 1))], axis=-1)
    return np.dot(homog, affine.T)


def warp_annotations(results, trans, scale, num_joints, scale_depth, abs_dz, use_bbox_center):
    """Annotation half of GlobalRotScaleTransPose (what transforms_3d.py:988-1058 computes): roots, joints and boxes
    move with the image's affine map `trans`; joints that leave the image become invisible; persons whose root leaves
    it (or, with `use_bbox_center`, that keep fewer than three visible joints) are dropped. Returns the updated
    results, or None when fewer than two persons remain (the sample is redrawn)."""
    height, width, _ = results['img_shape']
    rows = results['gt_poses_3d']
    n, split = len(rows), 3 + 3 * num_joints
    points = np.array(rows[:, :split]).reshape(n, num_joints + 1, 3)       # root first, then the joints
    vis = np.array(rows[:, split:]).reshape(n, num_joints)
    assert ((0 <= vis) & (vis <= 1)).all()
    depth = points[..., 2:]
    if scale_depth:
        if abs_dz:
            depth[0] = depth[0] * scale      # (first person only: the reference indexes the person axis here)
        else:
            depth = depth * scale
    moved = np.concatenate([_move_points(points[..., :2].reshape(-1, 2), trans).reshape(n, -1, 2), depth], axis=-1)

    boxes = results['gt_bboxes']
    corners = boxes[:, [[0, 1], [2, 3], [0, 3], [2, 1]]]                   # (n, 4, 2)
    corners = _move_points(corners.reshape(-1, 2), trans).reshape(n, 4, 2)
    hull = np.concatenate([corners.min(axis=1), corners.max(axis=1)], axis=-1)
    hull[:, 0::2] = hull[:, 0::2].clip(0, width - 1)
    hull[:, 1::2] = hull[:, 1::2].clip(0, height - 1)

    outside = (moved[..., 0] < 0) | (moved[..., 0] > width - 1) | (moved[..., 1] < 0) | (moved[..., 1] > height - 1)
    vis[outside[:, 1:]] = 0
    if use_bbox_center:
        root = np.stack([hull[:, 0::2].mean(-1), hull[:, 1::2].mean(-1), moved[:, 0, -1]], axis=-1)
        keep = (vis.sum(-1) >= 3) & ((boxes[:, 2:] - boxes[:, :2]).prod() > 64)
    else:
        root = moved[:, 0]
        keep = ~outside[:, 0]
        if keep.sum() < 2:
            return None
    table = np.concatenate([root, moved[:, 1:].reshape(n, -1), vis], axis=-1).astype(np.float32)
    results['gt_poses_3d'] = table[keep].copy()
    results['gt_bboxes'] = hull[keep]
    results['centers2d'] = table[:, :2][keep].copy()
    results['depths'] = table[:, 2][keep].copy()
    results['gt_labels'] = results['gt_labels'][keep]
    results['gt_labels_3d'] = results['gt_labels_3d'][keep]
    results['transform_mat'] = trans
    return results


@PIPELINES.register_module()
class GlobalRotScaleTransPose:
    """transforms_3d.py:901-1129: one random (rotation, scale, translation) affine map for image and annotations."""

    def __init__(self, rot_range=[-0.78539816, 0.78539816], scale_ratio_range=[0.95, 1.05], translation_std=[0, 0, 0],
                 shift_height=False, num_joints=None, scale_depth=False, abs_dz=False, img_norm_cfg=None,
                 use_bbox_center=False):
        if not isinstance(rot_range, (list, tuple, np.ndarray)):
            rot_range = [-rot_range, rot_range]
        if not isinstance(translation_std, (list, tuple, np.ndarray)):
            translation_std = [translation_std] * 3
        assert all(std >= 0 for std in translation_std)
        self.rot_range, self.scale_ratio_range, self.translation_std = rot_range, scale_ratio_range, translation_std
        self.num_joints, self.scale_depth, self.abs_dz, self.use_bbox_center = num_joints, scale_depth, abs_dz, use_bbox_center
        if abs_dz:
            assert scale_depth
        if img_norm_cfg is not None:
            self.img_mean = img_norm_cfg['mean']
            if img_norm_cfg['to_rgb']:
                self.img_mean = self.img_mean[::-1]
        else:
            self.img_mean = [127.5, 127.5, 127.5]

    def draw(self, results):
        """_rot_points, _random_scale, _trans_points: in this order (transforms_3d.py:1112-1116)."""
        noise_rotation = np.random.uniform(self.rot_range[0], self.rot_range[1])
        rot_sin, rot_cos = np.sin(noise_rotation), np.cos(noise_rotation)
        results['pcd_rotation'] = np.array([[rot_cos, -rot_sin, 0], [rot_sin, rot_cos, 0], [0, 0, 1]]).T
        results['pcd_rot'] = noise_rotation / math.pi * 180
        assert 'pcd_scale_factor' not in results
        results['pcd_scale_factor'] = np.random.uniform(self.scale_ratio_range[0], self.scale_ratio_range[1])
        results['pcd_trans'] = np.random.normal(scale=np.array(self.translation_std, dtype=np.float32), size=2).T

    def matrix(self, results):
        h, w, _ = results['img_shape']
        center = np.array([w / 2, h / 2], dtype=float) * (1 + results['pcd_trans'])
        new_scale = np.array([w, h], dtype=float) * results['pcd_scale_factor']
        return window_to_frame_affine(center, new_scale, results['pcd_rot'], [w, h])

    def __call__(self, results):
        results.setdefault('transformation_3d_flow', [])
        self.draw(results)
        trans = self.matrix(results)
        h, w, _ = results['img_shape']
        for key in results.get('img_fields', ['img']):
            img = results[key]
            assert tuple(img.shape[:2]) == (h, w)
            results[key] = _img_ops().warp_affine(img, trans, (int(w), int(h)), self.img_mean)
        results = warp_annotations(results, trans, results['pcd_scale_factor'], self.num_joints, self.scale_depth,
                                   self.abs_dz, self.use_bbox_center)
        if results is None:
            return None
        results['transformation_3d_flow'].extend(['R', 'S', 'T'])
        return results


# ---------------------------------------------------------------------------------------------- normalize / pad / format
@PIPELINES.register_module()
class Normalize:
    """mmdet Normalize: records the parameters; the arithmetic runs in DefaultFormatBundlePose3D's single pass."""

    def __init__(self, mean, std, to_rgb=True):
        self.mean, self.std, self.to_rgb = np.array(mean, dtype=np.float32), np.array(std, dtype=np.float32), to_rgb

    def __call__(self, results):
        results['img_norm_cfg'] = dict(mean=self.mean, std=self.std, to_rgb=self.to_rgb)
        results['_pending_normalize'] = True
        return results


@PIPELINES.register_module()
class Pad:
    """mmdet Pad(size / size_divisor, pad_val=0): records the padded shape (see Normalize)."""

    def __init__(self, size=None, size_divisor=None, pad_val=0):
        assert (size is None) != (size_divisor is None) and pad_val == 0
        self.size, self.size_divisor = size, size_divisor

    def __call__(self, results):
        h, w = results['img_shape'][:2]
        if self.size is not None:
            ph, pw = self.size
        else:
            d = self.size_divisor
            ph, pw = int(np.ceil(h / d)) * d, int(np.ceil(w / d)) * d
        results['pad_shape'] = (ph, pw, 3)
        results['pad_fixed_size'] = self.size
        results['pad_size_divisor'] = self.size_divisor
        return results


def materialize_image(results, out=None):
    """Normalize + Pad + HWC -> CHW of results['img'] in one kernel pass; `out` = a (3, Hp, Wp) slot of a batch tensor."""
    cfg = results.get('img_norm_cfg') if results.pop('_pending_normalize', False) else None
    mean = cfg['mean'] if cfg is not None else (0.0, 0.0, 0.0)
    std = cfg['std'] if cfg is not None else (1.0, 1.0, 1.0)
    ph, pw = results.get('pad_shape', results['img_shape'])[:2]
    return _img_ops().normalize_pad_chw(results['img'], mean, std, bool(cfg is not None and cfg['to_rgb']), (ph, pw), out=out)


@PIPELINES.register_module()
class DefaultFormatBundlePose3D:
    """formating.py:383-442: image to a CHW tensor (here: with the pending Normalize / Pad, on the device), annotation
    arrays to tensors. (mmcv's DataContainer is the collate protocol of mmcv's dataloader; plain tensors here.)"""

    def __init__(self, class_names, with_gt=True, with_label=True):
        self.class_names, self.with_gt, self.with_label = class_names, with_gt, with_label

    def __call__(self, results):
        if 'img' in results:
            results['img'] = materialize_image(results)
        for key in ['gt_bboxes', 'gt_bboxes_ignore', 'gt_labels', 'gt_poses_3d', 'gt_labels_3d', 'centers2d', 'depths',
                    'transform_mat']:
            if key in results:
                results[key] = torch.from_numpy(np.ascontiguousarray(results[key]))
        return results


@PIPELINES.register_module()
class Collect3D:
    """formating.py:83-...: the keys the detector takes plus `img_metas` (the meta keys the model reads)."""

    def __init__(self, keys, meta_keys=('filename', 'ori_shape', 'img_shape', 'pad_shape', 'scale_factor', 'flip',
                                        'flip_direction', 'img_norm_cfg', 'transformation_3d_flow', 'pcd_scale_factor',
                                        'pcd_rot', 'pcd_trans', 'transform_mat', 'cam'), debug=False, num_joints=15):
        self.keys, self.meta_keys = keys, meta_keys

    def __call__(self, results):
        data = {'img_metas': {k: results[k] for k in self.meta_keys if k in results}}
        for k in self.keys:
            data[k] = results[k]
        return data
