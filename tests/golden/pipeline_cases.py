"""Seeded synthetic annotations for the pose data pipeline fixtures (tests/golden/pipeline_*.npz): the inputs are rebuilt
from these seeds on both sides (reference at authoring time, das_amd in the tests)."""
import numpy as np

J = 15
FLIP_PAIRS = [[3, 9], [4, 10], [5, 11], [6, 12], [7, 13], [8, 14]]
IMG_NORM = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)


def annotations(seed, n=4, h=540, w=960):
    r = np.random.RandomState(seed)
    centers = np.stack([r.uniform(0.15 * w, 0.85 * w, n), r.uniform(0.2 * h, 0.8 * h, n)], 1).astype(np.float32)
    depths = r.uniform(150, 600, n).astype(np.float32)
    joints = np.zeros((n, J, 3), dtype=np.float32)
    joints[..., 0] = centers[:, None, 0] + r.uniform(-90, 90, (n, J))
    joints[..., 1] = centers[:, None, 1] + r.uniform(-160, 160, (n, J))
    joints[..., 2] = r.uniform(-40, 40, (n, J))
    vis = (r.uniform(size=(n, J)) > 0.2).astype(np.float32)
    gt = np.concatenate([centers, depths[:, None], joints.reshape(n, -1), vis], 1).astype(np.float32)
    x1y1 = joints[..., :2].min(1) - 5
    x2y2 = joints[..., :2].max(1) + 5
    boxes = np.concatenate([x1y1, x2y2], 1).astype(np.float32)
    return dict(gt_poses_3d=gt, centers2d=gt[:, :2].copy(), depths=gt[:, 2].copy(), gt_bboxes=boxes,
                gt_labels=np.zeros(n, dtype=np.int64), gt_labels_3d=np.zeros(n, dtype=np.int64),
                img_shape=(h, w, 3), bbox_fields=['gt_bboxes'])


CASES = [
    # name, seed, resize scale_factor (w, h), ResizePose(scale_depth, abs_dz), warp (rot deg, scale, trans), use_bbox_center
    ('a', 11, (0.9481481, 0.9479167), (True, False), (0.0, 0.83, (0.07, -0.11)), False),
    ('b', 12, (1.1851852, 1.1854166), (True, True), (0.0, 1.31, (-0.2, 0.05)), False),
    ('c', 13, (0.75, 0.75), (False, False), (6.5, 0.95, (0.12, 0.18)), True),
    ('d', 14, (1.0, 1.0), (True, False), (-4.0, 0.62, (0.3, -0.3)), False),
    ('e', 15, (1.0, 1.0), (True, False), (0.0, 1.4, (0.9, 0.9)), False),      # every root leaves the image: sample dropped
]
