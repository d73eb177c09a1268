#!/usr/bin/env python
"""CMU Panoptic raw capture -> COCO-style annotation files for `CMUPanopticDataset` (the
reference's mytools/panoptic2coco.py, itself adapted from voxelpose-pytorch).

    python tools/convert_panoptic.py --root data/panoptic

Per split it writes <root>/annotations/<split>.json with, per person and HD camera (0,16) / (0,30): the 15 joints in
camera-world millimetres (`joints3d`), their distorted projection + depth (`joints3d_img`), visibility, a box grown
from the visible joints, and the camera (K, R, t) on the image record. Training keeps every second frame of six
sequences and drops a frame if any of its persons is invalid; the four validation splits sample a fixed number of
frames evenly per (sequence, camera).
"""
import argparse
import glob
import json
import os

import numpy as np

JOINT_NAMES = ['neck', 'nose', 'mid-hip', 'l-shoulder', 'l-elbow', 'l-wrist', 'l-hip', 'l-knee', 'l-ankle', 'r-shoulder',
               'r-elbow', 'r-wrist', 'r-hip', 'r-knee', 'r-ankle']
LIMBS = [[0, 1], [0, 2], [0, 3], [3, 4], [4, 5], [0, 9], [9, 10], [10, 11], [2, 6], [2, 12], [6, 7], [7, 8], [12, 13],
         [13, 14]]
ROOT_JOINT = 2
CAMERAS = [(0, 16), (0, 30)]
WIDTH, HEIGHT = 1920, 1080
# panoptic world (y up) -> the camera-maths frame used downstream
AXES = np.array([[1.0, 0.0, 0.0], [0.0, 0.0, -1.0], [0.0, 1.0, 0.0]])

SPLITS = {
    'train': dict(sequences=['160224_haggling1', '160226_mafia2', '160224_mafia1', '160224_mafia2', '160224_ultimatum1',
                             '160224_ultimatum2'], interval=2, total=None, strict=True),
    'haggling': dict(sequences=['160226_haggling1', '160422_haggling1'], interval=None, total=2400, strict=False),
    'mafia': dict(sequences=['160226_mafia1', '160422_mafia2'], interval=None, total=2400, strict=False),
    'ultimatum': dict(sequences=['160422_ultimatum1'], interval=None, total=2400, strict=False),
    'pizza': dict(sequences=['160906_pizza1'], interval=None, total=2400, strict=False),
}


def project(X, K, R, t, dist):
    """3xN world points -> 3xN [u, v, depth] with the radial / tangential distortion model of cv2.projectPoints."""
    x = R @ X + t
    x[0:2] = x[0:2] / (x[2] + 1e-5)
    r = x[0] * x[0] + x[1] * x[1]
    radial = 1 + dist[0] * r + dist[1] * r * r + dist[4] * r * r * r
    xd = x[0] * radial + 2 * dist[2] * x[0] * x[1] + dist[3] * (r + 2 * x[0] * x[0])
    yd = x[1] * radial + 2 * dist[3] * xd * x[1] + dist[2] * (r + 2 * x[1] * x[1])
    x[0], x[1] = xd, yd
    u = K[0, 0] * x[0] + K[0, 1] * x[1] + K[0, 2]
    v = K[1, 0] * u + K[1, 1] * x[1] + K[1, 2]
    x[0], x[1] = u, v
    return x


def load_cameras(root, seq):
    with open(os.path.join(root, seq, f'calibration_{seq}.json')) as f:
        calib = json.load(f)
    cams = {}
    for cam in calib['cameras']:
        key = (cam['panel'], cam['node'])
        if key in CAMERAS:
            cams[key] = dict(K=np.array(cam['K']), distCoef=np.array(cam['distCoef']), R=np.array(cam['R']).dot(AXES),
                             t=np.array(cam['t']).reshape((3, 1)))
    return cams


def _plain(v):
    if isinstance(v, np.ndarray):
        if v.dtype == bool or v.dtype == np.int64:
            v = v.astype(np.int32)
        return v.tolist()
    return int(v) if isinstance(v, np.integer) else v


def person_record(body, key, cam):
    """One body of a frame seen by one camera -> annotation fields, or None when it is not a valid instance."""
    pose3d = np.array(body[key]).reshape((-1, 4))[:len(JOINT_NAMES)]
    vis = pose3d[:, -1] > 0.1
    if key == 'joints19':
        vis[1] = 0
    pose3d[:, 0:3] = pose3d[:, 0:3].dot(AXES)
    joints3d = pose3d[:, 0:3] * 10.0
    vis3d = np.repeat(np.reshape(vis, (-1, 1)), 3, axis=1)
    for_box = vis.copy()
    pose_img = project(pose3d[:, 0:3].transpose(), cam['K'], cam['R'], cam['t'], cam['distCoef']).transpose()
    pose2d = np.zeros((pose3d.shape[0], 2))
    pose2d[:, :2] = pose_img[:, :2]
    inside = (pose2d[:, 0] >= 0) & (pose2d[:, 0] <= WIDTH - 1) & (pose2d[:, 1] >= 0) & (pose2d[:, 1] <= HEIGHT - 1)
    vis[np.logical_not(inside)] = 0
    vis2d = np.repeat(np.reshape(vis, (-1, 1)), 2, axis=1)
    if for_box.sum() < 3:
        return None
    xmin, ymin = np.min(pose2d[for_box], axis=0)
    xmax, ymax = np.max(pose2d[for_box], axis=0)
    w, h = xmax - xmin, ymax - ymin
    if key == 'joints19':
        ymin, ymax = ymin - 0.30 * h, ymax + 0.15 * h
    else:
        ymin, ymax = ymin - 0.02 * h, ymax + 0.07 * h
    xmin, xmax = xmin - 0.15 * w, xmax + 0.15 * w
    xmin, xmax = np.array([xmin, xmax]).clip(0, WIDTH - 1)
    ymin, ymax = np.array([ymin, ymax]).clip(0, HEIGHT - 1)
    xmin, xmax, ymin, ymax = np.array([xmin, xmax, ymin, ymax]).tolist()
    w, h = xmax - xmin + 1, ymax - ymin + 1
    if w <= 1 or h <= 1 or w * h <= 64:
        return None
    return dict(category_id=1, area=w * h, bbox=[xmin, ymin, w, h], iscrowd=0, joints2d=pose2d, joints2d_vis=vis2d,
                joints3d_img=pose_img, joints3d=joints3d, joints3d_vis=vis3d, center2d=pose_img[ROOT_JOINT],
                num_keypoints=vis.sum())


def convert_split(root, sequences, interval=None, total=None, strict=False, log=print):
    images, annos = [], []
    next_img, next_ann, sampled = 1, 1, 0
    for seq in sequences:
        cams = load_cameras(root, seq)
        per_cam = total // len(sequences) // len(cams) if total else None
        assert (per_cam is None) != (interval is None)
        files = sorted(glob.iglob(os.path.join(root, seq, 'hdPose3d_stage1_coco19') + '/*.json'))
        key = 'joints19'
        if not files:
            files = sorted(glob.iglob(os.path.join(root, seq, 'hdPose3d_stage1') + '/*.json'))
            key = 'joints15'
        log(seq)
        for cam_id, cam in cams.items():
            for i, path in enumerate(files):
                if not (per_cam or (interval and i % interval == 0)):
                    continue
                try:
                    with open(path) as f:
                        bodies = json.load(f)['bodies']
                except Exception as e:       # (truncated files exist in the capture)
                    log(e)
                    continue
                if not bodies:
                    continue
                prefix = '{:02d}_{:02d}'.format(*cam_id)
                name = os.path.join(seq, 'hdImgs', prefix, prefix + os.path.basename(path).replace('body3DScene', ''))
                name = name.replace('json', 'jpg')
                if not os.path.exists(os.path.join(root, name)):
                    log('WARNING: File Not Exist', os.path.join(root, name))
                    continue
                recs = [person_record(b, key, cam) for b in bodies]
                good = [r for r in recs if r is not None]
                if not good or (strict and len(good) != len(recs)):
                    continue
                images.append(dict(id=next_img, width=WIDTH, height=HEIGHT, file_name=name,
                                   cam=dict(K=cam['K'].tolist(), R=cam['R'].tolist(), t=cam['t'].tolist())))
                for r in good:
                    rec = dict(id=next_ann, image_id=next_img)
                    rec.update({k: _plain(v) for k, v in r.items()})
                    annos.append(rec)
                    next_ann += 1
                next_img += 1
            if per_cam:      # even sampling of this (sequence, camera)'s frames
                fresh, kept = images[sampled:], images[:sampled]
                assert len(fresh) >= per_cam, (seq, cam_id, len(fresh), per_cam)
                pick = np.linspace(0, len(fresh) - 1, per_cam).astype(int)
                images = kept + [fresh[j] for j in pick]
                ids = set(x['id'] for x in images)
                annos = [a for a in annos if a['image_id'] in ids]
                sampled += per_cam
    cats = [dict(supercategory='person', id=1, name='person', keypoints=list(JOINT_NAMES), skeleton=LIMBS)]
    return dict(images=images, annotations=annos, categories=cats)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--root', default='data/panoptic')
    ap.add_argument('--splits', nargs='*', default=list(SPLITS))
    args = ap.parse_args()
    for name in args.splits:
        db = convert_split(args.root, **SPLITS[name])
        print('db size:', len(db['images']))
        out = os.path.join(args.root, 'annotations', f'{name}.json')
        os.makedirs(os.path.dirname(out), exist_ok=True)
        with open(out, 'w') as f:
            json.dump(db, f, indent=4)


if __name__ == '__main__':
    main()
