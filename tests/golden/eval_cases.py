"""Seeded synthetic annotations / detections for the evaluator fixtures (shared by make_golden_eval.py, which feeds
them to the REFERENCE's dataset classes, and by tests/test_evaluation.py, which feeds them to das_amd's)."""
import os

import numpy as np

PANOPTIC_J, PANOPTIC_ROOT = 15, 2
MUPOTS_J = 17


def _world2pixel(X, K, R, t):
    x = R @ X + t
    z = x[2].copy()
    u = K[0, 0] * x[0] / z + K[0, 2]
    v = K[1, 1] * x[1] / z + K[1, 2]
    return np.stack([u, v, z])


def _person(rs, J, centre, spread):
    return centre[:, None] + rs.normal(0, spread, (3, J))


def panoptic_annotation(seed=3, n_img=7):
    """COCO-style dict as mytools/panoptic2coco.py writes it (fields used by cmupanoptic_mono_dataset.py): world
    coordinates in cm inside the camera maths, `joints3d` stored in mm."""
    rs = np.random.RandomState(seed)
    images, anns = [], []
    aid = 1
    for i in range(n_img):
        ang = rs.uniform(-0.3, 0.3)
        R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
        t = np.array([[rs.uniform(-20, 20)], [rs.uniform(-10, 10)], [rs.uniform(250, 400)]])
        K = np.array([[1400.0 + 20 * i, 0, 960.0], [0, 1390.0 + 10 * i, 540.0], [0, 0, 1]])
        images.append(dict(id=100 + i, file_name=f'160422_ultimatum1/00_{i:02d}/frame_{i:06d}.jpg', width=1920, height=1080,
                           cam=dict(K=K.tolist(), R=R.tolist(), t=t.tolist())))
        G = [3, 1, 0, 2, 4, 2, 1][i % 7]
        for g in range(G):
            world = _person(rs, PANOPTIC_J, np.array([rs.uniform(-150, 150), rs.uniform(-40, 40), rs.uniform(-80, 80)]), 25.0)
            img3 = _world2pixel(world, K, R, t).T                     # (J, 3) [u, v, depth cm]
            vis = (rs.uniform(0, 1, (PANOPTIC_J, 1)) > 0.15).astype(float)
            if g == 1 and i == 0:
                vis[PANOPTIC_ROOT] = 0                               # invisible root: goes to bboxes_ignore
            vis2 = np.repeat(vis, 2, 1)
            x1, y1 = img3[:, 0].min(), img3[:, 1].min()
            w, h = img3[:, 0].max() - x1, img3[:, 1].max() - y1
            anns.append(dict(id=aid, image_id=100 + i, category_id=1, iscrowd=0, bbox=[float(x1), float(y1), float(w), float(h)],
                             area=float(w * h), joints3d_img=img3.tolist(), joints2d_vis=vis2.tolist(),
                             joints3d=(world.T * 10).tolist(), joints3d_vis=np.repeat(vis, 3, 1).tolist()))
            aid += 1
    return dict(images=images, annotations=anns, categories=[dict(id=1, name='person')])


def panoptic_outputs(ann, parsed, seed=4, depth_factor=1):
    """Detector outputs in the head's convention for images of `ann`: GT poses of `parsed` (per image gt_poses_3d as
    _parse_ann_info returns them: [c, depth, J x (u, v, dz), vis]) plus noise, a missed person, a false positive."""
    import torch
    rs = np.random.RandomState(seed)
    outs = []
    for im, gp in zip(ann['images'], parsed):
        J = PANOPTIC_J
        poses, scores = [], []
        for k, row in enumerate(gp):
            if k == 2:
                continue                                              # a missed person
            uvd = row[3:3 + 3 * J].reshape(J, 3).copy()
            uvd[:, :2] += rs.normal(0, 6.0, (J, 2))
            z = row[2] + uvd[:, 2] + rs.normal(0, 0.002, J)           # root depth (normalised) + dz
            poses.append(np.concatenate([uvd[:, :2], z[:, None]], 1))
            scores.append(float(rs.uniform(0.3, 0.9)))
        if len(gp) and rs.uniform() < 0.6:                            # a false positive
            poses.append(poses[0] + rs.normal(0, 80.0, (J, 3)) * np.array([1, 1, 0.0005]))
            scores.append(0.1)
        if len(gp) == 0 or (im['id'] == 103):
            poses, scores = ([], []) if im['id'] == 103 else (poses, scores)
        p = np.array(poses, dtype=np.float32).reshape(-1, J, 3)
        outs.append(dict(poses=torch.from_numpy(p), vis=torch.ones(len(p), J), scores=scores,
                         image_paths=['/data/panoptic/' + im['file_name']]))
    return outs


def mupots_annotation(seed=5, frames=2):
    """COCO-style dict as mytools/muco2coco.py-like tooling writes the MuPoTS test split + per-sequence annot.mat
    content (camera-space mm)."""
    rs = np.random.RandomState(seed)
    images, anns, mats = [], [], {}
    aid, iid = 1, 1
    for ts in range(20):
        P = 2 + ts % 2
        fx, fy, cx, cy = 1500.0 + ts, 1490.0 + ts, 1024.0, 1024.0
        seq = np.empty((frames, P), dtype=object)
        occ = np.empty((frames, P), dtype=object)
        for fidx in range(frames):
            images.append(dict(id=iid, file_name='TS%d/img_%06d.jpg' % (ts + 1, fidx), width=2048, height=2048,
                               intrinsic=[fx, fy, cx, cy]))
            for pidx in range(P):
                cam = _person(rs, MUPOTS_J, np.array([rs.uniform(-1200, 1200), rs.uniform(-300, 300), rs.uniform(3000, 6000)]),
                              220.0)                                 # (3, 17) mm
                u = fx * cam[0] / cam[2] + cx
                v = fy * cam[1] / cam[2] + cy
                valid = 0 if (ts == 3 and fidx == 1 and pidx == 0) else 1
                seq[fidx, pidx] = dict(annot2=np.stack([u, v]), annot3=cam, univ_annot3=cam * 1.02,
                                       isValidFrame=np.array([[valid]]))
                occ[fidx, pidx] = (rs.uniform(0, 1, (1, MUPOTS_J)) > 0.8).astype(float)
                if valid:
                    x1, y1 = u.min(), v.min()
                    anns.append(dict(id=aid, image_id=iid, category_id=1, iscrowd=0,
                                     bbox=[float(x1), float(y1), float(u.max() - x1), float(v.max() - y1)],
                                     keypoints_img=np.stack([u, v], 1).tolist(), keypoints_cam=cam.T.tolist(),
                                     keypoints_vis=np.ones((MUPOTS_J, 1)).tolist()))
                    aid += 1
            iid += 1
        mats[ts] = (seq, occ)
    return dict(images=images, annotations=anns, categories=[dict(id=1, name='person')]), mats


def write_mupots_mats(root, mats):
    import scipy.io as sio
    for ts, (seq, occ) in mats.items():
        d = os.path.join(root, 'TS%d' % (ts + 1))
        os.makedirs(d, exist_ok=True)
        sio.savemat(os.path.join(d, 'annot.mat'), {'annotations': seq})
        sio.savemat(os.path.join(d, 'occlusion.mat'), {'occlusion_labels': occ})


def mupots_outputs(ann, root, seed=6):
    """Detector outputs (21 joints as the MuCo head predicts; the first 17 are MuPoTS's) = GT + noise; one image with a
    missed person, one far-off prediction."""
    import torch
    rs = np.random.RandomState(seed)
    by_img = {}
    for a in ann['annotations']:
        by_img.setdefault(a['image_id'], []).append(a)
    outs = []
    for im in ann['images']:
        poses, scores = [], []
        for k, a in enumerate(by_img.get(im['id'], [])):
            uv = np.array(a['keypoints_img']) + rs.normal(0, 8.0, (MUPOTS_J, 2))
            z = np.array(a['keypoints_cam'])[:, 2] + rs.normal(0, 60.0, MUPOTS_J)
            if im['id'] == 7 and k == 0:
                z = z + 1500.0
            p17 = np.concatenate([uv, z[:, None]], 1)
            poses.append(np.concatenate([p17, p17[:4] + 5.0], 0))      # 21 joints
            scores.append(float(rs.uniform(0.3, 0.9)))
        if im['id'] == 11:          # (an image WITHOUT any detection makes the reference's procrustes() fail on NaNs:
            poses, scores = poses[:1], scores[:1]     # keep one — the other person counts as missed)
        p = np.array(poses, dtype=np.float32).reshape(-1, 21, 3)
        outs.append(dict(poses=torch.from_numpy(p), vis=torch.ones(len(p), 21), scores=scores,
                         image_paths=[os.path.join(root, im['file_name'])]))
    return outs
