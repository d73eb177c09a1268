"""Seeded synthetic COCO-style annotations for the training-dataset fixtures (MuCo-3DHP, COCO keypoints) and a synthetic
raw CMU-Panoptic / MuCo tree for the converter fixtures. Shared by tests/golden/make_golden_datasets.py (reference side,
authoring container) and tests/test_datasets.py (das_amd side)."""
import json
import os

import numpy as np


def muco_annotation(seed=5, n_img=6):
    rs = np.random.RandomState(seed)
    images, anns, aid = [], [], 1
    for i in range(n_img):
        f, c = [1500.0 + 3 * i, 1495.0 + i], [1024.0, 1024.0]
        images.append(dict(id=10 + i, file_name=f'augmented_set/{i:06d}.jpg', width=2048, height=2048, f=f, c=c))
        for g in range([2, 3, 1, 0, 2, 1][i]):
            cam = np.stack([rs.uniform(-900, 900, 21), rs.uniform(-900, 900, 21), rs.uniform(2500, 5000, 21)], 1)
            img = np.stack([cam[:, 0] / cam[:, 2] * f[0] + c[0], cam[:, 1] / cam[:, 2] * f[1] + c[1]], 1)
            vis = (rs.uniform(size=21) > 0.1).astype(float)
            if i == 0 and g == 1:
                vis[14] = 0          # invisible root -> bboxes_ignore
            x1, y1 = img.min(0)
            w, h = img.max(0) - img.min(0)
            anns.append(dict(id=aid, image_id=10 + i, category_id=1, iscrowd=int(i == 4 and g == 1),
                             bbox=[float(x1), float(y1), float(w), float(h)], keypoints_img=img.tolist(),
                             keypoints_cam=cam.tolist(), keypoints_vis=vis.tolist()))
            aid += 1
    return dict(images=images, annotations=anns, categories=[dict(id=1, name='person')])


def coco_annotation(seed=6, n_img=6):
    rs = np.random.RandomState(seed)
    images, anns, aid = [], [], 1
    for i in range(n_img):
        W, H = 640, 480
        images.append(dict(id=500 + i, file_name=f'train2017/{i:012d}.jpg', width=W, height=H))
        for g in range([2, 1, 3, 0, 1, 2][i]):
            cx, cy = rs.uniform(120, 520), rs.uniform(120, 360)
            kp = np.zeros((17, 3))
            kp[:, 0] = cx + rs.uniform(-60, 60, 17)
            kp[:, 1] = cy + rs.uniform(-100, 100, 17)
            kp[:, 2] = rs.choice([0, 1, 2], 17, p=[0.15, 0.25, 0.6])
            if i == 2 and g == 0:
                kp[11, 2] = 0         # a hip missing: person dropped
            if i == 4:
                kp[:, 2] = 0
                kp[[11, 12], 2] = 2   # fewer than six joints after the remap: sample dropped
            kp[kp[:, 2] == 0, :2] = 0
            x1, y1 = cx - 80, cy - 120
            w, h = (160.0, 240.0) if not (i == 5 and g == 1) else (6.0, 7.0)     # tiny box: filtered
            anns.append(dict(id=aid, image_id=500 + i, category_id=1, iscrowd=int(i == 0 and g == 1), area=float(w * h),
                             bbox=[float(x1), float(y1), float(w), float(h)], keypoints=kp.reshape(-1).tolist(),
                             num_keypoints=int((kp[:, 2] > 0).sum())))
            aid += 1
    return dict(images=images, annotations=anns, categories=[dict(id=1, name='person')])


# ------------------------------------------------------------------ raw trees for the converters
def write_panoptic_tree(root, seed=8):
    """A miniature CMU-Panoptic layout: <seq>/calibration_<seq>.json, <seq>/hdPose3d_stage1_coco19/body3DScene_*.json,
    <seq>/hdImgs/<panel>_<node>/<panel>_<node>_*.jpg (empty files: the converter only checks that they exist)."""
    rs = np.random.RandomState(seed)
    seqs = ['160224_haggling1', '160226_mafia2', '160224_mafia1', '160224_mafia2', '160224_ultimatum1', '160224_ultimatum2',
            '160226_haggling1', '160422_haggling1']
    for seq in seqs:
        cams = []
        for panel, node in ((0, 16), (0, 30), (0, 3)):
            ang = rs.uniform(-0.4, 0.4)
            R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
            cams.append(dict(panel=panel, node=node, K=[[1400.0 + node, 0, 960.0], [0, 1390.0 + node, 540.0], [0, 0, 1]],
                             distCoef=rs.uniform(-0.02, 0.02, 5).tolist(), R=R.tolist(),
                             t=[[rs.uniform(-20, 20)], [rs.uniform(100, 140)], [rs.uniform(250, 320)]]))
        os.makedirs(os.path.join(root, seq), exist_ok=True)
        with open(os.path.join(root, seq, f'calibration_{seq}.json'), 'w') as f:
            json.dump(dict(cameras=cams), f)
        pdir = os.path.join(root, seq, 'hdPose3d_stage1_coco19')
        os.makedirs(pdir, exist_ok=True)
        nframes = 60 if seq in ('160226_haggling1', '160422_haggling1') else 9
        for fr in range(nframes):
            bodies = []
            for b in range(int(rs.randint(0, 4)) if fr % 7 else 0):
                j = np.zeros((19, 4))
                j[:, 0] = rs.uniform(-120, 120) + rs.normal(0, 25, 19)
                j[:, 1] = rs.uniform(-160, -60) + rs.normal(0, 35, 19)
                j[:, 2] = rs.uniform(-80, 80) + rs.normal(0, 25, 19)
                j[:, 3] = rs.uniform(0, 1, 19)
                bodies.append(dict(id=b, joints19=j.reshape(-1).tolist()))
            with open(os.path.join(pdir, f'body3DScene_{fr:08d}.json'), 'w') as f:
                json.dump(dict(version=0.7, bodies=bodies), f)
            for panel, node in ((0, 16), (0, 30)):
                if fr % 11 == 5 and node == 30:
                    continue          # a missing image: skipped with a warning
                idir = os.path.join(root, seq, 'hdImgs', f'{panel:02d}_{node:02d}')
                os.makedirs(idir, exist_ok=True)
                open(os.path.join(idir, f'{panel:02d}_{node:02d}_{fr:08d}.jpg'), 'w').close()


def write_muco_tree(root, seed=9):
    rs = np.random.RandomState(seed)
    images = [dict(id=i, file_name=('unaugmented_set' if i % 3 else 'augmented_set') + f'/{i:05d}.jpg', width=2048,
                   height=2048) for i in range(23)]
    anns = [dict(id=k, image_id=int(rs.randint(0, 23)), category_id=int(rs.randint(0, 5)), bbox=rs.uniform(0, 100, 4).tolist())
            for k in range(60)]
    os.makedirs(os.path.join(root, 'annotations'), exist_ok=True)
    with open(os.path.join(root, 'annotations/MuCo-3DHP.json'), 'w') as f:
        json.dump(dict(images=images, annotations=anns), f)
