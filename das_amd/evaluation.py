"""Task metrics of the DAS reference, as plain numpy functions:

  * camera helpers `pixel2world` / `world2pixel`                      (mytools/vis_3d.py:4-26)
  * the COCO-style keypoint result records `result_keypoints.json`    (cmupanoptic_mono_dataset.py:313-356,
                                                                        mupots_3dhp.py:243-286)
  * root-aligned, matched MPJPE of CMU Panoptic                        (cmupanoptic_mono_dataset.py:358-424)
  * the MuPoTS-3D protocol: person matching with bone-length normalisation, Procrustes alignment, per-joint errors,
    3DPCK@150mm / AUC per joint group, relative and absolute          (mupots_3dhp.py:289-690)

Evaluation is host-side bookkeeping over a few hundred poses per image (numpy, as in the reference): nothing here is on
the GPU hot path. Each function cites the reference lines it restates; tests/golden/eval_*.npz pin them to outputs of the
reference's own functions (tests/golden/make_golden_eval.py).
"""
import json
import os
from collections import defaultdict

import numpy as np


# ------------------------------------------------------------------ camera helpers (mytools/vis_3d.py)
def world2pixel(X, K, R, t):
    """(3, N) world points -> (3, N) [u, v, depth] (vis_3d.py:4-13; note: v uses the already-updated u, as there)."""
    x = np.dot(R, X) + t
    x[0:2, :] = x[0:2, :] / (x[2, :] + 1e-5)
    x[0, :] = K[0, 0] * x[0, :] + K[0, 1] * x[1, :] + K[0, 2]
    x[1, :] = K[1, 0] * x[0, :] + K[1, 1] * x[1, :] + K[1, 2]
    return x


def pixel2world(x, K, R, t):
    """(3, N) [u, v, depth] -> (normalised image coords, camera coords, world coords) (vis_3d.py:16-26)."""
    X = x.copy()
    X[0, :] = X[0, :] - K[0, 2]
    X[1, :] = X[1, :] - K[1, 2]
    X[:2] = np.dot(np.linalg.inv(K[:2, :2]), X[:2])
    x1 = X.copy()
    X[0:2, :] = X[0:2, :] * X[2, :]
    x2 = X.copy()
    X = np.dot(np.linalg.inv(R), (X - t))
    return x1, x2, X.copy()


# ------------------------------------------------------------------ result records
def collect_keypoints(outputs, image_id_of, num_joints=None):
    """Detector outputs (list of dicts with poses (K,J,3), vis (K,J), scores, image_paths) -> per-image lists of
    {keypoints, score, vis, image_id, area} (cmupanoptic_mono_dataset.py:277-306, mupots_3dhp.py:205-236).
    image_id_of: callable path -> image id."""
    kpts = defaultdict(list)
    for out in outputs:
        poses = np.asarray(out['poses'].cpu().numpy() if hasattr(out['poses'], 'cpu') else out['poses'])
        vis = np.asarray(out['vis'].cpu().numpy() if hasattr(out['vis'], 'cpu') else out['vis'])
        if num_joints is not None:
            poses, vis = poses[:, :num_joints], vis[:, :num_joints]
        image_id = image_id_of(out['image_paths'][0])
        for p, kpt in enumerate(poses):
            area = (np.max(kpt[:, 0]) - np.min(kpt[:, 0])) * (np.max(kpt[:, 1]) - np.min(kpt[:, 1]))
            kpts[image_id].append({'keypoints': kpt[:, 0:3], 'score': out['scores'][p], 'vis': vis[p],
                                   'image_id': image_id, 'area': area})
    return list(kpts.values())


def coco_keypoint_results(keypoints, num_joints, cat_id=1):
    """`_coco_keypoint_results_one_category_kernel` (cmupanoptic_mono_dataset.py:327-356): flat keypoint list, score
    and the [x, y, w, h] box of the joints per detection."""
    cat_results = []
    for img_kpts in keypoints:
        if len(img_kpts) == 0:
            continue
        key_points = np.array([k['keypoints'] for k in img_kpts]).reshape(-1, num_joints * 3)
        for img_kpt, key_point in zip(img_kpts, key_points):
            kpt = key_point.reshape((num_joints, 3))
            left_top, right_bottom = np.amin(kpt, axis=0), np.amax(kpt, axis=0)
            w, h = right_bottom[0] - left_top[0], right_bottom[1] - left_top[1]
            cat_results.append({'image_id': img_kpt['image_id'], 'category_id': cat_id, 'keypoints': key_point.tolist(),
                                'score': float(img_kpt['score']),
                                'bbox': np.array([left_top[0], left_top[1], w, h]).tolist()})
    return cat_results


def write_keypoint_results(results, res_file):
    """result_keypoints.json as the reference writes it (sorted keys, indent 4; :313-325)."""
    os.makedirs(os.path.dirname(res_file) or '.', exist_ok=True)
    with open(res_file, 'w') as f:
        json.dump(results, f, sort_keys=True, indent=4)


# ------------------------------------------------------------------ CMU Panoptic: matched, root-aligned MPJPE
def denormalise_depth(pred_img, norm_depth_f, root_idx, norm_depth, abs_dz, depth_factor):
    """Undo the dataset's depth normalisation on predictions (cmupanoptic_mono_dataset.py:388-396): in place."""
    if norm_depth:
        if abs_dz:
            root_depth = pred_img[:, [root_idx], 2]
            dz = pred_img[..., 2] - root_depth
            pred_img[..., 2] = root_depth * norm_depth_f + dz
            pred_img[..., 2] *= depth_factor
        else:
            pred_img[..., 2] *= norm_depth_f * depth_factor
    return pred_img


def match_by_mean_distance(preds, gts, vis):
    """For every GT person the prediction with the smallest mean visible-joint distance (`vectorize_distance`, :358-363)."""
    d = np.sqrt(((gts[:, None] - preds[None]) ** 2).sum(axis=-1)) * vis[:, None]
    return d.mean(-1).argmin(1)


def joint_errors(preds, gts, vis):
    assert preds.shape == gts.shape == (*vis.shape, 3), (preds.shape, gts.shape, vis.shape)
    return np.sqrt(((preds[vis > 0] - gts[vis > 0]) ** 2).sum(axis=-1))


def panoptic_mpjpe(results, images, all_joints3d, all_joints3d_vis, num_joints, root_idx, norm_depth=True, abs_dz=True,
                   depth_factor=1):
    """`do_python_keypoint_eval` (cmupanoptic_mono_dataset.py:372-424).
    results: result records (coco_keypoint_results). images: per evaluated image, in dataset order, a dict
    {image_id, cam {K,R,t}, gt_poses_3d (G, 3+4J) as `_parse_ann_info` builds it}. all_joints3d (A,J,3) [mm],
    all_joints3d_vis (A,J,3): every annotation of the split, for the mean pose that stands in when an image has no
    detection. Returns MPJPE in mm (average over GT persons)."""
    all_pose = np.array(all_joints3d, dtype=float) / 10
    all_vis = np.array(all_joints3d_vis, dtype=float)
    all_pose = all_pose - all_pose[:, [root_idx], :]
    with np.errstate(invalid='ignore', divide='ignore'):
        mean_pose = (all_pose * all_vis).sum(0) / all_vis.sum(0)
    mean_pose[np.isnan(mean_pose)] = 0
    tot, cnt = 0.0, 0
    for im in images:
        res = [x for x in results if x['image_id'] == im['image_id']]
        cam = {k: np.array(v) for k, v in im['cam'].items()}
        f = np.sqrt(cam['K'][0, 0] * cam['K'][1, 1])
        pred_img = np.array([x['keypoints'] for x in res]).reshape(-1, num_joints, 3)
        denormalise_depth(pred_img, f, root_idx, norm_depth, abs_dz, depth_factor)
        pred = pixel2world(pred_img.reshape(-1, 3).T, cam['K'], cam['R'], cam['t'])[-1].T.reshape(pred_img.shape)
        gtp = np.asarray(im['gt_poses_3d'])
        gt_img = gtp[:, 3:3 + num_joints * 3].reshape(-1, num_joints, 3).copy()
        if norm_depth and abs_dz:
            gt_img[..., 2] += gtp[:, [2]] * f
        gt = pixel2world(gt_img.reshape(-1, 3).T, cam['K'], cam['R'], cam['t'])[-1].T.reshape(gt_img.shape)
        gt_vis = gtp[:, 3 + num_joints * 3:]
        if len(gt) == 0:
            continue
        pred = pred - pred[:, [root_idx]]
        if len(pred) == 0:
            pred = np.concatenate([pred, mean_pose[None]])
        gt = gt - gt[:, [root_idx]]
        idx = match_by_mean_distance(pred, gt, gt_vis)
        jpe = joint_errors(pred[idx], gt, gt_vis)
        if len(jpe) > 0:
            tot += jpe.mean() * 10 * len(gt)     # cm -> mm, weighted by the number of GT persons (AverageMeter)
            cnt += len(gt)
    return tot / cnt if cnt else 0.0


# ------------------------------------------------------------------ MuPoTS-3D protocol
MPII_JOINT_GROUPS = [['Head', [0]], ['Neck', [1]], ['Shou', [2, 5]], ['Elbow', [3, 6]], ['Wrist', [4, 7]],
                     ['Hip', [8, 11]], ['Knee', [9, 12]], ['Ankle', [10, 13]]]
MPII_ALL_JOINTS = sum((g[1] for g in MPII_JOINT_GROUPS), [])
# 'relavant' joint set of the MPI-INF-3DHP tools (mupots_3dhp.py:403-427), 0-based
MPII_O1 = [i - 1 for i in [2, 16, 2, 3, 4, 2, 6, 7, 15, 9, 10, 15, 12, 13, 15, 15, 2]]
SAFE_TRAVERSAL = [i - 1 for i in [15, 16, 2, 1, 17, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14]]
MUPOTS_ROOT = 14


def mpii_compute_3d_pck(seq_err, pck_thresh=150):
    """Per sequence: PCK curve (0..195 mm in steps of 5), PCK@150 and AUC per joint group and over all 14 evaluated
    joints (mupots_3dhp.py:435-475)."""
    thresh = np.arange(0, 200, 5)
    curves, pcks, aucs = [], [], []
    for err in seq_err:
        err = np.array(err).astype(np.float32)
        curve, pck_seq, auc_seq = [], [], []
        for _, joints in MPII_JOINT_GROUPS:
            sel = err[:, joints]
            buff = [np.float32(sel < t).sum() / len(joints) / len(err) for t in thresh]
            curve.append(buff)
            auc_seq.append(sum(buff) / len(buff))
            pck_seq.append(np.float32(sel < pck_thresh).sum() / len(joints) / len(err))
        curve.append([np.float32(err[:, MPII_ALL_JOINTS] < t).sum() / len(err) / len(MPII_ALL_JOINTS) for t in thresh])
        pck_seq.append(np.float32(err[:, MPII_ALL_JOINTS] < pck_thresh).sum() / len(err) / len(MPII_ALL_JOINTS))
        curves.append(curve)
        pcks.append(pck_seq)
        aucs.append(auc_seq)
    return curves, pcks, aucs


def _bone_table(o1, trav):
    """(child, parent) index arrays of the walk. The reference pairs the i-th joint of the traversal with the i-th
    entry of the parent list (not the child's own parent, mupots_3dhp.py:483-486), so the table is built the same way
    and the walk below keeps its sequential meaning: a child is placed relative to wherever its listed parent
    currently is."""
    child = np.asarray(trav, dtype=np.int64)
    return child, np.asarray(o1, dtype=np.int64)[:len(child)]


def rescale_bones(pose, gt, o1, trav):
    """Give every bone of `pose` the length it has in `gt`, keeping its direction (mupots_3dhp.py:480-489).
    pose (..., 3, J) and gt (..., 3, J) broadcast against each other, so all (GT person, prediction) pairs of a
    frame go through one call. Directions and lengths of all bones come from two vectorised differences; only the
    placement walks the table (sixteen adds on the whole batch)."""
    child, parent = _bone_table(o1, trav)
    pose, gt = np.broadcast_arrays(pose, gt)
    bone = pose[..., child] - pose[..., parent]                              # (..., 3, nb)
    want = np.sqrt(np.square(gt[..., child] - gt[..., parent]).sum(axis=-2))  # (..., nb)
    have = np.sqrt(np.square(bone).sum(axis=-2))
    step = bone * (want / have)[..., None, :]
    out = np.array(pose)
    for b in range(len(child)):
        out[..., child[b]] = out[..., parent[b]] + step[..., b]
    return out


def norm_by_bone_length(pred, gt, o1, trav):
    """Single-pair form under the reference's name."""
    return rescale_bones(pred, gt, o1, trav)


def procrustes(pose, target):
    """Least-squares similarity transform (scale, proper rotation, translation) of pose (..., 3, J) onto
    target (..., 3, J), batched over the leading axes (the reference's `procrustes`, mupots_3dhp.py:492-528, one pair
    at a time; computed here in column form from the un-normalised cross-covariance: Kabsch / Umeyama)."""
    pose, target = np.broadcast_arrays(np.asarray(pose, dtype=np.float64), np.asarray(target, dtype=np.float64))
    c_pose = pose.mean(axis=-1, keepdims=True)
    c_tgt = target.mean(axis=-1, keepdims=True)
    a, b = pose - c_pose, target - c_tgt
    left, sing, right = np.linalg.svd(b @ np.swapaxes(a, -1, -2))            # b a^T = left diag(sing) right
    flip = np.sign(np.linalg.det(left @ right))                             # -1 where the best fit is a reflection
    left = left.copy()
    left[..., :, -1] *= flip[..., None]
    rot = left @ right
    gain = (sing[..., :-1].sum(axis=-1) + flip * sing[..., -1]) / np.square(a).sum(axis=(-1, -2))
    return gain[..., None, None] * (rot @ a) + c_tgt


def match_people(gt_poses, pred_poses, o1, trav, threshold=250):
    """For every GT person the index of the closest prediction, root-relative and absolute, -1 when even the closest is
    further than `threshold` mm on average (`match`, mupots_3dhp.py:531-566). Distances are taken after the prediction's
    x / y have been scaled by the ratio of the root depths and its bones set to the GT's lengths. All G x P pairs at
    once, in float32 as the reference computes them. gt: G arrays (3, 17); pred (P, 3, 17)."""
    gt = np.stack([np.asarray(g) for g in gt_poses]).astype(np.float32)      # (G, 3, J)
    pr = np.asarray(pred_poses).astype(np.float32)                           # (P, 3, J)
    gt_root = gt[:, :, MUPOTS_ROOT:MUPOTS_ROOT + 1]
    pr_root = pr[:, :, MUPOTS_ROOT:MUPOTS_ROOT + 1]
    gt_rel = (gt - gt_root)[:, None]                                         # (G, 1, 3, J)
    cand = np.repeat((pr - pr_root)[None], len(gt), axis=0)                  # (G, P, 3, J)
    cand[:, :, :2] *= (gt_root[:, None, 2:3] / pr_root[None, :, 2:3])
    cand = rescale_bones(cand, gt_rel, o1, trav)
    rel = np.sqrt(np.square(cand - gt_rel).sum(axis=2)).mean(axis=-1)        # (G, P)
    absolute = np.sqrt(np.square(cand + pr_root[None] - gt_rel - gt_root[:, None]).sum(axis=2)).mean(axis=-1)

    def pick(dist):
        best = dist.argmin(axis=1)
        # the reference's predicate (`diffs.min() > threshold` -> -1, :556-565): a NaN distance (a frame whose only
        # "prediction" is the all-zero stand-in: root depth 0) is NOT greater than the threshold and keeps index 0
        ok = ~(dist[np.arange(len(dist)), best] > threshold)
        return [int(i) if k else -1 for i, k in zip(best, ok)]
    return pick(rel), pick(absolute)


def eval_mupots_sequence(annots, name2pred, ts, eval_mode='all'):
    """`eval_mupots_abs` (:569-690) for test sequence `ts` (0-based). annots: [person][frame] dicts with annot3 (3,17)
    and is_valid, as `load_mupots_annot` returns them. name2pred: 'TS%d/img_%06d.jpg' -> (P, 17, 3) camera-space
    predictions in mm. Returns (per-joint errors relative, per-joint errors absolute): lists of (17,) arrays."""
    o1 = MPII_O1
    trav = SAFE_TRAVERSAL[1:]
    all_mode = eval_mode == 'all'
    pje, pje_abs = [], []
    num_person, num_frames = len(annots), len(annots[0])
    for i in range(num_frames):
        gt_p3d = [annots[k][i]['annot3'] for k in range(num_person) if annots[k][i]['is_valid'] == 1]
        if not gt_p3d:
            continue
        pred = np.asarray(name2pred['TS%d/img_%06d.jpg' % (ts + 1, i)]).transpose(0, 2, 1)
        invalid = pred[:, 2, MUPOTS_ROOT] == 0
        if invalid.sum() > 0:
            pred = pred[~invalid]
        if len(pred) == 0:
            pred = np.zeros((1, 3, 17))
        matches, _ = match_people(gt_p3d, pred, o1, trav)
        for k, mk in enumerate(matches):
            gt_abs = gt_p3d[k]
            gt_rel = gt_abs - gt_abs[:, MUPOTS_ROOT:MUPOTS_ROOT + 1]
            if mk != -1:
                pa = pred[mk]
                p = pa - pa[:, MUPOTS_ROOT:MUPOTS_ROOT + 1]
                p[:2] = p[:2] * (gt_abs[[2], [MUPOTS_ROOT]] / pa[[2], [MUPOTS_ROOT]])
                p = norm_by_bone_length(p, gt_rel, o1, trav)
                p_rel, p_abs = p, p + pa[:, MUPOTS_ROOT:MUPOTS_ROOT + 1]
            elif all_mode:
                p_rel = p_abs = 100000 * np.ones(gt_rel.shape)
            else:
                continue
            pje.append(np.sqrt(np.power(p_rel - gt_rel, 2).sum(axis=0)))
            pje_abs.append(np.sqrt(np.power(p_abs - gt_abs, 2).sum(axis=0)))
    return pje, pje_abs


def mupots_pck(seq_errors, seq_errors_abs):
    """PCK_MEAN / PCK_MEAN_ABS in percent: mean over sequences of the all-joints 3DPCK@150mm (:316-335)."""
    _, pck, _ = mpii_compute_3d_pck(seq_errors)
    _, pck_abs, _ = mpii_compute_3d_pck(seq_errors_abs)
    return (sum(p[-1] for p in pck) / len(pck) * 100, sum(p[-1] for p in pck_abs) / len(pck_abs) * 100)


def load_mupots_annot(fname):
    """annot.mat of a MuPoTS test sequence -> [person][frame] dicts (`load_annot`, :338-359)."""
    import scipy.io as sio
    data = sio.loadmat(fname)['annotations']
    out = []
    for j in range(data.shape[1]):
        buff = []
        for i in range(data.shape[0]):
            dt = data[i, j]
            buff.append({'annot2': dt['annot2'][0, 0], 'annot3': dt['annot3'][0, 0], 'annot3_univ': dt['univ_annot3'][0, 0],
                         'is_valid': dt['isValidFrame'][0, 0][0, 0]})
        out.append(buff)
    return out
