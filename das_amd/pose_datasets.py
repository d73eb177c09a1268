"""Dataset classes of the DAS reference: annotation parsing and evaluation.

  CMUPanopticDataset  mmdet3d/datasets/cmupanoptic_mono_dataset.py:37-424   (MPJPE)
  MuPots3DHP          mmdet3d/datasets/mupots_3dhp.py:17-336                (3DPCK, relative and absolute)

Both read the COCO-style json the reference's converters write (mytools/panoptic2coco.py, muco2coco.py) through
`CocoLite`, a minimal stand-in for pycocotools' index (not installed here). `get_ann_info` builds the training targets
`gt_poses_3d (G, 3+4J) = [cx, cy, depth, J x (u, v, dz), J x vis]`, `centers2d`, `depths` exactly as the reference's
`_parse_ann_info`; `evaluate(outputs, res_folder)` takes the detector's `simple_test` outputs, writes
`result_keypoints.json` and returns the metric dict. Image loading / augmentation lives in `das_amd.pipelines`.
"""
import json
import os
from collections import OrderedDict, defaultdict

import numpy as np

from . import evaluation as E
from .datasets import DATASETS


class CocoLite:
    """The part of pycocotools.COCO the reference's datasets use: imgs / anns / cats indices, get_ann_ids,
    get_cat_ids, load_imgs, load_anns."""

    def __init__(self, annotation):
        data = annotation
        if isinstance(annotation, (str, os.PathLike)):
            with open(annotation) as f:
                data = json.load(f)
        self.dataset = data
        self.imgs = {im['id']: im for im in data.get('images', [])}
        self.anns = {a['id']: a for a in data.get('annotations', [])}
        self.cats = {c['id']: c for c in data.get('categories', [])}
        self.img_to_anns = defaultdict(list)
        for a in data.get('annotations', []):
            self.img_to_anns[a['image_id']].append(a)

    def get_img_ids(self):
        return list(self.imgs.keys())

    def get_cat_ids(self, cat_names=()):
        if not cat_names:
            return list(self.cats.keys())
        return [c['id'] for c in self.cats.values() if c['name'] in cat_names]

    def get_ann_ids(self, img_ids=()):
        return [a['id'] for i in img_ids for a in self.img_to_anns.get(i, [])]

    def load_anns(self, ids):
        return [self.anns[i] for i in ids]

    def load_imgs(self, ids):
        return [self.imgs[i] for i in ids]


class _PoseCocoDataset:
    """What mmdet's CocoDataset gives the two subclasses: the annotation index, image list, category maps."""
    CLASSES = ('person',)

    def __init__(self, ann_file, pipeline=None, data_root=None, img_prefix='', test_mode=False, **kwargs):
        self.data_root = data_root
        if data_root is not None and isinstance(ann_file, str) and not os.path.isabs(ann_file):
            ann_file = os.path.join(data_root, ann_file)
        if data_root is not None and img_prefix and not os.path.isabs(img_prefix):
            img_prefix = os.path.join(data_root, img_prefix)
        self.ann_file, self.img_prefix, self.test_mode = ann_file, img_prefix, test_mode
        self.coco = CocoLite(ann_file)
        self.cat_ids = self.coco.get_cat_ids(cat_names=self.CLASSES)
        self.cat2label = {cat_id: i for i, cat_id in enumerate(self.cat_ids)}
        self.img_ids = self.coco.get_img_ids()
        self.data_infos = []
        for i in self.img_ids:
            info = dict(self.coco.load_imgs([i])[0])
            info['filename'] = info['file_name']
            self.data_infos.append(info)
        self.pipeline = pipeline

    def __len__(self):
        return len(self.data_infos)

    def get_ann_info(self, idx):
        img_id = self.data_infos[idx]['id']
        ann_info = self.coco.load_anns(self.coco.get_ann_ids(img_ids=[img_id]))
        return self._parse_ann_info(self.data_infos[idx], ann_info)

    def __getitem__(self, idx):
        if self.pipeline is None:
            raise RuntimeError('no pipeline configured: build the dataset with pipeline=[...] (das_amd.pipelines)')
        ann = self.get_ann_info(idx)
        if ann is None:
            return None
        results = dict(img_info=self.data_infos[idx], ann_info=ann, img_prefix=self.img_prefix, pose3d_fields=[],
                       bbox_fields=[], img_fields=[])
        return self.pipeline(results)

    # shared by both datasets (cmupanoptic_mono_dataset.py:267-356, mupots_3dhp.py:195-286)
    def _evaluate_results(self, outputs, res_folder, image_id_of, num_joints=None):
        kpts = E.collect_keypoints(outputs, image_id_of, num_joints)
        results = E.coco_keypoint_results(kpts, self.num_joints)
        E.write_keypoint_results(results, os.path.join(res_folder, 'result_keypoints.json'))
        return results

    IGNORE = 'ignore'     # a record that only marks a region (crowd, root joint not annotated)

    def _collect_persons(self, img_info, ann_info, person_of, need_area=True):
        """One pass over an image's annotation records -> the target arrays of the sample. `person_of(ann, xywh)` is
        the dataset's reading of one record: None (not a training person), IGNORE (its box goes to bboxes_ignore) or
        (root (3,), joints (J, 3), vis (J,)) with root = [u, v, depth] of the person's centre. A row of gt_poses_3d is
        [root | J x (u, v, dz) | J x vis] — the layout DASHead's target assignment reads (das_head.py:488-650);
        centers2d / depths are the root's columns."""
        boxes, labels, rows, ignored = [], [], [], []
        for ann in ann_info:
            if ann.get('ignore', False) or not self._box_ok(ann, img_info, need_area):
                continue
            x, y, w, h = ann['bbox']
            person = self.IGNORE if ann.get('iscrowd', False) else person_of(ann, (x, y, w, h))
            if person is None:
                continue
            if isinstance(person, str):
                ignored.append([x, y, x + w, y + h])
                continue
            boxes.append([x, y, x + w, y + h])
            labels.append(self.cat2label[ann['category_id']])
            rows.append(np.concatenate([np.asarray(part, dtype=float).reshape(-1) for part in person]))
        table = np.array(rows, dtype=np.float32).reshape(len(rows), 3 + self.num_joints * 4)
        labels = np.array(labels, dtype=np.int64)
        return dict(bboxes=np.array(boxes, dtype=np.float32).reshape(len(rows), 4), labels=labels,
                    gt_poses_3d=table, centers2d=table[:, :2].copy(), depths=table[:, 2].copy(),
                    gt_labels_3d=labels.copy(), bboxes_ignore=np.array(ignored, dtype=np.float32).reshape(len(ignored), 4))

    def _camera_person(self, joints, vis, focal, xywh):
        """A person annotated in camera space, joints = (J, 3) [u, v, Z] (edited in place), for the 3-D datasets:
        Z / depth_factor / focal length (`norm_depth`: depth in units of the focal length, so that it survives image
        resizing), joint depths relative to the root (`abs_dz`), the root joint — or the box centre — as the person's
        centre. Degenerate annotations (all coordinates within 10 units) are dropped, a person whose root joint is
        not annotated only marks its box as ignored."""
        depth, rel = joints[:, 2], None
        if self.norm_depth:
            depth /= self.depth_factor
            if self.abs_dz:
                rel = depth - depth[[self.ROOT_IDX]]
            depth /= focal
        if joints.max() - joints.min() < 10:
            return None
        root = joints[self.ROOT_IDX].copy()
        if self.use_bbox_center:
            x, y, w, h = xywh
            root[0], root[1] = x + 0.5 * w, y + 0.5 * h
        elif vis[self.ROOT_IDX] == 0:
            return self.IGNORE
        if rel is not None:
            joints[:, 2] = rel
        return root, joints, vis

    def _enough_visible(self, ann):
        """training samples need six visible joints over all persons (the reference's datasets re-draw otherwise)"""
        return ann['gt_poses_3d'][:, 3 + self.num_joints * 3:].sum() >= 6

    def _box_ok(self, ann, img_info, need_area=True):
        x1, y1, w, h = ann['bbox']
        inter_w = max(0, min(x1 + w, img_info['width']) - max(x1, 0))
        inter_h = max(0, min(y1 + h, img_info['height']) - max(y1, 0))
        if inter_w * inter_h == 0:
            return False
        if (ann['area'] <= 0 if need_area else ('area' in ann and ann['area'] <= 0)) or w < 1 or h < 1:
            return False
        return ann['category_id'] in self.cat_ids


@DATASETS.register_module()
class CMUPanopticDataset(_PoseCocoDataset):
    JOINTS_DEF = {'neck': 0, 'nose': 1, 'mid-hip': 2, 'l-shoulder': 3, 'l-elbow': 4, 'l-wrist': 5, 'l-hip': 6,
                  'l-knee': 7, 'l-ankle': 8, 'r-shoulder': 9, 'r-elbow': 10, 'r-wrist': 11, 'r-hip': 12, 'r-knee': 13,
                  'r-ankle': 14}
    skeleton = [[0, 1], [0, 2], [0, 3], [3, 4], [4, 5], [0, 9], [9, 10], [10, 11], [2, 6], [2, 12], [6, 7], [7, 8],
                [12, 13], [13, 14]]
    ROOT_IDX = 2

    def __init__(self, data_root=None, load_interval=1, use_bbox_center=False, norm_depth=True, abs_dz=True,
                 depth_factor=1, **kwargs):
        super().__init__(data_root=data_root, **kwargs)
        self.num_joints = len(self.JOINTS_DEF)
        self.load_interval, self.norm_depth, self.depth_factor, self.abs_dz = load_interval, norm_depth, depth_factor, abs_dz
        if abs_dz:
            assert norm_depth
        self.name2id = {os.path.basename(self.coco.load_imgs([i])[0]['file_name']): i for i in self.img_ids}
        self.use_bbox_center = use_bbox_center

    def _parse_ann_info(self, img_info, ann_info):
        """Targets of one frame (what cmupanoptic_mono_dataset.py:166-264 computes): joints3d_img = [u, v, Z_cam], focal
        length = geometric mean of fx, fy; see `_camera_person`."""
        K = img_info['cam']['K']
        focal = np.sqrt(K[0][0] * K[1][1])

        def person_of(ann, xywh):
            return self._camera_person(np.array(ann['joints3d_img'], dtype=float),
                                       np.array(ann['joints2d_vis'], dtype=float)[:, 0], focal, xywh)
        ann = self._collect_persons(img_info, ann_info, person_of)
        if not self.test_mode and (len(ann['labels']) == 0 or not self._enough_visible(ann)):
            return None
        if 'cam' in img_info:
            ann['cam'] = img_info['cam']
        return ann

    def evaluate(self, outputs, res_folder='tmp', metric='mpjpe', **kwargs):
        for m in (metric if isinstance(metric, list) else [metric]):
            if m.lower() not in ('mpjpe',):
                raise KeyError(f'metric {m.lower()} is not supported')
        results = self._evaluate_results(outputs, res_folder, lambda p: self.name2id[os.path.basename(p)])
        return OrderedDict(self.do_python_keypoint_eval(results))

    def do_python_keypoint_eval(self, results):
        if isinstance(results, str):
            if os.path.isdir(results):
                results = os.path.join(results, 'result_keypoints.json')
            with open(results) as f:
                results = json.load(f)
        images = []
        for img_id in self.img_ids:
            img = self.coco.load_imgs([img_id])[0]
            parsed = self._parse_ann_info(img, self.coco.load_anns(self.coco.get_ann_ids(img_ids=[img_id])))
            images.append(dict(image_id=img_id, cam=parsed['cam'], gt_poses_3d=parsed['gt_poses_3d']))
        anns = list(self.coco.anns.values())
        mpjpe = E.panoptic_mpjpe(results, images, [a['joints3d'] for a in anns], [a['joints3d_vis'] for a in anns],
                                 self.num_joints, self.ROOT_IDX, self.norm_depth, self.abs_dz, self.depth_factor)
        return [['MPJPE:', f'{mpjpe:.2f}mm']]


@DATASETS.register_module()
class MuPots3DHP(_PoseCocoDataset):
    joint_num = 21          # MuCo-3DHP (training) joints; MuPoTS-3D annotates the first 17
    joints_name = ('Head_top', 'Thorax', 'R_Shoulder', 'R_Elbow', 'R_Wrist', 'L_Shoulder', 'L_Elbow', 'L_Wrist', 'R_Hip',
                   'R_Knee', 'R_Ankle', 'L_Hip', 'L_Knee', 'L_Ankle', 'Pelvis', 'Spine', 'Head', 'R_Hand', 'L_Hand',
                   'R_Toe', 'L_Toe')
    original_joints_name = joints_name[:17]
    flip_pairs = ((2, 5), (3, 6), (4, 7), (8, 11), (9, 12), (10, 13))
    JOINTS_DEF = {k: i for i, k in enumerate(original_joints_name)}
    ROOT_IDX = joints_name.index('Pelvis')

    def __init__(self, use_bbox_center=False, norm_depth=False, abs_dz=False, depth_factor=1, **kwargs):
        super().__init__(**kwargs)
        self.num_joints = len(self.JOINTS_DEF)
        self.name2id = {self.coco.load_imgs([i])[0]['file_name']: i for i in self.img_ids}
        self.use_bbox_center, self.norm_depth, self.depth_factor, self.abs_dz = use_bbox_center, norm_depth, depth_factor, abs_dz
        if abs_dz:
            assert norm_depth

    def _parse_ann_info(self, img_info, ann_info):
        """Targets of one frame (what mupots_3dhp.py:67-175 computes): pseudo camera from `intrinsic` = (fx, fy, cx, cy),
        joints = [u, v, Z_cam] from keypoints_img / keypoints_cam."""
        f, c = img_info['intrinsic'][:2], img_info['intrinsic'][2:]
        focal = np.sqrt(f[0] * f[1])

        def person_of(ann, xywh):
            uv, cam_xyz = np.array(ann['keypoints_img'], dtype=float), np.array(ann['keypoints_cam'], dtype=float)
            return self._camera_person(np.concatenate([uv, cam_xyz[:, 2:]], axis=1),
                                       np.array(ann['keypoints_vis'], dtype=float).reshape(-1), focal, xywh)
        ann = self._collect_persons(img_info, ann_info, person_of, need_area=False)
        ann['cam'] = dict(K=np.array([[f[0], 0., c[0]], [0., f[1], c[1]]]), R=np.eye(3), t=np.zeros((3, 1)))
        return ann

    def evaluate(self, outputs, res_folder='tmp', metric='pck', eval_mode='all', **kwargs):
        for m in (metric if isinstance(metric, list) else [metric]):
            if m.lower() not in ('pck',):
                raise KeyError(f'metric {m.lower()} is not supported')
        root = self.data_root if self.data_root[-1] == '/' else self.data_root + '/'
        results = self._evaluate_results(outputs, res_folder, lambda p: self.name2id[p.replace(root, '')], self.num_joints)
        return OrderedDict(self.do_python_keypoint_eval(results, eval_mode=eval_mode))

    def predictions_by_name(self, results):
        """result records -> file name -> (P, 17, 3) camera-space predictions in mm (mupots_3dhp.py:289-322)."""
        id2res = defaultdict(list)
        for r in results:
            id2res[r['image_id']].append(r)
        name2pred = {}
        for img_id in self.img_ids:
            res = id2res[img_id]
            img_info = self.coco.imgs[img_id]
            cam = self._parse_ann_info(img_info, self.coco.load_anns(self.coco.get_ann_ids(img_ids=[img_id])))['cam']
            f = np.sqrt(cam['K'][0, 0] * cam['K'][1, 1])
            if len(res) == 0:
                pred = np.zeros([1, self.num_joints, 3])
            else:
                pred_img = np.array([x['keypoints'] for x in res]).reshape(len(res), -1, 3)[:, :self.num_joints]
                E.denormalise_depth(pred_img, f, self.ROOT_IDX, self.norm_depth, self.abs_dz, self.depth_factor)
                pred = E.pixel2world(pred_img.reshape(-1, 3).T, cam['K'], cam['R'], cam['t'])[-1].T.reshape(pred_img.shape)
            name2pred[img_info['file_name']] = pred
        return name2pred

    def do_python_keypoint_eval(self, results, eval_mode='all', sequences=range(20)):
        if isinstance(results, str):
            if os.path.isdir(results):
                results = os.path.join(results, 'result_keypoints.json')
            with open(results) as f:
                results = json.load(f)
        name2pred = self.predictions_by_name(results)
        seq_err, seq_err_abs = [], []
        for ts in sequences:      # (the reference forks one process per sequence; the work is a few ms each)
            annots = E.load_mupots_annot(os.path.join(self.data_root, 'TS%d/annot.mat' % (ts + 1)))
            pje, pje_abs = E.eval_mupots_sequence(annots, name2pred, ts, eval_mode)
            seq_err.append(pje)
            seq_err_abs.append(pje_abs)
        pck, pck_abs = E.mupots_pck(seq_err, seq_err_abs)
        return [('PCK_MEAN:', f'{pck:.2f}'), ('PCK_MEAN_ABS:', f'{pck_abs:.2f}')]


@DATASETS.register_module()
class MuCo3DHPDataset(_PoseCocoDataset):
    """muco_3dhp.py: the MuCo-3DHP training set of the exp_mupots config (21 joints, root = Pelvis, pseudo camera from
    the per-image focal length `f` and principal point `c`)."""
    muco_joint_num = 21
    muco_joints_name = MuPots3DHP.joints_name
    muco_flip_pairs = ((2, 5), (3, 6), (4, 7), (8, 11), (9, 12), (10, 13), (17, 18), (19, 20))
    JOINTS_DEF = {k: i for i, k in enumerate(muco_joints_name)}
    ROOT_IDX = muco_joints_name.index('Pelvis')

    def __init__(self, ann_file, pipeline=None, use_bbox_center=False, norm_depth=False, depth_factor=1, abs_dz=False,
                 **kwargs):
        if abs_dz:
            assert norm_depth
        super().__init__(ann_file, pipeline, **kwargs)
        self.norm_depth, self.depth_factor, self.abs_dz, self.use_bbox_center = norm_depth, depth_factor, abs_dz, use_bbox_center
        self.num_joints = len(self.JOINTS_DEF)
        self.name2id = {os.path.basename(self.coco.load_imgs([i])[0]['file_name']): i for i in self.img_ids}
        if self.test_mode:            # muco_3dhp.py:63-66: every fourth image that has annotations
            keep = [i for i, info in enumerate(self.data_infos) if self.coco.get_ann_ids(img_ids=[info['id']])][::4]
            self.data_infos = [self.data_infos[i] for i in keep]

    def _parse_ann_info(self, img_info, ann_info):
        """Targets of one composited frame (what muco_3dhp.py:124-246 computes): pseudo camera from the per-image focal
        length `f` and principal point `c` (axes: MuCo's y-up world), joints = [u, v, Z_cam]."""
        f, c = img_info['f'], img_info['c']
        focal = np.sqrt(f[0] * f[1])

        def person_of(ann, xywh):
            uv, cam_xyz = np.array(ann['keypoints_img'], dtype=float), np.array(ann['keypoints_cam'], dtype=float)
            return self._camera_person(np.concatenate([uv, cam_xyz[:, 2:]], axis=1),
                                       np.array(ann['keypoints_vis'], dtype=float).reshape(-1), focal, xywh)
        ann = self._collect_persons(img_info, ann_info, person_of, need_area=False)
        if not self.test_mode and (len(ann['labels']) == 0 or not self._enough_visible(ann)):
            return None
        ann['cam'] = dict(K=np.array([[f[0], 0., c[0]], [0., f[1], c[1]]]),
                          R=np.array([[1.0, 0.0, 0.0], [0.0, 0.0, -1.0], [0.0, 1.0, 0.0]]), t=np.array([[0.], [0.], [0.]]))
        return ann

    def evaluate(self, *a, **k):
        raise NotImplementedError      # (as in the reference, muco_3dhp.py:248-249)


@DATASETS.register_module()
class COCOKeypointsDataset(_PoseCocoDataset):
    """coco_keypoints_dataset.py: COCO person keypoints as 2-D-only training samples (all dz = 0, which is what routes
    them to the 2-D flows of the RLE loss), re-indexed into the target joint set (`convert_ids`: 'muco' 21 joints,
    'panoptic' 15 joints, :229-271)."""
    JOINTS_DEF = {k: i for i, k in enumerate((
        'nose', 'left_eye', 'right_eye', 'left_ear', 'right_ear', 'left_shoulder', 'right_shoulder', 'left_elbow',
        'right_elbow', 'left_wrist', 'right_wrist', 'left_hip', 'right_hip', 'left_knee', 'right_knee', 'left_ankle',
        'right_ankle'))}
    CONVERT = {'muco': [-1, -1, 6, 8, 10, 5, 7, 9, 12, 14, 16, 11, 13, 15, -1, -1, -1, -1, -1, -1, -1],
               'panoptic': [-1, 0, -1, 5, 7, 9, 11, 13, 15, 6, 8, 10, 12, 14, 16]}

    def __init__(self, data_root=None, load_interval=1, use_nms=False, use_bbox_center=False, convert_ids=None, **kwargs):
        super().__init__(data_root=data_root, **kwargs)
        assert convert_ids in (None, 'muco', 'panoptic')
        self.num_joints = len(self.JOINTS_DEF)
        self.load_interval, self.convert_ids, self.use_nms, self.use_bbox_center = load_interval, convert_ids, use_nms, use_bbox_center
        self.name2id = {os.path.basename(self.coco.load_imgs([i])[0]['file_name']): i for i in self.img_ids}

    def _parse_ann_info(self, img_info, ann_info):
        """Targets of one COCO image (what coco_keypoints_dataset.py:133-287 computes): 2-D-only persons (every depth
        0), centre = midpoint of the hips (both must be annotated) or the clipped box centre; boxes under 2 px a side
        or 64 px2 are dropped; optionally re-indexed into the MuCo / Panoptic joint set."""
        lim = np.array([img_info['width'] - 1, img_info['height'] - 1], dtype=float)
        L_HIP, R_HIP = self.JOINTS_DEF['left_hip'], self.JOINTS_DEF['right_hip']

        def person_of(ann, xywh):
            x, y, w, h = xywh
            kp = np.array(ann['keypoints']).reshape(self.num_joints, 3)
            vis = (kp[..., 2] > 0).astype(float)
            corners = np.array([x, y, x + w, y + h], dtype=float).reshape(2, 2)
            corners[:, 0] = corners[:, 0].clip(0, lim[0])
            corners[:, 1] = corners[:, 1].clip(0, lim[1])
            side = corners[1] - corners[0]
            if (side < 2).any() or side.prod() < 64:
                return None
            joints = kp.copy()
            joints[..., 2] = 0
            if self.use_bbox_center:
                root = np.zeros(3, dtype=float)
                root[:2] = corners.mean(0)
            elif vis[L_HIP] == 0 or vis[R_HIP] == 0:
                return None
            else:
                root = 0.5 * (joints[L_HIP] + joints[R_HIP])
            return root, joints, vis
        out = self._collect_persons(img_info, ann_info, person_of)
        if len(out['labels']) == 0:
            return None
        if self.convert_ids is not None:
            src = np.array(self.CONVERT[self.convert_ids], dtype=np.int64)    # target joint -> COCO joint, -1 = none
            have = src >= 0
            g, n = out['gt_poses_3d'], len(out['labels'])
            uvd = g[:, 3:3 + self.num_joints * 3].reshape(n, self.num_joints, 3)
            vis = g[:, 3 + self.num_joints * 3:]
            new_uvd = np.zeros((n, len(src), 3), dtype=np.float32)
            new_vis = np.zeros((n, len(src)), dtype=np.float32)
            new_uvd[:, have], new_vis[:, have] = uvd[:, src[have]], vis[:, src[have]]
            out['gt_poses_3d'] = np.concatenate([g[:, :3], new_uvd.reshape(n, -1), new_vis], axis=1).astype(np.float32)
            if new_vis.sum() < 6:
                return None
        return out
