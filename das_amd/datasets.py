"""Synthetic multi-person pose data with the statistics BASELINE.md section 2 prescribes.

GT row layout follows the reference datasets (mmdet3d/datasets/cmupanoptic_mono_dataset.py:218-222):
`gt_poses_3d (G, 3+4J) = [cx, cy, depth, J x (u, v, dz), J x vis]`, `centers2d (G,2)`, `depths (G,)`,
`gt_labels_3d (G,)`; images are already normalised (`img ~ N(0,1)`), `scale_factor = (1,1,1,1)`.
Real-data loaders / CPU augmentation are out of scope."""
import numpy as np
import torch


class SyntheticPoseDataset:
    CLASSES = ('person',)

    def __init__(self, num_joints=15, img_shape=(512, 832), length=1024, seed=0, root_idx=2, test_mode=False,
                 max_persons=6, **kwargs):
        self.J, self.img_shape, self.length, self.seed = num_joints, tuple(img_shape), length, seed
        self.root_idx, self.test_mode, self.max_persons = root_idx, test_mode, max_persons

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        rs = np.random.RandomState((self.seed * 7919 + idx) % (2 ** 31))
        H, W = self.img_shape
        J = self.J
        img = torch.from_numpy(rs.standard_normal((3, H, W)).astype(np.float32))
        G = int(rs.randint(1, self.max_persons + 1))
        c = np.stack([rs.uniform(40, W - 40, G), rs.uniform(40, H - 40, G)], 1)
        depth = rs.uniform(0.2, 0.7, G)
        uv = c[:, None] + rs.normal(0, 60.0, (G, J, 2))
        dz = rs.normal(0, 20.0, (G, J, 1))
        dz[:, self.root_idx] = 0
        dz[rs.uniform(0, 1, G) < 0.1] = 0  # 10% of persons carry 2-D annotation only
        vis = np.ones((G, J), dtype=np.float32)
        poses = np.concatenate([c, depth[:, None], np.concatenate([uv, dz], -1).reshape(G, 3 * J), vis], 1)
        meta = dict(filename=f'synthetic_{self.seed}_{idx}', scale_factor=np.ones(4, dtype=np.float32),
                    img_shape=(H, W, 3), pad_shape=(H, W, 3), ori_shape=(H, W, 3), flip=False)
        return dict(img=img, img_metas=meta, gt_bboxes=torch.zeros(G, 4), gt_labels=torch.zeros(G, dtype=torch.long),
                    gt_poses_3d=torch.from_numpy(poses.astype(np.float32)),
                    gt_labels_3d=torch.zeros(G, dtype=torch.long),
                    centers2d=torch.from_numpy(c.astype(np.float32)), depths=torch.from_numpy(depth.astype(np.float32)))


ANNOTATION_KEYS = ('gt_bboxes', 'gt_labels', 'gt_poses_3d', 'gt_labels_3d', 'centers2d', 'depths')


def mark_uploaded(tensors):
    """Tag device tensors with an event recorded NOW on the current stream (right after the copies that filled them): a
    consumer on another stream waits for THAT (`uploaded_events`) instead of for everything the training stream has queued
    since — the detector computes the loss's ground-truth half on a side stream and reads three counts back to the host; behind
    an event recorded at forward time that read waits for the whole previous step, i.e. the host runs in lockstep with the GPU
    and every host hiccup of a few ms stalls it."""
    ts = [t for t in tensors if torch.is_tensor(t) and t.is_cuda]
    if ts:
        ev = torch.cuda.Event()
        ev.record()
        for t in ts:
            t._das_uploaded = ev
    return tensors


def uploaded_events(*groups):
    """The distinct upload events of the tensors in `groups` (lists of tensors or None), or None if any tensor carries none."""
    evs = []
    for g in groups:
        for t in (g or ()):
            ev = getattr(t, '_das_uploaded', None)
            if ev is None:
                return None
            if all(ev is not e for e in evs):
                evs.append(ev)
    return evs


def pack_to_device(arrays, device):
    """numpy arrays -> device tensors of the same shapes / dtypes with ONE host-to-device copy: the arrays are laid out
    back to back (16-byte aligned) in one byte buffer and the results are views of its device copy."""
    offs, total = [], 0
    for a in arrays:
        total = (total + 15) // 16 * 16
        offs.append(total)
        total += a.nbytes
    host = np.zeros(max(total, 16), dtype=np.uint8)
    for a, o in zip(arrays, offs):
        if a.nbytes:
            host[o:o + a.nbytes] = np.ascontiguousarray(a).reshape(-1).view(np.uint8)
    dev = torch.from_numpy(host).to(device, non_blocking=True)
    out = []
    for a, o in zip(arrays, offs):
        dt = torch.from_numpy(np.empty(0, dtype=a.dtype)).dtype
        if a.nbytes == 0:
            out.append(torch.empty(a.shape, dtype=dt, device=device))
        else:
            out.append(dev[o:o + a.nbytes].view(dt).view(a.shape))
    return mark_uploaded(out)



def collate(samples, device=None):
    """Batch a list of samples the way mmcv's collate + scatter would hand them to `model(**data)`: images of
    different (padded) sizes are zero-padded bottom / right to the largest one (mmcv.parallel.collate on stacked
    DataContainers), annotations stay per-image lists. With a GPU `device` the ~100 small annotation tensors of a
    batch go up in ONE host-to-device copy (`pack_to_device`) instead of one each."""
    def dev(t):
        return t.to(device, non_blocking=True) if device is not None else t
    imgs = [s['img'] for s in samples]
    if len({tuple(i.shape) for i in imgs}) == 1:
        img = torch.stack(imgs)
    else:
        hm, wm = max(i.shape[-2] for i in imgs), max(i.shape[-1] for i in imgs)
        img = imgs[0].new_zeros((len(imgs), imgs[0].shape[0], hm, wm))
        for b, i in enumerate(imgs):
            img[b, :, :i.shape[-2], :i.shape[-1]] = i
    out = dict(img=dev(img), img_metas=[s['img_metas'] for s in samples])
    keys = [k for k in ANNOTATION_KEYS if k in samples[0]]
    if device is not None and torch.device(device).type == 'cuda' and \
            all(torch.is_tensor(s[k]) and not s[k].is_cuda for s in samples for k in keys):
        flat = pack_to_device([s[k].numpy() for k in keys for s in samples], device)
        for j, k in enumerate(keys):
            out[k] = flat[j * len(samples):(j + 1) * len(samples)]
    else:
        for k in keys:
            out[k] = [dev(s[k]) for s in samples]
        mark_uploaded([t for k in keys for t in out[k]])
    return out


from .registry import Registry  # noqa: E402

DATASETS = Registry('dataset')      # mmdet.datasets.builder.DATASETS in the reference (mmdet3d/datasets/builder.py)
PIPELINES = Registry('pipeline')
DATASETS.register_module(module=SyntheticPoseDataset)


class ConcatDataset:
    """mmdet.datasets.ConcatDataset: the datasets back to back (Panoptic + COCO keypoints in the reference's train set)."""

    def __init__(self, datasets):
        self.datasets = list(datasets)
        self.cumulative_sizes = list(np.cumsum([len(d) for d in self.datasets]))
        self.CLASSES = getattr(self.datasets[0], 'CLASSES', None)

    def __len__(self):
        return int(self.cumulative_sizes[-1]) if self.datasets else 0

    def __getitem__(self, idx):
        if idx < 0:
            idx += len(self)
        d = int(np.searchsorted(self.cumulative_sizes, idx, side='right'))
        return self.datasets[d][idx - (int(self.cumulative_sizes[d - 1]) if d else 0)]


def build_dataset(cfg, default_args=None):
    """mmdet3d.datasets.build_dataset (tools/train.py:196): `type=` resolves through DATASETS; a list under
    `pipeline` becomes a `das_amd.pipelines.Compose`."""
    from . import pipelines, pose_datasets  # noqa: F401  (register CMUPanopticDataset / MuPots3DHP and the transforms)
    if isinstance(cfg, (list, tuple)):      # mmdet: a list of dataset configs is their concatenation (exp_panoptic.py:161-184)
        return ConcatDataset([build_dataset(c, default_args) for c in cfg])
    cfg = dict(cfg)
    if cfg.get('type') not in DATASETS:
        raise KeyError(f"dataset type {cfg.get('type')} is not registered; available: {sorted(DATASETS.module_dict)}")
    if isinstance(cfg.get('pipeline'), (list, tuple)):
        cfg['pipeline'] = pipelines.Compose(cfg['pipeline'])
    return DATASETS.build(cfg, default_args)


def collect_results(parts, size=None):
    """Results of a multi-rank test run back in dataset order. Rank r evaluated samples r, r + world, r + 2 * world, ...
    (`parts[r]`, in that order), so the dataset order is the round-robin interleave of the parts; ranks at the tail of a
    dataset whose length is not a multiple of the world size hold one result fewer, and every result is kept. The
    reference (tools/test.py:205-206 -> mmdet `multi_gpu_test` / `collect_results`) pads the sampler so that all parts
    are equally long, interleaves with zip and trims to `len(dataset)`: the same list. `size` trims likewise."""
    out = []
    longest = max((len(part) for part in parts), default=0)
    for i in range(longest):
        for part in parts:
            if i < len(part):
                out.append(part[i])
    return out if size is None else out[:size]
