"""das_amd.loader.PrefetchLoader on the CPU (plain-Python samples): batch order, re-draw of dropped samples, exception
hand-over, bounded look-ahead — the host logic tools/train.py relies on (the reference's dataloader workers:
configs/das/exp_panoptic.py:159-160)."""
import threading
import time

import pytest

from das_amd.loader import PrefetchLoader


class Toy:
    def __init__(self, n, drop=(), fail=(), delay=None):
        self.n, self.drop, self.fail, self.delay = n, set(drop), set(fail), delay or {}
        self.calls, self.lock = [], threading.Lock()

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        with self.lock:
            self.calls.append(i)
        time.sleep(self.delay.get(i, 0.0))
        if i in self.fail:
            raise ValueError(f'sample {i}')
        return None if i in self.drop else dict(idx=i)


def keep(samples, device=None):
    return [s['idx'] for s in samples]


@pytest.mark.parametrize('workers', [0, 1, 3])
def test_batches_arrive_in_order_whatever_the_worker_timing(workers):
    ds = Toy(12, delay={0: 0.05, 5: 0.03})
    batches = [[0, 1], [2, 3], [4, 5], [6, 7], [8, 9], [10, 11]]
    assert list(PrefetchLoader(ds, batches, keep, device='cpu', workers=workers)) == batches
    assert sorted(ds.calls) == list(range(12))


@pytest.mark.parametrize('workers', [0, 2])
def test_dropped_samples_are_redrawn_from_the_following_indices(workers):
    ds = Toy(6, drop={1, 2, 5})
    out = list(PrefetchLoader(ds, [[0, 1], [4, 5]], keep, device='cpu', workers=workers))
    assert out == [[0, 3], [4, 0]]          # 1 -> 2 -> 3; 5 wraps around to 0


@pytest.mark.parametrize('workers', [0, 2])
def test_a_worker_exception_surfaces_at_its_batch(workers):
    ds = Toy(8, fail={5})
    it = iter(PrefetchLoader(ds, [[0, 1], [2, 3], [4, 5], [6, 7]], keep, device='cpu', workers=workers))
    assert next(it) == [0, 1] and next(it) == [2, 3]
    with pytest.raises(ValueError, match='sample 5'):
        next(it)


def test_look_ahead_is_bounded():
    ds = Toy(40)
    it = iter(PrefetchLoader(ds, [[i] for i in range(40)], keep, device='cpu', workers=2, depth=2))
    assert next(it) == [0]
    time.sleep(0.2)
    assert len(ds.calls) <= 1 + 2 + 2 + 1     # consumed + workers + depth (+ the permit the consumer just returned)
    assert [b for b in it] == [[i] for i in range(1, 40)]


def test_pack_to_device_keeps_shapes_dtypes_and_values():
    """One buffer, one copy: the packed views must be the arrays (mixed dtypes, odd sizes, an empty one in the middle)."""
    import numpy as np
    from das_amd.datasets import pack_to_device
    rs = np.random.RandomState(0)
    arrays = [rs.randn(3, 15, 4).astype(np.float32), np.arange(3, dtype=np.int64), np.zeros((0, 4), np.float32),
              rs.randn(5).astype(np.float64), rs.randint(0, 9, (2, 7)).astype(np.int32), rs.randn(1, 1).astype(np.float32)]
    out = pack_to_device(arrays, 'cpu')
    assert len(out) == len(arrays)
    for a, t in zip(arrays, out):
        assert tuple(t.shape) == a.shape and t.numpy().dtype == a.dtype
        np.testing.assert_array_equal(t.numpy(), a)


def test_deferred_pipeline_records_the_image_ops_without_a_gpu(tmp_path):
    """What a ProcessLoader worker does: with `pipelines.DEFER_IMAGE_OPS` the reference's train pipeline runs on a box
    without a GPU — the image comes out as a FramePlan (decoded frame + the recorded ops, with the shape every stage
    would have produced), the annotations as ever; the plan survives pickling and a re-run with the same seed records
    the same plan."""
    import copy
    import os
    import pickle
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import pipeline_cases as PC
    from das_amd import pipelines as P
    from das_amd.image_ops import FramePlan
    h, w = 270, 480
    np.save(tmp_path / 'frame.npy', np.random.RandomState(5).randint(0, 256, (h, w, 3)).astype(np.uint8))
    ann = PC.annotations(21, n=5, h=h, w=w)
    pipe = P.Compose([
        dict(type='LoadImageFromFile', to_float32=True),
        dict(type='LoadAnnotationsPose3D', with_bbox=True, with_label=True),
        dict(type='ResizePose', scale_depth=True, abs_dz=False, img_scale=[(667, 256), (667, 320)], multiscale_mode='range',
             keep_ratio=True),
        dict(type='RandomFlipPose3D', flip_ratio_bev_horizontal=1.0, flip_pairs=PC.FLIP_PAIRS, num_joints=PC.J),
        dict(type='PhotoMetricDistortion'),
        dict(type='GlobalRotScaleTransPose', scale_depth=True, abs_dz=False, rot_range=[-0.1, 0.1], scale_ratio_range=[0.9, 1.1],
             translation_std=[0.02, 0.02], num_joints=PC.J, img_norm_cfg=PC.IMG_NORM, use_bbox_center=False),
        dict(type='Normalize', **PC.IMG_NORM), dict(type='Pad', size_divisor=32),
        dict(type='DefaultFormatBundlePose3D', class_names=['person']),
        dict(type='Collect3D', keys=['img', 'gt_bboxes', 'gt_labels', 'gt_poses_3d', 'gt_labels_3d', 'centers2d', 'depths']),
    ])
    src = dict(img_info=dict(filename=str(tmp_path / 'frame.npy')), img_prefix=None,
               ann_info=dict(bboxes=ann['gt_bboxes'], labels=ann['gt_labels'], centers2d=ann['centers2d'],
                             depths=ann['depths'], gt_poses_3d=ann['gt_poses_3d'], gt_labels_3d=ann['gt_labels_3d']))
    P.DEFER_IMAGE_OPS = True
    try:
        outs = []
        for _ in range(2):
            np.random.seed(1)
            outs.append(pipe(copy.deepcopy(src)))
    finally:
        P.DEFER_IMAGE_OPS = False
    a, b = outs
    plan = a['img']
    assert isinstance(plan, FramePlan) and not plan.rgb and plan.to_float
    assert [op for op, _ in plan.ops] == ['resize_bilinear', 'flip_horizontal', 'photometric_', 'warp_affine', 'normalize_pad_chw']
    new_h, new_w = a['img_metas']['img_shape'][:2]
    assert plan.ops[0][1] == ((new_w, new_h),) and tuple(plan.shape) == (3,) + tuple(a['img_metas']['pad_shape'][:2])
    assert plan.shape[1] % 32 == 0 and plan.shape[2] % 32 == 0 and plan.frame.shape == (h, w, 3)
    again = pickle.loads(pickle.dumps(plan))
    assert again.ops == plan.ops == b['img'].ops and np.array_equal(again.frame, plan.frame)
    assert len(pickle.dumps(plan.with_frame(None))) < 2000          # (what travels in the queue message beside the ring slot)
    for k in ('gt_poses_3d', 'gt_bboxes', 'depths'):
        assert a[k].shape[0] == b[k].shape[0] > 0 and bool((a[k] == b[k]).all())


def test_frame_ring_fits_itself_to_the_shared_memory_it_finds(monkeypatch, tmp_path):
    """48 x 8 MiB per worker does not fit a 64 MiB /dev/shm. The ring shrinks to what fits (keeping a quarter
    free), or moves to the ordinary temp directory when not even the minimum fits — with a warning, never a SIGBUS — and
    its file disappears as soon as `unlink()` is called (the mappings stay valid)."""
    import os
    import warnings
    from collections import namedtuple
    import numpy as np
    from das_amd import loader
    Stat = namedtuple('Stat', 'f_bavail f_frsize')
    real = os.statvfs

    def fake(free):
        return lambda p: Stat(free // 4096, 4096) if str(p) == '/dev/shm' else real(p)
    slot = 1 << 20
    monkeypatch.setattr(os, 'statvfs', fake(24 << 20))          # room for 18 slots of 1 MiB (three quarters of 24)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        ring = loader._FrameRing.create(48, slot, pin=False)
    assert ring.slots == 18 and any('shrunk' in str(x.message) for x in w)
    assert os.path.exists(ring.path) and ring.path.startswith('/dev/shm')
    ring.slot(17, 16)[:] = np.arange(16, dtype=np.uint8)
    path = ring.path
    ring.unlink()
    assert ring.path is None and not os.path.exists(path)
    assert ring.slot(17, 16).tolist() == list(range(16))         # the mapping outlives the name
    ring.close()
    monkeypatch.setattr(os, 'statvfs', fake(4 << 20))            # not even 8 slots: the temp directory takes the ring
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        ring = loader._FrameRing.create(48, slot, pin=False)
    assert not ring.path.startswith('/dev/shm') and 8 <= ring.slots <= 16 and any('too little' in str(x.message) for x in w)
    ring.close()
    assert ring.path is None
