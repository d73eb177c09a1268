"""Dev tool: can the data path keep up with the trainer? (missing #2)
Builds a synthetic CMU-Panoptic-style tree (1920 x 1080 JPEG frames, COCO-style annotations, 3 persons per frame),
runs CMUPanopticDataset + the reference's train pipeline (exp_panoptic.py: multi-scale resize to ~832 x 512, flip,
photometric distortion, rotation / scale / translation, normalize, pad; augmentation on the GPU) through
das_amd.loader.PrefetchLoader with 0 / 1 / 2 / 4 / 8 worker threads and reports images per second
  (a) of the loader alone, (b) beside the 4-stage train step at samples_per_gpu = 16 (the bench workload).
usage: python tools/dev/loader_bench.py [--frames 64] [--batches 12]"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

import bench  # noqa: E402
import das_amd  # noqa: E402,F401
import eval_cases as EC  # noqa: E402
from das_amd.datasets import build_dataset, collate  # noqa: E402
from das_amd.loader import PrefetchLoader, ProcessLoader  # noqa: E402
from das_amd.optim import FlatSGD, train_iteration  # noqa: E402


def make_tree(root, frames):
    from PIL import Image
    ann = EC.panoptic_annotation(n_img=7)
    rs = np.random.RandomState(0)
    proto = [im for im in ann['images'] if sum(a['image_id'] == im['id'] for a in ann['annotations']) >= 2]
    images, anns, aid = [], [], 1
    # smooth random frames (JPEG of white noise decodes slower and compresses worse than a camera frame)
    base = rs.randint(0, 256, (68, 120, 3)).astype(np.uint8)
    for i in range(frames):
        src = proto[i % len(proto)]
        im = dict(src, id=1000 + i, file_name=f'seq/00_{i % 31:02d}/frame_{i:06d}.jpg')
        f = os.path.join(root, im['file_name'])
        os.makedirs(os.path.dirname(f), exist_ok=True)
        frame = np.asarray(Image.fromarray(np.roll(base, i, 1)).resize((1920, 1080), Image.BICUBIC))
        Image.fromarray(frame).save(f, quality=90)
        images.append(im)
        for a in ann['annotations']:
            if a['image_id'] == src['id']:
                anns.append(dict(a, id=aid, image_id=im['id']))
                aid += 1
    os.makedirs(os.path.join(root, 'annotations'), exist_ok=True)
    with open(os.path.join(root, 'annotations', 'train.json'), 'w') as f:
        json.dump(dict(images=images, annotations=anns, categories=ann['categories']), f)


def dataset_cfg(root):
    norm = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)
    pairs = [[3, 9], [4, 10], [5, 11], [6, 12], [7, 13], [8, 14]]
    pipeline = [
        dict(type='LoadImageFromFile', to_float32=True),
        dict(type='LoadAnnotationsPose3D', with_bbox=True, with_label=True),
        dict(type='ResizePose', scale_depth=True, abs_dz=True, img_scale=[(1333, 512), (1333, 512)], multiscale_mode='range',
             keep_ratio=True),
        dict(type='RandomFlipPose3D', flip_ratio_bev_horizontal=0.5, flip_pairs=pairs, num_joints=15),
        dict(type='PhotoMetricDistortion', brightness_delta=32, contrast_range=(0.7, 1.3), saturation_range=(0.7, 1.3),
             hue_delta=18),
        dict(type='GlobalRotScaleTransPose', scale_depth=True, abs_dz=True, rot_range=[-0.0, 0.0],
             scale_ratio_range=[0.95, 1.05], translation_std=[0.01, 0.01], num_joints=15, img_norm_cfg=norm,
             use_bbox_center=False),
        dict(type='Normalize', **norm),
        dict(type='Pad', size_divisor=32),
        dict(type='DefaultFormatBundlePose3D', class_names=['person']),
        dict(type='Collect3D', keys=['img', 'gt_bboxes', 'gt_labels', 'gt_poses_3d', 'gt_labels_3d', 'centers2d', 'depths']),
    ]
    return dict(type='CMUPanopticDataset', data_root=root + '/', ann_file=root + '/annotations/train.json',
                img_prefix=root + '/', pipeline=pipeline, use_bbox_center=False, abs_dz=True, norm_depth=True,
                depth_factor=1)


class Usage:
    """CPU seconds of this process (+ children) and the cgroup's throttling counters over a phase."""

    def __init__(self):
        self.t0, self.c0, self.s0 = time.perf_counter(), self._cpu(), self._stat()

    @staticmethod
    def _cpu():
        t = os.times()
        return t.user + t.system + t.children_user + t.children_system

    @staticmethod
    def _stat():
        for f in ('/sys/fs/cgroup/cpu.stat', '/sys/fs/cgroup/cpu/cpu.stat'):
            try:
                return {k: int(v) for k, v in (ln.split() for ln in open(f))}
            except OSError:
                continue
        return {}

    def __str__(self):
        dt = time.perf_counter() - self.t0
        s1 = self._stat()
        d = {k: s1[k] - self.s0.get(k, 0) for k in s1}
        thr = d.get('throttled_usec', d.get('throttled_time', 0) / 1e3) / 1e6
        grp = d.get('usage_usec', 0) / 1e6
        return (f'[{(self._cpu() - self.c0) / dt:4.1f} cores busy in this process tree, cgroup {grp / dt:4.1f} cores, '
                f'throttled {d.get("nr_throttled", 0)} periods / {thr:.2f} s of {dt:.1f} s]')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=64)
    ap.add_argument('--batches', type=int, default=40)
    ap.add_argument('--spg', type=int, default=16)
    ap.add_argument('--no-train', action='store_true')
    ap.add_argument('--probe', action='store_true', help='diagnostics: trainer on a resident batch beside the loader variants')
    ap.add_argument('--no-graphs', action='store_true', help='queue the trunk launch by launch instead of replaying its hipGraphs')
    args = ap.parse_args()
    root = tempfile.mkdtemp(prefix='das_loader_')
    t0 = time.time()
    make_tree(root, args.frames)
    print(f'{args.frames} frames of 1920 x 1080 written in {time.time() - t0:.1f} s', flush=True)
    ds = build_dataset(dataset_cfg(root))
    order = [i % len(ds) for i in range(args.batches * args.spg)]
    batches = [order[b:b + args.spg] for b in range(0, len(order), args.spg)]
    np.random.seed(0)
    one = next(iter(PrefetchLoader(ds, [[0]], collate, workers=0)))     # (re-draws when the augmentation drops the sample)
    print('sample image', tuple(one['img'].shape), flush=True)

    for workers in (0, 4):
        torch.cuda.synchronize()
        t0, use = time.perf_counter(), Usage()
        for data in PrefetchLoader(ds, batches, collate, workers=workers):
            pass
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f'loader alone, workers={workers}: {len(order) / dt:7.1f} img/s  {use}', flush=True)
    if args.no_train:
        return
    model = bench.build_model(torch.device('cuda', 0), num_stages=4, train=True)
    opt = FlatSGD(model, lr=2e-3, momentum=0.9, weight_decay=1e-4, bias_lr_mult=2.0, bias_decay_mult=0.0, max_grad_norm=35.0)
    first = next(iter(PrefetchLoader(ds, batches[:1], collate, workers=0)))
    for _ in range(3):
        train_iteration(model, opt, first, 2e-3)
    if not args.no_graphs:
        from das_amd.graphs import enable_trunk_graphs
        enable_trunk_graphs(model, opt, first['img'])
        print('trunk captured as hipGraphs', flush=True)
        train_iteration(model, opt, first, 2e-3)
    torch.cuda.synchronize()
    t0, use = time.perf_counter(), Usage()
    for _ in range(len(batches)):
        train_iteration(model, opt, first, 2e-3)
    torch.cuda.synchronize()
    print(f'train step alone (resident batch {tuple(first["img"].shape)}): {len(order) / (time.perf_counter() - t0):7.1f} img/s  {use}',
          flush=True)
    if args.probe:
        # what slows the trainer down beside the loader? (a) the full loader, its batches discarded; (b) decode only
        # (no GPU-side stages); (c) threads that only sleep
        import copy
        cfg_dec = dataset_cfg(root)
        cfg_dec['pipeline'] = cfg_dec['pipeline'][:1]
        ds_dec = build_dataset(cfg_dec)

        class Sleepy:
            def __len__(self):
                return len(ds)

            def __getitem__(self, i):
                time.sleep(0.004)
                return dict(idx=i)
        variants = (('full pipeline, batches discarded', ds, collate), ('decode + upload only', ds_dec, lambda smp, device=None: smp),
                    ('sleeping threads', Sleepy(), lambda smp, device=None: smp))
        for name, dset, coll in variants:
            for workers in (2, 4):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in PrefetchLoader(dset, batches, coll, workers=workers):
                    train_iteration(model, opt, first, 2e-3)
                torch.cuda.synchronize()
                print(f'trainer on a resident batch beside [{name}], workers={workers}: {len(order) / (time.perf_counter() - t0):7.1f} img/s',
                      flush=True)
        return
    for workers in (4,):
        for data in PrefetchLoader(ds, batches[:3], collate, workers=workers):     # (warm: allocator pools of the streams)
            train_iteration(model, opt, data, 2e-3)
        torch.cuda.synchronize()
        t0, use = time.perf_counter(), Usage()
        for data in PrefetchLoader(ds, batches, collate, workers=workers):
            train_iteration(model, opt, data, 2e-3)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f'loader threads + 4-stage train step, workers={workers}: {len(order) / dt:7.1f} img/s  {use}', flush=True)
    for workers in (4,):
        t0 = time.perf_counter()
        pl = ProcessLoader(dataset_cfg(root), workers=workers)
        print(f'{workers} worker processes up in {time.perf_counter() - t0:.1f} s', flush=True)
        for data in pl.batches(batches[:3]):      # (warm: allocator pools, kernels)
            pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for data in pl.batches(batches):
            pass
        torch.cuda.synchronize()
        print(f'loader processes alone, workers={workers}: {len(order) / (time.perf_counter() - t0):7.1f} img/s', flush=True)
        kept = list(pl.batches(batches))
        for data in kept[:3]:
            train_iteration(model, opt, data, 2e-3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for data in kept:
            train_iteration(model, opt, data, 2e-3)
        torch.cuda.synchronize()
        print(f'train step alone on {len(kept)} DIFFERENT resident batches: {len(order) / (time.perf_counter() - t0):7.1f} img/s', flush=True)
        del kept, data
        for data in pl.batches(batches[:3]):
            train_iteration(model, opt, data, 2e-3)
        torch.cuda.synchronize()
        t0, use, h0, w0, l0 = time.perf_counter(), Usage(), pl.host_seconds, pl.wait_seconds, pl.late_batches
        it, t_next, t_step = iter(pl.batches(batches)), 0.0, 0.0
        while True:
            a = time.perf_counter()
            data = next(it, None)
            b = time.perf_counter()
            if data is None:
                break
            train_iteration(model, opt, data, 2e-3)
            t_next, t_step = t_next + b - a, t_step + time.perf_counter() - b
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f'   per batch: {t_next / len(batches) * 1e3:.1f} ms in next(loader), {t_step / len(batches) * 1e3:.1f} ms in train_iteration (host), '
              f'{dt / len(batches) * 1e3:.1f} ms wall; blocked for a batch {(pl.wait_seconds - w0) / len(batches) * 1e3:.1f} ms per batch, '
              f'{pl.late_batches - l0} of {len(batches)} batches not ready one step ahead', flush=True)
        print(f'loader processes + 4-stage train step, workers={workers}: {len(order) / dt:7.1f} img/s  {use}  '
              f'(upload + replay of the image ops: {(pl.host_seconds - h0) / len(batches) * 1e3:.1f} ms of this thread per batch)', flush=True)
        pl.close()


if __name__ == '__main__':
    main()
