"""tools/train.py and tools/test.py keep the reference's command-line surface: a short synthetic-data training run
writes an mmcv-style checkpoint (dense OIHW tensors + meta) that tools/test.py loads and runs inference with.
GPU only."""
import os
import pickle
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(*argv):
    r = subprocess.run([sys.executable, *argv], cwd=ROOT, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


def test_train_then_test_cli(tmp_path):
    work = str(tmp_path / 'work')
    small = ['model.backbone.num_stages=1', 'data.samples_per_gpu=2', 'data.train.type=SyntheticPoseDataset',
             'data.train.length=8', 'data.train.img_shape=(256,384)', 'data.test.type=SyntheticPoseDataset',
             'data.test.length=3', 'data.test.img_shape=(256,384)', 'runner.max_epochs=1', 'log_config.interval=1',
             'data.val.type=SyntheticPoseDataset', 'data.val.length=2', 'data.val.img_shape=(256,384)',
             'data.workers_per_gpu=2']
    out = run('tools/train.py', 'configs/das/exp_panoptic.py', '--work-dir', work, '--max-iters', '3', '--cfg-options',
              *small)
    assert 'loss_pose' in out
    assert 'Epoch(val) [1]' in out          # validation after the epoch (the reference's EvalHook; --no-validate skips it)
    ck = os.path.join(work, 'epoch_1.pth')
    sd = torch.load(ck, map_location='cpu', weights_only=False)
    assert set(sd) == {'state_dict', 'optimizer', 'meta'} and sd['meta']['iter'] == 3
    w = sd['state_dict']['backbone.top.top.0.conv.weight']
    assert w.shape == (64, 3, 7, 7) and w.is_contiguous()           # dense OIHW, loadable by the reference
    res = str(tmp_path / 'res.pkl')
    out = run('tools/test.py', 'configs/das/exp_panoptic.py', ck, '--out', res, '--cfg-options', *small)
    assert '0 missing / 0 unexpected' in out and '3 images' in out
    with open(res, 'rb') as f:
        results = pickle.load(f)
    assert len(results) == 3 and set(results[0]) >= {'poses', 'vis', 'centers', 'image_paths', 'scores'}


def test_resume_restores_momentum_and_iteration(tmp_path):
    """--resume-from continues where the checkpoint stopped: iteration counter, momentum, and a 9-image dataset on
    one rank runs ceil(9 / 2) = 5 iterations per epoch (padded by wrap-around like DistributedSampler)."""
    work = str(tmp_path / 'w')
    small = ['model.backbone.num_stages=1', 'data.samples_per_gpu=2', 'data.train.type=SyntheticPoseDataset',
             'data.train.length=9', 'data.train.img_shape=(128,192)', 'runner.max_epochs=2', 'log_config.interval=1']
    run('tools/train.py', 'configs/das/exp_panoptic.py', '--work-dir', work, '--no-validate', '--cfg-options', *small[:-2],
        'runner.max_epochs=1', 'log_config.interval=1')
    ck1 = torch.load(os.path.join(work, 'epoch_1.pth'), map_location='cpu', weights_only=False)
    assert ck1['meta']['iter'] == 5 and ck1['optimizer']['das_steps'] == 5
    # torch SGD's state_dict layout (one group per parameter, named_parameters order): loadable by the reference's runner
    opt_sd = ck1['optimizer']
    assert set(opt_sd) >= {'state', 'param_groups'} and len(opt_sd['param_groups']) == len(opt_sd['das_names'])
    i = opt_sd['das_names'].index('backbone.top.top.0.conv.weight')
    mom = opt_sd['state'][i]['momentum_buffer']
    assert mom.shape == (64, 3, 7, 7) and float(mom.abs().max()) > 0
    assert opt_sd['param_groups'][i]['params'] == [i] and opt_sd['param_groups'][i]['momentum'] == 0.9
    run('tools/train.py', 'configs/das/exp_panoptic.py', '--work-dir', work, '--no-validate', '--resume-from',
        os.path.join(work, 'epoch_1.pth'), '--cfg-options', *small)
    ck2 = torch.load(os.path.join(work, 'epoch_2.pth'), map_location='cpu', weights_only=False)
    assert ck2['meta']['iter'] == 10 and ck2['meta']['epoch'] == 2 and ck2['optimizer']['das_steps'] == 10


def test_bench_self_launches_two_ranks():
    """`python bench.py --gpus 2` without torchrun starts its own ranks; --share-gpu puts both on
    cuda:0 over gloo so that this runs on a 1-GPU box. All three workloads."""
    import json
    for wl, extra in (('train', ['--stages', '1', '--batch', '2']), ('infer', ['--batch', '2']),
                      ('decode', ['--batch', '16'])):
        out = run('bench.py', '--gpus', '2', '--share-gpu', '--steps', '2', '--warmup', '1', '--no-cpu-baseline',
                  '--workload', wl, *extra)
        line = json.loads([l for l in out.splitlines() if l.startswith('{')][-1])
        assert line['n_gpus'] == 2 and line['steps'] == 2 and line['value'] > 0, line
        assert line['config']['parallelism'] in ('dp2', 'replicas2')
        for key in ('roofline', 'roofline_hbm'):    # the measurement objects of the contract (roofline_hbm: conv workloads)
            r = line.get(key)
            if key == 'roofline_hbm' and wl == 'decode':
                continue
            assert r is not None, (wl, key)
            assert r['bound'] in ('hbm', 'mfma') and r['unit'] == ('GB/s' if r['bound'] == 'hbm' else 'TFLOP/s')
            assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3 and 'traffic' in r


def test_test_cli_on_a_real_dataset_class_with_pipeline_and_evaluator(tmp_path):
    """tools/test.py end to end on the non-synthetic path: CMUPanopticDataset over a COCO-style annotation file,
    the reference's test pipeline (MultiScaleFlipAug / Resize / Normalize / Pad ...) on the GPU, `--eval mpjpe`."""
    import json
    import numpy as np
    from PIL import Image
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    import eval_cases as EC
    ann = EC.panoptic_annotation(n_img=3)
    root = tmp_path / 'panoptic'
    rs = np.random.RandomState(0)
    for im in ann['images']:
        f = root / im['file_name']
        f.parent.mkdir(parents=True, exist_ok=True)
        Image.fromarray(rs.randint(0, 256, (im['height'], im['width'], 3)).astype(np.uint8)).save(f, quality=60)
    (root / 'annotations').mkdir()
    with open(root / 'annotations/val.json', 'w') as f:
        json.dump(ann, f)
    cfg = tmp_path / 'cfg.py'
    cfg.write_text(f"""
_base_ = ['{ROOT}/configs/das/exp_panoptic.py']
img_norm_cfg = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)
test_pipeline = [
    dict(type='LoadImageFromFile'),
    dict(type='LoadAnnotationsPose3D', with_pose_3d=True, with_label_3d=False),
    dict(type='MultiScaleFlipAug', img_scale=(667, 320), flip=False, transforms=[
        dict(type='Resize', keep_ratio=True),
        dict(type='RandomFlipPose3D', flip_ratio_bev_horizontal=0.0,
             flip_pairs=[[3, 9], [4, 10], [5, 11], [6, 12], [7, 13], [8, 14]], num_joints=15),
        dict(type='Normalize', **img_norm_cfg),
        dict(type='Pad', size_divisor=32),
        dict(type='DefaultFormatBundlePose3D', class_names=['person'], with_label=False),
        dict(type='Collect3D', keys=['img', 'gt_poses_3d', 'depths']),
    ])]
data = dict(samples_per_gpu=1, test=dict(_delete_=True, type='CMUPanopticDataset', data_root='{root}/',
                                         ann_file='{root}/annotations/val.json', img_prefix='{root}/', pipeline=test_pipeline,
                                         test_mode=True, use_bbox_center=False, abs_dz=True, norm_depth=True, depth_factor=1))
""")
    out = run('tools/test.py', str(cfg), 'none', '--eval', 'mpjpe', '--cfg-options', 'model.backbone.num_stages=1',
              'model.test_cfg.score_thr=0.0', 'model.test_cfg.nms_post=5')
    assert '3 images' in out and 'MPJPE' in out, out[-800:]


def test_train_cli_on_a_real_dataset_class_with_the_train_pipeline(tmp_path):
    """tools/train.py on CMUPanopticDataset + the reference's train pipeline (multi-scale resize, flip, photometric
    distortion, rotation / scale / translation, normalize, pad) running on the GPU; samples the augmentation drops are
    replaced, images of different sizes are padded to a common batch shape."""
    import json
    import numpy as np
    from PIL import Image
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    import eval_cases as EC
    ann = EC.panoptic_annotation(n_img=6)
    root = tmp_path / 'panoptic'
    rs = np.random.RandomState(1)
    for im in ann['images']:
        f = root / im['file_name']
        f.parent.mkdir(parents=True, exist_ok=True)
        Image.fromarray(rs.randint(0, 256, (im['height'], im['width'], 3)).astype(np.uint8)).save(f, quality=60)
    (root / 'annotations').mkdir()
    with open(root / 'annotations/train.json', 'w') as f:
        json.dump(ann, f)
    cfg = tmp_path / 'cfg.py'
    cfg.write_text(f"""
_base_ = ['{ROOT}/configs/das/exp_panoptic.py']
img_norm_cfg = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)
train_pipeline = [
    dict(type='LoadImageFromFile', to_float32=True),
    dict(type='LoadAnnotationsPose3D', with_bbox=True, with_label=True),
    dict(type='ResizePose', scale_depth=True, abs_dz=True, img_scale=[(480, 256), (480, 288)], multiscale_mode='range',
         keep_ratio=True),
    dict(type='RandomFlipPose3D', flip_ratio_bev_horizontal=0.5,
         flip_pairs=[[3, 9], [4, 10], [5, 11], [6, 12], [7, 13], [8, 14]], num_joints=15),
    dict(type='PhotoMetricDistortion', brightness_delta=32, contrast_range=(0.7, 1.3), saturation_range=(0.7, 1.3),
         hue_delta=18),
    dict(type='GlobalRotScaleTransPose', scale_depth=True, abs_dz=True, rot_range=[-0.0, 0.0], scale_ratio_range=[0.9, 1.1],
         translation_std=[0.02, 0.02], num_joints=15, img_norm_cfg=img_norm_cfg, use_bbox_center=False),
    dict(type='Normalize', **img_norm_cfg),
    dict(type='Pad', size_divisor=32),
    dict(type='DefaultFormatBundlePose3D', class_names=['person']),
    dict(type='Collect3D', keys=['img', 'gt_bboxes', 'gt_labels', 'gt_poses_3d', 'gt_labels_3d', 'centers2d', 'depths']),
]
data = dict(samples_per_gpu=2, train=dict(_delete_=True, type='CMUPanopticDataset', data_root='{root}/',
                                          ann_file='{root}/annotations/train.json', img_prefix='{root}/',
                                          pipeline=train_pipeline, use_bbox_center=False, abs_dz=True, norm_depth=True,
                                          depth_factor=1))
""")
    # (workers_per_gpu=4 from the base config; worker_mode=thread: decode + augmentation run in prefetch threads of the
    # trainer's process, das_amd.loader.PrefetchLoader)
    out = run('tools/train.py', str(cfg), '--work-dir', str(tmp_path / 'w'), '--max-iters', '3', '--no-validate',
              '--cfg-options', 'model.backbone.num_stages=1', 'runner.max_epochs=1', 'log_config.interval=1',
              'data.worker_mode=thread')
    assert 'loss_pose' in out and 'nan' not in out.lower(), out[-800:]
    # the same with worker PROCESSES (the default: CPU-only workers decode and move the annotations, the recorded image
    # ops are replayed here on the GPU, das_amd.loader.ProcessLoader); two epochs through one pool
    out = run('tools/train.py', str(cfg), '--work-dir', str(tmp_path / 'wp'), '--max-iters', '5', '--no-validate',
              '--cfg-options', 'model.backbone.num_stages=1', 'runner.max_epochs=2', 'log_config.interval=1',
              'data.workers_per_gpu=2')
    assert 'loss_pose' in out and 'nan' not in out.lower() and 'Epoch [2]' in out, out[-800:]


def test_train_cli_with_trunk_graphs(tmp_path):
    """`hip_graphs=True`: tools/train.py captures the backbone + neck after the first step and trains on through the
    graphs (fixed-size synthetic batches), checkpointing as usual."""
    out = run('tools/train.py', 'configs/das/exp_panoptic.py', '--work-dir', str(tmp_path / 'g'), '--max-iters', '4',
              '--no-validate', '--cfg-options', 'model.backbone.num_stages=1', 'runner.max_epochs=1', 'log_config.interval=1',
              'hip_graphs=True', 'data.samples_per_gpu=2', 'data.train.type=SyntheticPoseDataset', 'data.train.length=8',
              'data.train.img_shape=(256,384)', 'data.workers_per_gpu=0')
    assert 'trunk captured as hipGraphs' in out and 'nan' not in out.lower(), out[-800:]
    assert out.count('loss_pose') >= 4 and os.path.exists(tmp_path / 'g' / 'epoch_1.pth')


def test_process_loader_hands_device_batches_over_in_order():
    """das_amd.loader.ProcessLoader: worker processes (own interpreter, own HIP context on the same GPU) build the batches,
    the collated CUDA tensors arrive here by IPC handle, in batch order, and a failing batch raises here."""
    from das_amd.loader import ProcessLoader
    cfg = dict(type='SyntheticPoseDataset', num_joints=15, img_shape=(128, 192), length=12, seed=3, max_persons=3)
    # started WITHOUT a device (how tools/train.py starts it: before its first GPU call), bound afterwards
    pl = ProcessLoader(cfg, device=None, workers=2)
    try:
        assert pl.length == 12 and pl.side is None
        assert all(r.path is None and not r.registered for r in pl.rings)     # ring files unlinked once the workers mapped them
        import glob
        assert not glob.glob('/dev/shm/das_frames_*')
        pl.bind('cuda:0')
        assert pl.side is not None and all(r.registered for r in pl.rings)
        batches = [[0, 1], [2, 3], [4, 5], [6, 7], [8, 9]]
        got = list(pl.batches(batches))
        assert len(got) == 5
        from das_amd.datasets import SyntheticPoseDataset, collate
        ds = SyntheticPoseDataset(num_joints=15, img_shape=(128, 192), length=12, seed=3, max_persons=3)
        for b, data in zip(batches, got):
            ref = collate([ds[i] for i in b], device='cuda')
            assert data['img'].is_cuda and torch.equal(data['img'], ref['img'])
            assert all(torch.equal(x, y) for x, y in zip(data['gt_poses_3d'], ref['gt_poses_3d']))
        with pytest.raises(RuntimeError, match='loader worker failed on batch 0'):
            list(pl.batches([['not an index']]))
        del got, data
    finally:
        pl.close()
