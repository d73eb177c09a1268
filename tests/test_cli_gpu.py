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
    r = subprocess.run([sys.executable, *argv], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


def test_train_then_test_cli(tmp_path):
    work = str(tmp_path / 'work')
    small = ['model.backbone.num_stages=1', 'data.samples_per_gpu=2', 'data.train.type=SyntheticPoseDataset',
             'data.train.length=8', 'data.train.img_shape=(256,384)', 'data.test.type=SyntheticPoseDataset',
             'data.test.length=3', 'data.test.img_shape=(256,384)', 'runner.max_epochs=1', 'log_config.interval=1']
    out = run('tools/train.py', 'configs/das/exp_panoptic.py', '--work-dir', work, '--max-iters', '3', '--cfg-options',
              *small)
    assert 'loss_pose' in out
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
    run('tools/train.py', 'configs/das/exp_panoptic.py', '--work-dir', work, '--cfg-options', *small[:-2], 'runner.max_epochs=1',
        'log_config.interval=1')
    ck1 = torch.load(os.path.join(work, 'epoch_1.pth'), map_location='cpu', weights_only=False)
    assert ck1['meta']['iter'] == 5 and ck1['optimizer']['steps'] == 5
    mom = ck1['optimizer']['momentum_buffer']['backbone.top.top.0.conv.weight']
    assert mom.shape == (64, 3, 7, 7) and float(mom.abs().max()) > 0
    run('tools/train.py', 'configs/das/exp_panoptic.py', '--work-dir', work, '--resume-from', os.path.join(work, 'epoch_1.pth'),
        '--cfg-options', *small)
    ck2 = torch.load(os.path.join(work, 'epoch_2.pth'), map_location='cpu', weights_only=False)
    assert ck2['meta']['iter'] == 10 and ck2['meta']['epoch'] == 2 and ck2['optimizer']['steps'] == 10


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
