#!/usr/bin/env python
"""Training entry point with the reference's command line (tools/train.py:25-91):
  python tools/train.py CONFIG [--work-dir D] [--resume-from CKPT] [--no-validate] [--gpus N | --gpu-ids ...]
                        [--seed S] [--deterministic] [--cfg-options k=v ...]
                        [--launcher {none,pytorch,slurm,mpi}] [--local_rank N] [--autoscale-lr]
One process per GPU (`python -m torch.distributed.run --nproc-per-node N tools/train.py CFG --launcher pytorch`),
gradients all-reduced over RCCL. Data comes from cfg.data.train (SyntheticPoseDataset in this repo)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import das_amd  # noqa: E402
from das_amd.config import parse_cfg_options  # noqa: E402
from das_amd.datasets import build_dataset, collate, collect_results  # noqa: E402
from das_amd.loader import PrefetchLoader, ProcessLoader  # noqa: E402
from das_amd.optim import GcPark, build_optimizer, finish_checks, step_lr, train_iteration  # noqa: E402


def parse_args():
    p = argparse.ArgumentParser(description='DAS (MI355X) train a detector')
    p.add_argument('config')
    p.add_argument('--work-dir')
    p.add_argument('--resume-from')
    p.add_argument('--no-validate', action='store_true')
    g = p.add_mutually_exclusive_group()
    g.add_argument('--gpus', type=int)
    g.add_argument('--gpu-ids', type=int, nargs='+')
    p.add_argument('--seed', type=int, default=0)
    p.add_argument('--deterministic', action='store_true')
    p.add_argument('--cfg-options', nargs='+')
    p.add_argument('--launcher', choices=['none', 'pytorch', 'slurm', 'mpi'], default='none')
    p.add_argument('--local_rank', type=int, default=0)
    p.add_argument('--autoscale-lr', action='store_true')
    p.add_argument('--max-iters', type=int, default=None, help='stop after this many iterations (smoke runs)')
    return p.parse_args()


def validate(model, dataset, cfg, rank, world, work_dir):
    """mmcv's (Dist)EvalHook as the reference wires it (tools/train.py:218 `validate=not args.no_validate` ->
    mmdet `train_detector`; `evaluation = dict(interval=1)`, exp_panoptic.py:218): the validation set through the test
    pipeline one image at a time, rank r taking samples r, r + world, ..., results collected in dataset order on
    rank 0 and scored by the dataset's own evaluator (MPJPE / PCK, das_amd/evaluation.py)."""
    was_training = model.training
    model.eval()
    results = []
    with torch.no_grad():
        for i in range(rank, len(dataset), world):
            sample = dataset[i]
            if isinstance(sample['img'], (list, tuple)):   # MultiScaleFlipAug: one entry per test-time augmentation
                imgs = [t.unsqueeze(0).cuda() for t in sample['img']]
                metas = [[m] for m in sample['img_metas']]
            else:
                data = collate([sample], device='cuda')
                imgs, metas = [data['img']], [data['img_metas']]
            for r in model(return_loss=False, rescale=True, img=imgs, img_metas=metas):
                results.append({k: (v.cpu() if torch.is_tensor(v) else v) for k, v in r.items()})
    model.train(was_training)
    if world > 1:
        gathered = [None] * world
        torch.distributed.all_gather_object(gathered, results)
        results = collect_results(gathered, len(dataset))
    if rank != 0:
        return None
    ev = dict(cfg.get('evaluation', {}))
    ev.pop('interval', None)
    if hasattr(dataset, 'evaluate'):
        return dataset.evaluate(results, res_folder=os.path.join(work_dir, 'val'), **ev)
    return dict(images=len(results), poses=sum(len(r['scores']) for r in results))


def main():
    args = parse_args()
    cfg = das_amd.Config.fromfile(args.config)
    if args.cfg_options:
        cfg.merge_from_dict(parse_cfg_options(args.cfg_options))
    work_dir = args.work_dir or cfg.get('work_dir') or os.path.join('./work_dirs', os.path.splitext(
        os.path.basename(args.config))[0])
    distributed = args.launcher != 'none'
    local_rank = int(os.environ.get('LOCAL_RANK', args.local_rank))
    # The loader's worker processes start BEFORE this process makes its first GPU call (they are CPU-only and never
    # create a HIP context; starting processes out of a GPU-initialised parent is what some hosts are fragile about).
    # data.worker_mode: 'process' (default, the reference's model: `workers_per_gpu` worker processes — decode, random
    # draws and annotation arithmetic; the image ops they record are replayed on this GPU one batch ahead,
    # das_amd.loader.ProcessLoader) or 'thread' (threads of this process running the whole pipeline,
    # das_amd.loader.PrefetchLoader: no start-up cost, but they share the interpreter lock with the trainer)
    pool = None
    if cfg.data.get('workers_per_gpu', 0) > 0 and cfg.data.get('worker_mode', 'process') == 'process':
        pool = ProcessLoader(cfg.data.train, device=None, workers=cfg.data.get('workers_per_gpu', 0),
                             seed=args.seed + 1000 * int(os.environ.get('RANK', 0)))
    try:
        _train(args, cfg, work_dir, distributed, local_rank, pool)
    finally:        # (an exception, Ctrl-C or a failed capture must not leave worker processes or shared memory behind)
        if pool is not None:
            pool.close()


def _train(args, cfg, work_dir, distributed, local_rank, pool):
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.distributed.init_process_group(cfg.get('dist_params', {}).get('backend', 'nccl'))
    rank = torch.distributed.get_rank() if distributed else 0
    world = torch.distributed.get_world_size() if distributed else 1
    torch.manual_seed(args.seed)  # set_random_seed (tools/train.py:172-177)
    if rank == 0:
        os.makedirs(work_dir, exist_ok=True)

    pretrained = cfg.model.get('pretrained')
    if pretrained and not os.path.isfile(pretrained):
        if rank == 0:
            print(f'pretrained weights {pretrained} not found: training from random init')
        cfg.model.pretrained = None
    model = das_amd.build_model(cfg.model, train_cfg=cfg.get('train_cfg'), test_cfg=cfg.get('test_cfg'))
    model.init_weights()
    model.cuda().train()
    dataset = build_dataset(cfg.data.train)
    model.CLASSES = dataset.CLASSES

    if args.autoscale_lr:
        cfg.optimizer['lr'] = cfg.optimizer['lr'] * world / 8
    opt = build_optimizer(model, cfg)
    start_epoch, it = 0, 0
    if args.resume_from:   # mmcv `runner.resume`: weights, optimizer state, epoch and iteration counters
        ck = torch.load(args.resume_from, map_location='cpu', weights_only=False)
        model.load_state_dict(ck['state_dict'])
        if 'optimizer' in ck:
            opt.load_state_dict(ck['optimizer'])
        start_epoch = ck.get('meta', {}).get('epoch', 0)
        it = ck.get('meta', {}).get('iter', 0)

    val_dataset = None
    eval_every = cfg.get('evaluation', {}).get('interval', 1)
    if not args.no_validate and cfg.data.get('val') is not None:
        val_cfg = dict(cfg.data.val)
        val_cfg['test_mode'] = True
        val_dataset = build_dataset(val_cfg)
    spg = cfg.data.get('samples_per_gpu', 4)
    workers = cfg.data.get('workers_per_gpu', 0)
    if pool is not None:
        pool.bind(f'cuda:{local_rank}')      # (side stream, page-locked frame rings: the workers have been running since start-up)
    lrc = cfg.get('lr_config', {})
    want_graphs = bool(cfg.get('hip_graphs', False))
    max_epochs = cfg.get('runner', {}).get('max_epochs', 12)
    log_every = cfg.get('log_config', {}).get('interval', 50)
    gcp = GcPark()      # cyclic GC parked after warm-up, collected at logging / checkpoint points (as bench.py measures)
    for epoch in range(start_epoch, max_epochs):
        order = torch.randperm(len(dataset), generator=torch.Generator().manual_seed(args.seed + epoch)).tolist()
        # torch DistributedSampler semantics (mmdet's samplers build on it): pad by wrapping around to a multiple of
        # world x samples_per_gpu, so every rank runs the SAME number of iterations (unequal counts would leave
        # the last gradient all-reduce of the epoch waiting for a rank that has already finished)
        chunk = world * spg
        total = (len(order) + chunk - 1) // chunk * chunk
        order = (order * (total // max(len(order), 1) + 1))[:total]
        order = order[rank::world]
        t0 = time.time()
        # decode + GPU-side augmentation + collate run `workers_per_gpu` threads ahead of the training thread
        index_batches = [order[b:b + spg] for b in range(0, len(order), spg)]
        loader = pool.batches(index_batches) if pool is not None else \
            PrefetchLoader(dataset, index_batches, collate, device='cuda', workers=workers)
        for bi, data in enumerate(loader):
            b = bi * spg
            lr = step_lr(opt.base_lr, epoch, it, steps=lrc.get('step', (16, 20)), warmup_iters=lrc.get('warmup_iters', 0),
                         warmup_ratio=lrc.get('warmup_ratio', 1.0))
            out = train_iteration(model, opt, data, lr)
            it += 1
            gcp.step()
            if want_graphs and getattr(model, '_graphed_trunk', None) is None:
                # `hip_graphs=True` (fixed input size; with several ranks a trunk without SyncBN layers): after the first step — every workspace and
                # schedule exists — the backbone + neck forward / backward of batches shaped like this one are captured
                # as two hipGraphs (das_amd/graphs.py); other shapes keep the launch-by-launch path
                from das_amd.graphs import enable_trunk_graphs
                lv0 = dict(out['log_vars'].items())
                del out                                       # (no autograd graph may be alive during the capture)
                try:
                    enable_trunk_graphs(model, opt, data['img'])
                    print(f'trunk captured as hipGraphs for batches of shape {tuple(data["img"].shape)}', flush=True)
                except RuntimeError as e:
                    want_graphs = False
                    print(f'hip_graphs: capture refused ({e}); continuing launch by launch', flush=True)
                out = dict(log_vars=lv0)
            if rank == 0 and it % log_every == 0:
                lv = ', '.join(f'{k}: {v:.4f}' for k, v in out['log_vars'].items())
                print(f'Epoch [{epoch + 1}][{b // spg + 1}/{len(order) // spg}] lr: {lr:.3e}, '
                      f'{(time.time() - t0) / (b // spg + 1):.3f} s/it, {lv}', flush=True)
            if it % log_every == 0:
                gcp.collect()
            if args.max_iters and it >= args.max_iters:
                break
        finish_checks()      # (SyncBN row-count answers still in flight: a mismatch must not reach a checkpoint unreported)
        gcp.collect()
        if rank == 0:  # CheckpointHook(interval=1): state_dict + meta (tools/train.py:200-210)
            # dense OIHW copies on the host: the live parameters are channels-last views of the optimizer's
            # flat buffer, and a checkpoint must load into the reference (mmcv) as well
            torch.save(dict(state_dict={k: v.detach().contiguous().cpu() for k, v in model.state_dict().items()},
                            optimizer=opt.state_dict(),
                            meta=dict(epoch=epoch + 1, iter=it, config=cfg.text, CLASSES=model.CLASSES,
                                      das_amd_version=das_amd.__version__)),
                       os.path.join(work_dir, f'epoch_{epoch + 1}.pth'))
        if val_dataset is not None and (epoch + 1) % eval_every == 0:
            metrics = validate(model, val_dataset, cfg, rank, world, work_dir)
            if rank == 0:
                print(f'Epoch(val) [{epoch + 1}] {metrics}', flush=True)
        if args.max_iters and it >= args.max_iters:
            break
    finish_checks()
    gcp.release()
    if distributed:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
