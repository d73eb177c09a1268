#!/usr/bin/env python
"""Inference entry point with the reference's command line (tools/test.py:21-100):
  python tools/test.py CONFIG CHECKPOINT [--out results.pkl] [--eval mpjpe] [--cfg-options k=v ...]
                       [--launcher {none,pytorch}] [--local_rank N] [--seed S] [--deterministic]
Runs DAS.simple_test over cfg.data.test on the GPU(s). CHECKPOINT may be 'none' (random init)."""
import argparse
import os
import pickle
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import das_amd  # noqa: E402
from das_amd.config import parse_cfg_options  # noqa: E402
from das_amd.datasets import build_dataset, collate, collect_results  # noqa: E402


def parse_args():
    p = argparse.ArgumentParser(description='DAS (MI355X) test a model')
    p.add_argument('config')
    p.add_argument('checkpoint')
    p.add_argument('--out')
    p.add_argument('--fuse-conv-bn', action='store_true', help='accepted; eval-mode BN is always folded')
    p.add_argument('--format-only', action='store_true')
    p.add_argument('--eval', type=str, nargs='+')
    p.add_argument('--gpu-collect', action='store_true')
    p.add_argument('--tmpdir')
    p.add_argument('--seed', type=int, default=0)
    p.add_argument('--deterministic', action='store_true')
    p.add_argument('--cfg-options', nargs='+')
    p.add_argument('--eval-options', nargs='+')
    p.add_argument('--launcher', choices=['none', 'pytorch', 'slurm', 'mpi'], default='none')
    p.add_argument('--local_rank', type=int, default=0)
    p.add_argument('--max-batches', type=int, default=None)
    return p.parse_args()


def main():
    args = parse_args()
    cfg = das_amd.Config.fromfile(args.config)
    if args.cfg_options:
        cfg.merge_from_dict(parse_cfg_options(args.cfg_options))
    cfg.model.pretrained = None
    distributed = args.launcher != 'none'
    local_rank = int(os.environ.get('LOCAL_RANK', args.local_rank))
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.distributed.init_process_group(cfg.get('dist_params', {}).get('backend', 'nccl'))
    rank = torch.distributed.get_rank() if distributed else 0
    world = torch.distributed.get_world_size() if distributed else 1
    torch.manual_seed(args.seed)

    dataset = build_dataset(cfg.data.test)
    model = das_amd.build_model(cfg.model, test_cfg=cfg.get('test_cfg'))
    if args.checkpoint.lower() != 'none':
        ckpt = torch.load(args.checkpoint, map_location='cpu')
        sd = ckpt.get('state_dict', ckpt)
        sd = {k[7:] if k.startswith('module.') else k: v for k, v in sd.items()}
        missing, unexpected = model.load_state_dict(sd, strict=False)
        if rank == 0:
            print(f'loaded {args.checkpoint}: {len(missing)} missing / {len(unexpected)} unexpected keys')
        if 'meta' in ckpt and 'CLASSES' in ckpt['meta']:
            model.CLASSES = ckpt['meta']['CLASSES']
    else:
        model.init_weights()
    model.cuda().eval()

    results = []
    idxs = list(range(rank, len(dataset), world))
    if args.max_batches:
        idxs = idxs[:args.max_batches]
    for i in idxs:  # samples_per_gpu=1 as in the reference (tools/test.py:160-166)
        sample = dataset[i]
        if isinstance(sample['img'], (list, tuple)):   # MultiScaleFlipAug: one entry per test-time augmentation
            imgs = [t.unsqueeze(0).cuda() for t in sample['img']]
            metas = [[m] for m in sample['img_metas']]
        else:
            data = collate([sample], device='cuda')
            imgs, metas = [data['img']], [data['img_metas']]
        out = model(return_loss=False, rescale=True, img=imgs, img_metas=metas)
        for r in out:
            results.append({k: (v.cpu() if torch.is_tensor(v) else v) for k, v in r.items()})
    if distributed:
        gathered = [None] * world
        torch.distributed.all_gather_object(gathered, results)
        results = collect_results(gathered, len(dataset)) if world > 1 else results
    if rank == 0:
        n = sum(len(r['scores']) for r in results)
        print(f'{len(results)} images, {n} poses')
        if args.out:
            with open(args.out, 'wb') as f:
                pickle.dump(results, f)
        if args.eval:
            if hasattr(dataset, 'evaluate'):   # CMUPanopticDataset (mpjpe) / MuPots3DHP (pck): das_amd/evaluation.py
                kw = {}
                for kv in args.eval_options or []:     # k=v pairs, as mmcv's DictAction parses them
                    k, v = kv.split('=', 1)
                    kw[k] = v
                print(dataset.evaluate(results, metric=args.eval[0] if len(args.eval) == 1 else args.eval, **kw))
            else:
                print(f'--eval {args.eval}: {type(dataset).__name__} has no evaluator (synthetic data)')


if __name__ == '__main__':
    main()
