"""World-size-2 `gloo` checks of the data-parallel plumbing on CPU (no HIP compute involved):
flat gradient buckets are summed across ranks, parameter groups follow the reference's paramwise rule,
and the LR schedule equals mmcv's step + linear warm-up."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import das_amd
    from das_amd.optim import FlatSGD
    torch.manual_seed(0)
    net = das_amd.FPN([16] * 4, 24, 4, start_level=1, add_extra_convs='on_output', norm_cfg=dict(type='BN'))
    opt = FlatSGD(net, lr=0.1, bias_lr_mult=2.0, bias_decay_mult=0.0, max_grad_norm=35.0, bucket_mb=0)
    assert opt.world == world and len(opt.buckets) > 1  # bucket_mb=0 -> one element per bucket floor -> many buckets
    for i, p in enumerate(net.parameters()):
        p.grad.fill_(float(rank + 1) * (i + 1))
    opt.all_reduce_grads()
    ok = all(torch.allclose(p.grad, torch.full_like(p.grad, 3.0 * (i + 1))) for i, p in enumerate(net.parameters()))
    # BN affine parameters named 'bias' are NOT in the bias group (norm layer), conv biases would be
    names = {n for n, _ in net.named_parameters()}
    main = opt.groups[0]
    ret[rank] = (ok, main['end'] - main['start'], sum(p.numel() for p in net.parameters()), len(names))
    dist.destroy_process_group()


def test_bucketed_allreduce_world2_gloo():
    mp.set_start_method('spawn', force=True)
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + os.getpid() % 1000
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    for r in (0, 1):
        ok, main_n, total, _ = ret[r]
        assert ok
        assert main_n == total  # FPN with BN: every 'bias' belongs to a norm layer -> nothing in the bias group


def test_paramwise_groups_and_lr_schedule():
    import das_amd
    from das_amd.optim import FlatSGD, step_lr
    net = das_amd.FPN([16] * 4, 24, 4, start_level=1, add_extra_convs='on_output', norm_cfg=None)  # conv biases
    opt = FlatSGD(net, lr=0.1, bias_lr_mult=2.0, bias_decay_mult=0.0)
    nbias = sum(p.numel() for n, p in net.named_parameters() if n.endswith('.bias'))
    g = {x['key']: x for x in opt.groups}
    assert g['bias']['end'] - g['bias']['start'] == nbias and g['bias']['wd'] == 0.0 and g['bias']['lr_mult'] == 2.0
    # parameters are views of the flat buffer
    p0 = next(net.parameters())
    assert p0.data_ptr() == opt.flat_p.data_ptr() and p0.grad.data_ptr() == opt.flat_g.data_ptr()
    # schedule: lr/3 at it 0, full lr at it 250, x0.1 at epochs 16 and 20 (exp_panoptic.py:206-212)
    assert step_lr(2e-3, 0, 0) == pytest.approx(2e-3 / 3)
    assert step_lr(2e-3, 0, 125) == pytest.approx(2e-3 * (1 - 0.5 * (2 / 3)))
    assert step_lr(2e-3, 0, 250) == pytest.approx(2e-3)
    assert step_lr(2e-3, 16, 10 ** 6) == pytest.approx(2e-4) and step_lr(2e-3, 20, 10 ** 6) == pytest.approx(2e-5)
