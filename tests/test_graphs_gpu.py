"""HIP graphs of the trunk (das_amd/graphs.py): a training run that replays the captured backbone + neck forward /
backward graphs follows the eager run — same losses, same parameters up to the f32 run-to-run floor (float atomics in
the BatchNorm statistics) — the capture leaves no trace in the model (running statistics, gradients), and a batch of
another shape falls back to the eager path."""
import numpy as np
import pytest
import torch

from test_train_step_gpu import make

pytestmark = pytest.mark.gpu
KW = dict(lr=2e-3, momentum=0.9, weight_decay=1e-4, bias_lr_mult=2.0, bias_decay_mult=0.0, max_grad_norm=35.0)


def run(steps, graphs, dtype='f32'):
    from das_amd.graphs import enable_trunk_graphs
    from das_amd.optim import FlatSGD, train_iteration
    model, data = make(dtype)
    opt = FlatSGD(model, **KW)
    losses = [train_iteration(model, opt, data, 2e-3)['log_vars']['loss']]
    if graphs:
        before = {n: b.detach().clone() for n, b in model.named_buffers()}
        g0 = opt.flat_g.clone()
        trunk = enable_trunk_graphs(model, opt, data['img'])
        assert trunk is not None and model._graphed_trunk is trunk
        for n, b in model.named_buffers():      # the capture's warm-up iterations were undone
            assert torch.equal(b, before[n]), n
        assert torch.equal(opt.flat_g, g0)
    for _ in range(steps - 1):
        losses.append(train_iteration(model, opt, data, 2e-3)['log_vars']['loss'])
    torch.cuda.synchronize()
    return model, opt, data, losses


def test_graphed_trunk_follows_the_eager_run():
    _, oe, _, le = run(4, False)
    _, oe2, _, le2 = run(4, False)
    mg, og, data, lg = run(4, True)
    floor = float((oe.flat_p - oe2.flat_p).abs().max())
    err = float((og.flat_p - oe.flat_p).abs().max())
    moved = float((oe.flat_p - run(1, False)[1].flat_p).abs().max())
    assert np.allclose(lg, le, rtol=2e-3), (lg, le)
    assert err <= max(20 * floor, 2e-2 * moved), (err, floor, moved)
    # running statistics advanced by the replays, not frozen at their capture-time values
    bn = mg.backbone.top.top[0].bn
    assert int(bn.num_batches_tracked) == 4


def test_other_batch_shape_takes_the_eager_path():
    from das_amd.datasets import SyntheticPoseDataset, collate
    from das_amd.optim import train_iteration
    model, opt, data, _ = run(2, True)
    ds = SyntheticPoseDataset(num_joints=15, img_shape=(128, 192), length=8, seed=3, max_persons=3)
    small = collate([ds[i] for i in range(2)], device='cuda')     # batch 2 instead of 4
    assert not model._graphed_trunk.matches(small['img'])
    out = train_iteration(model, opt, small, 2e-3)
    assert np.isfinite(out['log_vars']['loss'])
    out = train_iteration(model, opt, data, 2e-3)                  # ... and back on the graphs
    assert np.isfinite(out['log_vars']['loss'])


def test_bf16_graphed_training_decreases_the_loss():
    _, _, _, losses = run(8, True, 'bf16')
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
