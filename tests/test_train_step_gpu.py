"""End-to-end train step of the DAS detector on the GPU (tiny widths): losses are finite, every trainable
parameter that the reference trains receives a gradient, SGD moves the weights and the loss goes down
when the same batch is repeated; bf16 and f32 compute agree on the loss values."""
import numpy as np
import pytest
import torch

from test_model_gpu import tiny_detector_cfg

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def make(dtype):
    import das_amd
    from das_amd.datasets import SyntheticPoseDataset, collate
    torch.manual_seed(0)
    cfg = tiny_detector_cfg()
    cfg['backbone']['compute_dtype'] = dtype
    model = das_amd.build_model(cfg)
    model.init_weights()
    model.to(DEV).train()
    ds = SyntheticPoseDataset(num_joints=15, img_shape=(128, 192), length=8, seed=3, max_persons=3)
    data = collate([ds[i] for i in range(4)], device=DEV)
    return model, data


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_train_step_decreases_loss(dtype):
    from das_amd.optim import FlatSGD, train_iteration
    model, data = make(dtype)
    opt = FlatSGD(model, lr=2e-3, momentum=0.9, weight_decay=1e-4, bias_lr_mult=2.0, bias_decay_mult=0.0,
                  max_grad_norm=35.0)
    losses = []
    for it in range(8):
        out = train_iteration(model, opt, data, 2e-3)
        assert set(out['log_vars']) == {'loss_cls', 'loss_depth', 'loss_pose', 'loss_centerness', 'loss'}
        assert all(np.isfinite(v) for v in out['log_vars'].values()), out['log_vars']
        assert out['num_samples'] == 4
        losses.append(out['log_vars']['loss'])
    assert losses[-1] < losses[0], losses
    # unused by construction: stride-4 output unit of the last stage, 2-D flows without 2-D samples
    no_grad = [n for n, p in model.named_parameters() if float(p.grad.abs().max()) == 0.0]
    # ... and the root-offset branch: pose_pred[:, :2] enters no loss term in the reference either
    # (das_head.py:341-471 uses depth, uvd, sigma, centerness, cls), hence its `find_unused_parameters=True`
    allowed = ('up4', 'flow2d', 'flow3d', 'conv_reg_prevs.0.', 'conv_regs.0.')
    # (per-level Scale parameters only receive gradient from levels that own positives in this batch)
    odd = [n for n in no_grad if not any(a in n for a in allowed) and not n.startswith('bbox_head.scales')]
    assert not odd, odd[:12]
    assert len(no_grad) < 0.35 * len(list(model.parameters()))


def test_bf16_and_f32_losses_agree():
    m32, data = make('f32')
    mbf, _ = make('bf16')
    mbf.load_state_dict(m32.state_dict())
    with torch.no_grad():
        l32 = m32.train_step(data)['log_vars']
        lbf = mbf.train_step(data)['log_vars']
    for k in l32:
        assert abs(l32[k] - lbf[k]) <= 0.08 * abs(l32[k]) + 0.05, (k, l32[k], lbf[k])


def test_backward_that_raises_does_not_poison_the_next_step():
    """A backward pass that raises half way (out of memory, an assert inside a Function) skips the autograd engine's
    end-of-backward callbacks; the deferred weight-gradient queue and the 'callback queued' flags must not survive into
    the next step (FlatSGD.zero_grad resets them, all_reduce_grads / step flush and join unconditionally): the step
    after the failure produces the gradients of a clean run."""
    from das_amd import autograd as ag
    from das_amd.optim import FlatSGD
    kw = dict(lr=2e-3, momentum=0.9, weight_decay=1e-4, bias_lr_mult=2.0, bias_decay_mult=0.0, max_grad_norm=35.0)
    ref_model, data = make('f32')
    ref = FlatSGD(ref_model, **kw)
    ref.zero_grad()
    ref_model.train_step(data, None)['loss'].backward()
    ref.all_reduce_grads()
    torch.cuda.synchronize()
    g_ref = ref.flat_g.clone()
    assert float(g_ref.abs().max()) > 0

    model, _ = make('f32')
    model.load_state_dict(ref_model.state_dict())
    opt = FlatSGD(model, **kw)
    opt.zero_grad()

    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.view_as(x)

        @staticmethod
        def backward(ctx, g):
            raise RuntimeError('boom')

    # fail in the middle of backward: the head's gradients (incl. queued weight gradients) are out, the backbone's are not
    feats = model.extract_feat(data['img'])
    feats = tuple(Boom.apply(f) for f in feats)
    losses = model.bbox_head.forward_train(feats, data['img_metas'], *[data[k] for k in (
        'gt_bboxes', 'gt_labels', 'gt_poses_3d', 'gt_labels_3d', 'centers2d', 'depths')])
    loss = sum(v for k, v in losses.items() if 'loss' in k)
    with pytest.raises(RuntimeError, match='boom'):
        loss.backward()
    assert ag._pending or ag._flush_queued[0] or any(e[2] for e in ag._side.values()), \
        'the failed backward left nothing behind: the test no longer exercises the reset'
    del feats, losses, loss

    opt.zero_grad()
    assert not ag._pending and not ag._flush_queued[0]
    model.train_step(data, None)['loss'].backward()
    opt.all_reduce_grads()
    torch.cuda.synchronize()
    assert not ag._pending
    # run-to-run floor of a clean step (float atomics in the BatchNorm statistics reorder sums): a second clean model
    clean, _ = make('f32')
    clean.load_state_dict(ref_model.state_dict())
    copt = FlatSGD(clean, **kw)
    copt.zero_grad()
    clean.train_step(data, None)['loss'].backward()
    copt.all_reduce_grads()
    torch.cuda.synchronize()
    scale = float(g_ref.abs().max())
    floor = float((copt.flat_g - g_ref).abs().max()) / scale
    err = float((opt.flat_g - g_ref).abs().max()) / scale
    # (stale weight-gradient operands added into this step would double the head's gradients: an O(1) error)
    assert err < max(10 * floor, 2e-3), (err, floor)
    # and the running statistics / parameters can still step
    opt.step(2e-3)
    torch.cuda.synchronize()
    assert torch.isfinite(opt.flat_p).all()


def test_optimizer_state_is_torch_sgd_layout_both_ways():
    """FlatSGD.state_dict() is what torch.optim.SGD (built the reference's way: mmcv paramwise_cfg -> one param group per
    parameter, named_parameters order) loads, and what torch writes FlatSGD loads: checkpoints resume across the two
    (the reference's `runner.resume` -> `optimizer.load_state_dict`)."""
    from das_amd.optim import FlatSGD, train_iteration
    kw = dict(lr=2e-3, momentum=0.9, weight_decay=1e-4, bias_lr_mult=2.0, bias_decay_mult=0.0, max_grad_norm=35.0)
    model, data = make('f32')
    opt = FlatSGD(model, **kw)
    for _ in range(2):
        train_iteration(model, opt, data, 2e-3)
    torch.cuda.synchronize()
    sd = opt.state_dict()
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    clones = [torch.nn.Parameter(p.detach().clone().contiguous()) for _, p in named]
    topt = torch.optim.SGD([dict(params=[c]) for c in clones], lr=2e-3, momentum=0.9)
    topt.load_state_dict(dict(state=sd['state'], param_groups=sd['param_groups']))
    for i, ((n, p), c) in enumerate(zip(named, clones)):
        buf = topt.state[c]['momentum_buffer']
        assert buf.shape == p.shape, n
        assert torch.equal(buf.cpu(), sd['state'][i]['momentum_buffer']), n
    # bias_lr_mult = 2 / bias_decay_mult = 0 reach conv biases, not the norm layers' (mmcv DefaultOptimizerConstructor)
    g = topt.param_groups[[n for n, _ in named].index('bbox_head.conv_cls.bias')]
    assert abs(g['lr'] - 4e-3) < 1e-12 and g['weight_decay'] == 0.0
    g = topt.param_groups[[n for n, _ in named].index('backbone.top.top.0.bn.bias')]
    assert abs(g['lr'] - 2e-3) < 1e-12 and abs(g['weight_decay'] - 1e-4) < 1e-12
    # ... and back: what torch writes, a fresh FlatSGD reads
    model2, _ = make('f32')
    opt2 = FlatSGD(model2, **kw)
    opt2.load_state_dict(topt.state_dict())
    assert torch.equal(opt2.flat_m, opt.flat_m)
    # the round-2 layout of this repo still loads
    opt3 = FlatSGD(make('f32')[0], **kw)
    opt3.load_state_dict(dict(momentum_buffer=opt._dense_momentum(), steps=2, base_lr=2e-3))
    assert torch.equal(opt3.flat_m, opt.flat_m) and opt3.steps == 2


def test_optimizer_state_with_a_frozen_parameter_matches_torch_layout():
    """mmcv's constructor makes a param group for a frozen parameter too (no momentum buffer); FlatSGD's file
    must have that layout, and a file written by torch over ALL parameters must load."""
    from das_amd.optim import FlatSGD, train_iteration
    kw = dict(lr=2e-3, momentum=0.9, weight_decay=1e-4, bias_lr_mult=2.0, bias_decay_mult=0.0, max_grad_norm=35.0)
    model, data = make('f32')
    frozen = 'backbone.top.top.0.bn.weight'
    dict(model.named_parameters())[frozen].requires_grad_(False)
    opt = FlatSGD(model, **kw)
    for _ in range(2):
        train_iteration(model, opt, data, 2e-3)
    torch.cuda.synchronize()
    sd = opt.state_dict()
    names = [n for n, _ in model.named_parameters()]
    assert len(sd['param_groups']) == len(names) and names.index(frozen) not in sd['state']
    clones = [torch.nn.Parameter(p.detach().clone().contiguous(), requires_grad=p.requires_grad) for _, p in model.named_parameters()]
    topt = torch.optim.SGD([dict(params=[c]) for c in clones], lr=2e-3, momentum=0.9)
    topt.load_state_dict(dict(state=sd['state'], param_groups=sd['param_groups']))     # torch accepts the group count
    model2, _ = make('f32')
    dict(model2.named_parameters())[frozen].requires_grad_(False)
    opt2 = FlatSGD(model2, **kw)
    opt2.load_state_dict(topt.state_dict())
    assert torch.equal(opt2.flat_m, opt.flat_m)


def test_lazy_log_vars_landing_buffers_are_reused_without_mixing_steps_up():
    """detectors.LazyLogVars reuses eight page-locked landing buffers round-robin (a fresh pinned allocation per step drains
    the device while the host runs ahead): values read late — after their buffer has been handed to a later step — must still be
    the values of THEIR step (the previous owner is read out before a slot is taken again)."""
    from das_amd.detectors import LazyLogVars
    kept = []
    for i in range(40):
        vals = torch.tensor([float(i), 2.0 * i, -1.0], device='cuda')
        kept.append(LazyLogVars(['a', 'b', 'c'], vals))
        if i % 7 == 3:
            assert kept[i]['b'] == 2.0 * i          # (some are read at once, most much later)
    torch.cuda.synchronize()
    for i, lv in enumerate(kept):
        assert dict(lv) == {'a': float(i), 'b': 2.0 * i, 'c': -1.0}, i
    assert len(LazyLogVars._RING) == 8
