"""MSPN's `up_conv` branch (bilinear upsample -> bias-free 1x1 conv -> train-mode BatchNorm, mspn_mmpose.py:385-389) evaluated
with the conv BEFORE the upsampling (autograd.UpConvBNTrainFn), and the unit's whole merge relu(BN(in_skip(x)) + BN(...)) as
one autograd node that writes neither normalised branch (autograd.UpMergeTrainFn, csrc/upmerge.hip): the upsampling kernel
that reduces the BatchNorm statistics, and both against the reference's order on one upsample unit."""
import numpy as np
import pytest
import torch

import cases

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('shape', [(2, 8, 13, 256, 16, 26), (1, 5, 7, 64, 9, 13), (2, 32, 52, 256, 64, 104), (1, 3, 4, 2048, 6, 8),
                                   (3, 1, 1, 128, 4, 4), (2, 64, 104, 256, 128, 208)])
def test_upsample_with_statistics_kernel(shape, dtype):
    """das_upsample_bilinear_ac_stats: the tensor is bit-identical to das_upsample_bilinear_ac's; the statistics are the sums of
    the STORED values (f64 sums of the output as the yardstick); y == NULL leaves the statistics unchanged."""
    from das_amd import ops
    from das_amd.nn import bn_stats_buffer_rows
    B, H, W, C, Ho, Wo = shape
    dt = torch.float32 if dtype == 'f32' else torch.bfloat16
    x = cases.randn(3, B, H, W, C).to(DEV).to(dt).contiguous()
    ref = ops.upsample_bilinear_ac(x, Ho, Wo)
    rows = B * Ho * Wo
    for stats in (bn_stats_buffer_rows(rows, C, x.device), torch.zeros(3 * 2 * C, device=DEV)):
        stats.zero_()
        y = ops.upsample_bilinear_ac(x, Ho, Wo, stats=stats)
        assert torch.equal(y, ref)
        got = stats.view(-1, 2, C).double().sum(0)
        yd = ref.double().view(-1, C)
        want = torch.stack([yd.sum(0), (yd * yd).sum(0)])
        scale = torch.stack([yd.abs().sum(0), (yd * yd).sum(0)]) + 1e-30
        assert float(((got - want).abs() / scale).max()) < 2e-6
        only = torch.zeros_like(stats)
        assert ops.upsample_bilinear_ac(x, Ho, Wo, stats=only, stats_only=True) is None
        got2 = only.view(-1, 2, C).double().sum(0)
        assert float(((got2 - want).abs() / scale).max()) < 2e-6


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('shape', [(2, 8, 13, 256, 16, 26), (1, 5, 7, 64, 9, 13), (2, 32, 52, 256, 64, 104), (3, 1, 1, 128, 4, 4),
                                   (1, 6, 5, 32, 6, 5), (2, 4, 4, 16, 13, 9)])
def test_statistics_of_the_upsampled_tensor_from_the_low_resolution(shape, dtype):
    """das_upsample_stats_lowres: sum and sum of squares of upsample(z) from z alone (w = upsample^T 1 and the 3 x 3 stencil of
    upsample^T upsample) against f64 sums over torch's interpolation of the same z."""
    import torch.nn.functional as F
    from das_amd import ops
    B, H, W, C, Ho, Wo = shape
    dt = torch.float32 if dtype == 'f32' else torch.bfloat16
    z = (cases.randn(5, B, H, W, C) + 0.3).to(DEV).to(dt).contiguous()
    up = F.interpolate(z.double().permute(0, 3, 1, 2), size=(Ho, Wo), mode='bilinear', align_corners=True).permute(0, 2, 3, 1)
    want = torch.stack([up.reshape(-1, C).sum(0), (up * up).reshape(-1, C).sum(0)])
    scale = torch.stack([up.abs().reshape(-1, C).sum(0), (up * up).reshape(-1, C).sum(0)]) + 1e-30
    for slots in (1, 4):
        stats = torch.zeros(slots * 2 * C, device=DEV)
        ops.upsample_stats_lowres(z, Ho, Wo, stats)
        got = stats.view(slots, 2, C).double().sum(0)
        assert float(((got - want).abs() / scale).max()) < 5e-6, (slots, float(((got - want).abs() / scale).max()))


def _unit(seed, dtype):
    from das_amd.backbones import UpsampleUnit
    torch.manual_seed(seed)
    m = UpsampleUnit(1, 4, 512, 256, gen_skip=True)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.weight.data.uniform_(0.5, 1.5)
            mod.bias.data.uniform_(-0.3, 0.3)
    return m.to(DEV).train()


def _run_unit(mode, dtype, state):
    """mode: False / 'ref' = the reference's order, kernel by kernel; 'lowres' = up_conv before the upsampling, the BatchNorm
    passes as everywhere else; 'merge' = the whole merge as one node; True / 'deferred' = that, and the two skip branches
    normalised by their consumer (the default path). The unit's skips meet a third tensor as in the next stage's add."""
    from das_amd import nn as nnops
    low_res = mode not in (False, 'ref')
    from das_amd.autograd import reset_step_state
    dt = torch.float32 if dtype == 'f32' else torch.bfloat16
    m = _unit(0, dtype)
    m.load_state_dict(state) if state is not None else None
    x = cases.randn(11, 2, 16, 26, 512).to(DEV).to(dt).requires_grad_(True)
    up_x = cases.randn(12, 2, 8, 13, 256).to(DEV).to(dt).requires_grad_(True)
    g = [cases.randn(13 + i, *s).to(DEV).to(dt) for i, s in enumerate([(2, 16, 26, 256), (2, 16, 26, 512)])]
    xn = cases.randn(15, 2, 16, 26, 512).to(DEV).to(dt).requires_grad_(True)
    nnops.UPCONV_AT_LOW_RES = low_res
    nnops.UPMERGE_FUSED = mode in (True, 'merge', 'deferred')
    nnops.DEFERRED_SKIPS = mode in (True, 'deferred')
    try:
        reset_step_state()
        out, s1, s2, _ = m(x, up_x)
        assert isinstance(s1, nnops.DeferredBN) == isinstance(s2, nnops.DeferredBN) == nnops.DEFERRED_SKIPS
        nxt = nnops.skip_add(xn, s1, s2)
        (out.float() * g[0].float()).sum().add((nxt.float() * g[1].float()).sum()).backward()
        torch.cuda.synchronize()
    finally:
        nnops.UPCONV_AT_LOW_RES = nnops.UPMERGE_FUSED = nnops.DEFERRED_SKIPS = True
    res = {'out': out.detach().float(), 'next': nxt.detach().float(), 'dx': x.grad.float(), 'dup_x': up_x.grad.float(),
           'dxn': xn.grad.float()}
    res.update({'g.' + n: p.grad.detach().float() for n, p in m.named_parameters()})
    res.update({'b.' + n: b.detach().float().clone() for n, b in m.named_buffers()})
    return res, {k: v.detach().clone() for k, v in m.state_dict().items()}


def test_conv_before_upsampling_equals_the_reference_order_f32():
    """One upsample unit (in_skip + up_conv + the two skip convs), train mode, f32: output, every gradient and every BatchNorm
    buffer with the 1x1 conv moved in front of the upsampling against the reference's order. The two differ by the order of
    f32 summation only: 2e-5 of the tensor's largest magnitude (a flipped ReLU at a value of ~1e-6 moves nothing visible)."""
    _, state = _run_unit(False, 'f32', None)
    state = {k: v for k, v in _unit(0, 'f32').state_dict().items()}
    a, _ = _run_unit('ref', 'f32', state)
    for mode in ('lowres', 'merge', 'deferred'):
        b, _ = _run_unit(mode, 'f32', state)
        assert set(a) == set(b)
        worst = {}
        for k in a:
            scale = float(a[k].abs().max())
            if scale == 0:
                assert float(b[k].abs().max()) == 0, k
                continue
            worst[k] = float((a[k] - b[k]).abs().max()) / scale
        print(mode, {k: '%.1e' % v for k, v in worst.items()})
        assert max(worst.values()) < 2e-5, (mode, max(worst.items(), key=lambda kv: kv[1]))
        assert worst['out'] > 0, 'the switch did not change the path'


def test_conv_before_upsampling_bf16_band():
    """bf16: the exchanged order rounds at different places (conv output at low resolution, then the interpolated value)
    — both orders against the f32 run of the reference's order: the new order's error is within 1.5x the old one's."""
    state = {k: v for k, v in _unit(0, 'f32').state_dict().items()}
    ref, _ = _run_unit(False, 'f32', state)
    old, _ = _run_unit(False, 'bf16', state)
    for mode in ('lowres', 'merge', 'deferred'):
        _bf16_band(ref, old, _run_unit(mode, 'bf16', state)[0], mode)


def _bf16_band(ref, old, new, mode):
    for k in ('out', 'next', 'dx', 'dup_x', 'g.out_skip1.conv.weight', 'g.out_skip2.bn.weight', 'g.up_conv.conv.weight', 'g.up_conv.bn.weight', 'g.in_skip.conv.weight', 'b.up_conv.bn.running_var'):
        scale = float(ref[k].abs().max())
        e_old = float((old[k] - ref[k]).abs().max()) / scale
        e_new = float((new[k] - ref[k]).abs().max()) / scale
        print('%-7s %-28s old %.2e new %.2e' % (mode, k, e_old, e_new))
        assert e_new <= max(1.5 * e_old, 4e-3), (mode, k, e_old, e_new)


def test_unused_finest_unit_advances_the_same_running_statistics():
    """UpsampleUnit.forward_unused (the last stage's finest map nobody reads): running statistics through the statistics-only
    upsampling kernel against the full unit's."""
    from das_amd.backbones import UpsampleUnit
    torch.manual_seed(1)
    full = UpsampleUnit(3, 4, 256, 256).to(DEV).train()
    lean = UpsampleUnit(3, 4, 256, 256).to(DEV).train()
    lean.load_state_dict(full.state_dict())
    x = cases.randn(21, 2, 32, 52, 256).to(DEV)
    up_x = cases.randn(22, 2, 16, 26, 256).to(DEV)
    with torch.no_grad():
        full(x, up_x)
        assert lean.forward_unused(x, up_x) is None
    for (n, a), (_, b) in zip(full.named_buffers(), lean.named_buffers()):
        assert torch.allclose(a.float(), b.float(), rtol=1e-5, atol=1e-7), n
        assert 'num_batches' not in n or int(a) == 1


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('rows,C', [(2 * 16 * 26, 512), (333, 64), (1000, 2048), (7, 8), (4 * 32 * 52, 256)])
def test_bn_relu_add3_kernels_vs_torch_f64(rows, C, dtype):
    """das_bn_relu_add3_forward / _backward against torch autograd in f64 through x + relu(batch_norm(raw1)) +
    relu(batch_norm(raw2)) (training mode: the gradient passes through the batch statistics)."""
    import torch.nn.functional as F
    from das_amd import ops
    dt = torch.float32 if dtype == 'f32' else torch.bfloat16
    mk = lambda seed: cases.randn(seed, rows, C).to(dt)
    x, r1, r2, g = mk(1), mk(2) * 1.5 + 0.2, mk(3) * 0.7 - 0.1, mk(4)
    ga = [torch.rand(C, generator=torch.Generator().manual_seed(5 + i)) + 0.5 for i in range(2)]
    be = [torch.rand(C, generator=torch.Generator().manual_seed(7 + i)) - 0.5 for i in range(2)]
    leaves = [t.double().requires_grad_(True) for t in (x, r1, r2, ga[0], be[0], ga[1], be[1])]
    X, R1, R2, G1, B1, G2, B2 = leaves
    ref = X + F.relu(F.batch_norm(R1, None, None, G1, B1, True, 0.1, 1e-5)) + F.relu(F.batch_norm(R2, None, None, G2, B2, True, 0.1, 1e-5))
    ref.backward(g.double())
    par = []
    for r, gam, bet in ((r1, ga[0], be[0]), (r2, ga[1], be[1])):
        rd = r.double()
        mean, var = rd.mean(0), rd.var(0, unbiased=False)
        par.append(tuple(t.float().to(DEV).contiguous() for t in (mean, (var + 1e-5).rsqrt(), gam, bet)))
    xd, r1d, r2d, gd = (t.to(DEV).contiguous() for t in (x, r1, r2, g))
    out = ops.bn_relu_add3_forward(xd, r1d, par[0], r2d, par[1])
    acc = tuple(torch.full((C,), 0.25, device=DEV) for _ in range(4))
    d1, d2, sums = ops.bn_relu_add3_backward(gd, r1d, par[0], r2d, par[1], acc=acc)
    tol = 2e-5 if dtype == 'f32' else 1.2e-2

    def close(a, b, what, t=tol):
        scale = float(b.abs().max()) + 1e-30
        err = float((a.double().cpu() - b).abs().max()) / scale
        assert err < t, (what, err)

    close(out, ref.detach(), 'out')
    close(d1, R1.grad, 'd raw1')
    close(d2, R2.grad, 'd raw2')
    want = [B1.grad, G1.grad, B2.grad, G2.grad]
    for i, w in enumerate(want):
        close(sums[i * C:(i + 1) * C], w, 'sums %d' % i, 5e-5 if dtype == 'f32' else tol)
    for a, w, n in zip(acc, (G1.grad, B1.grad, G2.grad, B2.grad), ('dgamma1', 'dbeta1', 'dgamma2', 'dbeta2')):
        close(a - 0.25, w, n, 1e-4 if dtype == 'f32' else tol)


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('rows,C', [(2 * 16 * 26, 512), (333, 64), (50, 2048), (7, 8)])
def test_bn_dual_apply_vs_the_two_pass_form(rows, C, dtype):
    """das_bn_dual_apply = relu(BN3(raw3) + BNd(rawd)) (a bottleneck's last BatchNorm with the projection shortcut normalised on
    the fly) against f64, and against the two-pass form (shortcut normalised and stored, then added in bn3's pass) in the
    storage type's rounding."""
    from das_amd import ops
    dt = torch.float32 if dtype == 'f32' else torch.bfloat16
    r3, rd = (cases.randn(31, rows, C) * 1.3 + 0.1).to(dt).to(DEV), (cases.randn(32, rows, C) * 0.8 - 0.2).to(dt).to(DEV)
    par = []
    for i, r in enumerate((r3, rd)):
        x = r.double()
        gam = (torch.rand(C, generator=torch.Generator().manual_seed(40 + i)) + 0.5).to(DEV)
        bet = (torch.rand(C, generator=torch.Generator().manual_seed(50 + i)) - 0.5).to(DEV)
        par.append((x.mean(0).float(), (x.var(0, unbiased=False) + 1e-5).rsqrt().float(), gam, bet))
    aff = lambda r, p: (r.double() - p[0].double()) * p[1].double() * p[2].double() + p[3].double()
    want = torch.relu(aff(r3, par[0]) + aff(rd, par[1]))
    got = ops.bn_dual_apply(r3.view(1, 1, rows, C), par[0], rd.view(1, 1, rows, C), par[1], relu=True).view(rows, C)
    scale = float(want.abs().max())
    assert float((got.double() - want).abs().max()) / scale < (1e-6 if dtype == 'f32' else 4e-3)
    lin = ops.bn_dual_apply(r3.view(1, 1, rows, C), par[0], rd.view(1, 1, rows, C), par[1], relu=False).view(rows, C)
    assert float((lin.double() - (aff(r3, par[0]) + aff(rd, par[1]))).abs().max()) / scale < (1e-6 if dtype == 'f32' else 4e-3)
