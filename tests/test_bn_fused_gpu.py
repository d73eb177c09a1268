"""BatchNorm backward with its reduction folded into the producing data-gradient conv (DasConvDesc.bnb_*,
das_bn_backward_apply, autograd.BottleneckChainFn) against torch autograd of conv -> BatchNorm(train) -> ReLU
(mspn_mmpose.py:126-157) and against the unfused per-unit path."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import cases

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def nhwc(t, dtype):
    return t.permute(0, 2, 3, 1).contiguous().to(dtype).to(DEV)


def nchw(t):
    return t.float().cpu().permute(0, 3, 1, 2).contiguous()


def q(t, dtype):
    return t.to(dtype).float()


# (kernel forced, B, H, W, Cin (= channels of dY), Cout (= channels of the BatchNorm layer), k, with residual / y mask)
CASES = [
    ('conv_glds_kernel', 2, 11, 13, 64, 72, 3, False),
    ('conv_glds_kernel', 2, 11, 13, 64, 128, 1, True),
    ('conv_glds3_kernel', 2, 19, 23, 128, 128, 3, False),
    ('conv_glds4_kernel<pp>', 2, 13, 17, 64, 256, 1, True),
    ('conv_glds4_kernel', 1, 16, 26, 256, 264, 3, False),
    ('conv1x1_stream_kernel', 2, 91, 93, 64, 256, 1, True),     # mode 3: residual + y mask (conv1's data gradient)
    ('conv1x1_stream_kernel', 2, 91, 93, 256, 1024, 1, True),
    ('conv1x1_stream_kernel', 2, 91, 93, 256, 64, 1, False),    # mode 4: recomputed mask (conv3's data gradient, layer1)
    ('conv1x1_stream_kernel', 2, 91, 93, 128, 512, 1, False),   # mode 3 without y: no ReLU on the producing layer
]
FORCE = {
    'conv_glds_kernel': {'conv.big_minblocks': 1 << 30, 'conv.glds4_minblocks': 0, 'conv.stream_minrows': 0},
    'conv_glds3_kernel': {'conv.big_minblocks': 1, 'conv.glds4_minblocks': 0, 'conv.stream_minrows': 0},
    'conv_glds4_kernel': {'conv.glds4_minblocks': 1, 'conv.glds4_pp': 0, 'conv.stream_minrows': 0},
    'conv_glds4_kernel<pp>': {'conv.glds4_minblocks': 1, 'conv.glds4_pp': 1, 'conv.stream_minrows': 0},
    'conv1x1_stream_kernel': {},
}


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('case', CASES)
def test_dgrad_epilogue_reduces_bn_backward(case, dtype):
    from das_amd import ops as o
    kernel, B, H, W, Cin, Cout, k, with_res = case
    if dtype == torch.float32 and kernel != 'conv_glds_kernel':
        pytest.skip('the 256-row tiles and the persistent kernel are bf16 kernels')
    no_relu = kernel == 'conv1x1_stream_kernel' and Cout == 512
    dy = cases.randn(201, B, Cin, H, W)
    wf = cases.randn(202, Cout, Cin, k, k) / (Cin * k * k) ** 0.5      # already the data-gradient ("flipped") weights
    res = cases.randn(203, B, Cout, H, W) if with_res else None
    raw = cases.randn(204, B, Cout, H, W) * 1.5 + 0.3
    gamma, beta = cases.randn(205, Cout).abs() + 0.5, cases.randn(206, Cout) * 0.5
    rawq = q(raw, dtype)
    mean = rawq.mean((0, 2, 3))
    invstd = 1.0 / torch.sqrt(rawq.var((0, 2, 3), unbiased=False) + 1e-5)
    aff = (rawq - mean[None, :, None, None]) * invstd[None, :, None, None] * gamma[None, :, None, None] + beta[None, :, None, None]
    # y as the forward would have stored it: relu(affine (+ something positive-ish when a residual entered))
    y = q(F.relu(aff + (cases.randn(207, B, Cout, H, W) if with_res else 0)), dtype)
    g = q(F.conv2d(q(dy, dtype), q(wf, dtype), None, 1, k // 2), dtype)
    if with_res:
        g = g + q(res, dtype)
    mask = torch.ones_like(g) if no_relu else ((y > 0).float() if with_res else (aff > 0).float())
    dz_ref = q(g * mask, dtype)
    xhat = (rawq - mean[None, :, None, None]) * invstd[None, :, None, None]
    s_ref = torch.cat([dz_ref.sum((0, 2, 3)), (dz_ref * xhat).sum((0, 2, 3))])

    rows = B * H * W
    slots = 3
    sums = torch.zeros(slots * 2 * Cout, device=DEV)
    bnb = o.BnBwd(nhwc(raw, dtype), nhwc(y, dtype) if (with_res and not no_relu) else None, mean.to(DEV), invstd.to(DEV),
                  gamma.to(DEV), beta.to(DEV), not no_relu)
    with o.tuning(**FORCE[kernel]):
        dz = o.conv2d(nhwc(dy, dtype), o.pack_weight(wf.to(DEV), dtype), k, k, 1, k // 2,
                      residual=nhwc(res, dtype) if with_res else None, bn_bwd=bnb, stats=sums)
        assert o.last_kernel() == kernel, o.last_kernel()
    tol = dict(rtol=1.6e-2, atol=1.6e-2) if dtype == torch.bfloat16 else dict(rtol=2e-5, atol=2e-5)
    got = nchw(dz)
    # elements whose mask sits on a rounding knife edge (|affine| ~ 0) may legitimately flip: ignore |aff| < 1e-3
    safe = torch.ones_like(aff, dtype=torch.bool) if (with_res or no_relu) else (aff.abs() > 1e-3)
    np.testing.assert_allclose((got * safe).numpy(), (dz_ref * safe).numpy(), **tol)
    folded = sums.view(slots, -1).sum(0).cpu()
    # the sums are those of the kernel's own stored dZ
    s_self = torch.cat([got.sum((0, 2, 3)), (got * xhat).sum((0, 2, 3))])
    np.testing.assert_allclose(folded.numpy() / rows, s_self.numpy() / rows, rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(folded.numpy() / rows, s_ref.numpy() / rows, rtol=2e-2, atol=5e-3)   # (knife-edge masks)

    # the apply pass on (dZ, sums) equals torch autograd through BatchNorm(train)+ReLU given the same dZ
    draw = o.bn_backward_apply(dz, bnb.raw, bnb.mean, bnb.invstd, bnb.gamma, sums)
    s1, s2 = s_self[:Cout], s_self[Cout:]
    ref = (gamma * invstd)[None, :, None, None] * (got - s1[None, :, None, None] / rows - xhat * s2[None, :, None, None] / rows)
    np.testing.assert_allclose(nchw(draw).numpy(), q(ref, dtype).numpy(), **(tol if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=1e-4)))


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_chain_backward_equals_per_unit_backward(dtype):
    """MSPN2 (2 stages, tiny widths, [2,2,2,2] blocks so that block-to-block fusion is exercised) train-mode backward:
    BottleneckChainFn vs one autograd node per conv+BN unit — same forward bits, gradients equal to summation order."""
    import das_amd
    from das_amd import backbones
    res = {}
    for fused in (False, True):
        backbones.FUSED_LAYER_BACKWARD = fused
        try:
            torch.manual_seed(0)
            m = das_amd.MSPN2(unit_channels=16, num_stages=2, num_blocks=[2, 2, 2, 2], compute_dtype=dtype)
            cases.det_fill(m.state_dict(), 5)
            m.to(DEV).train()
            x = cases.randn(7, 2, 3, 64, 96).to(DEV)
            outs = m(x)
            gs = [cases.randn(80 + i, 2, 16, 16 >> i, 24 >> i).to(DEV) for i in range(4)]
            sum((o.float() * g).sum() for o, g in zip(outs, gs)).backward()
            res[fused] = ([o.detach().float().clone() for o in outs],
                          {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
        finally:
            backbones.FUSED_LAYER_BACKWARD = True
    (o0, g0), (o1, g1) = res[False], res[True]
    for a, b in zip(o0, o1):
        assert torch.equal(a, b)                      # identical forward kernels
    if dtype == 'f32':
        tol = 2e-3
    else:
        tol = 6e-2   # bf16 storage of dZ / dRaw: both paths round the same tensors, but a flipped ulp moves a ReLU mask
    errs = []
    assert set(g0) == set(g1)
    for n in g0:
        a, b = g0[n].float(), g1[n].float()
        if float(a.abs().max()) == 0:
            assert float(b.abs().max()) == 0, n
            continue
        errs.append(float((a - b).abs().max() / a.abs().max()))
    errs = np.sort(np.array(errs))
    # (train-mode BN nets are ill-conditioned, see test_train_gpu.py: the bulk must agree tightly, the tail loosely)
    assert errs[int(0.9 * len(errs))] < tol and errs[-1] < 20 * tol, (errs[int(0.9 * len(errs))], errs[-1])


def test_chain_backward_with_flat_optimizer_direct_accumulation():
    """With FlatSGD the chain adds weight gradients and BatchNorm parameter gradients straight into the flat buffer;
    the result equals the autograd-delivered gradients of the same chain."""
    import das_amd
    from das_amd.optim import FlatSGD
    grads = {}
    for flat in (False, True):
        torch.manual_seed(0)
        m = das_amd.MSPN2(unit_channels=16, num_stages=1, num_blocks=[2, 1, 2, 1], compute_dtype='f32')
        cases.det_fill(m.state_dict(), 5)
        m.to(DEV).train()
        if flat:
            opt = FlatSGD(m, lr=1e-3)
            opt.zero_grad()
        outs = m(cases.randn(7, 2, 3, 64, 96).to(DEV))
        sum((o.float() ** 2).sum() for o in outs).backward()
        torch.cuda.synchronize()
        grads[flat] = {n: p.grad.detach().float().clone() for n, p in m.named_parameters()}
    errs = []
    for n in grads[False]:
        a, b = grads[False][n], grads[True][n]
        errs.append(float((a - b).abs().max()) / max(float(a.abs().max()), 1e-6))
    errs = np.sort(np.array(errs))
    assert errs[int(0.9 * len(errs))] < 2e-3 and errs[-1] < 4e-2, (errs[int(0.9 * len(errs))], errs[-1])
