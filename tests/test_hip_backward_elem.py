"""Backward of GroupNorm / max-pool / bilinear / nearest ops vs torch autograd (CPU). GPU only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import cases

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def ops():
    from das_amd import ops as o
    return o


def nhwc(t, dtype=torch.float32):
    return t.permute(0, 2, 3, 1).contiguous().to(dtype).to(DEV)


def nchw(t):
    return t.float().cpu().permute(0, 3, 1, 2).contiguous()


def rnd(t, dtype):
    return t.to(dtype).float()


def close(a, b, dtype, scale=1.0):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    tol = (5e-5 if dtype == torch.float32 else 1.6e-2) * scale
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)
    assert err < tol, err


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('C,G', [(256, 32), (64, 32), (72, 9)])
def test_groupnorm_backward_ragged(dtype, C, G):
    o = ops()
    B = 2
    sizes = [(9, 13), (5, 7), (2, 3)]
    xs = [rnd(cases.randn(21 + i, B, C, h, w) * 2 + 0.5, dtype) for i, (h, w) in enumerate(sizes)]
    dys = [rnd(cases.randn(31 + i, B, C, h, w), dtype) for i, (h, w) in enumerate(sizes)]
    gamma, beta = cases.randn(22, C), cases.randn(23, C)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    xr = [x.clone().requires_grad_(True) for x in xs]
    ys = [F.relu(F.group_norm(x, G, gr, br, 1e-5)) for x in xr]
    sum((y * d).sum() for y, d in zip(ys, dys)).backward()
    X = o.Ragged.from_levels([nhwc(x, dtype) for x in xs])
    Y, st = o.groupnorm(X, gamma.to(DEV), beta.to(DEV), G, relu=True, out=X.new(C), return_stats=True)
    DY = o.Ragged.from_levels([nhwc(d, dtype) for d in dys])
    dx, dgamma, dbeta = o.groupnorm_backward(DY, Y, X, st, gamma.to(DEV), G, relu=True)
    for l, x in enumerate(xr):
        close(nchw(dx.level(l)).numpy(), x.grad.numpy(), dtype, scale=2.0)
    close(dgamma.cpu().numpy(), gr.grad.numpy(), dtype, scale=2.0)
    close(dbeta.cpu().numpy(), br.grad.numpy(), dtype, scale=2.0)
    # the training path: mask recomputed from x (y is neither kept nor read): the same mask — the forward's own arithmetic —
    # so the input gradient has the same bits wherever y is not a rounding knife edge (|y| tiny), and the same quality
    dx2, dgamma2, dbeta2 = o.groupnorm_backward(DY, None, X, st, gamma.to(DEV), G, relu=True, beta=beta.to(DEV))
    for l, x in enumerate(xr):
        close(nchw(dx2.level(l)).numpy(), x.grad.numpy(), dtype, scale=2.0)
        same = (dx2.level(l) == dx.level(l)).float().mean()
        assert float(same) > 0.999, float(same)
    close(dgamma2.cpu().numpy(), gr.grad.numpy(), dtype, scale=2.0)
    close(dbeta2.cpu().numpy(), br.grad.numpy(), dtype, scale=2.0)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_pool_and_upsample_backward(dtype):
    o = ops()
    # max-pool with many exact ties (post-ReLU zeros), odd sizes
    a = F.relu(rnd(cases.randn(11, 2, 16, 13, 17), dtype)).requires_grad_(True)
    y = F.max_pool2d(a, 3, 2, 1)
    dy = rnd(cases.randn(12, *y.shape), dtype)
    y.backward(dy)
    dx = o.maxpool3x3s2_backward(nhwc(a.detach(), dtype), nhwc(dy, dtype))
    close(nchw(dx).numpy(), a.grad.numpy(), dtype)
    # the route the training path takes since round 4: the forward records the winning tap (one byte per output), the backward
    # gathers from (dy, idx). Same forward values, same gradient BITS as the route that re-derives the maxima from x, ties included.
    for (hh, ww) in ((13, 17), (16, 26), (7, 1)):
        a2 = F.relu(rnd(cases.randn(18, 2, 16, hh, ww), dtype))
        y2, idx = o.maxpool3x3s2(nhwc(a2, dtype), return_argmax=True)
        assert torch.equal(y2, o.maxpool3x3s2(nhwc(a2, dtype))) and idx.dtype == torch.uint8 and int(idx.max()) <= 8
        dy2 = nhwc(rnd(cases.randn(19, *nchw(y2).shape), dtype), dtype)
        assert torch.equal(o.maxpool3x3s2_backward_argmax(dy2, idx, hh, ww), o.maxpool3x3s2_backward(nhwc(a2, dtype), dy2))

    b = rnd(cases.randn(13, 2, 16, 7, 9), dtype).requires_grad_(True)
    up = F.interpolate(b, size=(13, 17), mode='bilinear', align_corners=True)
    dup = rnd(cases.randn(14, *up.shape), dtype)
    up.backward(dup)
    db = o.upsample_bilinear_ac_backward(nhwc(dup, dtype), 7, 9)
    close(nchw(db).numpy(), b.grad.numpy(), dtype)
    # exact 2x (the MSPN case) and a 1-row source
    # (... and the shapes around the five-tap kernel's limit of 2.5 x: 2 x with odd sizes, 2.4 x, 2.6 x (generic kernel), identity,
    # reductions, one column)
    for (h, w, Ho, Wo) in [(16, 26, 32, 52), (1, 4, 3, 9), (5, 7, 9, 13), (64, 104, 128, 208), (6, 6, 14, 14), (6, 6, 15, 16),
                           (7, 9, 7, 9), (13, 17, 7, 9), (9, 1, 18, 1), (3, 2, 6, 5)]:
        b = rnd(cases.randn(15, 1, 8, h, w), dtype).requires_grad_(True)
        up = F.interpolate(b, size=(Ho, Wo), mode='bilinear', align_corners=True)
        dup = rnd(cases.randn(16, *up.shape), dtype)
        up.backward(dup)
        close(nchw(o.upsample_bilinear_ac_backward(nhwc(dup, dtype), h, w)).numpy(), b.grad.numpy(), dtype)

    c = rnd(cases.randn(17, 2, 16, 7, 9), dtype).requires_grad_(True)
    nn_up = F.interpolate(c, size=(13, 17), mode='nearest')
    dnn = rnd(cases.randn(18, *nn_up.shape), dtype)
    nn_up.backward(dnn)
    dc = o.upsample_nearest_backward(nhwc(dnn, dtype), 7, 9)
    close(nchw(dc).numpy(), c.grad.numpy(), dtype)
