"""Backward kernels vs torch autograd (CPU, fp32) on the same seeded inputs. GPU only."""
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
    tol = (2e-5 if dtype == torch.float32 else 1.2e-2) * scale
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)
    assert err < tol, err


CASES = [
    # B, H, W, Cin, Cout, k, stride, pad
    (2, 9, 11, 16, 24, 3, 1, 1),
    (2, 17, 13, 64, 256, 1, 1, 0),
    (2, 16, 20, 64, 128, 1, 2, 0),
    (1, 18, 22, 128, 136, 3, 2, 1),
    (2, 13, 15, 64, 64, 3, 2, 1),     # odd sizes, stride 2
    (3, 8, 13, 256, 256, 3, 1, 1),    # K = 2304, Cout = 256: the ping-pong 256 x 256 weight-gradient kernel (bf16)
    (2, 9, 7, 128, 512, 3, 1, 1),     # same kernel: two Cout tiles, K = 1152 ends inside a 256-column tile
    (1, 20, 24, 8, 64, 7, 2, 3),      # stem (only wgrad is needed in training; dgrad checked anyway)
]


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('case', CASES)
def test_conv_dgrad_wgrad(case, dtype):
    B, H, W, Cin, Cout, k, s, p = case
    o = ops()
    x = cases.randn(1, B, Cin, H, W)
    w = cases.randn(2, Cout, Cin, k, k) / (Cin * k * k) ** 0.5
    xr, wr = rnd(x, dtype).requires_grad_(True), rnd(w, dtype).requires_grad_(True)
    y = F.conv2d(xr, wr, None, s, p)
    dy = rnd(cases.randn(3, *y.shape), dtype)
    y.backward(dy)
    dx = o.conv2d_dgrad(nhwc(dy, dtype), o.pack_weight_dgrad(w.to(DEV), dtype), k, k, s, p, (H, W))
    assert tuple(dx.shape) == (B, H, W, Cin)
    close(nchw(dx).numpy(), xr.grad.numpy(), dtype)
    dw = o.conv2d_wgrad(nhwc(x, dtype), nhwc(dy, dtype), k, k, s, p)
    assert tuple(dw.shape) == (Cout, k, k, Cin) and dw.dtype == torch.float32
    close(dw.permute(0, 3, 1, 2).cpu().numpy(), wr.grad.numpy(), dtype, scale=0.5)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('C,O', [(64, 72), (128, 256)])   # (128, 256): K = 1152 -> the ping-pong kernel in bf16
def test_conv_wgrad_ragged_and_bias(dtype, C, O):
    o = ops()
    B = 2
    sizes = [(12, 20), (6, 10), (3, 5)]
    xs = [cases.randn(50 + i, B, C, h, w) for i, (h, w) in enumerate(sizes)]
    dys = [cases.randn(60 + i, B, O, h, w) for i, (h, w) in enumerate(sizes)]
    w = (cases.randn(70, O, C, 3, 3) / 24)
    wr = rnd(w, dtype).requires_grad_(True)
    tot = 0
    for x, dy in zip(xs, dys):
        tot = tot + (F.conv2d(rnd(x, dtype), wr, None, 1, 1) * rnd(dy, dtype)).sum()
    tot.backward()
    xr = o.Ragged.from_levels([nhwc(x, dtype) for x in xs])
    dyr = o.Ragged.from_levels([nhwc(d, dtype) for d in dys])
    dw = o.conv2d_wgrad(xr, dyr, 3, 3, 1, 1)
    close(dw.permute(0, 3, 1, 2).cpu().numpy(), wr.grad.numpy(), dtype, scale=0.5)
    db = o.colsum(dyr)
    ref = sum(rnd(d, dtype).sum((0, 2, 3)) for d in dys)
    close(db.cpu().numpy(), ref.numpy(), dtype, scale=0.5)
    # ragged dgrad == per-level dgrad
    wd = o.pack_weight_dgrad(w.to(DEV), dtype)
    dx = o.conv2d_dgrad(dyr, wd, 3, 3, 1, 1, None)
    for l, d in enumerate(dys):
        ref = o.conv2d_dgrad(nhwc(d, dtype), wd, 3, 3, 1, 1, sizes[l])
        np.testing.assert_array_equal(dx.level(l).float().cpu().numpy(), ref.float().cpu().numpy())


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('relu,res', [(True, True), (False, False), (True, False)])
def test_bn_train_backward(dtype, relu, res):
    o = ops()
    B, H, W, C = 3, 10, 12, 64
    raw = rnd(cases.randn(14, B, C, H, W) * 1.5 + 0.3, dtype)
    gamma, beta = cases.randn(16, C).abs() + 0.5, cases.randn(17, C)
    r = rnd(cases.randn(20, B, C, H, W), dtype)
    dy = rnd(cases.randn(21, B, C, H, W), dtype)
    rawr, gr, br, rr = raw.clone().requires_grad_(True), gamma.clone().requires_grad_(True), \
        beta.clone().requires_grad_(True), r.clone().requires_grad_(True)
    z = F.batch_norm(rawr, None, None, gr, br, True, 0.1, 1e-5)
    if res:
        z = z + rr
    y = F.relu(z) if relu else z
    y.backward(dy)
    mean = raw.mean((0, 2, 3))
    invstd = 1.0 / torch.sqrt(raw.var((0, 2, 3), unbiased=False) + 1e-5)
    yd = nhwc(rnd(y.detach(), dtype), dtype)
    draw, dres, dgamma, dbeta = o.bn_train_backward(nhwc(dy, dtype), yd, nhwc(raw, dtype), mean.to(DEV), invstd.to(DEV),
                                                    gamma.to(DEV), relu, res)
    close(nchw(draw).numpy(), rawr.grad.numpy(), dtype)
    close(dgamma.cpu().numpy(), gr.grad.numpy(), dtype, scale=0.5)
    close(dbeta.cpu().numpy(), br.grad.numpy(), dtype, scale=0.5)
    if res:
        close(nchw(dres).numpy(), rr.grad.numpy(), dtype)
