"""Backward of the head-specific kernels vs torch autograd through the CPU oracle. GPU only."""
import numpy as np
import pytest
import torch

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


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)


@pytest.mark.parametrize('shape', [
    (2, 16, 24, 9, 11, 1.5),     # tile overhang, offsets mostly inside the LDS window
    (1, 136, 8, 19, 21, 4.0),    # two channel slabs (second one partial), offsets often beyond the 2-px halo
    (2, 8, 8, 17, 8, 0.0),       # zero offsets: every sample on the integer grid
])
def test_deform_im2col_backward_vs_oracle_autograd(shape):
    from oracle.nn_ops import modulated_deform_conv2d
    o = ops()
    B, C, O, H, W, oscale = shape
    x = cases.randn(24, B, C, H, W).requires_grad_(True)
    w = (cases.randn(25, O, C, 3, 3) / 12)
    om = cases.randn(27, B, 27, H, W)
    om[:, :18] *= oscale
    om = om.requires_grad_(True)
    y = modulated_deform_conv2d(x, om[:, :18], torch.sigmoid(om[:, 18:]), w, None)
    dy = cases.randn(28, *y.shape)
    y.backward(dy)
    # dcol = dY x W : (rows, 9C), tap-major
    dcol = torch.einsum('boyx,ockl->byxklc', dy, w).reshape(B, H, W, 9 * C).contiguous().to(DEV)
    omd = torch.zeros(B, H, W, 32, device=DEV)
    omd[..., :27] = om.detach().permute(0, 2, 3, 1).to(DEV)
    dx, dom = o.deform_im2col3x3_backward(nhwc(x.detach()), omd, dcol)
    assert rel(nchw(dx).numpy(), x.grad.numpy()) < 1e-4
    assert rel(nchw(dom[..., :27]).numpy(), om.grad.numpy()) < 1e-4
    assert float(dom[..., 27:].abs().max()) == 0.0


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_deform_im2col_backward_ragged_equals_per_level(dtype):
    """All FPN levels in one launch (ragged rows) == one launch per level."""
    o = ops()
    B, C = 2, 64
    sizes = [(16, 26), (8, 13), (4, 7)]
    xs = [nhwc(cases.randn(50 + i, B, C, h, w), dtype) for i, (h, w) in enumerate(sizes)]
    oms = [torch.cat([nhwc(cases.randn(60 + i, B, 27, h, w) * 1.2), torch.zeros(B, h, w, 5, device=DEV)], -1)
           for i, (h, w) in enumerate(sizes)]
    dcs = [nhwc(cases.randn(70 + i, B, 9 * C, h, w), dtype) for i, (h, w) in enumerate(sizes)]
    R = o.Ragged.from_levels
    dx, dom = o.deform_im2col3x3_backward(R(xs), R(oms), R(dcs))
    for l in range(len(sizes)):
        dxl, doml = o.deform_im2col3x3_backward(xs[l], oms[l], dcs[l])
        assert rel(dx.level(l).float().cpu().numpy(), dxl.float().cpu().numpy()) < 1e-5
        assert rel(dom.level(l).cpu().numpy(), doml.cpu().numpy()) < 1e-5


@pytest.mark.parametrize('J,h,w,scale', [(3, 10, 14, 1.5), (15, 16, 26, 4.0)])
def test_offset_sample_backward_vs_oracle_autograd(J, h, w, scale):
    from oracle.head import offset_sample
    o = ops()
    B = 2
    uvd = (cases.randn(31, B, J * 3, h, w) * 2).requires_grad_(True)
    so = (cases.randn(32, B, J * 8, h, w) * scale).requires_grad_(True)
    conf = cases.randn(33, B, J * 3, h, w).requires_grad_(True)
    out = offset_sample(uvd, so, conf, J, 4)
    g = cases.randn(34, *out.shape)
    out.backward(g)
    d_uvd, d_so, d_conf = o.offset_sample_backward(nhwc(uvd.detach()), nhwc(so.detach()), nhwc(conf.detach()), nhwc(g), J)
    assert rel(nchw(d_uvd).numpy(), uvd.grad.numpy()) < 2e-4
    assert rel(nchw(d_so).numpy(), so.grad.numpy()) < 2e-4
    assert rel(nchw(d_conf).numpy(), conf.grad.numpy()) < 2e-4


def test_blend_and_assemble_backward():
    o = ops()
    J, root, B, H, W = 5, 2, 2, 4, 6
    off = cases.randn(37, B, 3 * J, H, W).requires_grad_(True)
    wl = cases.randn(38, B, 3 * J, H, W).requires_grad_(True)
    nxt = cases.randn(39, B, 3 * J, H, W).requires_grad_(True)
    g = torch.sigmoid(wl)
    out = (1 - g) * off + g * nxt
    go = cases.randn(40, *out.shape)
    out.backward(go)
    d_off, d_w, d_nxt = o.sigmoid_blend_backward(nhwc(off.detach()), nhwc(wl.detach()), nhwc(nxt.detach()), nhwc(go))
    for a, b in ((d_off, off), (d_w, wl), (d_nxt, nxt)):
        assert rel(nchw(a).numpy(), b.grad.numpy()) < 1e-5

    RAW = 16 + 3 * J + 1 + 3 * J + 5
    uvd_c, sigma_c = 16, 16 + 3 * J + 1
    raw = cases.randn(41, B, RAW, H, W).requires_grad_(True)
    sc = torch.tensor([1.1, 0.9, 1.2, 0.8], requires_grad=True)
    r_uvd = raw[:, uvd_c:uvd_c + 3 * J].reshape(B, J, 3, H, W)
    r_uvd = torch.cat([r_uvd[:, :, :2] * sc[2], r_uvd[:, :, 2:] * sc[3]], 2)
    zm = torch.ones(J, 3)
    zm[root, 2] = 0
    r_uvd = (r_uvd * zm[None, :, :, None, None]).reshape(B, 3 * J, H, W)
    r_sig = raw[:, sigma_c:sigma_c + 3 * J].reshape(B, J, 3, H, W) * zm[None, :, :, None, None] + (1 - zm)[None, :, :, None, None]
    pose = torch.cat([raw[:, 8:10] * sc[0], raw[:, 12:13] * sc[1], r_uvd, r_sig.reshape(B, -1, H, W)], 1)
    gp, gu = cases.randn(42, *pose.shape), cases.randn(43, *r_uvd.shape)
    ((pose * gp).sum() + (r_uvd * gu).sum()).backward()
    rawd = nhwc(raw.detach())
    desc = o.head_desc(J, root, RAW, 8, 12, uvd_c, sigma_c, [sc.detach().tolist()], [16.0], 50.0, 20.0)
    d_raw, d_scale = o.head_assemble_backward(rawd, nhwc(gp), nhwc(gu), desc)
    assert rel(nchw(d_raw).numpy(), raw.grad.numpy()) < 1e-5
    assert rel(d_scale[0].cpu().numpy(), sc.grad.numpy()) < 1e-4
