"""Kernel-level parity: every C-ABI entry point of libdas_hip.so against the CPU oracle /
plain torch fp32 on the same seeded inputs. GPU only (`-m gpu`)."""
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


def tol(dtype):
    return dict(rtol=2e-5, atol=2e-5) if dtype == torch.float32 else dict(rtol=1.6e-2, atol=1.6e-2)


CONV_CASES = [
    # B, H, W, Cin, Cout, k, stride, pad
    (2, 9, 11, 16, 24, 3, 1, 1),
    (1, 16, 20, 32, 136, 3, 1, 1),
    (2, 17, 13, 64, 64, 1, 1, 0),
    (2, 17, 13, 64, 256, 1, 2, 0),
    (1, 20, 24, 8, 64, 7, 2, 3),      # stem-like, Cin not a multiple of BK
    (3, 8, 13, 256, 256, 3, 1, 1),    # head shape at the coarsest level
    (2, 12, 12, 48, 8, 3, 2, 1),      # tiny Cout (BN=32 path), stride-2 3x3
    (1, 33, 47, 128, 192, 3, 2, 1),
    (1, 5, 7, 24, 40, 1, 1, 0),
]


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_plain(case, dtype):
    B, H, W, Cin, Cout, k, s, p = case
    x = cases.randn(1, B, Cin, H, W)
    w = cases.randn(2, Cout, Cin, k, k) / (Cin * k * k) ** 0.5
    ref = F.conv2d(rnd(x, dtype), rnd(w, dtype), None, s, p)
    o = ops()
    y = o.conv2d(nhwc(x, dtype), o.pack_weight(w.to(DEV), dtype), k, k, s, p)
    assert y.dtype == dtype
    np.testing.assert_allclose(nchw(y).numpy(), ref.numpy(), **tol(dtype))


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_conv_epilogue_scale_shift_residual_relu_stats(dtype):
    B, H, W, Cin, Cout = 2, 14, 18, 64, 200
    x, w = cases.randn(3, B, Cin, H, W), cases.randn(4, Cout, Cin, 3, 3) / 24
    scale, shift = cases.randn(5, Cout).abs() + 0.5, cases.randn(6, Cout)
    res = cases.randn(7, B, Cout, H, W)
    o = ops()
    conv = F.conv2d(rnd(x, dtype), rnd(w, dtype), None, 1, 1) * scale[None, :, None, None] + shift[None, :, None, None]
    conv_q = rnd(conv, dtype)
    ref = F.relu(conv_q + rnd(res, dtype))
    stats = torch.zeros(2 * Cout, device=DEV)
    y = o.conv2d(nhwc(x, dtype), o.pack_weight(w.to(DEV), dtype), 3, 3, 1, 1, scale=scale.to(DEV), shift=shift.to(DEV),
                 residual=nhwc(res, dtype), relu=True, stats=stats)
    np.testing.assert_allclose(nchw(y).numpy(), ref.numpy(), **tol(dtype))
    n = B * H * W
    s_ref = torch.cat([conv_q.sum((0, 2, 3)), (conv_q ** 2).sum((0, 2, 3))])
    np.testing.assert_allclose(stats.cpu().numpy() / n, s_ref.numpy() / n, rtol=3e-3 if dtype != torch.float32 else 1e-4,
                               atol=3e-3 if dtype != torch.float32 else 1e-4)


@pytest.mark.parametrize('Cin,Cout', [(64, 64), (64, 256), (128, 128), (256, 64), (256, 512), (128, 256)])
def test_conv_stream1x1(Cin, Cout):
    """The persistent 1x1 kernel (bf16, K <= 256, >= 16384 rows): scale / shift / residual / ReLU / statistics,
    a channel-slice input, a last tile that overhangs M."""
    dtype = torch.bfloat16
    B, H, W = 2, 91, 93    # 16926 rows = 132 tiles of 128 + 30 rows
    x, w = cases.randn(11, B, Cin + 32, H, W), cases.randn(12, Cout, Cin, 1, 1) / Cin ** 0.5
    scale, shift = cases.randn(13, Cout).abs() + 0.5, cases.randn(14, Cout)
    res = cases.randn(15, B, Cout, H, W)
    o = ops()
    xs = nhwc(x, dtype)[..., 16:16 + Cin]
    wq = o.pack_weight(w.to(DEV), dtype)
    conv = F.conv2d(rnd(x[:, 16:16 + Cin], dtype), rnd(w, dtype))
    y = o.conv2d(xs, wq, 1, 1)
    np.testing.assert_allclose(nchw(y).numpy(), conv.numpy(), **tol(dtype))
    # mode 0: BatchNorm statistics of the stored values, in slots
    stats = torch.zeros(4, 2 * Cout, device=DEV)
    y = o.conv2d(xs, wq, 1, 1, stats=stats.view(-1))
    conv_q = rnd(conv, dtype)
    np.testing.assert_allclose(nchw(y).numpy(), conv_q.numpy(), **tol(dtype))
    n = B * H * W
    s_ref = torch.cat([conv_q.sum((0, 2, 3)), (conv_q ** 2).sum((0, 2, 3))])
    np.testing.assert_allclose(stats.sum(0).cpu().numpy() / n, s_ref.numpy() / n, rtol=3e-3, atol=3e-3)
    # mode 1: scale / shift + ReLU (eval-mode BatchNorm folded into the conv)
    aff_q = rnd(conv * scale[None, :, None, None] + shift[None, :, None, None], dtype)
    y = o.conv2d(xs, wq, 1, 1, scale=scale.to(DEV), shift=shift.to(DEV), relu=True)
    np.testing.assert_allclose(nchw(y).numpy(), F.relu(aff_q).numpy(), **tol(dtype))
    # mode 2: residual add (+ ReLU), the data-gradient epilogue
    y = o.conv2d(xs, wq, 1, 1, residual=nhwc(res, dtype))
    np.testing.assert_allclose(nchw(y).numpy(), rnd(conv_q + rnd(res, dtype), dtype).numpy(), **tol(dtype))
    y = o.conv2d(xs, wq, 1, 1, residual=nhwc(res, dtype), relu=True)
    np.testing.assert_allclose(nchw(y).numpy(), F.relu(conv_q + rnd(res, dtype)).numpy(), **tol(dtype))
    # every epilogue feature at once is not the persistent kernel's: the tile kernels take it
    y = o.conv2d(xs, wq, 1, 1, scale=scale.to(DEV), shift=shift.to(DEV), residual=nhwc(res, dtype), relu=True)
    np.testing.assert_allclose(nchw(y).numpy(), F.relu(aff_q + rnd(res, dtype)).numpy(), **tol(dtype))


def test_conv_relu_in_slices_and_f32_out():
    """bf16 in, f32 out (head predictors), reading a channel slice and writing into a slice."""
    o = ops()
    B, H, W = 2, 6, 9
    big = cases.randn(8, B, 96, H, W)
    w = cases.randn(9, 16, 32, 1, 1) / 6
    xb = nhwc(big, torch.bfloat16)
    out = torch.zeros(B, H, W, 40, device=DEV)
    o.conv2d(xb[..., 32:64], o.pack_weight(w.to(DEV), torch.bfloat16), 1, 1, relu_in=True, out_dtype=torch.float32,
             out=out[..., 8:24])
    ref = F.conv2d(F.relu(rnd(big[:, 32:64], torch.bfloat16)), rnd(w, torch.bfloat16))
    np.testing.assert_allclose(nchw(out[..., 8:24]).numpy(), ref.numpy(), rtol=1e-4, atol=1e-4)
    assert float(out[..., :8].abs().max()) == 0 and float(out[..., 24:].abs().max()) == 0


def test_conv_rejects_bad_args():
    from das_amd import _lib
    o = ops()
    x = torch.zeros(1, 4, 4, 12, device=DEV)  # Cin % 8 != 0
    w = torch.zeros(8, 1, 1, 12, device=DEV)
    with pytest.raises(_lib.DasHipError):
        o.conv2d(x, w, 1, 1)
    with pytest.raises(_lib.DasHipError):
        o.conv2d(torch.zeros(1, 4, 4, 8), torch.zeros(8, 1, 1, 8), 1, 1)  # CPU tensors: no fallback


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_pack_pool_upsample_add(dtype):
    o = ops()
    img = cases.randn(10, 2, 3, 19, 23)
    x = o.pack_image(img.to(DEV), dtype, 8)
    assert x.shape == (2, 19, 23, 8)
    np.testing.assert_array_equal(nchw(x)[:, :3].numpy(), rnd(img, dtype).numpy())
    assert float(x[..., 3:].float().abs().max()) == 0
    np.testing.assert_array_equal(o.to_nchw_f32(x, 1, 2).cpu().numpy(), rnd(img, dtype)[:, 1:3].numpy())

    a = cases.randn(11, 2, 16, 13, 17)
    np.testing.assert_array_equal(nchw(o.maxpool3x3s2(nhwc(a, dtype))).numpy(),
                                  F.max_pool2d(rnd(a, dtype), 3, 2, 1).numpy())
    up = o.upsample_bilinear_ac(nhwc(a, dtype), 26, 33)
    ref = F.interpolate(rnd(a, dtype), size=(26, 33), mode='bilinear', align_corners=True)
    np.testing.assert_allclose(nchw(up).numpy(), ref.numpy(), **(dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32
                                                                  else dict(rtol=8e-3, atol=8e-3)))
    b = cases.randn(12, 2, 16, 7, 9)
    y = o.add_upsample_nearest(nhwc(a, dtype), nhwc(b, dtype))
    ref = rnd(rnd(a, dtype) + F.interpolate(rnd(b, dtype), size=(13, 17), mode='nearest'), dtype)
    np.testing.assert_allclose(nchw(y).numpy(), ref.numpy(), rtol=0, atol=0)
    c = cases.randn(13, 2, 16, 13, 17)
    y = o.add3(nhwc(a, dtype), nhwc(c, dtype), nhwc(a * 0.5, dtype), relu=True)
    ref = F.relu(rnd(rnd(rnd(a, dtype) + rnd(c, dtype), dtype) + rnd(a * 0.5, dtype), dtype))
    np.testing.assert_allclose(nchw(y).numpy(), ref.numpy(), rtol=0, atol=0)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_bn_train_apply(dtype):
    o = ops()
    B, H, W, Cin, Cout = 3, 10, 12, 32, 64
    x, w = cases.randn(14, B, Cin, H, W), cases.randn(15, Cout, Cin, 1, 1) / 5
    gamma, beta = cases.randn(16, Cout).abs() + 0.5, cases.randn(17, Cout)
    rm, rv = cases.randn(18, Cout) * 0.1, cases.randn(19, Cout).abs() + 0.5
    res = cases.randn(20, B, Cout, H, W)
    conv = rnd(F.conv2d(rnd(x, dtype), rnd(w, dtype)), dtype)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    ref = F.relu(rnd(F.batch_norm(conv, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5), dtype) + rnd(res, dtype))
    stats = torch.zeros(2 * Cout, device=DEV)
    raw = o.conv2d(nhwc(x, dtype), o.pack_weight(w.to(DEV), dtype), 1, 1, stats=stats)
    rm_d, rv_d = rm.to(DEV), rv.to(DEV)
    y, mean, invstd = o.bn_train_apply(raw, stats, gamma.to(DEV), beta.to(DEV), rm_d, rv_d, residual=nhwc(res, dtype),
                                       relu=True)
    t = dict(rtol=1e-4, atol=1e-4) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    np.testing.assert_allclose(nchw(y).numpy(), ref.numpy(), **t)
    np.testing.assert_allclose(rm_d.cpu().numpy(), rm_ref.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(rv_d.cpu().numpy(), rv_ref.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(mean.cpu().numpy(), conv.mean((0, 2, 3)).numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_bn_statistics_slots_equal_single_array(dtype):
    """Workgroup slots (DasConvDesc.stats_slots) only regroup the same partial sums: conv + BatchNorm through 5 slots
    equals the single-array path; the SyncBN helper folds the slots before the all-reduce."""
    from das_amd import nn as dnn
    o = ops()
    B, H, W, Cin, Cout = 2, 37, 41, 64, 72
    x, w = cases.randn(30, B, Cin, H, W), cases.randn(31, Cout, Cin, 3, 3) / 24
    gamma, beta = (cases.randn(32, Cout).abs() + 0.5).to(DEV), cases.randn(33, Cout).to(DEV)
    xd, wd = nhwc(x, dtype), o.pack_weight(w.to(DEV), dtype)
    s1, s5 = torch.zeros(2 * Cout, device=DEV), torch.zeros(5 * 2 * Cout, device=DEV)
    raw1 = o.conv2d(xd, wd, 3, 3, 1, 1, stats=s1)
    raw5 = o.conv2d(xd, wd, 3, 3, 1, 1, stats=s5)
    assert torch.equal(raw1, raw5)
    np.testing.assert_allclose(s5.view(5, -1).sum(0).cpu().numpy(), s1.cpu().numpy(), rtol=2e-5, atol=1e-3)
    assert int((s5.view(5, -1).abs().sum(1) > 0).sum()) == 5          # every slot took part
    folded = dnn.sync_stats(s5.clone(), Cout, lambda t: None)          # (world of one: the all-reduce is a no-op)
    np.testing.assert_allclose(folded.cpu().numpy(), s1.cpu().numpy(), rtol=2e-5, atol=1e-3)
    y1, m1, i1 = o.bn_train_apply(raw1, s1, gamma, beta, None, None, relu=True)
    y5, m5, i5 = o.bn_train_apply(raw5, s5, gamma, beta, None, None, relu=True)
    np.testing.assert_allclose(m5.cpu().numpy(), m1.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(i5.cpu().numpy(), i1.cpu().numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(y5.float().cpu().numpy(), y1.float().cpu().numpy(), **tol(dtype))


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('C,rows,slots,res,relu', [(256, 3001, 16, True, True), (64, 1500, 16, False, True),
                                                   (512, 777, 5, True, False), (2048, 130, 1, True, True)])
def test_bn_apply_stream_path_equals_the_looping_kernel(dtype, C, rows, slots, res, relu):
    """The large layers' apply pass (slot fold as its own launch + bn_apply_stream_kernel, from `bn.stream_minbytes` on)
    must give the bits of bn_apply_kernel: output, saved mean / invstd, running statistics, batch counter — with and
    without residual / ReLU, odd row counts (a ragged last workgroup), 1 / 5 / 16 slots, C up to 2048 (f32: 512)."""
    o = ops()
    if dtype == torch.float32 and C > 1024:
        C = 1024
    g = torch.Generator(device='cpu').manual_seed(C + rows)
    raw = (torch.randn(1, rows, 1, C, generator=g) * 2 + 0.3).to(dtype).to(DEV)
    r = torch.randn(1, rows, 1, C, generator=g).to(dtype).to(DEV) if res else None
    f = raw.float().reshape(rows, C)
    tot = torch.stack([f.sum(0), (f * f).sum(0)]).reshape(-1)
    parts = torch.rand(slots, 2 * C, generator=g).to(DEV)
    stats = (parts / parts.sum(0, keepdim=True) * tot).contiguous().reshape(-1)     # slots that add up to the totals
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(DEV), torch.randn(C, generator=g).to(DEV)
    out = {}
    for name, minbytes in (('loop', 0), ('stream', 1)):
        rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
        nbt = torch.zeros((), dtype=torch.int64, device=DEV)
        with o.tuning(**{'bn.stream_minbytes': minbytes}):
            y, m, i = o.bn_train_apply(raw, stats.clone(), gamma, beta, rm, rv, 0.1, 1e-5, residual=r, relu=relu,
                                       num_batches_tracked=nbt)
            assert o.last_kernel() == ('bn_apply_stream_kernel' if minbytes else 'bn_apply_kernel')
        out[name] = (y, m.clone(), i.clone(), rm, rv, nbt)
    for a, b in zip(out['loop'], out['stream']):
        assert torch.equal(a, b)
    assert int(out['stream'][5]) == 1


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('C,rows,slots', [(256, 3001, 16), (64, 1500, 16), (512, 777, 5), (1024, 130, 1)])
def test_bn_backward_apply_stream_path_equals_the_looping_kernel(dtype, C, rows, slots):
    """das_bn_backward_apply on large tensors (slot fold as its own launch + bn_bwd_apply_dz_stream_kernel) against the
    looping kernel: d raw and the parameter-gradient accumulators, bit for bit."""
    o = ops()
    g = torch.Generator(device='cpu').manual_seed(C * 7 + rows)
    raw = (torch.randn(1, rows, 1, C, generator=g) * 2 + 0.3).to(dtype).to(DEV)
    dz = torch.randn(1, rows, 1, C, generator=g).to(dtype).to(DEV)
    mean, invstd = torch.randn(C, generator=g).to(DEV), (torch.rand(C, generator=g) + 0.5).to(DEV)
    gamma = (torch.rand(C, generator=g) + 0.5).to(DEV)
    sums = torch.randn(slots * 2 * C, generator=g).to(DEV) * rows / slots
    out = {}
    for name, minbytes in (('loop', 0), ('stream', 1)):
        dg, db = torch.full((C,), 0.25, device=DEV), torch.full((C,), -0.5, device=DEV)
        with o.tuning(**{'bn.stream_minbytes': minbytes}):
            draw = o.bn_backward_apply(dz, raw, mean, invstd, gamma, sums.clone(), dg, db)
            assert o.last_kernel() == ('bn_bwd_apply_dz_stream_kernel' if minbytes else 'bn_bwd_apply_dz_kernel')
        out[name] = (draw, dg, db)
    for a, b in zip(out['loop'], out['stream']):
        assert torch.equal(a, b)
    assert float((out['stream'][1] - 0.25).abs().max()) > 0


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('C,G', [(256, 32), (64, 32), (32, 32), (768, 96)])
def test_groupnorm(dtype, C, G):
    o = ops()
    x = cases.randn(21, 2, C, 9, 13) * 2 + 0.5
    gamma, beta = cases.randn(22, C), cases.randn(23, C)
    ref = F.relu(F.group_norm(rnd(x, dtype), G, gamma, beta, 1e-5))
    xd = nhwc(x, dtype)
    y = o.groupnorm(xd, gamma.to(DEV), beta.to(DEV), G, relu=True, out=torch.empty_like(xd))
    t = dict(rtol=2e-4, atol=2e-4) if dtype == torch.float32 else dict(rtol=1.6e-2, atol=1.6e-2)
    np.testing.assert_allclose(nchw(y).numpy(), ref.numpy(), **t)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_dcnv2_im2col_plus_gemm(dtype):
    from oracle.nn_ops import modulated_deform_conv2d
    o = ops()
    B, C, O, H, W = 2, 32, 40, 11, 14
    x, w, bias = cases.randn(24, B, C, H, W), cases.randn(25, O, C, 3, 3) / 17, cases.randn(26, O)
    om = cases.randn(27, B, 27, H, W)
    om[:, :18] *= 2.0
    ref = modulated_deform_conv2d(rnd(x, dtype), om[:, :18], torch.sigmoid(om[:, 18:]), rnd(w, dtype), bias)
    omd = torch.zeros(B, H, W, 32, device=DEV)
    omd[..., :27] = om.permute(0, 2, 3, 1).to(DEV)
    col = o.deform_im2col3x3(nhwc(x, dtype), omd)
    wp = o.pack_weight(w.to(DEV), dtype).reshape(O, 1, 1, 9 * C)
    y = o.conv2d(col, wp, 1, 1, shift=bias.to(DEV))
    np.testing.assert_allclose(nchw(y).numpy(), ref.numpy(), **(dict(rtol=1e-4, atol=1e-4) if dtype == torch.float32
                                                                  else dict(rtol=3e-2, atol=3e-2)))


@pytest.mark.parametrize('case', [(2, 64, 40, 11, 14, False), (1, 128, 256, 37, 23, False), (2, 256, 256, 0, 0, True),
                                  (1, 64, 8, 9, 130, False)])
def test_dcnv2_fused_forward(case):
    """das_dcn3x3_fused (sampling straight into the GEMM's LDS stage) against the oracle's modulated_deform_conv2d and
    against the two-kernel path on the same operands: `col` side output BIT-identical to das_deform_im2col3x3's, y equal to the
    GEMM over that col up to f32 accumulation order (bf16-exact bar); plain NHWC (tile overhang, Cout < 256) and the head's
    four ragged levels (offsets that leave the plane, level boundaries inside a 128-pixel tile)."""
    from oracle.nn_ops import modulated_deform_conv2d
    o = ops()
    B, C, O, H, W, ragged = case
    bf = torch.bfloat16
    w, bias = cases.randn(225, O, C, 3, 3) / (3 * C ** 0.5), cases.randn(226, O)
    sizes = [(16, 26), (8, 13), (4, 7), (2, 4)] if ragged else [(H, W)]
    xs = [cases.randn(224 + i, B, C, h, ww) for i, (h, ww) in enumerate(sizes)]
    oms = [cases.randn(227 + i, B, 27, h, ww) for i, (h, ww) in enumerate(sizes)]
    for t in oms:
        t[:, :18] *= 2.5
    omd = []
    for t in oms:
        z = torch.zeros(t.shape[0], t.shape[2], t.shape[3], 32, device=DEV)
        z[..., :27] = t.permute(0, 2, 3, 1).to(DEV)
        omd.append(z)
    if ragged:
        xr = o.Ragged.from_levels([nhwc(t, bf) for t in xs])
        omr = o.Ragged.from_levels(omd)
    else:
        xr, omr = nhwc(xs[0], bf), omd[0]
    wp = o.pack_weight(w.to(DEV), bf).reshape(O, 1, 1, 9 * C)
    shift = bias.to(DEV)
    y, col = o.dcn3x3_fused(xr, omr, wp, shift, want_col=True)
    y_only = o.dcn3x3_fused(xr, omr, wp, shift)
    col2 = o.deform_im2col3x3(xr, omr)
    y2 = o.conv2d(col2, wp, 1, 1, shift=shift)
    dd = (lambda t: t.data) if ragged else (lambda t: t)
    assert torch.equal(dd(col), dd(col2))
    assert torch.equal(dd(y), dd(y_only))
    ya, yb = dd(y).float().cpu().numpy(), dd(y2).float().cpu().numpy()
    scale = float(np.abs(yb).max())
    assert float(np.abs(ya - yb).max()) <= 2 ** -7 * scale, float(np.abs(ya - yb).max()) / scale   # one bf16 step of the largest value
    assert float((ya != yb).mean()) < 0.08     # (f32 accumulation order differs between the two GEMMs: rounding midpoints only)
    for i, (xl, ol) in enumerate(zip(xs, oms)):
        ref = modulated_deform_conv2d(rnd(xl, bf), ol[:, :18], torch.sigmoid(ol[:, 18:]), rnd(w, bf), bias)
        got = nchw(y.level(i) if ragged else y)
        np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=3e-2, atol=3e-2)


def test_dcnv2_fused_through_the_module_train_and_eval():
    """nn.dcn_v2 with the one-kernel forward switched on (autograd.DCN_FUSED for the training graph, das_tuning key
    dcn.fused_minrows for eval) against the two-kernel path on the same module and input, bf16: the output within one bf16 step,
    the gradients of the input, the DCN weight / bias and the offset conv equal up to the GEMMs' accumulation order."""
    from das_amd import autograd as ag, nn as dnn
    o = ops()
    torch.manual_seed(3)
    mod = dnn.ConvModule(64, 64, 3, padding=1, conv_cfg=dict(type='DCNv2'), norm_cfg=None, act_cfg=None).to(DEV)
    with torch.no_grad():
        mod.conv.conv_offset.weight.normal_(0, 0.05)
        mod.conv.conv_offset.bias.normal_(0, 0.5)
    x0 = (cases.randn(301, 2, 64, 19, 23)).permute(0, 2, 3, 1).contiguous().to(DEV).to(torch.bfloat16)
    gy = cases.randn(302, 2, 19, 23, 64).to(DEV).to(torch.bfloat16)
    res = {}
    for fused in (False, True):
        ag.DCN_FUSED = fused
        try:
            for p_ in mod.parameters():
                p_.grad = None
            x = x0.clone().requires_grad_(True)
            y = mod(x)
            assert o.last_kernel() == ('dcn3x3_fused_kernel' if fused else o.last_kernel())
            (y.float() * gy.float()).sum().backward()
            with torch.no_grad(), o.tuning(**{'dcn.fused_minrows': 1 if fused else 0}):
                ye = mod(x0)
            res[fused] = (y.detach().float(), ye.float(), x.grad.float(), {n: p_.grad.float().clone() for n, p_ in mod.named_parameters()})
        finally:
            ag.DCN_FUSED = True
    (ya, yea, dxa, ga), (yb, yeb, dxb, gb) = res[False], res[True]
    step = 2 ** -7 * float(ya.abs().max())
    assert float((ya - yb).abs().max()) <= step and float((yea - yeb).abs().max()) <= step
    assert torch.equal(yb, yeb)                               # train-graph forward == eval forward of the fused kernel
    assert float((dxa - dxb).abs().max()) <= 4e-2 * float(dxa.abs().max())
    for n in ga:
        assert float((ga[n] - gb[n]).abs().max()) <= 4e-2 * float(ga[n].abs().max()) + 1e-6, n


def test_offset_sample_matches_oracle(golden_dir):
    import os
    from oracle.head import offset_sample
    o = ops()
    B, Jn, heads, h, w = 2, 3, 4, 10, 14
    uvd, so, conf = cases.randn(31, B, Jn * 3, h, w) * 2, cases.randn(32, B, Jn * heads * 2, h, w) * 1.5, \
        cases.randn(33, B, Jn * 3, h, w)
    y = o.offset_sample(nhwc(uvd), nhwc(so), nhwc(conf), Jn)
    z = np.load(os.path.join(golden_dir, 'offset_sample.npz'))
    np.testing.assert_allclose(nchw(y).numpy(), z['out'], rtol=1e-4, atol=1e-5)  # reference fixture
    np.testing.assert_allclose(nchw(y).numpy(), offset_sample(uvd, so, conf, Jn, heads).numpy(), rtol=1e-4, atol=1e-5)
    # J = 15 at a full-size level, offsets large enough to leave the map
    B, Jn, h, w = 1, 15, 32, 52
    uvd, so, conf = cases.randn(34, B, Jn * 3, h, w) * 20, cases.randn(35, B, Jn * 8, h, w) * 6, cases.randn(36, B, Jn * 3, h, w)
    y = o.offset_sample(nhwc(uvd), nhwc(so), nhwc(conf), Jn)
    # values reach |80| with pixel-to-pixel slopes of ~30: f32 coordinate rounding alone moves samples by ~1e-4
    np.testing.assert_allclose(nchw(y).numpy(), offset_sample(uvd, so, conf, Jn, 4).numpy(), rtol=1e-4, atol=2e-3)


def test_blend_assemble_finalize():
    o = ops()
    J, root = 5, 2
    B, H, W = 2, 4, 6
    off, wl, nxt = cases.randn(37, B, 3 * J, H, W), cases.randn(38, B, 3 * J, H, W), cases.randn(39, B, 3 * J, H, W)
    y = o.sigmoid_blend(nhwc(off), nhwc(wl), nhwc(nxt))
    g = torch.sigmoid(wl)
    np.testing.assert_allclose(nchw(y).numpy(), ((1 - g) * off + g * nxt).numpy(), rtol=1e-5, atol=1e-6)

    raw = cases.randn(40, B, 16 + 3 * J + 1 + 3 * J + 5, H, W)
    uvd_c, sigma_c = 16, 16 + 3 * J + 1
    sc = (1.1, 0.9, 1.2, 0.8)
    rawd = nhwc(raw)
    desc = o.head_desc(J, root, rawd.shape[-1], 8, 12, uvd_c, sigma_c, [sc], [16.0], 50.0, 20.0)
    pose, uvd = o.head_assemble(rawd, desc)
    r_uvd = raw[:, uvd_c:uvd_c + 3 * J].clone().reshape(B, J, 3, H, W)
    r_uvd[:, :, :2] *= sc[2]
    r_uvd[:, :, 2] *= sc[3]
    r_uvd[:, root, 2] = 0
    r_sig = raw[:, sigma_c:sigma_c + 3 * J].clone().reshape(B, J, 3, H, W)
    r_sig[:, root, 2] = 1
    ref = torch.cat([raw[:, 8:10] * sc[0], raw[:, 12:13] * sc[1], r_uvd.reshape(B, -1, H, W), r_sig.reshape(B, -1, H, W)], 1)
    np.testing.assert_allclose(nchw(pose).numpy(), ref.numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(nchw(uvd).numpy(), r_uvd.reshape(B, -1, H, W).numpy(), rtol=1e-6, atol=1e-7)

    refd = cases.randn(41, B, 3 * J, H, W)
    rd = nhwc(refd)
    o.head_finalize(pose, rd, desc, eval_mode=True)
    e = refd.clone().reshape(B, J, 3, H, W)
    e[:, root, 2] = 0
    np.testing.assert_array_equal(nchw(rd).numpy(), e.reshape(B, -1, H, W).numpy())
    e[:, :, :2] *= 16.0
    e[:, :, 2] *= 50.0
    ref2 = ref.clone()
    ref2[:, 3:3 + 3 * J] = e.reshape(B, -1, H, W)
    ref2[:, 2] = ref[:, 2] / 20.0
    np.testing.assert_allclose(nchw(pose).numpy(), ref2.numpy(), rtol=1e-6, atol=1e-7)


def _decode_hip(cls, pose, ctr, sf, J, strides, cfg):
    o = ops()
    out = o.decode([nhwc(c) for c in cls], [nhwc(c) for c in ctr], [nhwc(p) for p in pose], strides,
                   torch.tensor(sf, dtype=torch.float32, device=DEV), J, cfg['nms_pre'], cfg['nms_post'],
                   cfg['score_thr'], cfg['nms_thr'], nms_soft=cfg.get('nms_type', 'hard') != 'hard')
    return {k: v.cpu() for k, v in out.items()}


def _check_decode_vs_oracle(cls, pose, ctr, sfs, J, strides, cfg, golden=None):
    from oracle import decode as od
    metas = [dict(scale_factor=np.array([s[0], s[1], s[0], s[1]], dtype=np.float32), filename='x') for s in sfs]
    ref = od.get_poses(cls, pose, ctr, metas, J, strides, cfg, return_index=True)
    out = _decode_hip(cls, pose, ctr, sfs, J, strides, cfg)
    for b, r in enumerate(ref):
        K = int(out['count'][b])
        assert K == r['poses'].shape[0]
        # bit-exact kept indices, in order
        np.testing.assert_array_equal(out['index'][b, :K].numpy(), r['index'].numpy())
        np.testing.assert_allclose(out['poses'][b, :K].numpy(), r['poses'].numpy(), rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(out['centers'][b, :K].numpy(), r['centers'].numpy(), rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(out['scores'][b, :K].numpy(), np.array(r['scores'], dtype=np.float32), rtol=1e-5)
        if golden is not None:
            np.testing.assert_allclose(out['poses'][b, :K].numpy(), golden[f'poses{b}'], rtol=1e-4, atol=1e-4)
            np.testing.assert_allclose(out['scores'][b, :K].numpy(), golden[f'scores{b}'], rtol=1e-5)
    return out


def test_decode_full_size_vs_oracle_and_reference_fixture(golden_dir):
    import os
    z = np.load(os.path.join(golden_dir, 'decode_full.npz'))
    cls, pose, ctr = cases.full_decode_inputs()
    out = _check_decode_vs_oracle(cls, pose, ctr, [(1.3, 1.3), (1.0, 1.0)], cases.FULL_J, cases.FULL_STRIDES,
                                  cases.FULL_TEST_CFG, golden=z)
    assert int(out['count'].min()) > 20


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_decode_soft_nms_vs_oracle_and_reference_fixture(golden_dir, tag):
    """nms_type='soft' (pose_nms.py:128-194): kept indices bit-exact vs the oracle, values vs the reference fixture."""
    import os
    z = np.load(os.path.join(golden_dir, 'decode_soft.npz'))
    cfg = cases.SOFT_DECODE_CFGS[tag]
    cls, pose, ctr = cases.full_decode_inputs(**cases.SOFT_DECODE_INPUTS[tag])
    out = _check_decode_vs_oracle(cls, pose, ctr, [(1.3, 1.3), (1.0, 1.0)], cases.FULL_J, cases.FULL_STRIDES, cfg,
                                  golden={k[2:]: z[k] for k in z.files if k.startswith(tag + '_')})
    assert out['count'].tolist() == [cfg['nms_post']] * 2


def test_decode_soft_nms_edge_cases():
    # fewer candidates than nms_post (everything is kept, in rescored order), the nms_pre truncation path, J = 21, none
    cfg = dict(cases.FULL_TEST_CFG, nms_type='soft', nms_thr=0.2)
    cls, pose, ctr = cases.full_decode_inputs(seed=5, B=1, bias=-5.0)
    out = _check_decode_vs_oracle(cls, pose, ctr, [(1.0, 1.0)], cases.FULL_J, cases.FULL_STRIDES, cfg)
    assert 0 < int(out['count'][0]) < cfg['nms_post']
    cls, pose, ctr = cases.full_decode_inputs(seed=77, B=1, bias=1.0)
    _check_decode_vs_oracle(cls, pose, ctr, [(1.0, 1.0)], cases.FULL_J, cases.FULL_STRIDES,
                            dict(cfg, nms_pre=300, nms_post=50))
    sizes = [(96, 128), (48, 64), (24, 32), (12, 16)]
    cls, pose, ctr = cases.full_decode_inputs(seed=6, B=1, Jn=21, sizes=sizes, bias=-4.5)
    _check_decode_vs_oracle(cls, pose, ctr, [(0.8, 0.8)], 21, cases.FULL_STRIDES, cfg)
    cls, pose, ctr = cases.full_decode_inputs(seed=78, B=2, bias=-20.0)
    out = _decode_hip(cls, pose, ctr, [(1.0, 1.0)] * 2, cases.FULL_J, cases.FULL_STRIDES, cfg)
    assert out['count'].tolist() == [0, 0]


def test_decode_topk_truncation_and_empty():
    # many candidates above threshold on level 0 -> the nms_pre truncation path is exercised
    cls, pose, ctr = cases.full_decode_inputs(seed=77, B=1, bias=1.0)
    cfg = dict(cases.FULL_TEST_CFG, nms_pre=300, nms_thr=0.3)
    _check_decode_vs_oracle(cls, pose, ctr, [(1.0, 1.0)], cases.FULL_J, cases.FULL_STRIDES, cfg)
    cls, pose, ctr = cases.full_decode_inputs(seed=78, B=2, bias=-20.0)
    out = _decode_hip(cls, pose, ctr, [(1.0, 1.0)] * 2, cases.FULL_J, cases.FULL_STRIDES, cases.FULL_TEST_CFG)
    assert out['count'].tolist() == [0, 0]


def test_decode_mupots_topology():
    """config 5 geometry: J=21, 768x1024 input -> levels 96x128 ... 12x16 (16320 locations)."""
    sizes = [(96, 128), (48, 64), (24, 32), (12, 16)]
    cls, pose, ctr = cases.full_decode_inputs(seed=5, B=1, Jn=21, sizes=sizes, bias=-4.5)
    _check_decode_vs_oracle(cls, pose, ctr, [(0.8, 0.8)], 21, cases.FULL_STRIDES, cases.FULL_TEST_CFG)


HD_SIZES = [(135, 240), (68, 120), (34, 60), (17, 30)]      # 1080 x 1920 input at strides 8 / 16 / 32 / 64: 43 110 locations


def _passing_per_level(cls, ctr, thr):
    return [int(((c.sigmoid() * k.sigmoid()) > thr).flatten(1).sum(1).max()) for c, k in zip(cls, ctr)]


@pytest.mark.parametrize('case', ['typical', 'level0_topk_in_lds', 'level0_radix_select', 'keep_all_6k', 'keep_all_over_16k'])
def test_decode_1080p_beyond_the_former_lds_caps(case):
    """The reference's `_get_poses_single` (das_head.py:690-796) has no size limit; until round 5 das_decode refused a level
    of more than 16 384 locations or more than 4096 candidates in total. 1080 x 1920 frames (level 0 =
    135 x 240 = 32 400 locations), J = 15, kept indices bit-exact against the oracle on every path of the selection:
      typical              a few hundred candidates (one sweep, everything in LDS)
      level0_topk_in_lds   level 0 has more candidates than nms_pre = 1000 but fewer than the 16 384 keys LDS holds
      level0_radix_select  more than 16 384 locations of level 0 pass the threshold: its nms_pre-th key by radix select
      keep_all_6k          nms_pre = -1 (no per-level cut): ~6000 candidates, suppression flags in the workspace
      keep_all_over_16k    nms_pre = -1 and more than 16 384 candidates: the global order from a sort over the workspace"""
    cfg = dict(cases.FULL_TEST_CFG)
    bias, B = -4.5, 2
    if case == 'level0_topk_in_lds':
        bias = -2.6
    elif case == 'level0_radix_select':
        bias, B = 1.5, 1
    elif case == 'keep_all_6k':
        bias, B, cfg = -3.0, 1, dict(cfg, nms_pre=-1, nms_post=40)
    elif case == 'keep_all_over_16k':
        bias, B, cfg = 0.0, 1, dict(cfg, nms_pre=-1, nms_post=20)
    cls, pose, ctr = cases.full_decode_inputs(seed=31, B=B, sizes=HD_SIZES, bias=bias)
    n = _passing_per_level(cls, ctr, cfg['score_thr'])
    print(case, 'candidates above the threshold per level (max over images):', n)
    if case == 'typical':
        assert 50 < sum(n) < 1000
    elif case == 'level0_topk_in_lds':
        assert 1000 < n[0] <= 16384
    elif case == 'level0_radix_select':
        assert n[0] > 16384
    elif case == 'keep_all_6k':
        assert 4096 < sum(n) <= 16384
    else:
        assert sum(n) > 16384
    # (oracle_early_stop: the oracle's greedy loop stops at nms_post kept poses — the same prefix as the reference's run to the
    # end + truncation, oracle/decode.py oks_nms; at these candidate counts the full loop would take minutes to hours)
    out = _check_decode_vs_oracle(cls, pose, ctr, [(1.5, 1.5)] * B, cases.FULL_J, cases.FULL_STRIDES,
                                  dict(cfg, oracle_early_stop=True))
    assert int(out['count'].min()) > 10


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_ragged_multilevel_conv_gn_dcn_match_per_level(dtype):
    """One launch over the rows of all levels == the same op applied level by level."""
    o = ops()
    B, C, O = 2, 64, 72
    sizes = [(12, 20), (6, 10), (3, 5), (2, 3)]
    lv = [nhwc(cases.randn(50 + i, B, C, h, w), dtype) for i, (h, w) in enumerate(sizes)]
    r = o.Ragged.from_levels(lv)
    w3 = o.pack_weight((cases.randn(60, O, C, 3, 3) / 24).to(DEV), dtype)
    bias = cases.randn(61, O).to(DEV)
    y = o.conv2d(r, w3, 3, 3, 1, 1, shift=bias, relu=True)
    for l, x in enumerate(lv):
        ref = o.conv2d(x, w3, 3, 3, 1, 1, shift=bias, relu=True)
        np.testing.assert_array_equal(y.level(l).float().cpu().numpy(), ref.float().cpu().numpy())
    # GroupNorm statistics are per (level, image)
    gamma, beta = cases.randn(62, O).to(DEV), cases.randn(63, O).to(DEV)
    g = o.groupnorm(y.like(y.data.clone()), gamma, beta, 9, relu=True)
    for l in range(len(sizes)):
        ref = o.groupnorm(y.level(l).clone(), gamma, beta, 9, relu=True)
        np.testing.assert_allclose(g.level(l).float().cpu().numpy(), ref.float().cpu().numpy(), rtol=1e-5, atol=1e-5)
    # deformable im2col + offset_sample geometry per row
    om = o.Ragged.from_levels([nhwc(cases.randn(70 + i, B, 32, h, w) * 1.5) for i, (h, w) in enumerate(sizes)])
    col = o.deform_im2col3x3(r, om)
    for l, x in enumerate(lv):
        ref = o.deform_im2col3x3(x, om.level(l).contiguous())
        np.testing.assert_array_equal(col.level(l).float().cpu().numpy(), ref.float().cpu().numpy())
    J = 3
    uvd = o.Ragged.from_levels([nhwc(cases.randn(80 + i, B, 3 * J, h, w) * 2) for i, (h, w) in enumerate(sizes)])
    so = o.Ragged.from_levels([nhwc(cases.randn(90 + i, B, 8 * J, h, w)) for i, (h, w) in enumerate(sizes)])
    cf = o.Ragged.from_levels([nhwc(cases.randn(95 + i, B, 3 * J, h, w)) for i, (h, w) in enumerate(sizes)])
    out = o.offset_sample(uvd, so, cf, J)
    for l in range(len(sizes)):
        ref = o.offset_sample(uvd.level(l).contiguous(), so.level(l).contiguous(), cf.level(l).contiguous(), J)
        np.testing.assert_array_equal(out.level(l).cpu().numpy(), ref.cpu().numpy())
