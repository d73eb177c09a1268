"""Parity of the tile kernels the benchmark actually runs — conv_glds3_kernel (256 x 128, 3 stages),
conv_glds4_kernel (256 x 256, 4 stages; plain and ping-pong schedule), conv_glds_kernel with its K loop split
over workgroups, conv1x1_stream_kernel, conv_wgrad_kernel / conv_wgrad_pp_kernel at their production split — against
F.conv2d on bf16-rounded operands ("what's weak" #1).

Two kinds of cases:
  * forced: a small problem is pushed through a given kernel by lowering its dispatch threshold
    (ops.tuning -> das_tuning_set), which covers the 3x3 tap walk and its border predicates, stride 2, ragged
    multi-level rows, overhanging M / N tiles, K loops far longer than the pipeline depth, every epilogue;
  * real: production-size layers with the default thresholds.
Every case asserts WHICH kernel the launcher picked (ops.last_kernel -> das_last_kernel).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import cases

pytestmark = pytest.mark.gpu

DEV = 'cuda'
BF = torch.bfloat16
TOL = dict(rtol=1.6e-2, atol=1.6e-2)


def ops():
    from das_amd import ops as o
    return o


def nhwc(t, dtype=BF):
    return t.permute(0, 2, 3, 1).contiguous().to(dtype).to(DEV)


def nchw(t):
    return t.float().cpu().permute(0, 3, 1, 2).contiguous()


def q(t):
    return t.to(BF).float()


def assert_bf16_exact(y, ref, frac=0.05):
    """the bf16-appropriate SHARP check for a kernel that accumulates in f32 and stores bf16. `ref` is the
    f32 result on the same bf16-rounded operands; only the accumulation order differs (~1e-6 relative), so the stored
    value is round_bf16(ref) except where ref sits at a rounding midpoint, and then it is the NEIGHBOURING bf16 value:
    every element within ONE bf16 step of round_bf16(ref) (a single missing product of a K = 2304 reduction is ~0.02 =
    2.5 steps at |y| < 2), and at most `frac` of the elements off at all. TOL (1.6e-2 relative + absolute) stays as the
    coarse net below it."""
    yf, rf = y.float(), ref.float()
    rq = rf.to(BF).float()
    step = torch.pow(2.0, torch.floor(torch.log2(rq.abs().clamp_min(1e-30))) - 7)      # bf16 spacing at round(ref)
    err = (yf - rq).abs()
    tol = torch.maximum(step * 1.001, torch.full_like(step, 3e-5))     # (one step; floor: f32 accumulation noise near zero)
    bad = err > tol
    assert not bool(bad.any()), (int(bad.sum()), float((err / tol).max()), float(err.max()))
    off = float((err > 3e-5).float().mean())
    assert off <= frac, off


def conv_ref(x, w, s, p):
    """f32 reference on the GPU's own f32 path would be another kernel of ours; use torch on the CPU."""
    return F.conv2d(q(x), q(w), None, s, p)


FORCE = {
    'conv_glds3_kernel': {'conv.big_minblocks': 1, 'conv.glds4_minblocks': 0, 'conv.splitk_target': 0,
                          'conv.glds3_pp_mink': -1, 'conv.stream_minrows': 0},
    'conv_glds3_kernel<pp>': {'conv.big_minblocks': 1, 'conv.glds4_minblocks': 0, 'conv.splitk_target': 0,
                              'conv.glds3_pp_mink': 0, 'conv.stream_minrows': 0},
    'conv_glds4_kernel': {'conv.glds4_minblocks': 1, 'conv.glds4_pp': 0, 'conv.splitk_target': 0, 'conv.glds4_mf': 8,
                          'conv.stream_minrows': 0},
    'conv_glds4_kernel<pp>': {'conv.glds4_minblocks': 1, 'conv.glds4_pp': 1, 'conv.splitk_target': 0,
                              'conv.glds4_mf': 8, 'conv.stream_minrows': 0},
    'conv_glds4_kernel<pp,288>': {'conv.glds4_minblocks': 1, 'conv.glds4_pp': 1, 'conv.splitk_target': 0,
                                  'conv.glds4_mf': 9, 'conv.stream_minrows': 0},
    # the same 256 x 256 ping-pong tile on v_mfma_f32_32x32x16_bf16 (round 6)
    'conv_glds4_kernel<pp,mf32>': {'conv.glds4_minblocks': 1, 'conv.glds4_pp': 1, 'conv.splitk_target': 0,
                                   'conv.glds4_mf': 8, 'conv.stream_minrows': 0, 'conv.glds4_mfma32': 1},
}

# B, H, W, Cin, Cout, k, stride, pad
SHAPES3 = [
    (2, 19, 23, 128, 128, 3, 1, 1),    # 18 K-steps: the steady-state 3-stage loop, borders on every side
    (1, 33, 47, 64, 96, 3, 2, 1),      # stride 2, N tile overhangs Cout
    (2, 9, 31, 512, 136, 1, 1, 0),     # 1x1, two N tiles (the second nearly empty), M overhang
    (1, 40, 40, 64, 128, 1, 1, 0),     # a single K-step (prologue == epilogue)
    (3, 7, 5, 192, 128, 3, 1, 1),      # planes smaller than a tile: several images per M tile
]
SHAPES4 = [
    (2, 13, 17, 256, 256, 3, 1, 1),    # head-like 3x3, 72 K-steps of 32
    (1, 29, 31, 96, 384, 3, 2, 1),     # Cin % 64 != 0, stride 2, the second N tile half empty
    (2, 8, 13, 2048, 256, 1, 1, 0),    # the 2048 -> 256 lateral (K = 2048)
    (1, 21, 10, 32, 512, 1, 1, 0),     # one K-step
    (1, 16, 26, 64, 256, 3, 1, 1),     # two K-steps per tap
    (3, 6, 7, 256, 264, 3, 1, 1),      # several images per tile, Cout = 256 + 8
]


def _run_forced(kernel, case, **kw):
    o = ops()
    B, H, W, Cin, Cout, k, s, p = case
    x = cases.randn(101, B, Cin, H, W)
    w = cases.randn(102, Cout, Cin, k, k) / (Cin * k * k) ** 0.5
    with o.tuning(**FORCE[kernel]):
        y = o.conv2d(nhwc(x), o.pack_weight(w.to(DEV), BF), k, k, s, p, **kw)
        assert o.last_kernel() == kernel, o.last_kernel()
    return x, w, y


@pytest.mark.parametrize('kernel', ['conv_glds3_kernel', 'conv_glds3_kernel<pp>'])
@pytest.mark.parametrize('case', SHAPES3)
def test_glds3_forced(kernel, case):
    x, w, y = _run_forced(kernel, case)
    np.testing.assert_allclose(nchw(y).numpy(), conv_ref(x, w, case[6], case[7]).numpy(), **TOL)
    assert_bf16_exact(nchw(y), conv_ref(x, w, case[6], case[7]))


@pytest.mark.parametrize('kernel', ['conv_glds4_kernel', 'conv_glds4_kernel<pp>', 'conv_glds4_kernel<pp,288>', 'conv_glds4_kernel<pp,mf32>'])
@pytest.mark.parametrize('case', SHAPES4)
def test_glds4_forced(kernel, case):
    x, w, y = _run_forced(kernel, case)
    np.testing.assert_allclose(nchw(y).numpy(), conv_ref(x, w, case[6], case[7]).numpy(), **TOL)
    assert_bf16_exact(nchw(y), conv_ref(x, w, case[6], case[7]))


@pytest.mark.parametrize('kernel', ['conv_glds3_kernel', 'conv_glds3_kernel<pp>', 'conv_glds4_kernel', 'conv_glds4_kernel<pp>', 'conv_glds4_kernel<pp,288>',
                                    'conv_glds4_kernel<pp,mf32>'])
def test_tile_epilogues_forced(kernel):
    """scale / shift / residual / ReLU, then the BatchNorm statistics of the stored values (in slots)."""
    o = ops()
    B, H, W, Cin, Cout = 2, 18, 15, 128, 256 if 'glds4' in kernel else 128
    x, w = cases.randn(111, B, Cin, H, W), cases.randn(112, Cout, Cin, 3, 3) / (9 * Cin) ** 0.5
    scale, shift = cases.randn(113, Cout).abs() + 0.5, cases.randn(114, Cout)
    res = cases.randn(115, B, Cout, H, W)
    conv = conv_ref(x, w, 1, 1)
    xd, wd = nhwc(x), o.pack_weight(w.to(DEV), BF)
    with o.tuning(**FORCE[kernel]):
        y = o.conv2d(xd, wd, 3, 3, 1, 1, scale=scale.to(DEV), shift=shift.to(DEV), residual=nhwc(res), relu=True)
        assert o.last_kernel() == kernel
        aff_q = q(conv * scale[None, :, None, None] + shift[None, :, None, None])
        np.testing.assert_allclose(nchw(y).numpy(), F.relu(aff_q + q(res)).numpy(), **TOL)
        stats = torch.zeros(3, 2 * Cout, device=DEV)
        y = o.conv2d(xd, wd, 3, 3, 1, 1, stats=stats.view(-1))
        assert o.last_kernel() == kernel
    yq = nchw(y)
    np.testing.assert_allclose(yq.numpy(), q(conv).numpy(), **TOL)
    n = B * H * W
    s_ref = torch.cat([yq.sum((0, 2, 3)), (yq ** 2).sum((0, 2, 3))])      # of the values as stored
    np.testing.assert_allclose(stats.sum(0).cpu().numpy() / n, s_ref.numpy() / n, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize('kernel', ['conv_glds3_kernel', 'conv_glds3_kernel<pp>', 'conv_glds4_kernel', 'conv_glds4_kernel<pp>', 'conv_glds4_kernel<pp,288>',
                                    'conv_glds4_kernel<pp,mf32>'])
def test_tile_ragged_levels_forced(kernel):
    """The head's mode: four FPN levels in one launch, shared 3x3 weights (das_head.py:176-178)."""
    o = ops()
    B, Cin, Cout = 2, 64, 256 if 'glds4' in kernel else 128
    sizes = [(16, 26), (8, 13), (4, 7), (2, 4)]
    w = cases.randn(120, Cout, Cin, 3, 3) / (9 * Cin) ** 0.5
    bias = cases.randn(121, Cout)
    xs = [cases.randn(122 + i, B, Cin, h, ww) for i, (h, ww) in enumerate(sizes)]
    rag = o.Ragged.from_levels([nhwc(t) for t in xs])
    with o.tuning(**FORCE[kernel]):
        y = o.conv2d(rag, o.pack_weight(w.to(DEV), BF), 3, 3, 1, 1, shift=bias.to(DEV), relu=True)
        assert o.last_kernel() == kernel
    for l, t in enumerate(xs):
        ref = F.relu(conv_ref(t, w, 1, 1) + bias[None, :, None, None])
        np.testing.assert_allclose(nchw(y.level(l)).numpy(), ref.numpy(), **TOL)


@pytest.mark.parametrize('kernel', ['conv_glds3_kernel', 'conv_glds3_kernel<pp>', 'conv_glds4_kernel<pp>', 'conv_glds4_kernel<pp,288>', 'conv_glds4_kernel<pp,mf32>'])
def test_tile_dgrad_forced(kernel):
    """Data gradient of a stride-1 3x3 conv on the tile kernels (flipped weights), with a second gradient of the
    same tensor added in the epilogue."""
    o = ops()
    B, H, W, Cin, Cout = 2, 14, 19, 256 if 'glds4' in kernel else 128, 128
    x = cases.randn(130, B, Cin, H, W).requires_grad_(True)
    w = cases.randn(131, Cout, Cin, 3, 3) / (9 * Cin) ** 0.5
    dy = cases.randn(132, B, Cout, H, W)
    other = cases.randn(133, B, Cin, H, W)
    F.conv2d(x, q(w), None, 1, 1).backward(q(dy))
    with o.tuning(**FORCE[kernel]):
        dx = o.conv2d_dgrad(nhwc(dy), o.pack_weight_dgrad(w.to(DEV), BF), 3, 3, 1, 1, (H, W), residual=nhwc(other))
        assert o.last_kernel() == kernel
    np.testing.assert_allclose(nchw(dx).numpy(), q(q(x.grad) + q(other)).numpy(), **TOL)


# ---------------------------------------------------------------- balanced pixel tiles (conv.balance_rows)
BALANCED = [
    # B, H, W, Cin, Cout, k, kernel, conv.balance_rows, expected rows per tile
    (16, 64, 104, 512, 128, 1, 'conv_glds3_kernel', 1, 208),        # 416 tiles of 256 rows on 256 CUs -> 512 of 208
    (16, 64, 104, 512, 256, 1, 'conv_glds4_kernel<pp>', 1, 208),    # the 256 x 256 tile kernel, same rows
    (16, 32, 52, 1024, 256, 1, 'conv_glds3_kernel<pp>', 1, 208),    # 104 x 2 tiles (one round, 81 % full) -> 128 x 2
    (16, 32, 52, 256, 256, 3, 'conv_glds3_kernel<pp>', 2, 208),     # 3x3 (borders inside the shorter tiles): only with 2
    (16, 32, 52, 256, 256, 3, 'conv_glds3_kernel<pp>', 1, 256),     # ... and left alone with 1
]


@pytest.mark.parametrize('case', BALANCED)
def test_balanced_pixel_tiles_store_what_the_full_tiles_store(case):
    """A 256-row tile launch whose last round of one-workgroup-per-CU tiles would be partly empty spreads its rows over the
    tiles of full rounds (ConvP::mstep, das_conv_last_tile_rows): every output element is still the same K loop in the same
    order, so the stored tensor is BIT-identical to the plain tiling's; the BatchNorm statistics (float atomics over other
    workgroups) agree to rounding; epilogue operands (residual, fused BatchNorm-backward with mask bits) follow the rows."""
    o = ops()
    if torch.cuda.get_device_properties(0).multi_processor_count != 256:
        pytest.skip('the cases are sized for 256 CUs')
    B, H, W, Cin, Cout, k, kernel, mode, rows = case
    x = nhwc(cases.randn(171, B, Cin, H, W))
    w = o.pack_weight((cases.randn(172, Cout, Cin, k, k) / (Cin * k * k) ** 0.5).to(DEV), BF)
    res = nhwc(cases.randn(173, B, Cout, H, W))
    raw = nhwc(cases.randn(174, B, Cout, H, W))
    bits = (torch.rand(B * H * W * Cout // 8, device=DEV) * 256).to(torch.uint8)
    mean, invstd = cases.randn(175, Cout).to(DEV) * 0.1, (cases.randn(176, Cout).abs() + 0.5).to(DEV)
    gamma, beta = (cases.randn(177, Cout).abs() + 0.5).to(DEV), (cases.randn(178, Cout) * 0.2).to(DEV)
    fuse = o.BnBwd(raw, None, mean, invstd, gamma, beta, True, bits=bits)
    out = {}
    for m in (0, mode):
        with o.tuning(**{'conv.balance_rows': m, 'conv.stream_minrows': 0}):
            s1, s2 = torch.zeros(2 * Cout, device=DEV), torch.zeros(2 * Cout, device=DEV)
            y1 = o.conv2d(x, w, k, k, 1, k // 2, stats=s1)
            assert o.last_kernel() == kernel, o.last_kernel()
            assert o.last_tile_rows() == (rows if m else 256), (m, o.last_tile_rows())
            y2 = o.conv2d(x, w, k, k, 1, k // 2, residual=res, bn_bwd=fuse, stats=s2)
            assert o.last_tile_rows() == (rows if m else 256), (m, o.last_tile_rows())
            out[m] = (y1, s1, y2, s2)
    assert torch.equal(out[0][0], out[mode][0]) and torch.equal(out[0][2], out[mode][2])
    n = B * H * W
    for i in (1, 3):
        np.testing.assert_allclose(out[mode][i].cpu().numpy() / n, out[0][i].cpu().numpy() / n, rtol=1e-4, atol=1e-5)
    assert float((out[mode][2] == 0).float().mean()) > 0.2      # the mask bits are in effect


# ---------------------------------------------------------------- 3x3, 64 -> 64 channels: patch kernel
C64 = {'conv.c64_mintiles': 1}


@pytest.mark.parametrize('case', [(2, 32, 48), (1, 16, 16), (3, 48, 16), (1, 128, 208), (2, 40, 60), (1, 128, 232)])
def test_c64_patch_kernel_forward_stats_residual(case):
    """conv3x3_c64_kernel (16 x 16-pixel tiles, input patch + all weights in LDS, persistent over tiles; production: the
    128 x 208 stage at B = 16, 1664 tiles on 256 workgroups) against torch: plain output + BatchNorm statistics in
    slots; scale / shift / residual / ReLU; more tiles than workgroups AND fewer (a single tile); image borders on
    every side of every image; heights / widths that are not multiples of 16 (border squares partly outside: 928-wide
    frames give W = 232); and the same launch on the 128-row tile kernel gives the same stored values."""
    o = ops()
    B, H, W = case
    x, w = cases.randn(141, B, 64, H, W), cases.randn(142, 64, 64, 3, 3) / 24
    conv = conv_ref(x, w, 1, 1)
    xd, wd = nhwc(x), o.pack_weight(w.to(DEV), BF)
    stats = torch.zeros(4, 128, device=DEV)
    with o.tuning(**C64):
        y = o.conv2d(xd, wd, 3, 3, 1, 1, stats=stats.view(-1))
        assert o.last_kernel() == 'conv3x3_c64_kernel', o.last_kernel()
    yq = nchw(y)
    np.testing.assert_allclose(yq.numpy(), q(conv).numpy(), **TOL)
    n = B * H * W
    s_ref = torch.cat([yq.sum((0, 2, 3)), (yq ** 2).sum((0, 2, 3))])
    np.testing.assert_allclose(stats.sum(0).cpu().numpy() / n, s_ref.numpy() / n, rtol=1e-3, atol=1e-3)
    with o.tuning(**{'conv.c64_mintiles': 0}):
        y_tile = o.conv2d(xd, wd, 3, 3, 1, 1)
        assert o.last_kernel() != 'conv3x3_c64_kernel'
    np.testing.assert_allclose(y.float().cpu().numpy(), y_tile.float().cpu().numpy(), rtol=8e-3, atol=8e-3)
    scale, shift = cases.randn(143, 64).abs() + 0.5, cases.randn(144, 64)
    res = cases.randn(145, B, 64, H, W)
    with o.tuning(**C64):
        y2 = o.conv2d(xd, wd, 3, 3, 1, 1, scale=scale.to(DEV), shift=shift.to(DEV), residual=nhwc(res), relu=True)
        assert o.last_kernel() == 'conv3x3_c64_kernel'
    aff_q = q(conv * scale[None, :, None, None] + shift[None, :, None, None])
    np.testing.assert_allclose(nchw(y2).numpy(), F.relu(aff_q + q(res)).numpy(), **TOL)


def test_c64_patch_kernel_data_gradient_and_channel_slices():
    """Data gradient (flipped weights) with a second gradient added in the epilogue; input and output as channel
    slices of wider tensors (pixel strides 128 / 96)."""
    o = ops()
    B, H, W = 2, 32, 32
    x = cases.randn(151, B, 64, H, W).requires_grad_(True)
    w = cases.randn(152, 64, 64, 3, 3) / 24
    dy, other = cases.randn(153, B, 64, H, W), cases.randn(154, B, 64, H, W)
    F.conv2d(x, q(w), None, 1, 1).backward(q(dy))
    wide_in = torch.zeros(B, H, W, 128, device=DEV, dtype=BF)
    wide_in[..., 32:96] = nhwc(dy)
    with o.tuning(**C64):
        dx = o.conv2d_dgrad(wide_in[..., 32:96], o.pack_weight_dgrad(w.to(DEV), BF), 3, 3, 1, 1, (H, W), residual=nhwc(other))
        assert o.last_kernel() == 'conv3x3_c64_kernel', o.last_kernel()
    np.testing.assert_allclose(nchw(dx).numpy(), q(q(x.grad) + q(other)).numpy(), **TOL)


# ---------------------------------------------------------------- the 7x7 stride-2 stem conv (8 stored channels -> 64)
@pytest.mark.parametrize('case', [(2, 64, 128), (1, 32, 64), (3, 70, 150), (1, 65, 97), (2, 128, 208), (5, 48, 64)])
def test_stem7x7_kernel_statistics_affine_borders(case):
    """conv_stem7x7_kernel (conv_stem.hip: weights in registers, 21 x 72-pixel input windows through LDS-DMA, persistent over
    8 x 32-pixel output tiles; production: B x 512 x 832 frames, 6656 tiles on 256 workgroups) against torch on the same
    bf16 operands: the training forward (plain output + BatchNorm statistics in slots) and the eval forward (folded
    BatchNorm + ReLU); frames whose output is a whole number of tiles and frames whose last tiles hang over the right /
    bottom border (odd heights and widths too), more tiles than workgroups' first round and a single tile; and the same
    launch on the generic kernel stores the same values up to accumulation order (reference layer:
    /root/reference/mmdet3d/models/backbones/mspn_mmpose.py:228-246)."""
    o = ops()
    B, H, W = case
    x = cases.randn(171, B, 3, H, W)
    w = cases.randn(172, 64, 3, 7, 7) / 12
    conv = conv_ref(x, w, 2, 3)
    x8 = torch.zeros(B, H, W, 8, dtype=BF, device=DEV)
    x8[..., :3] = nhwc(x)
    w8 = torch.zeros(64, 8, 7, 7)
    w8[:, :3] = w
    wd = o.pack_weight(w8.to(DEV), BF)
    stats = torch.zeros(4, 128, device=DEV)
    y = o.conv2d(x8, wd, 7, 7, 2, 3, stats=stats.view(-1))
    assert o.last_kernel() == 'conv_stem7x7_kernel', o.last_kernel()
    yq = nchw(y)
    assert yq.shape == conv.shape
    assert_bf16_exact(yq, conv)
    n = yq.numel() // 64
    s_ref = torch.cat([yq.sum((0, 2, 3)), (yq ** 2).sum((0, 2, 3))])
    np.testing.assert_allclose(stats.sum(0).cpu().numpy() / n, s_ref.numpy() / n, rtol=1e-3, atol=1e-3)
    with o.tuning(**{'conv.stem7x7': 0}):
        y_gen = o.conv2d(x8, wd, 7, 7, 2, 3)
        assert o.last_kernel() == 'conv_reg_kernel', o.last_kernel()
    np.testing.assert_allclose(y.float().cpu().numpy(), y_gen.float().cpu().numpy(), rtol=8e-3, atol=8e-3)
    scale, shift = cases.randn(173, 64).abs() + 0.5, cases.randn(174, 64)
    y2 = o.conv2d(x8, wd, 7, 7, 2, 3, scale=scale.to(DEV), shift=shift.to(DEV), relu=True)
    assert o.last_kernel() == 'conv_stem7x7_kernel', o.last_kernel()
    aff = conv * scale[None, :, None, None] + shift[None, :, None, None]
    assert_bf16_exact(nchw(y2), F.relu(aff), frac=0.08)
    # an output written into a channel slice of a wider tensor (pixel stride 96)
    wide = torch.full((B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, 96), 7.0, dtype=BF, device=DEV)
    o.conv2d(x8, wd, 7, 7, 2, 3, out=wide[..., 16:80])
    assert o.last_kernel() == 'conv_stem7x7_kernel', o.last_kernel()
    assert torch.equal(wide[..., 16:80], y) and bool((wide[..., :16] == 7).all()) and bool((wide[..., 80:] == 7).all())


def test_stem7x7_kernel_leaves_other_launches_alone():
    o = ops()
    x8 = nhwc(cases.randn(175, 1, 8, 64, 64))
    w7 = o.pack_weight((cases.randn(176, 64, 8, 7, 7) / 20).to(DEV), BF)
    o.conv2d(x8, w7, 7, 7, 1, 3)                                   # stride 1
    assert o.last_kernel() != 'conv_stem7x7_kernel'
    o.conv2d(x8, w7, 7, 7, 2, 3, residual=torch.zeros(1, 32, 32, 64, dtype=BF, device=DEV))   # a residual
    assert o.last_kernel() != 'conv_stem7x7_kernel'
    w128 = o.pack_weight((cases.randn(177, 128, 8, 7, 7) / 20).to(DEV), BF)
    o.conv2d(x8, w128, 7, 7, 2, 3)                                 # 128 output channels
    assert o.last_kernel() != 'conv_stem7x7_kernel'
    x16 = nhwc(cases.randn(178, 1, 16, 64, 64))
    w16 = o.pack_weight((cases.randn(179, 64, 16, 7, 7) / 20).to(DEV), BF)
    o.conv2d(x16, w16, 7, 7, 2, 3)                                 # 16 input channels
    assert o.last_kernel() != 'conv_stem7x7_kernel'
    tiny = nhwc(cases.randn(180, 1, 8, 10, 12))                    # one tile, mostly outside the frame
    o.conv2d(tiny, w7, 7, 7, 2, 3)
    assert o.last_kernel() != 'conv_stem7x7_kernel'


def test_c64_patch_kernel_leaves_other_shapes_alone():
    o = ops()
    with o.tuning(**C64):
        for (H, W, Cin, Cout, k) in [(17, 17, 64, 64, 3), (32, 32, 64, 128, 3), (32, 32, 128, 64, 3), (32, 32, 64, 64, 1)]:
            x = nhwc(cases.randn(161, 1, Cin, H, W))
            w = o.pack_weight((cases.randn(162, Cout, Cin, k, k) / 24).to(DEV), BF)
            o.conv2d(x, w, k, k, 1, k // 2)
            assert o.last_kernel() != 'conv3x3_c64_kernel', (H, W, Cin, Cout, k)


# ---------------------------------------------------------------- split-K (small M, long K)
SPLITK = {
    # kernel: (tuning that routes the case to it, bit in conv.splitk_kernels, cases B, H, W, Cin, Cout, k, stride, pad)
    'conv_glds_kernel': ({'conv.big_minblocks': 1 << 30, 'conv.glds4_minblocks': 0, 'conv.stream_minrows': 0,
                          'conv.splitk_target': 0}, 1, [
        (2, 16, 26, 512, 512, 3, 1, 1),    # layer4-like 3x3: 7 x 4 tiles, 72 K steps
        (1, 9, 13, 2048, 136, 1, 1, 0),    # 1x1, K = 2048, M and N overhang
        (2, 32, 52, 256, 256, 3, 2, 1),    # stride 2
        (3, 7, 5, 192, 64, 3, 1, 1),       # BN = 64 tile, several images per M tile, odd split boundaries (27 steps)
    ]),
    'conv_glds3_kernel<pp>': (FORCE['conv_glds3_kernel<pp>'], 2, [
        (2, 16, 26, 512, 512, 3, 1, 1), (3, 7, 5, 192, 128, 3, 1, 1),
    ]),
    'conv_glds3_kernel': (FORCE['conv_glds3_kernel'], 2, [
        (2, 16, 26, 512, 512, 3, 1, 1), (1, 9, 13, 2048, 136, 1, 1, 0), (2, 32, 52, 256, 256, 3, 2, 1),
        (3, 7, 5, 192, 128, 3, 1, 1),      # splits start in the middle of a tap (192 channels = 3 steps per tap)
    ]),
    'conv_glds4_kernel': (FORCE['conv_glds4_kernel'], 4, [
        (2, 16, 26, 512, 512, 3, 1, 1), (1, 9, 13, 2048, 264, 1, 1, 0), (3, 7, 5, 96, 256, 3, 1, 1),
    ]),
    'conv_glds4_kernel<pp>': (FORCE['conv_glds4_kernel<pp>'], 4, [
        (2, 16, 26, 512, 512, 3, 1, 1), (2, 32, 52, 256, 256, 3, 2, 1), (3, 7, 5, 96, 256, 3, 1, 1),
    ]),
}


@pytest.mark.parametrize('kernel,idx', [(k, i) for k, v in SPLITK.items() for i in range(len(v[2]))])
def test_splitk_matches_unsplit(kernel, idx):
    """Each tile kernel with its K loop split over blockIdx.y + splitk_finish_kernel, against F.conv2d and against the
    same kernel unsplit."""
    o = ops()
    force, bit, shapes = SPLITK[kernel]
    B, H, W, Cin, Cout, k, s, p = shapes[idx]
    x = cases.randn(201, B, Cin, H, W)
    w = cases.randn(202, Cout, Cin, k, k) / (Cin * k * k) ** 0.5
    xd, wd = nhwc(x), o.pack_weight(w.to(DEV), BF)
    with o.tuning(**force):
        y1 = o.conv2d(xd, wd, k, k, s, p)
        assert o.last_kernel() == kernel, o.last_kernel()
    with o.tuning(**{**force, 'conv.splitk_target': 512, 'conv.splitk_minsteps': 2, 'conv.splitk_kernels': bit}):
        y2 = o.conv2d(xd, wd, k, k, s, p)
        assert o.last_kernel() == kernel.replace('<pp>', '') + '<splitk>', o.last_kernel()
    np.testing.assert_allclose(nchw(y2).numpy(), conv_ref(x, w, s, p).numpy(), **TOL)
    # f32 partial sums regrouped: equal up to one bf16 rounding of the result
    np.testing.assert_allclose(nchw(y2).numpy(), nchw(y1).numpy(), rtol=8e-3, atol=2e-3)


@pytest.mark.parametrize('kernel', ['conv_glds3_kernel<pp>', 'conv_glds3_kernel'])
def test_splitk_finished_inside_the_kernel_is_bitwise_the_two_launch_result(kernel):
    """Tuning key conv.splitk_inkernel = 1 (round 6): the split-K sum of conv_glds3_kernel is finished by the LAST workgroup to
    arrive at a tile (agent-scope release / ticket / acquire, csrc/conv_igemm.hip splitk_arrive_and_reduce) instead of by
    splitk_finish_kernel. The slabs are summed in slab order whoever arrives last, so the result must be BITWISE the two-launch
    result — for every shape of the split-K set, with residual + ReLU and with BatchNorm statistics, and on every one of 40
    repetitions while another stream keeps 64 CUs busy in bursts (uneven arrival order; a stale slab or a torn counter shows as
    a differing element)."""
    import ctypes
    from das_amd import _lib
    o = ops()
    lib = _lib.load()
    force, bit, shapes = SPLITK[kernel]
    side = torch.cuda.Stream()
    for (B, H, W, Cin, Cout, k, s, p) in shapes:
        x = cases.randn(221, B, Cin, H, W)
        w = cases.randn(222, Cout, Cin, k, k) / (Cin * k * k) ** 0.5
        xd, wd = nhwc(x), o.pack_weight(w.to(DEV), BF)
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        res = nhwc(cases.randn(223, B, Cout, Ho, Wo))
        split = {**force, 'conv.splitk_target': 512, 'conv.splitk_minsteps': 2, 'conv.splitk_kernels': bit}
        with o.tuning(**split):
            y0 = o.conv2d(xd, wd, k, k, s, p, residual=res, relu=True)
            assert o.last_kernel().endswith('<splitk>'), o.last_kernel()
            st0 = torch.zeros(2 * Cout, device=DEV)
            r0 = o.conv2d(xd, wd, k, k, s, p, stats=st0)
        with o.tuning(**{**split, 'conv.splitk_inkernel': 1}):
            for rep in range(40):
                if rep % 4 == 0:     # bursts of foreign work beside the launch: the splits of a tile arrive in a different order
                    _lib.check(lib.das_dev_occupy_cus(64, 256, 4096, 30 + 7 * rep, ctypes.c_void_p(side.cuda_stream)), 'occupy')
                y1 = o.conv2d(xd, wd, k, k, s, p, residual=res, relu=True)
                assert o.last_kernel().endswith('<splitk>'), o.last_kernel()
                assert torch.equal(y1, y0), (kernel, (B, H, W, Cin, Cout, k, s, p), rep, float((y1.float() - y0.float()).abs().max()))
            st1 = torch.zeros(2 * Cout, device=DEV)
            r1 = o.conv2d(xd, wd, k, k, s, p, stats=st1)
            assert torch.equal(r1, r0)
            np.testing.assert_allclose(st1.cpu().numpy(), st0.cpu().numpy(), rtol=1e-5, atol=1e-3)   # (float atomics: order)
    torch.cuda.synchronize()


def test_splitk_epilogues_and_default_dispatch():
    """The layer4 3x3 at B=8 takes split-K by default; every epilogue runs in the finishing kernel."""
    o = ops()
    B, H, W, Cin, Cout = 8, 16, 26, 512, 512
    x, w = cases.randn(211, B, Cin, H, W), cases.randn(212, Cout, Cin, 3, 3) / (9 * Cin) ** 0.5
    scale, shift = cases.randn(213, Cout).abs() + 0.5, cases.randn(214, Cout)
    res = cases.randn(215, B, Cout, H, W)
    torch.set_num_threads(8)
    conv = conv_ref(x, w, 1, 1)
    xd, wd = nhwc(x), o.pack_weight(w.to(DEV), BF)
    y = o.conv2d(xd, wd, 3, 3, 1, 1, scale=scale.to(DEV), shift=shift.to(DEV), residual=nhwc(res), relu=True)
    assert o.last_kernel().endswith('<splitk>'), o.last_kernel()
    aff_q = q(conv * scale[None, :, None, None] + shift[None, :, None, None])
    np.testing.assert_allclose(nchw(y).numpy(), F.relu(aff_q + q(res)).numpy(), **TOL)
    stats = torch.zeros(2 * Cout, device=DEV)
    y = o.conv2d(xd, wd, 3, 3, 1, 1, stats=stats)
    assert o.last_kernel().endswith('<splitk>'), o.last_kernel()
    yq = nchw(y)
    n = B * H * W
    s_ref = torch.cat([yq.sum((0, 2, 3)), (yq ** 2).sum((0, 2, 3))])
    np.testing.assert_allclose(stats.cpu().numpy() / n, s_ref.numpy() / n, rtol=1e-3, atol=1e-3)
    # data gradient of a stride-2 3x3 (zero-upsampled dY) through the split path
    dy = cases.randn(216, B, Cout, H // 2, W // 2)
    xr = cases.randn(217, B, Cin, H, W).requires_grad_(True)
    F.conv2d(xr, q(w), None, 2, 1).backward(q(dy))
    dx = o.conv2d_dgrad(nhwc(dy), o.pack_weight_dgrad(w.to(DEV), BF), 3, 3, 2, 1, (H, W))
    assert o.last_kernel() == 'conv_glds_kernel<splitk>', o.last_kernel()
    np.testing.assert_allclose(nchw(dx).numpy(), q(xr.grad).numpy(), **TOL)


@pytest.mark.parametrize('cin,cout', [(64, 256), (128, 512), (256, 64)])
def test_stream_affine_residual(cin, cout):
    """conv1x1_stream_kernel mode 5: the closing 1x1 of an eval-mode bottleneck (folded BN, + identity, ReLU)."""
    o = ops()
    B, H, W = 2, 64, 130
    x, w = cases.randn(221, B, cin, H, W), cases.randn(222, cout, cin, 1, 1) / cin ** 0.5
    scale, shift = cases.randn(223, cout).abs() + 0.5, cases.randn(224, cout)
    res = cases.randn(225, B, cout, H, W)
    y = o.conv2d(nhwc(x), o.pack_weight(w.to(DEV), BF), 1, 1, 1, 0, scale=scale.to(DEV), shift=shift.to(DEV),
                 residual=nhwc(res), relu=True)
    assert o.last_kernel() == 'conv1x1_stream_kernel', o.last_kernel()
    aff_q = q(conv_ref(x, w, 1, 0) * scale[None, :, None, None] + shift[None, :, None, None])
    np.testing.assert_allclose(nchw(y).numpy(), F.relu(aff_q + q(res)).numpy(), **TOL)


@pytest.mark.parametrize('cin,cout', [(64, 256), (128, 512), (256, 64), (256, 1024), (64, 64)])
def test_stream_and_c64_plain_outputs_are_bf16_exact(cin, cout):
    """The two persistent kernels that carry most of the step's convolution bytes, held to the bf16-exact bar
    (assert_bf16_exact): conv1x1_stream_kernel on every (K, Cout) class of the step (plain / statistics mode stores the
    rounded f32 sum), and conv3x3_c64_kernel."""
    o = ops()
    B, H, W = 2, 96, 104
    x = cases.randn(231, B, cin, H, W)
    w = cases.randn(232, cout, cin, 1, 1) / cin ** 0.5
    stats = torch.zeros(2 * cout, device=DEV)
    y = o.conv2d(nhwc(x), o.pack_weight(w.to(DEV), BF), 1, 1, 1, 0, stats=stats)
    assert o.last_kernel() == 'conv1x1_stream_kernel', o.last_kernel()
    assert_bf16_exact(nchw(y), conv_ref(x, w, 1, 0))
    if cin == 64 and cout == 64:
        w3 = cases.randn(233, 64, 64, 3, 3) / 24
        with o.tuning(**{'conv.c64_mintiles': 1}):
            y3 = o.conv2d(nhwc(x), o.pack_weight(w3.to(DEV), BF), 3, 3, 1, 1)
            assert o.last_kernel() == 'conv3x3_c64_kernel', o.last_kernel()
        assert_bf16_exact(nchw(y3), conv_ref(x, w3, 1, 1))


# ---------------------------------------------------------------- stride-2 data gradient by output parity
S2 = [
    # B, H, W, Cin, Cout, k, pad
    (2, 32, 52, 128, 128, 3, 1),     # the Bottleneck's 3x3 stride 2
    (2, 32, 52, 256, 512, 1, 0),     # the downsample branch: only the even-even pixels receive a gradient
    (1, 17, 23, 64, 72, 3, 1),       # odd sizes: the four sub-grids differ in size
    (3, 9, 6, 64, 64, 1, 0),
]


@pytest.mark.parametrize('case', S2)
def test_dgrad_stride2_parity_classes(case):
    """conv2d_dgrad(stride 2) = one sub-grid launch per output parity (DasConvDesc.out_sub) against autograd, against the
    zero-upsampled form (in_up = 2), with a second gradient added, and accumulating in place."""
    o = ops()
    B, H, W, Cin, Cout, k, p = case
    x = cases.randn(230, B, Cin, H, W).requires_grad_(True)
    w = cases.randn(231, Cout, Cin, k, k) / (Cin * k * k) ** 0.5
    Ho, Wo = (H + 2 * p - k) // 2 + 1, (W + 2 * p - k) // 2 + 1
    dy = cases.randn(232, B, Cout, Ho, Wo)
    other = cases.randn(233, B, Cin, H, W)
    F.conv2d(x, q(w), None, 2, p).backward(q(dy))
    wd = o.pack_weight_dgrad(w.to(DEV), BF)
    dx = o.conv2d_dgrad(nhwc(dy), wd, k, k, 2, p, (H, W))
    np.testing.assert_allclose(nchw(dx).numpy(), q(x.grad).numpy(), **TOL)
    old = o.conv2d(nhwc(dy), wd, k, k, 1, k - 1 - p, in_up=2, out_hw=(H, W))
    np.testing.assert_allclose(nchw(dx).numpy(), nchw(old).numpy(), rtol=8e-3, atol=2e-3)
    dx2 = o.conv2d_dgrad(nhwc(dy), wd, k, k, 2, p, (H, W), residual=nhwc(other))
    np.testing.assert_allclose(nchw(dx2).numpy(), q(q(x.grad) + q(other)).numpy(), **TOL)
    acc = nhwc(other).clone()
    dx3 = o.conv2d_dgrad(nhwc(dy), wd, k, k, 2, p, (H, W), accumulate=acc)
    assert dx3.data_ptr() == acc.data_ptr()
    np.testing.assert_allclose(nchw(dx3).numpy(), nchw(dx2).numpy(), rtol=0, atol=0)


def test_dgrad_stride2_fused_bn_backward():
    """The 3x3 stride-2 data gradient with the BatchNorm-backward reduction folded in: same dZ and the same sums as the
    zero-upsampled launch."""
    o = ops()
    B, H, W, Cin, Cout = 2, 32, 52, 128, 128
    w = cases.randn(241, Cout, Cin, 3, 3) / (9 * Cin) ** 0.5
    dy = nhwc(cases.randn(242, B, Cout, H // 2, W // 2))
    raw = nhwc(cases.randn(243, B, Cin, H, W))
    mean, invstd = cases.randn(244, Cin).to(DEV) * 0.1, (cases.randn(245, Cin).abs() + 0.5).to(DEV)
    gamma, beta = (cases.randn(246, Cin).abs() + 0.5).to(DEV), (cases.randn(247, Cin) * 0.2).to(DEV)
    wd = o.pack_weight_dgrad(w.to(DEV), BF)
    fuse = o.BnBwd(raw, None, mean, invstd, gamma, beta, True)
    s_new, s_old = torch.zeros(2 * Cin, device=DEV), torch.zeros(2 * Cin, device=DEV)
    dz_new = o.conv2d_dgrad(dy, wd, 3, 3, 2, 1, (H, W), bn_bwd=fuse, stats=s_new)
    dz_old = o.conv2d(dy, wd, 3, 3, 1, 1, in_up=2, out_hw=(H, W), bn_bwd=fuse, stats=s_old)
    np.testing.assert_allclose(dz_new.float().cpu().numpy(), dz_old.float().cpu().numpy(), rtol=8e-3, atol=2e-3)
    n = B * H * W
    np.testing.assert_allclose(s_new.cpu().numpy() / n, s_old.cpu().numpy() / n, rtol=2e-3, atol=2e-4)
    assert float((dz_new == 0).float().mean()) > 0.2      # the mask is in effect


# ---------------------------------------------------------------- production sizes, default dispatch
REAL = [
    # B, H, W, Cin, Cout, k, stride, pad, expected kernel
    (2, 128, 208, 256, 256, 3, 1, 1, 'conv_glds4_kernel<pp>'),      # 208 tiles of 256 x 256, K = 2304
    (2, 128, 208, 64, 256, 3, 1, 1, 'conv_glds4_kernel<pp>'),       # K = 576
    (16, 16, 26, 2048, 2048, 1, 1, 0, 'conv_glds4_kernel<pp>'),     # out_skip1 of the coarsest unit: 26 x 8 tiles
    (4, 128, 208, 128, 128, 3, 1, 1, 'conv_glds3_kernel<pp>'),      # layer2-like 3x3 on a large map (K = 1152: ping-pong)
    (16, 32, 52, 1024, 128, 1, 1, 0, 'conv_glds3_kernel<pp>'),      # 104 tiles, K = 1024
    (2, 128, 208, 256, 64, 1, 1, 0, 'conv1x1_stream_kernel'),
]


@pytest.mark.parametrize('case', REAL)
def test_tiles_real_sizes(case):
    o = ops()
    B, H, W, Cin, Cout, k, s, p, expect = case
    x = cases.randn(141, B, Cin, H, W)
    w = cases.randn(142, Cout, Cin, k, k) / (Cin * k * k) ** 0.5
    torch.set_num_threads(8)
    ref = conv_ref(x, w, s, p)
    y = o.conv2d(nhwc(x), o.pack_weight(w.to(DEV), BF), k, k, s, p)
    assert o.last_kernel() == expect, o.last_kernel()
    np.testing.assert_allclose(nchw(y).numpy(), ref.numpy(), **TOL)
    assert_bf16_exact(nchw(y), ref)


def test_head_ragged_real_size():
    """The real head launch: 4 levels x B=16 x 256 channels, 3x3 256 -> 256 (187 tiles of 256 x 256, ping-pong)."""
    o = ops()
    B, Cc = 16, 256
    sizes = [(64, 104), (32, 52), (16, 26), (8, 13)]
    w = cases.randn(150, Cc, Cc, 3, 3) / (9 * Cc) ** 0.5
    xs = [cases.randn(151 + i, B, Cc, h, ww) for i, (h, ww) in enumerate(sizes)]
    rag = o.Ragged.from_levels([nhwc(t) for t in xs])
    y = o.conv2d(rag, o.pack_weight(w.to(DEV), BF), 3, 3, 1, 1)
    # 553 tiles of 256 rows = 3 rounds on 256 CUs; 492 tiles of 288 rows = 2 rounds: the launcher takes the taller tile
    assert o.last_kernel() == 'conv_glds4_kernel<pp,288>', o.last_kernel()
    torch.set_num_threads(8)
    for l in (3, 2, 1):      # (level 0 alone is 106k rows x 2304 x 256: skipped on the CPU side, levels share the code path)
        np.testing.assert_allclose(nchw(y.level(l)).numpy(), conv_ref(xs[l], w, 1, 1).numpy(), **TOL)
    # level 0: a band of rows across the top border, an interior band and the bottom border of the last image
    ref0 = conv_ref(xs[0][-1:], w, 1, 1)
    np.testing.assert_allclose(nchw(y.level(0)[-1:]).numpy(), ref0.numpy(), **TOL)


def test_tail_split_head_conv_at_inference_batch():
    """B = 8 head conv: 277 tiles of 256 x 256 on 256 CUs. Default: 246 tiles of 288 rows (one round). With the tile
    height pinned to 256 the 21 tiles of the second round go to a split-K tail launch (conv.tail_split). Same values
    as the single launch; spot-checked against the CPU reference in the rows either launch owns."""
    o = ops()
    B, Cc = 8, 256
    sizes = [(64, 104), (32, 52), (16, 26), (8, 13)]
    w = cases.randn(190, Cc, Cc, 3, 3) / (9 * Cc) ** 0.5
    xs = [cases.randn(191 + i, B, Cc, h, ww) for i, (h, ww) in enumerate(sizes)]
    rag = o.Ragged.from_levels([nhwc(t) for t in xs])
    assert (rag.rows + 255) // 256 == 277
    wq = o.pack_weight(w.to(DEV), BF)
    stats9 = torch.zeros(16 * 2 * Cc, device=DEV)
    y9 = o.conv2d(rag, wq, 3, 3, 1, 1, stats=stats9)
    assert o.last_kernel() == 'conv_glds4_kernel<pp,288>', o.last_kernel()
    with o.tuning(**{'conv.glds4_mf': 8}):
        stats = torch.zeros(16 * 2 * Cc, device=DEV)
        y = o.conv2d(rag, wq, 3, 3, 1, 1, stats=stats)
        assert o.last_kernel() == 'conv_glds4_kernel<pp>'
    with o.tuning(**{'conv.tail_split': 0, 'conv.glds4_mf': 8}):
        stats1 = torch.zeros(16 * 2 * Cc, device=DEV)
        y1 = o.conv2d(rag, wq, 3, 3, 1, 1, stats=stats1)
    for ya, sa in ((y, stats), (y9, stats9)):
        np.testing.assert_allclose(ya.data.float().cpu().numpy(), y1.data.float().cpu().numpy(), rtol=8e-3, atol=8e-3)
        np.testing.assert_allclose(sa.view(16, -1).sum(0).cpu().numpy() / rag.rows, stats1.view(16, -1).sum(0).cpu().numpy() / rag.rows,
                                   rtol=1e-3, atol=1e-4)
    torch.set_num_threads(8)
    for l in (3, 2, 1):            # levels 1-3 are the last rows: the tail launch owns levels 2, 3 and the end of level 1
        np.testing.assert_allclose(nchw(y.level(l)).numpy(), conv_ref(xs[l], w, 1, 1).numpy(), **TOL)
    np.testing.assert_allclose(nchw(y.level(0)[:1]).numpy(), conv_ref(xs[0][:1], w, 1, 1).numpy(), **TOL)


WGRAD_REAL = [
    # B, H, W, Cin, Cout, k, stride, pad, expected kernel
    (2, 128, 208, 64, 64, 3, 1, 1, 'conv_wgrad_c64_kernel'),    # 208 squares of 16 x 16 pixels on 208 workgroups
    (2, 128, 208, 64, 72, 3, 1, 1, 'conv_wgrad_kernel'),        # 5 tiles x 153 splits = 765 workgroups
    (3, 128, 208, 256, 256, 1, 1, 0, 'conv_wgrad_pp_kernel'),   # 1 tile x 256 splits
    (16, 32, 52, 256, 256, 3, 1, 1, 'conv_wgrad_pp_kernel'),    # 9 tiles x 28 splits
    (8, 64, 104, 256, 512, 1, 2, 0, 'conv_wgrad_pp_kernel'),    # stride-2 1x1 (the downsample branch)
    (4, 64, 104, 128, 128, 3, 2, 1, 'conv_wgrad_kernel'),       # stride-2 3x3
]


@pytest.mark.parametrize('case', WGRAD_REAL)
def test_wgrad_real_split(case):
    o = ops()
    B, H, W, Cin, Cout, k, s, p, expect = case
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    x = cases.randn(161, B, Cin, H, W)
    dy = cases.randn(162, B, Cout, Ho, Wo) / (B * Ho * Wo) ** 0.5
    torch.set_num_threads(8)
    w0 = torch.zeros(Cout, Cin, k, k, requires_grad=True)
    F.conv2d(q(x), w0, None, s, p).backward(q(dy))
    dw = o.conv2d_wgrad(nhwc(x), nhwc(dy), k, k, s, p)
    assert o.last_kernel() == expect, o.last_kernel()
    got = dw.cpu().permute(0, 3, 1, 2)
    np.testing.assert_allclose(got.numpy(), w0.grad.numpy(), rtol=2e-3, atol=2e-3 * float(w0.grad.abs().max()))
    # accumulate = 1 adds to what is there (the flat-gradient path), deterministically
    acc = dw.clone()
    o.conv2d_wgrad(nhwc(x), nhwc(dy), k, k, s, p, out=acc, accumulate=True)
    np.testing.assert_allclose(acc.cpu().numpy(), 2 * dw.cpu().numpy(), rtol=1e-5, atol=1e-5)
    dw2 = o.conv2d_wgrad(nhwc(x), nhwc(dy), k, k, s, p)
    np.testing.assert_allclose(dw2.cpu().numpy(), dw.cpu().numpy(), rtol=1e-5, atol=1e-6)


WGRAD_SHAPES = [
    # B, H, W, Cin, Cout, k, stride, pad, expected wave arrangement (1 = 64 x 256, 2 = 256 x 64)
    (2, 128, 208, 64, 256, 1, 1, 0, 2),     # K = 64: conv3 of the 128 x 208 stage's bottlenecks
    (2, 128, 208, 256, 64, 1, 1, 0, 1),     # Cout = 64: conv1 of the same
    (2, 64, 104, 256, 32, 3, 1, 1, 1),      # Cout = 32 (half of the 64-row tile empty), nine K tiles
    (1, 64, 96, 8, 64, 7, 2, 3, 1),         # the stem: K = 392 = one full K tile + 136 columns, 8-channel taps
    (2, 40, 60, 64, 200, 1, 1, 0, 2),       # Cout = 200 overhangs the 256-row tile
    (2, 40, 60, 72, 48, 3, 1, 1, 1),        # K = 648: the taps' 72 channels straddle the 16-byte slots' swizzle
]


@pytest.mark.parametrize('case', WGRAD_SHAPES)
def test_wgrad_wave_arrangements(case):
    """conv_wgrad_kernel<bf16> as 64 x 256 / 256 x 64 tiles (four waves of 64 x 64 in a row / a column) for the layers whose
    128 x 128 tile was half empty: against torch's weight gradient, against the 128 x 128 arrangement of the same op
    (wgrad.shapes = 0: f32 summation order only), accumulating, deterministic; das_wgrad_last_plan says which ran."""
    o = ops()
    B, H, W, Cin, Cout, k, s, p, shape = case
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    x = cases.randn(261, B, Cin, H, W)
    dy = cases.randn(262, B, Cout, Ho, Wo) / (B * Ho * Wo) ** 0.5
    torch.set_num_threads(8)
    w0 = torch.zeros(Cout, Cin, k, k, requires_grad=True)
    F.conv2d(q(x), w0, None, s, p).backward(q(dy))
    ref = w0.grad.permute(0, 2, 3, 1).numpy()
    xd, dyd = nhwc(x), nhwc(dy)
    dw = o.conv2d_wgrad(xd, dyd, k, k, s, p)
    assert o.last_kernel() == 'conv_wgrad_kernel' and o.last_wgrad_plan()['shape'] == shape, o.last_wgrad_plan()
    np.testing.assert_allclose(dw.cpu().numpy(), ref, rtol=2e-3, atol=2e-3 * float(np.abs(ref).max()))
    with o.tuning(**{'wgrad.shapes': 0}):
        sq = o.conv2d_wgrad(xd, dyd, k, k, s, p)
        assert o.last_wgrad_plan()['shape'] == 0
    np.testing.assert_allclose(dw.cpu().numpy(), sq.cpu().numpy(), rtol=1e-4, atol=2e-5 * float(np.abs(ref).max()))
    acc = dw.clone()
    o.conv2d_wgrad(xd, dyd, k, k, s, p, out=acc, accumulate=True)
    np.testing.assert_allclose(acc.cpu().numpy(), 2 * dw.cpu().numpy(), rtol=1e-5, atol=1e-6)
    again = o.conv2d_wgrad(xd, dyd, k, k, s, p)
    if o.last_wgrad_plan()['groups'] == 1:      # one reduce group per tile: fixed summation order
        assert torch.equal(again, dw)
    else:                                       # a lone small result cut into hundreds of runs: atomic reduce groups
        np.testing.assert_allclose(again.cpu().numpy(), dw.cpu().numpy(), rtol=1e-4, atol=1e-6 * float(np.abs(ref).max()) + 1e-7)
    # one launch with all three arrangements side by side == the single launches
    x2 = cases.randn(263, B, 128, H, W)
    dy2 = cases.randn(264, B, 128, H, W) / (B * H * W) ** 0.5
    outs = [torch.zeros(Cout, k, k, Cin, device=DEV), torch.zeros(128, 1, 1, 128, device=DEV)]
    o.conv2d_wgrad_batch([(xd, dyd, k, k, s, p, outs[0]), (nhwc(x2), nhwc(dy2), 1, 1, 1, 0, outs[1])])
    np.testing.assert_allclose(outs[0].cpu().numpy(), dw.cpu().numpy(), rtol=1e-4, atol=2e-5 * float(np.abs(ref).max()))
    single = o.conv2d_wgrad(nhwc(x2), nhwc(dy2), 1, 1, 1, 0)
    np.testing.assert_allclose(outs[1].cpu().numpy(), single.cpu().numpy(), rtol=1e-4, atol=1e-5)


def test_wgrad_wave_arrangement_ragged_predictor():
    """The head's 256 -> 48 predictor over the four ragged levels (Cout = 48: the 64 x 256 arrangement) against torch."""
    o = ops()
    B, Cc, Co = 2, 256, 48
    sizes = [(64, 104), (32, 52), (16, 26), (8, 13)]
    xs = [cases.randn(271 + i, B, Cc, h, ww) for i, (h, ww) in enumerate(sizes)]
    rows = sum(B * h * ww for h, ww in sizes)
    dys = [cases.randn(281 + i, B, Co, h, ww) / rows ** 0.5 for i, (h, ww) in enumerate(sizes)]
    torch.set_num_threads(8)
    w0 = torch.zeros(Co, Cc, 1, 1, requires_grad=True)
    for xl, dl in zip(xs, dys):
        F.conv2d(q(xl), w0, None, 1, 0).backward(q(dl))
    xr = o.Ragged.from_levels([nhwc(t) for t in xs])
    dyr = o.Ragged.from_levels([nhwc(t) for t in dys])
    dw = o.conv2d_wgrad(xr, dyr, 1, 1, 1, 0)
    assert o.last_kernel() == 'conv_wgrad_kernel' and o.last_wgrad_plan()['shape'] == 1
    got = dw.cpu().permute(0, 3, 1, 2)
    np.testing.assert_allclose(got.numpy(), w0.grad.numpy(), rtol=2e-3, atol=2e-3 * float(w0.grad.abs().max()))


@pytest.mark.parametrize('case', [(1, 16, 16), (3, 48, 32), (2, 40, 60), (2, 128, 232), (5, 64, 64)])
def test_wgrad_c64_patch_kernel(case):
    """conv_wgrad_c64_kernel (3x3, 64 -> 64: one wave per tap over 16 x 16-pixel squares, dY and the input patch loaded
    once per square, per-workgroup partials summed in a fixed order) against torch's weight gradient: a single square,
    more squares than workgroups (5 x 16 = 80... and 2 x 8 x 15 = 240 on <= 256), border squares partly outside the
    image, channel-slice operands (pixel strides 96 / 128), written and accumulated, and run-to-run bit-identical."""
    o = ops()
    B, H, W = case
    x = cases.randn(181, B, 64, H, W)
    dy = cases.randn(182, B, 64, H, W) / (B * H * W) ** 0.5
    torch.set_num_threads(8)
    w0 = torch.zeros(64, 64, 3, 3, requires_grad=True)
    F.conv2d(q(x), w0, None, 1, 1).backward(q(dy))
    ref = w0.grad.permute(0, 2, 3, 1)
    wide_x = torch.zeros(B, H, W, 96, device=DEV, dtype=BF)
    wide_x[..., 16:80] = nhwc(x)
    wide_dy = torch.zeros(B, H, W, 128, device=DEV, dtype=BF)
    wide_dy[..., 64:] = nhwc(dy)
    with o.tuning(**{'conv.c64_mintiles': 1}):
        dw = o.conv2d_wgrad(wide_x[..., 16:80], wide_dy[..., 64:], 3, 3, 1, 1)
        assert o.last_kernel() == 'conv_wgrad_c64_kernel', o.last_kernel()
        np.testing.assert_allclose(dw.cpu().numpy(), ref.numpy(), rtol=2e-3, atol=2e-3 * float(ref.abs().max()))
        acc = dw.clone()
        o.conv2d_wgrad(nhwc(x), nhwc(dy), 3, 3, 1, 1, out=acc, accumulate=True)
        np.testing.assert_allclose(acc.cpu().numpy(), 2 * dw.cpu().numpy(), rtol=1e-5, atol=1e-6)
        assert torch.equal(o.conv2d_wgrad(nhwc(x), nhwc(dy), 3, 3, 1, 1), dw)      # deterministic
    with o.tuning(**{'conv.c64_mintiles': 0}):
        other = o.conv2d_wgrad(nhwc(x), nhwc(dy), 3, 3, 1, 1)
        assert o.last_kernel() == 'conv_wgrad_kernel'
    np.testing.assert_allclose(dw.cpu().numpy(), other.cpu().numpy(), rtol=1e-3, atol=1e-3 * float(ref.abs().max()))


def test_wgrad_ragged_real_split():
    """Head weight gradient over the four ragged levels at B=16 (the ping-pong kernel's row walker crosses image
    planes and level boundaries inside a split)."""
    o = ops()
    B, Cc = 8, 256
    sizes = [(64, 104), (32, 52), (16, 26), (8, 13)]
    xs = [cases.randn(171 + i, B, Cc, h, ww) for i, (h, ww) in enumerate(sizes)]
    rows = sum(B * h * ww for h, ww in sizes)
    dys = [cases.randn(181 + i, B, Cc, h, ww) / rows ** 0.5 for i, (h, ww) in enumerate(sizes)]
    torch.set_num_threads(8)
    w0 = torch.zeros(Cc, Cc, 3, 3, requires_grad=True)
    for xl, dl in zip(xs[1:], dys[1:]):       # CPU reference over levels 1..3 (level 0 is 53k rows x 2304 x 256)
        F.conv2d(q(xl), w0, None, 1, 1).backward(q(dl))
    xr = o.Ragged.from_levels([nhwc(t) for t in xs])
    dyr = o.Ragged.from_levels([nhwc(t) if i else torch.zeros_like(nhwc(t)) for i, t in enumerate(dys)])
    dw = o.conv2d_wgrad(xr, dyr, 3, 3, 1, 1)
    assert o.last_kernel() == 'conv_wgrad_pp_kernel'
    got = dw.cpu().permute(0, 3, 1, 2)
    np.testing.assert_allclose(got.numpy(), w0.grad.numpy(), rtol=2e-3, atol=2e-3 * float(w0.grad.abs().max()))


def test_wgrad_batch_equals_single_launches():
    """das_conv2d_wgrad_batch: heterogeneous ops (both kernel classes, 3x3 / 1x1 / strided / ragged) sharing launches
    give what one launch per op gives (same kernels, a different pixel split: f32 summation order only)."""
    o = ops()
    items, singles = [], []
    specs = [(16, 32, 52, 256, 1024, 1, 1, 0), (16, 32, 52, 1024, 256, 1, 1, 0), (16, 32, 52, 256, 256, 3, 1, 1),
             (8, 64, 104, 128, 512, 1, 1, 0), (16, 16, 26, 512, 512, 3, 1, 1), (4, 64, 104, 128, 128, 3, 2, 1),
             (2, 128, 208, 64, 64, 3, 1, 1), (8, 64, 104, 256, 512, 1, 2, 0), (16, 16, 26, 512, 2048, 1, 1, 0),
             (2, 40, 40, 64, 72, 1, 1, 0)]
    for i, (B, H, W, Cin, Cout, k, s, p) in enumerate(specs):
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        x = torch.randn(B, H, W, Cin, device=DEV, dtype=BF)
        dy = torch.randn(B, Ho, Wo, Cout, device=DEV, dtype=BF) / (B * Ho * Wo) ** 0.5
        base = torch.randn(Cout, k, k, Cin, device=DEV)
        out = base.clone()
        items.append((x, dy, k, k, s, p, out))
        singles.append(base + o.conv2d_wgrad(x, dy, k, k, s, p))
    B, Cc = 4, 256
    sizes = [(32, 52), (16, 26), (8, 13), (4, 7)]
    xr = o.Ragged.from_levels([torch.randn(B, h, w, Cc, device=DEV, dtype=BF) for h, w in sizes])
    dyr = o.Ragged.from_levels([torch.randn(B, h, w, Cc, device=DEV, dtype=BF) / 50 for h, w in sizes])
    base = torch.zeros(Cc, 3, 3, Cc, device=DEV)
    items.append((xr, dyr, 3, 3, 1, 1, base.clone()))
    singles.append(o.conv2d_wgrad(xr, dyr, 3, 3, 1, 1))
    o.conv2d_wgrad_batch(items)
    for it, ref in zip(items, singles):
        got = it[6]
        np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=2e-4, atol=2e-4 * float(ref.abs().max()))
    from das_amd import _lib
    with pytest.raises(_lib.DasHipError):       # two ops adding into one buffer would race in the reduction
        o.conv2d_wgrad_batch([items[0], items[0]])


def test_wgrad_schedule_direct_units():
    """A launch with more tiles than workgroups: the host schedule gives most tiles their whole pixel reduction as ONE
    unit whose accumulators go straight into dW (no workspace, no reduction pass), several units per workgroup.
    Checked against F.conv2d's weight gradient (CPU), written (accumulate = 0) and added (accumulate = 1)."""
    o = ops()
    specs = [(4, 16, 26, 512, 2048, 1, 1, 0), (4, 16, 26, 2048, 512, 1, 1, 0), (4, 16, 26, 512, 512, 3, 1, 1),
             (4, 16, 26, 2048, 2048, 1, 1, 0), (4, 32, 52, 256, 1024, 1, 1, 0), (4, 32, 52, 1024, 256, 1, 1, 0),
             (4, 32, 52, 256, 256, 3, 1, 1), (4, 16, 26, 1024, 2048, 1, 2, 0), (4, 16, 26, 512, 2040, 1, 1, 0)]
    items, refs = [], []
    torch.set_num_threads(8)
    for i, (B, H, W, Cin, Cout, k, s, p) in enumerate(specs):
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        x = cases.randn(301 + i, B, Cin, H, W)
        dy = cases.randn(331 + i, B, Cout, Ho, Wo) / (B * Ho * Wo) ** 0.5
        w0 = torch.zeros(Cout, Cin, k, k, requires_grad=True)
        F.conv2d(q(x), w0, None, s, p).backward(q(dy))
        refs.append(w0.grad.permute(0, 2, 3, 1))
        items.append((nhwc(x), nhwc(dy), k, k, s, p, torch.full((Cout, k, k, Cin), 7.0, device=DEV)))
    with o.tuning(**{'wgrad.pp_blocks': 32}):                    # 197 tiles on 32 workgroups (a step's deferred ops have
        o.conv2d_wgrad_batch(items, accumulate=False)            # that ratio on 256)
        plan = o.last_wgrad_plan()
        assert o.last_kernel() == 'conv_wgrad_pp_kernel' and plan['cls'] == 0 and plan['grid'] == 32
        assert plan['longest'] >= 4 and plan['direct'] >= 0.9 * plan['units'], plan   # (the longest tiles may be cut)
        for it, ref in zip(items, refs):
            np.testing.assert_allclose(it[6].cpu().numpy(), ref.numpy(), rtol=2e-3, atol=2e-3 * float(ref.abs().max()))
        first = [it[6].clone() for it in items]
        o.conv2d_wgrad_batch(items)                              # same shapes: cached schedule, added this time
        for it, f in zip(items, first):
            np.testing.assert_allclose(it[6].cpu().numpy(), 2 * f.cpu().numpy(), rtol=1e-5, atol=1e-6)
    o.conv2d_wgrad_batch(items, accumulate=False)                # full grid: most tiles cut in two, same result
    plan = o.last_wgrad_plan()
    assert plan['grid'] == 256 and plan['partial'] > 0, plan
    for it, ref in zip(items, refs):
        np.testing.assert_allclose(it[6].cpu().numpy(), ref.numpy(), rtol=2e-3, atol=2e-3 * float(ref.abs().max()))


def test_wgrad_schedule_cut_tile_atomic_groups():
    """One small result with a long reduction alone in a launch: the tile is cut into hundreds of runs, the reduction
    pass sums them in several groups that add atomically (into a zeroed dW when accumulate = 0)."""
    o = ops()
    B, H, W, Cin, Cout = 4, 128, 208, 64, 64
    x, dy = cases.randn(361, B, Cin, H, W), cases.randn(362, B, Cout, H, W) / (B * H * W) ** 0.5
    torch.set_num_threads(8)
    w0 = torch.zeros(Cout, Cin, 1, 1, requires_grad=True)
    F.conv2d(q(x), w0, None, 1, 0).backward(q(dy))
    ref = w0.grad.permute(0, 2, 3, 1)
    out = torch.full((Cout, 1, 1, Cin), 3.0, device=DEV)
    o.conv2d_wgrad_batch([(nhwc(x), nhwc(dy), 1, 1, 1, 0, out)], accumulate=False)
    plan = o.last_wgrad_plan()
    assert plan['cls'] == 1 and plan['direct'] == 0 and plan['partial'] >= 64 and plan['groups'] > 1, plan
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=2e-3, atol=2e-3 * float(ref.abs().max()))
    o.conv2d_wgrad_batch([(nhwc(x), nhwc(dy), 1, 1, 1, 0, out)])
    np.testing.assert_allclose(out.cpu().numpy(), 2 * ref.numpy(), rtol=2e-3, atol=4e-3 * float(ref.abs().max()))


def test_tuning_api_rejects_unknown_key():
    from das_amd import _lib
    lib = _lib.load()
    assert lib.das_tuning_set(b'no.such.key', 1) == _lib.DAS_ERR_ARG
    assert lib.das_tuning_reset() == 0


# ---------------------------------------------------------------- persistent grids under a CU reserve
def test_persistent_grids_follow_the_cu_reserve():
    """comm.reserved_cus (the CUs left to a collective's kernels while gradient buckets are in flight): every persistent
    one-workgroup-per-CU grid shrinks with it — weight-gradient plans of both classes, the streaming 1x1 kernel, the
    3x3 c64 kernel — and the results do not change (bit-identical for the convolutions: a tile's arithmetic does not
    depend on which workgroup runs it; the weight gradient regroups f32 partial sums)."""
    o = ops()
    import ctypes as C
    from das_amd import _lib
    lib = _lib.load()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    B, H, W = 2, 64, 104
    x1, w1 = nhwc(cases.randn(301, B, 256, H, W)), o.pack_weight((cases.randn(302, 64, 256, 1, 1) / 16).to(DEV), BF)
    x3, w3 = nhwc(cases.randn(303, B, 64, H, W)), o.pack_weight((cases.randn(304, 64, 64, 3, 3) / 24).to(DEV), BF)
    xg, dyg = nhwc(cases.randn(305, B, 256, 32, 52)), nhwc(cases.randn(306, B, 256, 32, 52))
    xs_, dys_ = nhwc(cases.randn(307, B, 64, 32, 52)), nhwc(cases.randn(308, B, 128, 32, 52))
    out = {}
    for reserve in (0, 40):
        with o.tuning(**{'comm.reserved_cus': reserve, 'conv.stream_minrows': 1024, 'conv.c64_mintiles': 1}):
            y1 = o.conv2d(x1, w1, 1, 1, 1, 0)
            assert o.last_kernel() == 'conv1x1_stream_kernel'
            y3 = o.conv2d(x3, w3, 3, 3, 1, 1)
            assert o.last_kernel() == 'conv3x3_c64_kernel'
            dw_pp = o.conv2d_wgrad(xg, dyg, 3, 3, 1, 1)
            plan_pp = o.last_wgrad_plan()
            dw = o.conv2d_wgrad(xs_, dys_, 1, 1, 1, 0)
            plan = o.last_wgrad_plan()
        assert plan_pp['cls'] == 0 and plan_pp['grid'] == cus - reserve, plan_pp
        assert plan['cls'] == 1 and plan['grid'] == 3 * (cus - reserve), plan
        out[reserve] = (y1, y3, dw_pp, dw)
    assert torch.equal(out[0][0], out[40][0]) and torch.equal(out[0][1], out[40][1])
    for a, b in zip(out[0][2:], out[40][2:]):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-4, atol=2e-3)
    # the measurement aid itself: 8 workgroups hold their CUs for 2 ms on a side stream, the main stream keeps working
    side = torch.cuda.Stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        e0.record()
        _lib.check(lib.das_dev_occupy_cus(8, 256, 64 * 1024, 2000, C.c_void_p(side.cuda_stream)), 'das_dev_occupy_cus')
        e1.record()
    y = o.conv2d(x1, w1, 1, 1, 1, 0)
    torch.cuda.synchronize()
    assert 1.8 <= e0.elapsed_time(e1) <= 4.0 and torch.equal(y, out[0][0])
    assert lib.das_dev_occupy_cus(0, 256, 4096, 10, None) == _lib.DAS_ERR_ARG


def test_wgrad_pp_share_is_per_thread_and_follows_the_usable_cus():
    """das_wgrad_pp_share (round 6): the calling thread's share of the usable CUs for the ping-pong weight-gradient
    grid — what das_amd.autograd._on_side sets around side-stream launches instead of flipping the process-global key
    wgrad.pp_blocks. Half the usable CUs with share 2 (also under a CU reserve), the whole chip again with share 1, another
    thread never sees it, an explicit wgrad.pp_blocks wins; the gradient is the same up to the regrouping of f32 partial sums."""
    import threading
    from das_amd import _lib
    o = ops()
    lib = _lib.load()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    xg, dyg = nhwc(cases.randn(311, 2, 256, 32, 52)), nhwc(cases.randn(312, 2, 256, 32, 52))

    def grid():
        dw = o.conv2d_wgrad(xg, dyg, 3, 3, 1, 1)
        plan = o.last_wgrad_plan()
        assert plan['cls'] == 0, plan
        return plan['grid'], dw
    g1, dw1 = grid()
    assert g1 == cus
    assert lib.das_wgrad_pp_share(0) != 0 and lib.das_wgrad_pp_share(17) != 0      # (DAS_ERR_ARG: 1 .. 16)
    assert lib.das_wgrad_pp_share(2) == 0
    try:
        g2, dw2 = grid()
        assert g2 == cus // 2
        seen = []
        t = threading.Thread(target=lambda: seen.append(grid()[0]))
        t.start()
        t.join()
        assert seen == [cus]                      # another thread: the whole chip
        with o.tuning(**{'comm.reserved_cus': 40}):
            assert grid()[0] == (cus - 40) // 2
        with o.tuning(**{'wgrad.pp_blocks': 96}):
            assert grid()[0] == 96                # the explicit key wins over the share
    finally:
        assert lib.das_wgrad_pp_share(1) == 0
    assert grid()[0] == cus
    np.testing.assert_allclose(dw2.cpu().numpy(), dw1.cpu().numpy(), rtol=2e-3, atol=2e-3)


# (B, H, W, Cin, Cout): the step's long-K reduce convs and shapes whose rows / channels do not fill the last tile / column block
KSTREAM = [(2, 64, 104, 512, 128), (2, 32, 52, 1024, 256), (1, 37, 29, 512, 256), (3, 16, 26, 1024, 128), (1, 21, 17, 1024, 384)]


@pytest.mark.parametrize('case', KSTREAM)
def test_kstream_weight_stationary_1x1_forward_and_statistics(case):
    """conv1x1_kstream_kernel (round 6, tuning key conv.kstream): K = 512 / 1024 split over two wave groups, weights in
    registers, 16-row pixel tiles. Plain output against F.conv2d (bf16-exact up to the regrouped f32 sum of the two K halves),
    the BatchNorm statistics against the sums of the kernel's own stored output, and against the tile kernel it replaces."""
    o = ops()
    B, H, W, Cin, Cout = case
    x = cases.randn(401, B, Cin, H, W)
    w = cases.randn(402, Cout, Cin, 1, 1) / Cin ** 0.5
    xd, wd = nhwc(x), o.pack_weight(w.to(DEV), BF)
    with o.tuning(**{'conv.kstream': 0}):
        y_tile = o.conv2d(xd, wd, 1, 1, 1, 0)
        assert o.last_kernel() != 'conv1x1_kstream_kernel'
    with o.tuning(**{'conv.kstream': 31, 'conv.stream_minrows': 64}):
        stats = torch.zeros(2 * 2 * Cout, device=DEV)      # two slots
        y = o.conv2d(xd, wd, 1, 1, 1, 0, stats=stats)
        assert o.last_kernel() == 'conv1x1_kstream_kernel', o.last_kernel()
        y2 = o.conv2d(xd, wd, 1, 1, 1, 0)
        assert o.last_kernel() == 'conv1x1_kstream_kernel'
    assert torch.equal(y, y2)
    ref = conv_ref(x, w, 1, 0)
    np.testing.assert_allclose(nchw(y).numpy(), ref.numpy(), **TOL)
    assert_bf16_exact(nchw(y), ref)
    np.testing.assert_allclose(nchw(y).numpy(), nchw(y_tile).numpy(), rtol=8e-3, atol=2e-3)
    yq = nchw(y)
    n = B * H * W
    s_ref = torch.cat([yq.sum((0, 2, 3)), (yq ** 2).sum((0, 2, 3))])
    np.testing.assert_allclose(stats.view(2, -1).sum(0).cpu().numpy() / n, s_ref.numpy() / n, rtol=1e-3, atol=1e-3)
    # MODE 2: plain output + a residual (a data gradient with a second gradient of the same tensor)
    res = cases.randn(403, B, Cout, H, W)
    with o.tuning(**{'conv.kstream': 511, 'conv.stream_minrows': 64}):
        yr = o.conv2d(xd, wd, 1, 1, 1, 0, residual=nhwc(res))
        assert o.last_kernel() == 'conv1x1_kstream_kernel', o.last_kernel()
    # (the sum of two bf16 values, rounded once: exact given the conv's stored value)
    assert torch.equal(yr, (y.float() + nhwc(res).float()).to(BF))
