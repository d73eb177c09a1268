"""Tensor-level wrappers over the C ABI of libdas_hip.so.

PyTorch is plumbing here: it owns device memory and the stream; every op below hands raw
device pointers + sizes to a HIP kernel. Activations are NHWC tensors of shape (B, H, W, C),
bf16 or f32. Ops raise (never fall back) when a tensor is not on a GPU.
"""
import ctypes as C

import torch

from . import _lib

_DT = {torch.float32: _lib.DAS_F32, torch.bfloat16: _lib.DAS_BF16}

# bench.py sets this to a list to time every conv launch with HIP events recorded on the launch
# stream: entries are (kernel family tag, algorithmic FLOPs, start event, end event).
PROFILE = None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.DasHipError('das_amd ops run on the GPU only (no CPU fallback); got a CPU tensor')


def _ps(t):
    """pixel stride (elements) of an NHWC tensor or a channel-slice view of one"""
    assert t.shape[-1] == 1 or t.stride(-1) == 1, 'channel dim must be contiguous'
    ps = t.stride(-2)
    if t.dim() == 4:
        assert t.stride(1) == ps * t.shape[2] and t.stride(0) == ps * t.shape[1] * t.shape[2], 'rows must be dense'
    return ps


def pack_weight(w, dtype, cin_pad=None, cout_pad=None):
    """OIHW f32 parameter -> (Cout_pad, KH, KW, Cin_pad) K-contiguous tensor in `dtype`."""
    O, I, KH, KW = w.shape
    cin_pad = cin_pad or (I + 7) // 8 * 8
    cout_pad = cout_pad or (O + 7) // 8 * 8
    p = torch.zeros(cout_pad, KH, KW, cin_pad, dtype=dtype, device=w.device)
    p[:O, :, :, :I] = w.detach().permute(0, 2, 3, 1).to(dtype)
    return p


def conv2d(x, w, KH, KW, stride=1, pad=0, scale=None, shift=None, residual=None, relu=False, relu_in=False,
           out_dtype=None, stats=None, out=None):
    """x (B,H,W,Cin[view]) ; w packed (Cout,KH,KW,Cin). Returns y (B,Ho,Wo,Cout)."""
    _need_gpu(x, w)
    lib = _lib.load()
    B, H, W, Cin = x.shape
    Cout = w.shape[0]
    assert w.shape[1:] == (KH, KW, Cin), (w.shape, (KH, KW, Cin))
    assert w.dtype == x.dtype and w.is_contiguous()
    Ho = (H + 2 * pad - KH) // stride + 1
    Wo = (W + 2 * pad - KW) // stride + 1
    out_dtype = out_dtype or x.dtype
    if out is None:
        out = torch.empty(B, Ho, Wo, Cout, dtype=out_dtype, device=x.device)
    assert out.shape == (B, Ho, Wo, Cout) and out.dtype == out_dtype
    for t in (scale, shift):
        assert t is None or (t.dtype == torch.float32 and t.numel() == Cout and t.is_contiguous())
    d = _lib.DasConvDesc(
        dtype=_DT[x.dtype], out_dtype=_DT[out_dtype], B=B, H=H, W=W, Cin=Cin, x_pix_stride=_ps(x), Ho=Ho, Wo=Wo,
        Cout=Cout, y_pix_stride=_ps(out), KH=KH, KW=KW, stride=stride, pad=pad, relu_in=int(relu_in), relu=int(relu),
        scale=scale.data_ptr() if scale is not None else None, shift=shift.data_ptr() if shift is not None else None,
        residual=residual.data_ptr() if residual is not None else None,
        res_pix_stride=_ps(residual) if residual is not None else 0,
        stats=stats.data_ptr() if stats is not None else None)
    if residual is not None:
        assert residual.shape == out.shape and residual.dtype == out.dtype
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(lib.das_conv2d_nhwc(_ptr(x), _ptr(w), _ptr(out), C.byref(d), _stream()), 'das_conv2d_nhwc')
    if PROFILE is not None:
        e1.record()
        bn = 128 if Cout > 64 else (64 if Cout > 32 else 32)
        tag = f'conv_igemm<{str(x.dtype)[6:]},{str(out_dtype)[6:]},BN{bn}>'
        PROFILE.append((tag, 2.0 * B * Ho * Wo * Cout * KH * KW * Cin, e0, e1, (B, H, W, Cin, Cout, KH, stride, _ps(x))))
    return out


def pack_image(img, dtype, cpad=8):
    """NCHW f32 -> NHWC dtype with channels zero-padded to cpad."""
    _need_gpu(img)
    B, Cc, H, W = img.shape
    img = img.contiguous().float()
    y = torch.empty(B, H, W, cpad, dtype=dtype, device=img.device)
    _lib.check(_lib.load().das_pack_nchw_to_nhwc(_ptr(img), _ptr(y), _DT[dtype], B, Cc, H, W, cpad, _stream()),
               'das_pack_nchw_to_nhwc')
    return y


def to_nchw_f32(x, c0=0, Cn=None):
    """NHWC (slice) -> dense NCHW f32."""
    _need_gpu(x)
    B, H, W, Cc = x.shape
    Cn = Cn or Cc - c0
    y = torch.empty(B, Cn, H, W, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().das_unpack_nhwc_to_nchw(_ptr(x), _ptr(y), _DT[x.dtype], B, Cn, H, W, _ps(x), c0, _stream()),
               'das_unpack_nhwc_to_nchw')
    return y


def maxpool3x3s2(x):
    _need_gpu(x)
    B, H, W, Cc = x.shape
    assert x.is_contiguous()
    y = torch.empty(B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Cc, dtype=x.dtype, device=x.device)
    _lib.check(_lib.load().das_maxpool3x3s2(_ptr(x), _ptr(y), _DT[x.dtype], B, H, W, Cc, _stream()), 'das_maxpool3x3s2')
    return y


def upsample_bilinear_ac(x, Ho, Wo):
    _need_gpu(x)
    B, H, W, Cc = x.shape
    assert x.is_contiguous()
    y = torch.empty(B, Ho, Wo, Cc, dtype=x.dtype, device=x.device)
    _lib.check(_lib.load().das_upsample_bilinear_ac(_ptr(x), _ptr(y), _DT[x.dtype], B, H, W, Cc, Ho, Wo, _stream()),
               'das_upsample_bilinear_ac')
    return y


def add_upsample_nearest(a, b):
    _need_gpu(a, b)
    B, H, W, Cc = a.shape
    assert a.is_contiguous() and b.is_contiguous() and a.dtype == b.dtype
    y = torch.empty_like(a)
    _lib.check(_lib.load().das_add_upsample_nearest(_ptr(a), _ptr(b), _ptr(y), _DT[a.dtype], B, H, W, Cc, b.shape[1],
                                                    b.shape[2], _stream()), 'das_add_upsample_nearest')
    return y


def add3(a, b, c=None, relu=False):
    _need_gpu(a, b, c)
    assert a.is_contiguous() and b.is_contiguous() and (c is None or c.is_contiguous())
    y = torch.empty_like(a)
    _lib.check(_lib.load().das_add3(_ptr(a), _ptr(b), _ptr(c), _ptr(y), _DT[a.dtype], a.numel(), int(relu), _stream()),
               'das_add3')
    return y


def bn_train_apply(x, stats, gamma, beta, running_mean, running_var, momentum=0.1, eps=1e-5, residual=None,
                   relu=False):
    """x (B,H,W,C) raw conv output, stats f32[2C] from the conv epilogue. Returns y, mean, invstd."""
    _need_gpu(x, stats)
    assert x.is_contiguous()
    Cc = x.shape[-1]
    count = x.numel() // Cc
    y = torch.empty_like(x)
    mean = torch.empty(Cc, dtype=torch.float32, device=x.device)
    invstd = torch.empty_like(mean)
    _lib.check(_lib.load().das_bn_train_apply(_ptr(x), _ptr(y), _DT[x.dtype], count, Cc, _ptr(stats), _ptr(gamma),
                                              _ptr(beta), _ptr(running_mean), _ptr(running_var), momentum, eps,
                                              _ptr(residual), int(relu), _ptr(mean), _ptr(invstd), _stream()),
               'das_bn_train_apply')
    return y, mean, invstd


def groupnorm(x, gamma, beta, G, eps=1e-5, relu=True, out=None):
    """In/out NHWC (may be channel-slice views with a pixel stride)."""
    _need_gpu(x)
    B, H, W, Cc = x.shape
    out = x if out is None else out
    ws = torch.empty(B * G * 2, dtype=torch.float32, device=x.device)
    assert _ps(out) == _ps(x)
    _lib.check(_lib.load().das_groupnorm_nhwc(_ptr(x), _ptr(out), _DT[x.dtype], B, H * W, Cc, _ps(x), G, _ptr(gamma),
                                              _ptr(beta), eps, int(relu), _ptr(ws), _stream()), 'das_groupnorm_nhwc')
    return out


def deform_im2col3x3(x, om):
    """x (B,H,W,C[view]); om (B,H,W,>=27) f32 -> col (B,H,W,9*C)."""
    _need_gpu(x, om)
    B, H, W, Cc = x.shape
    assert om.dtype == torch.float32
    col = torch.empty(B, H, W, 9 * Cc, dtype=x.dtype, device=x.device)
    _lib.check(_lib.load().das_deform_im2col3x3(_ptr(x), _ptr(om), _ptr(col), _DT[x.dtype], B, H, W, Cc, _ps(x),
                                                _ps(om), _stream()), 'das_deform_im2col3x3')
    return col


def offset_sample(uvd, samp_off, conf, J, heads=4):
    _need_gpu(uvd, samp_off, conf)
    B, H, W, _ = uvd.shape
    out = torch.empty(B, H, W, 3 * J, dtype=torch.float32, device=uvd.device)
    _lib.check(_lib.load().das_offset_sample(_ptr(uvd), _ptr(samp_off), _ptr(conf), _ptr(out), B, H, W, J, heads,
                                             _ps(uvd), _ps(samp_off), _ps(conf), _ps(out), _stream()),
               'das_offset_sample')
    return out


def sigmoid_blend(off, w, nxt):
    _need_gpu(off, w, nxt)
    Cc = off.shape[-1]
    out = torch.empty(*off.shape[:-1], Cc, dtype=torch.float32, device=off.device)
    npix = off.numel() // Cc if off.is_contiguous() else off.shape[0] * off.shape[1] * off.shape[2]
    _lib.check(_lib.load().das_sigmoid_blend(_ptr(off), _ptr(w), _ptr(nxt), _ptr(out), npix, Cc, _ps(off), _ps(w),
                                             _ps(nxt), _ps(out), _stream()), 'das_sigmoid_blend')
    return out


def head_assemble(raw, J, root_idx, off_c, depth_c, uvd_c, sigma_c, scales):
    """raw (B,H,W,raw_ps) f32 -> pose_pred (B,H,W,3+6J), uvd (B,H,W,3J)."""
    _need_gpu(raw)
    B, H, W, _ = raw.shape
    pose = torch.empty(B, H, W, 3 + 6 * J, dtype=torch.float32, device=raw.device)
    uvd = torch.empty(B, H, W, 3 * J, dtype=torch.float32, device=raw.device)
    d = _lib.DasHeadAssembleDesc(J=J, root_idx=root_idx, raw_ps=_ps(raw), off_c=off_c, depth_c=depth_c, uvd_c=uvd_c,
                                 sigma_c=sigma_c, scale_off=scales[0], scale_depth=scales[1], scale_uv=scales[2],
                                 scale_d=scales[3])
    _lib.check(_lib.load().das_head_assemble(_ptr(raw), _ptr(pose), _ptr(uvd), B * H * W, C.byref(d), _stream()),
               'das_head_assemble')
    return pose, uvd


def head_finalize(pose, ref, J, root_idx, stride, z_norm, depth_factor, eval_mode):
    _need_gpu(pose, ref)
    npix = pose.numel() // pose.shape[-1]
    _lib.check(_lib.load().das_head_finalize(_ptr(pose), _ptr(ref), npix, J, root_idx, _ps(ref), float(stride),
                                             float(z_norm), float(depth_factor), int(eval_mode), _stream()),
               'das_head_finalize')
    return pose, ref


def decode(cls_list, ctr_list, pose_list, strides, scale_factors, J, nms_pre, nms_post, score_thr, nms_thr):
    """Eval-mode head outputs per level, NHWC f32: cls/ctr (B,H,W,c>=1) logits in channel 0,
    pose (B,H,W,>=3+3J). scale_factors: (B,2) f32 device tensor. Returns dict of device tensors:
    count (B,), scores (B,nms_post), poses (B,nms_post,J,3), centers (B,nms_post,3), index (B,nms_post)."""
    _need_gpu(*cls_list, *ctr_list, *pose_list, scale_factors)
    lib = _lib.load()
    B = cls_list[0].shape[0]
    L = len(cls_list)
    d = _lib.DasDecodeDesc(B=B, J=J, num_levels=L, nms_pre=nms_pre, nms_post=nms_post, score_thr=score_thr,
                           nms_thr=nms_thr, scale_factor=scale_factors.data_ptr())
    for l in range(L):
        d.H[l], d.W[l], d.stride[l] = cls_list[l].shape[1], cls_list[l].shape[2], strides[l]
        d.cls[l], d.ctr[l], d.pose[l] = cls_list[l].data_ptr(), ctr_list[l].data_ptr(), pose_list[l].data_ptr()
        d.cls_ps[l], d.ctr_ps[l], d.pose_ps[l] = _ps(cls_list[l]), _ps(ctr_list[l]), _ps(pose_list[l])
        for t in (cls_list[l], ctr_list[l], pose_list[l]):
            assert t.dtype == torch.float32
    assert scale_factors.dtype == torch.float32 and scale_factors.shape == (B, 2) and scale_factors.is_contiguous()
    cap = lib.das_decode_cap(C.byref(d))
    dev = cls_list[0].device
    ws = torch.empty(lib.das_decode_ws_bytes(B, cap, J), dtype=torch.uint8, device=dev)
    out = dict(count=torch.zeros(B, dtype=torch.int32, device=dev),
               scores=torch.zeros(B, nms_post, dtype=torch.float32, device=dev),
               poses=torch.zeros(B, nms_post, J, 3, dtype=torch.float32, device=dev),
               centers=torch.zeros(B, nms_post, 3, dtype=torch.float32, device=dev),
               index=torch.zeros(B, nms_post, dtype=torch.int32, device=dev))
    _lib.check(lib.das_decode(C.byref(d), _ptr(out['scores']), _ptr(out['poses']), _ptr(out['centers']),
                              _ptr(out['index']), _ptr(out['count']), _ptr(ws), _stream()), 'das_decode')
    return out
