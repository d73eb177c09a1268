"""Oracle building blocks (test infrastructure only — see oracle/__init__.py).

Restates the third-party layer arithmetic the reference path calls into
(mmcv-full 1.3.10 / mmdet 2.14.0, source absent from /root/reference):

* ``ConvModule``  = conv -> norm -> ReLU, ``bias='auto'`` means "bias iff no norm"
  (call sites: mspn_mmpose.py:254,327,340,352,361,372,546;
  anchor_free_mono3d_pose_head.py:116,136,157; das_head.py:112;
  recursive_update.py:177,243)
* ``ModulatedDeformConv2dPack`` (DCNv2) (das_head.py:107-108,
  anchor_free_mono3d_pose_head.py:111-112,131-132, recursive_update.py:177-178)
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
GN_EPS = 1e-5


def batch_norm(sd, p, x, train):
    """torch BatchNorm2d semantics; in train mode updates sd's running stats in place."""
    rm, rv = sd[p + '.running_mean'], sd[p + '.running_var']
    y = F.batch_norm(x, rm, rv, sd[p + '.weight'], sd[p + '.bias'], train, BN_MOMENTUM, BN_EPS)
    if train and (p + '.num_batches_tracked') in sd:
        sd[p + '.num_batches_tracked'] += 1
    return y


def conv_bn(sd, p, x, stride=1, padding=0, relu=False, train=False):
    """ConvModule(conv, BN, optional ReLU); the conv has no bias because a norm follows."""
    y = F.conv2d(x, sd[p + '.conv.weight'], None, stride, padding)
    y = batch_norm(sd, p + '.bn', y, train)
    return F.relu(y) if relu else y


def group_norm(sd, p, x, groups=32):
    return F.group_norm(x, groups, sd[p + '.weight'], sd[p + '.bias'], GN_EPS)


def bilinear_zero(x, py, px):
    """Sample x (B,C,H,W) at pixel coordinates (py,px) (B,Ho,Wo); corners outside the
    map contribute 0 (the DCNv2 `dmcn_im2col_bilinear` rule, corner order LL,LH,HL,HH)."""
    B, C, H, W = x.shape
    y0 = torch.floor(py)
    x0 = torch.floor(px)
    y1 = y0 + 1
    x1 = x0 + 1
    ly = py - y0
    lx = px - x0
    hy = 1 - ly
    hx = 1 - lx
    xf = x.reshape(B, C, H * W)
    shape = py.shape[1:]

    # mmcv's kernels gate the WHOLE sample (`if (h_im > -1 && w_im > -1 && h_im < height && w_im < width)` in
    # modulated_deformable_im2col / col2im / col2im_coord). The value is unchanged by the gate (the corner
    # weights vanish there), but the gradient at exactly py == -1 or px == -1 is zero, not the one-sided
    # derivative — which matters because zero-initialised offset convs put every border tap exactly there.
    gate = (py > -1) & (px > -1) & (py < H) & (px < W)

    def corner(yy, xx, wgt):
        valid = (yy >= 0) & (yy <= H - 1) & (xx >= 0) & (xx <= W - 1) & gate
        idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).long().reshape(B, 1, -1).expand(B, C, -1)
        v = torch.gather(xf, 2, idx).reshape(B, C, *shape)
        return v * (wgt * valid.to(x.dtype))[:, None]

    return corner(y0, x0, hy * hx) + corner(y0, x1, hy * lx) + corner(y1, x0, ly * hx) + corner(y1, x1, ly * lx)


def modulated_deform_conv2d(x, offset, mask, weight, bias, stride=1, padding=1, dilation=1):
    """DCNv2, deform_groups=1, groups=1.

    offset (B, 2*kh*kw, Ho, Wo): channel 2k = dy, 2k+1 = dx of tap k = i*kw + j.
    mask   (B,   kh*kw, Ho, Wo): already sigmoid-ed modulation.
    out[b,o,y,x] = bias[o] + sum_{c,k} weight[o,c,k] * mask[b,k,y,x] *
                   bilinear_zero(x[b,c], y*s - p + i*d + dy, x*s - p + j*d + dx)
    """
    B, C, H, W = x.shape
    O, _, kh, kw = weight.shape
    Ho = (H + 2 * padding - dilation * (kh - 1) - 1) // stride + 1
    Wo = (W + 2 * padding - dilation * (kw - 1) - 1) // stride + 1
    ys = torch.arange(Ho, dtype=x.dtype, device=x.device) * stride - padding
    xs = torch.arange(Wo, dtype=x.dtype, device=x.device) * stride - padding
    cols = []
    for i in range(kh):
        for j in range(kw):
            k = i * kw + j
            py = ys[None, :, None] + i * dilation + offset[:, 2 * k]
            px = xs[None, None, :] + j * dilation + offset[:, 2 * k + 1]
            cols.append(bilinear_zero(x, py, px) * mask[:, k:k + 1])
    col = torch.stack(cols, 2)  # (B,C,K,Ho,Wo)
    out = torch.einsum('ock,bckyx->boyx', weight.reshape(O, C, kh * kw), col)
    if bias is not None:
        out = out + bias[None, :, None, None]
    return out


def dcn_pack(sd, p, x, stride=1, padding=1):
    """ModulatedDeformConv2dPack.forward: conv_offset -> chunk(3) -> offset=cat(o1,o2),
    mask=sigmoid(o3) -> modulated deform conv."""
    out = F.conv2d(x, sd[p + '.conv_offset.weight'], sd[p + '.conv_offset.bias'], stride, padding)
    o1, o2, m = torch.chunk(out, 3, dim=1)
    offset = torch.cat((o1, o2), dim=1)
    mask = torch.sigmoid(m)
    return modulated_deform_conv2d(x, offset, mask, sd[p + '.weight'], sd.get(p + '.bias'), stride, padding)


def conv_gn_relu(sd, p, x, padding=1, dcn=False, groups=32):
    """ConvModule(conv|DCNv2, GN(32), ReLU). Bias present iff the key exists."""
    if dcn:
        y = dcn_pack(sd, p + '.conv', x, 1, padding)
    else:
        y = F.conv2d(x, sd[p + '.conv.weight'], sd.get(p + '.conv.bias'), 1, padding)
    return F.relu(group_norm(sd, p + '.gn', y, groups))
