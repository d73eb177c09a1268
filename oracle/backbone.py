"""Oracle: MSPN2 backbone + FPN-as-configured (test infrastructure only).

Follows /root/reference/mmdet3d/models/backbones/mspn_mmpose.py:
  ResNetTop :533-556, _Bottleneck.forward :126-157, DownsampleModule.forward :278-289,
  UpsampleUnit.forward :381-404, UpsampleModule.forward :458-477, MSPN2.forward :657-667.
FPN follows mmdet 2.14.0 `FPN.forward` (source absent; published algorithm) with the
merged config of configs/_base_/models/das.py:16-23 + configs/das/exp_panoptic.py:24-30.
"""
import torch
import torch.nn.functional as F

from .nn_ops import batch_norm, conv_bn

LAYER_STRIDES = (1, 2, 2, 2)


def bottleneck(sd, p, x, stride, train):
    """_Bottleneck 'pytorch' style: the stride sits on the 3x3 (mspn_mmpose.py:67-69)."""
    # conv/bn are stored flat as conv1/bn1 ... (not ConvModule), so spell them out:
    out = F.conv2d(x, sd[p + '.conv1.weight'])
    out = _bn(sd, p + '.bn1', out, train, relu=True)
    out = F.conv2d(out, sd[p + '.conv2.weight'], None, stride, 1)
    out = _bn(sd, p + '.bn2', out, train, relu=True)
    out = F.conv2d(out, sd[p + '.conv3.weight'])
    out = _bn(sd, p + '.bn3', out, train, relu=False)
    if (p + '.downsample.conv.weight') in sd:
        identity = conv_bn(sd, p + '.downsample', x, stride, 0, False, train)
    else:
        identity = x
    return F.relu(out + identity)


def _bn(sd, p, x, train, relu):
    y = batch_norm(sd, p, x, train)
    return F.relu(y) if relu else y


def downsample_module(sd, p, x, skip1, skip2, num_blocks, train):
    outs = []
    for i, nb in enumerate(num_blocks):
        for b in range(nb):
            x = bottleneck(sd, f'{p}.layer{i + 1}.{b}', x, LAYER_STRIDES[i] if b == 0 else 1, train)
        if skip1 is not None:
            x = x + skip1[i] + skip2[i]
        outs.append(x)
    return outs[::-1]  # coarse -> fine


def upsample_unit(sd, p, x, up_x, train):
    out = conv_bn(sd, p + '.in_skip', x, 1, 0, False, train)
    if up_x is not None:
        up = F.interpolate(up_x, size=x.shape[-2:], mode='bilinear', align_corners=True)
        out = out + conv_bn(sd, p + '.up_conv', up, 1, 0, False, train)
    out = F.relu(out)
    skip1 = skip2 = cross = None
    if (p + '.out_skip1.conv.weight') in sd:
        skip1 = conv_bn(sd, p + '.out_skip1', x, 1, 0, True, train)
        skip2 = conv_bn(sd, p + '.out_skip2', out, 1, 0, True, train)
    if (p + '.cross_conv.conv.weight') in sd:
        cross = conv_bn(sd, p + '.cross_conv', out, 1, 0, True, train)
    return out, skip1, skip2, cross


def upsample_module(sd, p, mids, train):
    outs, s1, s2, cross = [], [], [], None
    for i, m in enumerate(mids):
        o, a, b, c = upsample_unit(sd, f'{p}.up{i + 1}', m, outs[i - 1] if i > 0 else None, train)
        outs.append(o)
        s1.append(a)
        s2.append(b)
        if c is not None:
            cross = c
    return outs, s1[::-1], s2[::-1], cross


def mspn2_forward(sd, img, num_stages, num_blocks=(3, 4, 6, 3), prefix='', train=False):
    """Returns the last stage's 4 maps, fine -> coarse (strides 4, 8, 16, 32)."""
    p = prefix
    x = conv_bn(sd, p + 'top.top.0', img, 2, 3, True, train)
    x = F.max_pool2d(x, 3, 2, 1)
    skip1 = skip2 = None
    outs = None
    for s in range(num_stages):
        sp = f'{p}multi_stage_mspn.{s}'
        mids = downsample_module(sd, sp + '.downsample', x, skip1, skip2, num_blocks, train)
        outs, skip1, skip2, x = upsample_module(sd, sp + '.upsample', mids, train)
        if skip1[0] is None:
            skip1 = skip2 = None
    return outs[::-1]


def fpn_forward(sd, feats, prefix='', start_level=1, num_outs=4, train=False):
    """mmdet FPN with add_extra_convs='on_output', relu_before_extra_convs=True, BN, no act.

    laterals 1x1+BN on levels start_level..; top-down `+= nearest_up(size=finer)`;
    3x3+BN per level; the first extra level = 3x3 stride-2 +BN on the last output (no ReLU),
    further extra levels on relu(previous extra output) — mmdet 2.14.0 `FPN.forward` part 2.
    """
    p = prefix
    n = len(feats) - start_level
    lats = [conv_bn(sd, f'{p}lateral_convs.{i}', feats[i + start_level], 1, 0, False, train) for i in range(n)]
    for i in range(n - 1, 0, -1):
        lats[i - 1] = lats[i - 1] + F.interpolate(lats[i], size=lats[i - 1].shape[-2:], mode='nearest')
    outs = [conv_bn(sd, f'{p}fpn_convs.{i}', lats[i], 1, 1, False, train) for i in range(n)]
    for i in range(n, num_outs):
        src = outs[-1] if i == n else F.relu(outs[-1])
        outs.append(conv_bn(sd, f'{p}fpn_convs.{i}', src, 2, 1, False, train))
    return outs
