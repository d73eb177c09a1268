"""MSPN2 backbone (reference: mmdet3d/models/backbones/mspn_mmpose.py).

Same constructor arguments, module tree and state-dict keys as the reference class
(`top.top.0.{conv,bn}`, `multi_stage_mspn.{s}.downsample.layer{L}.{b}.{conv1,bn1,...}`,
`multi_stage_mspn.{s}.upsample.up{u}.{in_skip,up_conv,out_skip1,out_skip2,cross_conv}.{conv,bn}`);
every layer runs as an NHWC HIP kernel. Reference quirks kept on purpose:
  * blocks >= 1 of every layer ignore `norm_cfg` and use plain BN (mspn_mmpose.py:273-274);
  * `ResNetTop` always emits 64 channels and `cross_conv` always emits 64 (`:456,:615`);
  * `frozen_stages` / `norm_eval` are accepted and ignored (`:600`, `_frozen_stage` never called).
"""
import copy
import os

import torch
import torch.nn as nn

from . import nn as nnops
from . import ops
from .nn import ConvModule, as_nhwc, conv_bn, to_nchw_view
from .registry import BACKBONES


# training: a layer's Bottlenecks run as one autograd node whose backward folds the BatchNorm-backward reductions into
# the data-gradient convs (autograd.BottleneckChainFn); False = one node per conv+BN unit (the tests compare both)
FUSED_LAYER_BACKWARD = True


class Bottleneck(nn.Module):
    """ResNet bottleneck, 'pytorch' style (stride on the 3x3), expansion 4 (mspn_mmpose.py:17-157,196-210)."""
    expansion = 4

    def __init__(self, in_channels, out_channels, stride=1, downsample=None, norm_cfg=dict(type='BN')):
        super().__init__()
        mid = out_channels
        out = out_channels * self.expansion
        self.conv1 = nn.Conv2d(in_channels, mid, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(mid)
        self.conv2 = nn.Conv2d(mid, mid, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(mid)
        self.conv3 = nn.Conv2d(mid, out, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(out)
        for bn in (self.bn1, self.bn2, self.bn3):
            bn._das_sync = norm_cfg.get('type') == 'SyncBN'
        self.downsample = downsample

    def forward(self, x):
        if self.downsample is None:
            # identity path routed through conv1's autograd node: its gradient is added in conv1's
            # data-gradient epilogue instead of by a separate elementwise kernel
            out, identity = conv_bn(x, self.conv1, self.bn1, relu=True, skip_through=True)
        else:
            out = conv_bn(x, self.conv1, self.bn1, relu=True)
            identity = self.downsample(x)
        out = conv_bn(out, self.conv2, self.bn2, relu=True)
        # relu(bn3(conv3(out)) + identity): residual add and ReLU ride in the conv/BN epilogue
        return conv_bn(out, self.conv3, self.bn3, relu=True, residual=identity)


class DownsampleModule(nn.Module):
    def __init__(self, num_blocks, num_units=4, has_skip=False, norm_cfg=dict(type='BN'), in_channels=64):
        super().__init__()
        assert len(num_blocks) == num_units
        self.has_skip, self.num_units, self.norm_cfg = has_skip, num_units, copy.deepcopy(norm_cfg)
        self.in_channels = in_channels
        self.layer1 = self._make_layer(in_channels, num_blocks[0])
        for i in range(1, num_units):
            self.add_module(f'layer{i + 1}', self._make_layer(in_channels * 2 ** i, num_blocks[i], stride=2))

    def _make_layer(self, out_channels, blocks, stride=1):
        downsample = None
        if stride != 1 or self.in_channels != out_channels * Bottleneck.expansion:
            downsample = ConvModule(self.in_channels, out_channels * Bottleneck.expansion, 1, stride=stride, padding=0,
                                    norm_cfg=self.norm_cfg, act_cfg=None)
        units = [Bottleneck(self.in_channels, out_channels, stride=stride, downsample=downsample,
                            norm_cfg=self.norm_cfg)]
        self.in_channels = out_channels * Bottleneck.expansion
        for _ in range(1, blocks):
            units.append(Bottleneck(self.in_channels, out_channels))  # default BN: reference quirk
        return nn.Sequential(*units)

    def forward(self, x, skip1, skip2):
        out = []
        from . import autograd as ag
        for i in range(self.num_units):
            layer = getattr(self, f'layer{i + 1}')
            # x (for i > 0) also feeds the upsample module: it is handed THROUGH the layer's autograd node, so that
            # the gradient of those consumers arrives there and is added in a data-gradient epilogue (no separate add)
            y = ag.bottleneck_chain(x, layer, skip_through=i > 0) if FUSED_LAYER_BACKWARD else None
            if y is None:
                x = layer(x)
            elif i > 0:
                x, out[-1] = y
            else:
                x = y
            if self.has_skip:
                x = nnops.skip_add(x, skip1[i], skip2[i])
            out.append(x)
        out.reverse()
        return tuple(out)


class UpsampleUnit(nn.Module):
    def __init__(self, ind, num_units, in_channels, unit_channels=256, gen_skip=False, gen_cross_conv=False,
                 norm_cfg=dict(type='BN'), out_channels=64):
        super().__init__()
        self.ind, self.num_units = ind, num_units
        self.in_skip = ConvModule(in_channels, unit_channels, 1, norm_cfg=norm_cfg, act_cfg=None)
        if ind > 0:
            self.up_conv = ConvModule(unit_channels, unit_channels, 1, norm_cfg=norm_cfg, act_cfg=None)
        self.gen_skip = gen_skip
        if gen_skip:
            self.out_skip1 = ConvModule(in_channels, in_channels, 1, norm_cfg=norm_cfg)
            self.out_skip2 = ConvModule(unit_channels, in_channels, 1, norm_cfg=norm_cfg)
        self.gen_cross_conv = gen_cross_conv
        if ind == num_units - 1 and gen_cross_conv:
            self.cross_conv = ConvModule(unit_channels, out_channels, 1, norm_cfg=norm_cfg)

    def forward_unused(self, x, up_x):
        """This unit's output is consumed by nobody (the last stage's finest map when the neck starts at level 1: the
        reference computes it anyway, mspn_mmpose.py:381-404, and nothing reads it). Eval: nothing to do. Train: the two
        convs still run — their BatchNorm layers' running statistics are part of the state dict and must advance exactly
        as the reference's do — but the normalised tensors are never written (`conv_bn_stats_only`), and no autograd
        node is recorded (the reference's backward does not reach them either)."""
        assert not self.gen_skip and not (self.ind == self.num_units - 1 and self.gen_cross_conv)
        if not self.in_skip.bn.training:
            return None
        with torch.no_grad():
            nnops.conv_bn_stats_only(x, self.in_skip.conv, self.in_skip.bn)
            if self.ind > 0:
                nnops.upsample_conv_bn_stats_only(up_x, x.shape[1], x.shape[2], self.up_conv.conv, self.up_conv.bn)
        return None

    def forward(self, x, up_x):
        # x and out each feed several convs: all but the last consumer hand the tensor through their autograd node
        # (conv_bn(skip_through=True)), so the gradients meet in data-gradient epilogues instead of elementwise adds
        thru = self.gen_skip
        if self.ind > 0:
            # relu(in_skip(x) + up_conv(upsample(up_x))): in train mode one autograd node (nn.up_merge: up_conv runs before
            # the upsampling, on a quarter of the pixels, and neither normalised branch is written); otherwise add + ReLU
            # fused into up_conv's BatchNorm pass
            out = nnops.up_merge(x, up_x, self.in_skip, self.up_conv, skip_through=thru)
            if thru:
                out, x = out
        else:
            out = conv_bn(x, self.in_skip.conv, self.in_skip.bn, relu=True, skip_through=thru)
            if thru:
                out, x = out
        skip1 = skip2 = cross = None
        if self.gen_skip:
            # (train mode: normalised by their consumer, the next stage's add — nn.conv_bn_deferred / skip_add)
            # (the fused add handles ONE statistics span for both layers: a pair that mixes SyncBN and plain BN takes
            # the per-layer path — `partner`)
            skip1 = nnops.conv_bn_deferred(x, self.out_skip1, partner=self.out_skip2, hold=True)   # (finalized with skip2's)
            skip2, out = nnops.conv_bn_deferred(out, self.out_skip2, skip_through=True, partner=self.out_skip1)
        if self.ind == self.num_units - 1 and self.gen_cross_conv:
            m = self.cross_conv
            cross, out = conv_bn(out, m.conv, m.norm, relu=m.with_activation, skip_through=True)
        return out, skip1, skip2, cross


class UpsampleModule(nn.Module):
    def __init__(self, unit_channels=256, num_units=4, gen_skip=False, gen_cross_conv=False, norm_cfg=dict(type='BN'),
                 out_channels=64):
        super().__init__()
        self.in_channels = [Bottleneck.expansion * out_channels * 2 ** i for i in range(num_units)][::-1]
        self.num_units = num_units
        for i in range(num_units):
            self.add_module(f'up{i + 1}', UpsampleUnit(i, num_units, self.in_channels[i], unit_channels, gen_skip,
                                                        gen_cross_conv, norm_cfg=norm_cfg, out_channels=64))

    def forward(self, x, skip_finest=False):
        out, skip1, skip2, cross = [], [], [], None
        for i in range(self.num_units):
            unit = getattr(self, f'up{i + 1}')
            if skip_finest and i == self.num_units - 1:
                out.append(unit.forward_unused(x[i], out[i - 1] if i > 0 else None))
                skip1.append(None)
                skip2.append(None)
                continue
            o, s1, s2, c = unit(x[i], out[i - 1] if i > 0 else None)
            out.append(o)
            skip1.append(s1)
            skip2.append(s2)
            if c is not None:
                cross = c
        skip1.reverse()
        skip2.reverse()
        return out, skip1, skip2, cross


class SingleStageNetwork(nn.Module):
    def __init__(self, has_skip=False, gen_skip=False, gen_cross_conv=False, unit_channels=256, num_units=4,
                 num_blocks=(2, 2, 2, 2), norm_cfg=dict(type='BN'), in_channels=64):
        super().__init__()
        self.downsample = DownsampleModule(list(num_blocks), num_units, has_skip, norm_cfg, in_channels)
        self.upsample = UpsampleModule(unit_channels, num_units, gen_skip, gen_cross_conv, norm_cfg, in_channels)

    def forward(self, x, skip1, skip2, skip_finest=False):
        mid = self.downsample(x, skip1, skip2)
        return self.upsample(mid, skip_finest)


class ResNetTop(nn.Module):
    def __init__(self, norm_cfg=dict(type='BN'), channels=64):
        super().__init__()
        # index 1 (MaxPool2d) has no parameters; kept so that keys read top.top.0.*
        self.top = nn.Sequential(ConvModule(3, channels, 7, stride=2, padding=3, norm_cfg=norm_cfg),
                                 nn.MaxPool2d(kernel_size=3, stride=2, padding=1))

    def forward(self, x):
        return nnops.max_pool(self.top[0](x))


@BACKBONES.register_module()
class MSPN2(nn.Module):
    """forward(img NCHW float) -> list of the last stage's 4 maps, strides 4/8/16/32, 256 ch,
    as NCHW-shaped channels-last views (mspn_mmpose.py:657-667)."""

    def __init__(self, unit_channels=256, num_stages=4, num_units=4, num_blocks=[2, 2, 2, 2], norm_cfg=dict(type='BN'),
                 res_top_channels=64, frozen_stages=-1, norm_eval=False, pretrained=None, compute_dtype='bf16'):
        super().__init__()
        norm_cfg = copy.deepcopy(norm_cfg)
        num_blocks = copy.deepcopy(num_blocks)
        assert num_stages > 0 and num_units > 1 and num_units == len(num_blocks)
        self.unit_channels, self.num_stages, self.num_units = unit_channels, num_stages, num_units
        self.num_blocks, self.norm_cfg = num_blocks, norm_cfg
        self.top = ResNetTop(norm_cfg=norm_cfg)
        self.multi_stage_mspn = nn.ModuleList()
        for i in range(num_stages):
            last = i == num_stages - 1
            self.multi_stage_mspn.append(SingleStageNetwork(i > 0, not last, not last, unit_channels, num_units,
                                                            num_blocks, norm_cfg, res_top_channels))
        self.pretrained = pretrained
        self.compute_dtype = torch.bfloat16 if compute_dtype in ('bf16', torch.bfloat16) else torch.float32
        # True (set by the detector when its neck starts at level >= 1, as every DAS config's does): the last stage's
        # finest map — outputs[0], stride 4 — has no consumer; forward returns None in its place (UpsampleUnit.forward_unused)
        self.skip_unused_finest = False

    def forward(self, x):
        x = as_nhwc(x, self.compute_dtype)
        x = self.top(x)
        skip1 = skip2 = None
        out = None
        last = self.multi_stage_mspn[-1]
        for stage in self.multi_stage_mspn:
            out, skip1, skip2, x = stage(x, skip1, skip2, skip_finest=self.skip_unused_finest and stage is last)
        return [to_nchw_view(o) if o is not None else None for o in out[::-1]]

    def init_weights(self, pretrained=None):
        """mspn_mmpose.py:669-721, both branches.
          * `pretrained` starting with 'weights/' (the DAS configs' MSPN checkpoints, exp_panoptic.py:12): a full
            detector / MSPN checkpoint — keys under `backbone.` are loaded by name, non-strict.
          * otherwise: kaiming-normal (fan_out, relu) for every conv of the stages and of the stem, BatchNorm weight 1 /
            bias 0, Linear N(0, 0.01); then, if `pretrained` is a path, a ResNet-50 classification checkpoint
            (torchvision / MMPose key layout `conv1`, `bn1`, `layer{1-4}.{b}.*`, `layer{L}.0.downsample.{0,1}.*`) is
            mapped onto the stem (`conv1 -> top.0.conv`, `bn1 -> top.0.bn`) and onto the downsample module of EVERY
            stage (`downsample.0 -> downsample.conv`, `downsample.1 -> downsample.bn`).
        Unlike the reference (which ignores its argument and fails on pretrained=None), the argument is honoured and None
        means random initialisation."""
        pretrained = pretrained if pretrained is not None else self.pretrained
        if isinstance(pretrained, str) and pretrained.startswith('weights/'):
            if not os.path.isfile(pretrained):
                raise FileNotFoundError(f'pretrained backbone checkpoint {pretrained} not found')
            loaded = torch.load(pretrained, map_location='cpu', weights_only=False)['state_dict']
            sd = {k.replace('backbone.', ''): v for k, v in loaded.items() if k.startswith('backbone.')}
            return self.load_state_dict(sd, strict=False)
        for m in self.multi_stage_mspn.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, a=0, mode='fan_out', nonlinearity='relu')
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        for m in self.top.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, a=0, mode='fan_out', nonlinearity='relu')
        if not isinstance(pretrained, str):
            return None
        if not os.path.isfile(pretrained):
            raise FileNotFoundError(f'pretrained backbone checkpoint {pretrained} not found')
        tmp = resnet_state_dict(pretrained)
        top, bottlenecks = {}, {}
        for k, v in tmp.items():
            if k.startswith('layer'):
                if 'downsample.0' in k:
                    bottlenecks[k.replace('downsample.0', 'downsample.conv')] = v
                elif 'downsample.1' in k:
                    bottlenecks[k.replace('downsample.1', 'downsample.bn')] = v
                else:
                    bottlenecks[k] = v
            elif k.startswith('conv1'):
                top[k.replace('conv1', 'top.0.conv')] = v
            elif k.startswith('bn1'):
                top[k.replace('bn1', 'top.0.bn')] = v
        report = [self.top.load_state_dict(top, strict=False)]
        for stage in self.multi_stage_mspn:
            report.append(stage.downsample.load_state_dict(bottlenecks, strict=False))
        return report


def resnet_state_dict(filename, map_location='cpu'):
    """`get_state_dict` (mspn_mmpose.py:161-193): the checkpoint's `state_dict` (or the file itself) with the
    `module.backbone.` / `module.` / `backbone.` prefixes stripped."""
    ck = torch.load(filename, map_location=map_location, weights_only=False)
    if not isinstance(ck, dict):
        raise RuntimeError(f'No state_dict found in checkpoint file {filename}')
    sd = ck.get('state_dict', ck)
    out = {}
    for k, v in sd.items():
        for pre in ('module.backbone.', 'module.', 'backbone.'):
            if k.startswith(pre):
                k = k[len(pre):]
                break
        out[k] = v
    return out
