"""FPN neck as the DAS configs use it (mmdet 2.14.0 `FPN`; merged config
configs/_base_/models/das.py:16-23 + configs/das/exp_panoptic.py:24-30): laterals 1x1+BN from
`start_level`, top-down `+= nearest_upsample`, 3x3+BN per level; with `add_extra_convs='on_output'`
the first extra level is a stride-2 3x3+BN on the last output (no ReLU) and further extra levels
take `relu(previous)` when `relu_before_extra_convs` (mmdet `FPN.forward` part 2).
State-dict keys: `lateral_convs.{i}.{conv,bn}`, `fpn_convs.{i}.{conv,bn}`."""
import torch.nn as nn

from . import ops
from .nn import ConvModule, add_nearest, as_nhwc, to_nchw_view
from .registry import NECKS


@NECKS.register_module()
class FPN(nn.Module):
    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False,
                 relu_before_extra_convs=False, no_norm_on_lateral=False, conv_cfg=None, norm_cfg=None, act_cfg=None,
                 upsample_cfg=dict(mode='nearest'), init_cfg=None, compute_dtype=None):
        super().__init__()
        assert isinstance(in_channels, list)
        assert upsample_cfg.get('mode', 'nearest') == 'nearest' and 'scale_factor' not in upsample_cfg
        assert conv_cfg is None and act_cfg is None
        self.in_channels, self.out_channels, self.num_outs = in_channels, out_channels, num_outs
        self.num_ins = len(in_channels)
        self.relu_before_extra_convs = relu_before_extra_convs
        assert end_level == -1, 'the DAS configs use end_level=-1'
        self.backbone_end_level = self.num_ins
        assert num_outs >= self.num_ins - start_level
        self.start_level = start_level
        if add_extra_convs is True:
            add_extra_convs = 'on_input'
        assert add_extra_convs in (False, 'on_input', 'on_lateral', 'on_output')
        self.add_extra_convs = add_extra_convs
        self.lateral_convs = nn.ModuleList()
        self.fpn_convs = nn.ModuleList()
        for i in range(start_level, self.backbone_end_level):
            self.lateral_convs.append(ConvModule(in_channels[i], out_channels, 1,
                                                 norm_cfg=None if no_norm_on_lateral else norm_cfg, act_cfg=None))
            self.fpn_convs.append(ConvModule(out_channels, out_channels, 3, padding=1, norm_cfg=norm_cfg, act_cfg=None))
        extra = num_outs - self.backbone_end_level + start_level
        if add_extra_convs and extra >= 1:
            for i in range(extra):
                cin = in_channels[self.backbone_end_level - 1] if (i == 0 and add_extra_convs == 'on_input') \
                    else out_channels
                self.fpn_convs.append(ConvModule(cin, out_channels, 3, stride=2, padding=1, norm_cfg=norm_cfg,
                                                 act_cfg=None))
        elif extra >= 1:
            raise NotImplementedError('max-pool extra levels are not used by the DAS configs')

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)

    def forward(self, inputs):
        assert len(inputs) == len(self.in_channels)
        # (levels below start_level are never read: the backbone may hand None for them — MSPN2.skip_unused_finest)
        dtype = inputs[self.start_level].dtype
        feats = [as_nhwc(t, dtype) if t is not None else None for t in inputs]
        lats = [l(feats[i + self.start_level]) for i, l in enumerate(self.lateral_convs)]
        n = len(lats)
        for i in range(n - 1, 0, -1):
            lats[i - 1] = add_nearest(lats[i - 1], lats[i])
        outs = [self.fpn_convs[i](lats[i]) for i in range(n)]
        if self.num_outs > len(outs):
            if self.add_extra_convs == 'on_input':
                src = feats[self.backbone_end_level - 1]
            elif self.add_extra_convs == 'on_lateral':
                src = lats[-1]
            else:
                src = outs[-1]
            outs.append(self.fpn_convs[n](src))  # first extra level: no ReLU in mmdet 2.14.0
            for i in range(n + 1, self.num_outs):
                outs.append(self.fpn_convs[i](outs[-1], relu_in=self.relu_before_extra_convs))
        return tuple(to_nchw_view(o) for o in outs)
