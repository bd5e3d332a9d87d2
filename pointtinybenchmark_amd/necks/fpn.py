"""FPN neck with the fork's ``num_outs`` < #levels behaviour (T/mmdet/models/necks/fpn.py:67-218; fork edits
at :96,134,193): every lateral 1x1 conv + GN and the whole top-down nearest-upsample chain run, but only the
first ``num_outs`` 3x3 output convs exist.  GroupNorm-apply and the top-down add are ONE fused pass per level."""
import torch.nn as nn

from .. import ops
from ..layers import ConvModule, _PackCache, conv_gn
from ..registry import NECKS


@NECKS.register_module()
class FPN(nn.Module):
    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False,
                 extra_convs_on_inputs=True, relu_before_extra_convs=False, no_norm_on_lateral=False, conv_cfg=None,
                 norm_cfg=None, act_cfg=None, upsample_cfg=dict(mode='nearest'), init_cfg=None):
        super().__init__()
        assert isinstance(in_channels, (list, tuple))
        assert norm_cfg is not None and norm_cfg['type'] == 'GN' and not no_norm_on_lateral and act_cfg is None, \
            'the CPR/P2P configs use GN laterals without activation (SURVEY.md §8 a2)'
        assert upsample_cfg.get('mode', 'nearest') == 'nearest' and 'scale_factor' not in upsample_cfg
        self.in_channels, self.out_channels, self.num_outs = list(in_channels), out_channels, num_outs
        self.num_ins = len(in_channels)
        self.backbone_end_level = self.num_ins if end_level == -1 else end_level
        self.start_level, self.end_level = start_level, end_level
        self.add_extra_convs = add_extra_convs
        self.lateral_convs = nn.ModuleList()
        self.fpn_convs = nn.ModuleList()
        for i in range(start_level, self.backbone_end_level):
            self.lateral_convs.append(ConvModule(in_channels[i], out_channels, 1, norm_cfg=norm_cfg, act_cfg=None))
            if i < start_level + num_outs:  # fork change (fpn.py:134)
                self.fpn_convs.append(ConvModule(out_channels, out_channels, 3, padding=1, norm_cfg=norm_cfg,
                                                 act_cfg=None))
        extra = num_outs - self.backbone_end_level + start_level
        assert extra < 1, 'extra pyramid levels are not used by the CPR/P2P configs (num_outs=1)'
        self._cache = _PackCache()
        self.init_weights()

    def init_weights(self):
        """init_cfg=dict(type='Xavier', layer='Conv2d', distribution='uniform') (fpn.py:81-82): Xavier-uniform conv
        weights, zero biases; GroupNorm keeps weight 1 / bias 0."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def run_laterals(self, xs, tape=None):
        """xs: the NHWC stage outputs from ``start_level`` on -> the top-down lateral sums, finest first (fpn.py:166-188)."""
        c = self._cache
        # top-down: coarsest level first; GN-apply of level i and "+= upsample(level i+1)" in one kernel
        lat = [None] * len(xs)
        for i in range(len(xs) - 1, -1, -1):
            up = lat[i + 1] if i + 1 < len(xs) else None
            rec = None
            if tape is not None:
                rec = dict(kind='lateral', level=i)
                tape.append(rec)
            lat[i] = conv_gn(c, self.lateral_convs[i], xs[i], up=up, save=rec)
        return lat

    def _run(self, inputs, lazy, tape=None, out_b8=False):
        assert len(inputs) == len(self.in_channels)
        c = self._cache
        xs = [ops.from_nchw(inputs[i + self.start_level]) for i in range(len(self.lateral_convs))]
        lat = self.run_laterals(xs, tape)
        used = min(len(lat), self.num_outs)
        outs = []
        for i in range(used):
            rec = None
            if tape is not None:
                rec = dict(kind='out', level=i)
                tape.append(rec)
            outs.append(conv_gn(c, self.fpn_convs[i], lat[i], materialize=not lazy, save=rec, out_b8=out_b8 and lazy))
        return outs

    def forward(self, inputs):
        return tuple(ops.as_nchw(t) for t in self._run(inputs, lazy=False))

    def forward_lazy(self, inputs, tape=None, out_b8=False):
        """Internal fast path: per level (raw conv output NHWC, (a, b)) -- the consumer conv applies the GroupNorm
        affine while loading (no activation after the FPN convs: act_cfg=None).  tape: training records.
        out_b8: the consumer is a Winograd 3x3 layer (CPRHead tower) -- raw outputs channel-blocked where the layer can."""
        return self._run(inputs, lazy=True, tape=tape, out_b8=out_b8 and tape is None)
