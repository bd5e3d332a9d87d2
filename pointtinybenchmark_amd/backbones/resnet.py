"""ResNet backbone on the fp32-MFMA implicit-GEMM conv kernel.

Interface of T/mmdet/models/backbones/resnet.py:305-657 (ctor kwargs, ``forward(x) -> tuple`` of the
``out_indices`` stage outputs, ``frozen_stages`` / ``norm_eval`` train() semantics, state-dict keys).
BatchNorm is always evaluated with running statistics on this path (norm_eval=True in every CPR/P2P config)
and is folded into the conv epilogue; the bottleneck shortcut add + ReLU are fused into conv3's epilogue.
The backward of the trainable stages is driven by training.CprTrainer from the per-block records of ``forward(tape=)``."""
import os

import torch
import torch.nn as nn

from .. import ops
from ..layers import _PackCache, folded_bn, packed_conv
from ..registry import BACKBONES


FUSE_SHORTCUT = [os.environ.get('CPR_FUSE_SHORTCUT', '1') == '1']   # A/B switch (tests, tools)


class _Block(nn.Module):
    def __init__(self, kind, inplanes, planes, stride, downsample):
        super().__init__()
        self.kind = kind
        if kind == 'bottleneck':  # style='pytorch': the stride sits on the 3x3 (resnet.py:153-158)
            self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
            self.bn1 = nn.BatchNorm2d(planes)
            self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
            self.bn2 = nn.BatchNorm2d(planes)
            self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
            self.bn3 = nn.BatchNorm2d(planes * 4)
        else:
            self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
            self.bn1 = nn.BatchNorm2d(planes)
            self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
            self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def run(self, cache, x, save=None):
        """save (dict): training mode -- keeps the block's activations for the backward pass."""
        identity = x
        dt = x.dtype
        # forward-only fp32 bottleneck with a projection shortcut: the shortcut GEMM rides in conv3's launch (bit-identical,
        # ops.conv2d_dual); the training step keeps the two launches (its backward walks the recorded maps)
        fuse_shortcut = FUSE_SHORTCUT[0] and self.downsample is not None and save is None and self.kind == 'bottleneck' and \
            dt == torch.float32
        if self.downsample is not None and not fuse_shortcut:
            s, b = folded_bn(cache, self.downsample[1])
            identity = ops.conv2d(x, packed_conv(cache, self.downsample[0], x.dtype), scale=s, bias=b)
        s1, b1 = folded_bn(cache, self.bn1)
        o1 = ops.conv2d(x, packed_conv(cache, self.conv1, dt), scale=s1, bias=b1, relu=True)
        s2, b2 = folded_bn(cache, self.bn2)
        if self.kind == 'bottleneck':
            o2 = ops.conv2d(o1, packed_conv(cache, self.conv2, dt), scale=s2, bias=b2, relu=True)
            s3, b3 = folded_bn(cache, self.bn3)
            if fuse_shortcut:
                sd, bd = folded_bn(cache, self.downsample[1])
                out = ops.conv2d_dual(o2, packed_conv(cache, self.conv3, dt), x, packed_conv(cache, self.downsample[0], dt),
                                      scale=s3, bias=b3, scale2=sd, bias2=bd, relu=True)
            else:
                out = ops.conv2d(o2, packed_conv(cache, self.conv3, dt), scale=s3, bias=b3, residual=identity, relu=True)
        else:
            o2 = None
            out = ops.conv2d(o1, packed_conv(cache, self.conv2, dt), scale=s2, bias=b2, residual=identity, relu=True)
        if save is not None:
            save.update(block=self, x=x, o1=o1, o2=o2, identity=identity, out=out)
        return out


F32_STEM = [os.environ.get('CPR_F32_STEM', '1') != '0']       # 0: stem conv on the implicit-GEMM kernel + separate max-pool (A/B, tests)
BF16_STEM_POOL = [os.environ.get('CPR_BF16_STEM_POOL', '1') != '0']   # 0: stem conv and max-pool as two kernels
BF16_STEM = [os.environ.get('CPR_BF16_STEM', '1') != '0']     # 0: the bf16 mode keeps its stem on the fp32 kernel (A/B, tests)


@BACKBONES.register_module()
class ResNet(nn.Module):
    arch_settings = {18: ('basic', (2, 2, 2, 2)), 34: ('basic', (3, 4, 6, 3)), 50: ('bottleneck', (3, 4, 6, 3)),
                     101: ('bottleneck', (3, 4, 23, 3)), 152: ('bottleneck', (3, 8, 36, 3))}

    def __init__(self, depth, in_channels=3, stem_channels=None, base_channels=64, num_stages=4,
                 strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1), out_indices=(0, 1, 2, 3), style='pytorch',
                 deep_stem=False, avg_down=False, frozen_stages=-1, conv_cfg=None,
                 norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, dcn=None, stage_with_dcn=None,
                 plugins=None, with_cp=False, zero_init_residual=True, pretrained=None, init_cfg=None):
        super().__init__()
        if depth not in self.arch_settings:
            raise KeyError('invalid depth %s for resnet' % depth)
        assert style == 'pytorch' and not deep_stem and not avg_down and dcn is None and plugins is None, \
            'only the options used by the CPR/P2P configs are built (SURVEY.md §2a row 5)'
        assert tuple(dilations[:num_stages]) == (1,) * num_stages and norm_cfg.get('type') == 'BN'
        assert norm_eval, 'BatchNorm batch statistics are not on this path (every CPR/P2P config sets norm_eval=True)'
        self.depth, self.num_stages, self.out_indices = depth, num_stages, tuple(out_indices)
        self.frozen_stages, self.norm_eval = frozen_stages, norm_eval
        kind, blocks = self.arch_settings[depth]
        exp = 4 if kind == 'bottleneck' else 1
        stem = stem_channels or base_channels
        self.conv1 = nn.Conv2d(in_channels, stem, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(stem)
        inplanes = stem
        self.res_layers = []
        for i in range(num_stages):
            planes = base_channels * 2 ** i
            layer = []
            for bi in range(blocks[i]):
                stride = strides[i] if bi == 0 else 1
                ds = None
                if bi == 0 and (stride != 1 or inplanes != planes * exp):
                    ds = nn.Sequential(nn.Conv2d(inplanes, planes * exp, 1, stride, bias=False),
                                       nn.BatchNorm2d(planes * exp))
                layer.append(_Block(kind, inplanes, planes, stride, ds))
                inplanes = planes * exp
            name = 'layer%d' % (i + 1)
            self.add_module(name, nn.Sequential(*layer))
            self.res_layers.append(name)
        self.feat_dim = inplanes
        self.compute_dtype = torch.float32   # torch.bfloat16 = bf16 activations/weights from the stem output on
        self._cache = _PackCache()
        self.zero_init_residual = zero_init_residual
        self.init_weights()
        self._freeze_stages()

    def _freeze_stages(self):  # resnet.py:612-628
        if self.frozen_stages >= 0:
            for m in (self.conv1, self.bn1):
                for p in m.parameters():
                    p.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            for p in getattr(self, 'layer%d' % i).parameters():
                p.requires_grad = False

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        return self

    def init_weights(self):
        """The reference's default init_cfg when no checkpoint is given (resnet.py:404-424; the configs name
        torchvision://resnet50, which is not available offline, so weights normally arrive through load_state_dict):
        Kaiming normal (fan_out, relu) on every conv, BatchNorm weight 1 / bias 0, and -- zero_init_residual -- weight 0 on
        the last norm of every block."""
        from ..layers import bump_weight_epoch
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if self.zero_init_residual:
            for name in self.res_layers:
                for blk in getattr(self, name):
                    nn.init.constant_((blk.bn3 if blk.kind == 'bottleneck' else blk.bn2).weight, 0)
        bump_weight_epoch()

    def stem(self, x):
        """(N,3,H,W) image -> the NHWC map after conv1 + bn1 + ReLU + max-pool (resnet.py:630-637)."""
        c = self._cache
        s, b = folded_bn(c, self.bn1)
        c1 = self.conv1
        std7 = tuple(c1.weight.shape) == (64, 3, 7, 7) and c1.stride == (2, 2) and c1.padding == (3, 3)
        # the fused stem kernels read the three planes of a contiguous (N,3,H,W) fp32 image themselves (no nchw_to_nhwc4 pass)
        planar = std7 and x.dim() == 4 and x.dtype == torch.float32 and x.shape[1] == 3 and x.is_contiguous() and \
            (F32_STEM[0] if self.compute_dtype == torch.float32 else BF16_STEM[0])
        if not planar:
            # (N,3,H,W) float image -> NHWC4; a 4-channel channels-last view (datasets.GpuImagePipeline output) is taken as is
            x = ops.from_nchw(x) if (x.shape[1] > 4 or (x.shape[1] == 4 and x.stride(1) == 1)) else ops.nchw_to_nhwc(x)
        fused_in = std7 and x.dtype == torch.float32 and (planar or x.shape[-1] == 4)
        if self.compute_dtype == torch.bfloat16 and BF16_STEM[0] and fused_in and ops.stem_bf16_fits(x, planar):
            # bf16 compute mode: the stem on the bf16 matrix cores (csrc/stem_bf16.hip; round 4)
            wp = c.get(('stem_bf16', id(c1)), [c1.weight], lambda: ops.stem_weight_bf16(c1.weight))
            if BF16_STEM_POOL[0]:
                return ops.stem7x7s2_pool_bf16(x, wp, scale=s, bias=b, planar=planar)       # conv + BN + ReLU + max-pool, one kernel
            x = ops.stem7x7s2_bf16(x, wp, scale=s, bias=b, relu=True, planar=planar)
        elif self.compute_dtype == torch.float32 and F32_STEM[0] and fused_in:
            # conv + BN + ReLU + max-pool in one exact-fp32 kernel (csrc/stem_f32.hip; round 4)
            wp = c.get(('stem_f32', id(c1)), [c1.weight], lambda: ops.stem_weight_f32(c1.weight))
            return ops.stem7x7s2_pool_f32(x, wp, scale=s, bias=b, planar=planar)
        else:
            # (other stems: the implicit-GEMM kernel in its stem mode; in the bf16 mode it emits the bf16 map)
            if planar:          # (a bf16-mode map too large for the bf16 stem kernel's 32-bit offsets)
                x = ops.nchw_to_nhwc(x)
            x = ops.conv2d(x, packed_conv(c, c1), scale=s, bias=b, relu=True, out_dtype=self.compute_dtype)
        return ops.maxpool3x3s2(x)

    def run_stage(self, i, x, tape=None):
        """Stage ``i`` (``layer{i+1}``) on an NHWC map.  tape (list): one record per block with trainable parameters."""
        for blk in getattr(self, self.res_layers[i]):
            rec = None
            if tape is not None and blk.conv1.weight.requires_grad:
                rec = dict(stage=i)
                tape.append(rec)
            x = blk.run(self._cache, x, rec)
        return x

    def forward(self, x, tape=None):
        """x: (N,3,H,W) -> tuple of NCHW-shaped (channels_last) stage outputs.
        tape (list): training mode -- one record per block with trainable parameters, in forward order."""
        x = self.stem(x)
        outs = []
        for i in range(len(self.res_layers)):
            x = self.run_stage(i, x, tape)
            if i in self.out_indices:
                outs.append(ops.as_nchw(x))
        return tuple(outs)
