"""Load the reference's own hot-path classes from /root/reference (read-only).

TEST INFRASTRUCTURE ONLY -- works only where /root/reference exists (the build
container); it is used to validate ``oracle/cpr_oracle.py`` and to generate the
fixtures under ``tests/golden/`` (see ``oracle/gen_golden.py``).  Nothing that
runs on the GPU box imports this module.

How it works: ``import mmdet`` fails here (mmcv,
pycocotools, ... are not installed), so the leaf modules are imported one by
one: ``mmdet.*`` packages are registered as empty namespace modules whose
``__path__`` points into the reference tree (no ``__init__`` runs), and a small
pure-Python ``mmcv`` stand-in provides the registry / ConvModule / decorators
those files need.
"""
import importlib
import os
import sys
import types

import torch
import torch.nn as nn

REF_ROOT = os.environ.get('CPR_REFERENCE_ROOT', '/root/reference/TOV_mmdetection')


def available():
    return os.path.isdir(os.path.join(REF_ROOT, 'mmdet'))


# ----------------------------------------------------------------------------- mmcv stand-in
class _Registry:
    def __init__(self, name, parent=None, **kw):
        self.name, self._d, self.parent = name, {}, parent

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self._d[name or cls.__name__] = cls
            return cls
        return deco if module is None else deco(module)

    def get(self, k):
        return self._d.get(k) or (self.parent.get(k) if self.parent else None)

    def build(self, cfg, default_args=None):
        return _build_from_cfg(cfg, self, default_args)


def _build_from_cfg(cfg, registry, default_args=None):
    cfg = dict(cfg)
    t = cfg.pop('type')
    for k, v in (default_args or {}).items():
        cfg.setdefault(k, v)
    cls = registry.get(t) if isinstance(t, str) else t
    assert cls is not None, t
    return cls(**cfg)


class _BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg

    def init_weights(self):
        pass


def _identity_deco(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]
    return lambda f: f


def _build_norm_layer(cfg, num_features, postfix=''):
    cfg = dict(cfg)
    t = cfg.pop('type')
    rg = cfg.pop('requires_grad', True)
    if t == 'BN':
        layer, name = nn.BatchNorm2d(num_features, **cfg), 'bn'
    elif t == 'GN':
        layer, name = nn.GroupNorm(num_channels=num_features, **cfg), 'gn'
    else:
        raise KeyError(t)
    for p in layer.parameters():
        p.requires_grad = rg
    return name + str(postfix), layer


def _build_conv_layer(cfg, *a, **k):
    assert cfg is None or cfg.get('type') in (None, 'Conv2d')
    return nn.Conv2d(*a, **k)


class _ConvModule(nn.Module):
    """conv -> norm -> act, bias='auto' (bias only when there is no norm)."""

    def __init__(self, in_c, out_c, k, stride=1, padding=0, dilation=1, groups=1, bias='auto',
                 conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'), inplace=True, **kw):
        super().__init__()
        self.with_norm = norm_cfg is not None
        self.with_act = act_cfg is not None
        if bias == 'auto':
            bias = not self.with_norm
        self.conv = nn.Conv2d(in_c, out_c, k, stride, padding, dilation, groups, bias)
        if self.with_norm:
            self.norm_name, norm = _build_norm_layer(norm_cfg, out_c)
            self.add_module(self.norm_name, norm)
        if self.with_act:
            self.activate = nn.ReLU(inplace=inplace)

    def forward(self, x):
        x = self.conv(x)
        if self.with_norm:
            x = getattr(self, self.norm_name)(x)
        if self.with_act:
            x = self.activate(x)
        return x


def _torch_batched_nms(boxes, scores, idxs, nms_cfg, class_agnostic=False):
    """Restatement of mmcv 1.3.x ``batched_nms`` (third-party, un-vendored): class-offset
    trick, greedy IoU > thr suppression, offset=0.  Returns (dets(k,5), keep)."""
    nms_cfg = dict(nms_cfg)
    class_agnostic = nms_cfg.pop('class_agnostic', class_agnostic)
    thr = nms_cfg.get('iou_threshold', nms_cfg.get('iou_thr'))
    if class_agnostic:
        b = boxes
    else:
        off = idxs.to(boxes) * (boxes.max() + 1)
        b = boxes + off[:, None]
    order = torch.argsort(scores, descending=True, stable=True)
    b = b[order]
    n = len(b)
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    supp = torch.zeros(n, dtype=torch.bool)
    keep = []
    for i in range(n):
        if supp[i]:
            continue
        keep.append(i)
        xx1 = torch.maximum(b[i, 0], b[i + 1:, 0])
        yy1 = torch.maximum(b[i, 1], b[i + 1:, 1])
        xx2 = torch.minimum(b[i, 2], b[i + 1:, 2])
        yy2 = torch.minimum(b[i, 3], b[i + 1:, 3])
        inter = (xx2 - xx1).clamp(min=0) * (yy2 - yy1).clamp(min=0)
        iou = inter / (area[i] + area[i + 1:] - inter)
        supp[i + 1:] |= iou > thr
    keep = order[torch.tensor(keep, dtype=torch.long)]
    return torch.cat([boxes[keep], scores[keep, None]], -1), keep


_LOADED = None


def load():
    """Returns a namespace with the reference's classes (cached)."""
    global _LOADED
    if _LOADED is not None:
        return _LOADED
    if not available():
        raise RuntimeError('reference tree not found at %s' % REF_ROOT)
    R = REF_ROOT

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        if '.' in name:
            p, c = name.rsplit('.', 1)
            setattr(sys.modules[p], c, m)
        return m

    def ns(name, rel):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(R, rel)]
        sys.modules[name] = m
        if '.' in name:
            p, c = name.rsplit('.', 1)
            setattr(sys.modules[p], c, m)
        return m

    mmcv = mod('mmcv', __version__='1.3.9', jit=_identity_deco)
    mod('mmcv.cnn', ConvModule=_ConvModule, build_conv_layer=_build_conv_layer,
        build_norm_layer=_build_norm_layer, build_plugin_layer=None, MODELS=_Registry('model'))
    mod('mmcv.utils', Registry=_Registry, build_from_cfg=_build_from_cfg)
    mod('mmcv.runner', BaseModule=_BaseModule, force_fp32=_identity_deco, auto_fp16=_identity_deco,
        Sequential=nn.Sequential, ModuleList=nn.ModuleList)

    class _Stub:
        def __init__(self, *a, **k):
            raise NotImplementedError('mmcv.ops is third-party and not on the CPU path')

    mod('mmcv.ops', DeformConv2d=_Stub, sigmoid_focal_loss=None)
    mod('mmcv.ops.nms', batched_nms=_torch_batched_nms)

    for name, rel in [
            ('mmdet', 'mmdet'), ('mmdet.core', 'mmdet/core'), ('mmdet.core.bbox', 'mmdet/core/bbox'),
            ('mmdet.core.bbox.assigners', 'mmdet/core/bbox/assigners'),
            ('mmdet.core.bbox.match_costs', 'mmdet/core/bbox/match_costs'),
            ('mmdet.core.bbox.samplers', 'mmdet/core/bbox/samplers'),
            ('mmdet.core.bbox.iou_calculators', 'mmdet/core/bbox/iou_calculators'),
            ('mmdet.core.anchor', 'mmdet/core/anchor'), ('mmdet.core.utils', 'mmdet/core/utils'),
            ('mmdet.core.post_processing', 'mmdet/core/post_processing'),
            ('mmdet.models', 'mmdet/models'), ('mmdet.models.losses', 'mmdet/models/losses'),
            ('mmdet.models.dense_heads', 'mmdet/models/dense_heads'),
            ('mmdet.models.backbones', 'mmdet/models/backbones'), ('mmdet.models.necks', 'mmdet/models/necks'),
            ('mmdet.models.utils', 'mmdet/models/utils'), ('mmdet.models.point', 'mmdet/models/point'),
            ('mmdet.models.point.dense_heads', 'mmdet/models/point/dense_heads'),
            ('mmdet.models.detectors', 'mmdet/models/detectors'), ('mmdet.utils', 'mmdet/utils')]:
        ns(name, rel)

    I = importlib.import_module
    b = I('mmdet.core.bbox.builder')
    sys.modules['mmdet.core.bbox'].__dict__.update(build_assigner=b.build_assigner, build_sampler=b.build_sampler)
    iou = I('mmdet.core.bbox.iou_calculators.iou2d_calculator')
    sys.modules['mmdet.core.bbox.iou_calculators'].bbox_overlaps = iou.bbox_overlaps
    tr = I('mmdet.core.bbox.transforms')
    sys.modules['mmdet.core.bbox'].bbox_xyxy_to_cxcywh = tr.bbox_xyxy_to_cxcywh
    mcb = I('mmdet.core.bbox.match_costs.builder')
    sys.modules['mmdet.core.bbox.match_costs'].build_match_cost = mcb.build_match_cost
    mc = I('mmdet.core.bbox.match_costs.match_cost')
    um = I('mmdet.utils.util_mixins')
    sys.modules['mmdet.utils'].util_mixins = um
    ha = I('mmdet.core.bbox.assigners.hungarian_assigner')
    pa = I('mmdet.core.bbox.assigners.point_assigner')
    ps = I('mmdet.core.bbox.samplers.pseudo_sampler')
    I('mmdet.core.anchor.builder')
    pg = I('mmdet.core.anchor.point_generator')
    # core/utils/misc.py pulls mask structures (pycocotools) -> execute its text minus that import
    src = open(os.path.join(R, 'mmdet/core/utils/misc.py')).read().replace(
        'from ..mask.structures import BitmapMasks, PolygonMasks', '')
    misc = types.ModuleType('mmdet.core.utils.misc')
    exec(compile(src, 'misc.py', 'exec'), misc.__dict__)
    # post_processing/bbox_nms.py: imports mmcv.ops.nms.batched_nms + iou calculators
    nms = I('mmdet.core.post_processing.bbox_nms')
    core = sys.modules['mmdet.core']
    core.__dict__.update(PointGenerator=pg.PointGenerator, build_assigner=b.build_assigner,
                         build_sampler=b.build_sampler, images_to_levels=None, multi_apply=misc.multi_apply,
                         multiclass_nms=nms.multiclass_nms, unmap=misc.unmap, bbox2result=None,
                         bbox_mapping_back=None, merge_aug_proposals=None, bbox_overlaps=iou.bbox_overlaps)
    I('mmdet.models.builder')
    L = sys.modules['mmdet.models.losses']
    acc = I('mmdet.models.losses.accuracy')
    L.accuracy = acc.accuracy
    I('mmdet.models.losses.utils')
    I('mmdet.models.losses.cross_entropy_loss')
    fl = I('mmdet.models.losses.focal_loss')
    sl1 = I('mmdet.models.losses.smooth_l1_loss')
    L.FocalLoss = fl.FocalLoss
    mil = I('mmdet.models.losses.multi_instance_learning_loss')
    mmcv.cnn = sys.modules['mmcv.cnn']
    rl = I('mmdet.models.utils.res_layer')
    sys.modules['mmdet.models.utils'].ResLayer = rl.ResLayer
    rn = I('mmdet.models.backbones.resnet')
    fpn = I('mmdet.models.necks.fpn')
    cm = types.ModuleType('mmdet.utils.contextmanagers')
    cm.completed = None
    sys.modules['mmdet.utils.contextmanagers'] = cm
    sys.modules['mmdet.utils'].contextmanagers = cm
    I('mmdet.models.dense_heads.base_dense_head')
    I('mmdet.models.losses.iou_loss')
    afh = I('mmdet.models.dense_heads.anchor_free_head')
    sys.modules['mmdet.models.dense_heads'].AnchorFreeHead = afh.AnchorFreeHead
    cpr = I('mmdet.models.point.dense_heads.cpr_head')
    p2p = I('mmdet.models.point.dense_heads.p2p_head')

    _LOADED = types.SimpleNamespace(
        ResNet=rn.ResNet, FPN=fpn.FPN, CPRHead=cpr.CPRHead, P2PHead=p2p.P2PHead, MILLoss=mil.MILLoss,
        HungarianAssignerV2=ha.HungarianAssignerV2, PointAssigner=pa.PointAssigner,
        PseudoSampler=ps.PseudoSampler, FocalLossCost=mc.FocalLossCost, DisCostV2=mc.DisCostV2,
        PointGenerator=pg.PointGenerator, multiclass_nms=nms.multiclass_nms, FocalLoss=fl.FocalLoss,
        SmoothL1Loss=sl1.SmoothL1Loss, multi_apply=misc.multi_apply, unmap=misc.unmap,
        cpr_module=cpr, p2p_module=p2p, transforms=tr, batched_nms=_torch_batched_nms)
    return _LOADED


class AttrDict(dict):
    """train_cfg/test_cfg objects the reference heads read by attribute."""
    __getattr__ = dict.get


def init_like_reference(module, seed=0):
    """The stand-in BaseModule.init_weights is a no-op, so apply the init_cfg rules of the hot-path
    classes by hand under a fixed seed: Normal(0, 0.01) for every Conv2d/Linear of the heads
    (cpr_head.py:939-948, p2p_head.py:46-57), cls_out bias = -log((1-0.01)/0.01); Kaiming for the
    backbone convs; Xavier-uniform for the FPN convs (fpn.py:79-80)."""
    g = torch.Generator().manual_seed(seed)
    import math
    for name, m in module.named_modules():
        if isinstance(m, (nn.Conv2d, nn.Linear)):
            w = m.weight
            if 'backbone' in name or name.startswith('layer') or name == 'conv1':
                fan_out = w.shape[0] * (w[0][0].numel() if w.dim() == 4 else 1)
                w.data = torch.randn(w.shape, generator=g) * math.sqrt(2.0 / fan_out)
            elif 'neck' in name or 'lateral' in name or 'fpn_convs' in name:
                rf = w[0][0].numel() if w.dim() == 4 else 1
                a = math.sqrt(6.0 / (w.shape[1] * rf + w.shape[0] * rf))
                w.data = (torch.rand(w.shape, generator=g) * 2 - 1) * a
            else:
                w.data = torch.randn(w.shape, generator=g) * 0.01
            if m.bias is not None:
                m.bias.data.zero_()
                if name.endswith('cls_out'):
                    m.bias.data.fill_(-math.log((1 - 0.01) / 0.01))
    return module
