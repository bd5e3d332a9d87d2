"""Deterministic synthetic weights and inputs for the CPR / P2P hot path.

There is no network in the build or GPU environment, so neither datasets nor the
``torchvision://resnet50`` checkpoint the reference configs name are available.
This module produces (a) a state dict with the reference's key layout
(``backbone.layer{1-4}.{i}.conv{1-3}.weight`` ... ``bbox_head.cls_out.weight``)
and (b) synthetic 640x640 "tiny person" tiles + point annotations with the shapes the
reference pipeline produces (``T/configs2/TinyPersonV2/coarsepointv2/
coarse_point_refine_base_TinyPersonV2_640.py:17-27``: normalised image, Pad(32), 16x16 pseudo
boxes around coarse points).  Pure torch-CPU; no HIP, no oracle import.
"""
import math

import torch

ARCH = {  # T/mmdet/models/backbones/resnet.py:360-366
    18: ('basic', (2, 2, 2, 2)),
    34: ('basic', (3, 4, 6, 3)),
    50: ('bottleneck', (3, 4, 6, 3)),
    101: ('bottleneck', (3, 4, 23, 3)),
    152: ('bottleneck', (3, 8, 36, 3)),
}


def backbone_out_channels(depth):
    kind, _ = ARCH[depth]
    e = 4 if kind == 'bottleneck' else 1
    return [64 * e, 128 * e, 256 * e, 512 * e]


def _bn(sd, prefix, c, g):
    sd[prefix + '.weight'] = torch.rand(c, generator=g) * 0.5 + 0.5
    sd[prefix + '.bias'] = torch.randn(c, generator=g) * 0.1
    sd[prefix + '.running_mean'] = torch.randn(c, generator=g) * 0.1
    sd[prefix + '.running_var'] = torch.rand(c, generator=g) + 0.5
    sd[prefix + '.num_batches_tracked'] = torch.tensor(0, dtype=torch.long)


def _kaiming(shape, g):
    fan_out = shape[0] * shape[2] * shape[3]
    return torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_out)


def _xavier_uniform(shape, g):
    rf = shape[2] * shape[3]
    a = math.sqrt(6.0 / (shape[1] * rf + shape[0] * rf))
    return (torch.rand(shape, generator=g) * 2 - 1) * a


def _gn(sd, prefix, c, g):
    sd[prefix + '.weight'] = torch.rand(c, generator=g) * 0.5 + 0.75
    sd[prefix + '.bias'] = torch.randn(c, generator=g) * 0.1


def resnet_state_dict(depth=50, seed=0, prefix='backbone.'):
    g = torch.Generator().manual_seed(seed)
    kind, blocks = ARCH[depth]
    sd = {}
    sd[prefix + 'conv1.weight'] = _kaiming((64, 3, 7, 7), g)
    _bn(sd, prefix + 'bn1', 64, g)
    inplanes = 64
    for li, nb in enumerate(blocks):
        planes = 64 * 2 ** li
        stride = 1 if li == 0 else 2
        exp = 4 if kind == 'bottleneck' else 1
        for bi in range(nb):
            p = '%slayer%d.%d.' % (prefix, li + 1, bi)
            if kind == 'bottleneck':
                sd[p + 'conv1.weight'] = _kaiming((planes, inplanes, 1, 1), g)
                _bn(sd, p + 'bn1', planes, g)
                sd[p + 'conv2.weight'] = _kaiming((planes, planes, 3, 3), g)
                _bn(sd, p + 'bn2', planes, g)
                sd[p + 'conv3.weight'] = _kaiming((planes * 4, planes, 1, 1), g)
                _bn(sd, p + 'bn3', planes * 4, g)
            else:
                sd[p + 'conv1.weight'] = _kaiming((planes, inplanes, 3, 3), g)
                _bn(sd, p + 'bn1', planes, g)
                sd[p + 'conv2.weight'] = _kaiming((planes, planes, 3, 3), g)
                _bn(sd, p + 'bn2', planes, g)
            if bi == 0 and (stride != 1 or inplanes != planes * exp):
                sd[p + 'downsample.0.weight'] = _kaiming((planes * exp, inplanes, 1, 1), g)
                _bn(sd, p + 'downsample.1', planes * exp, g)
            inplanes = planes * exp
    return sd


def fpn_state_dict(in_channels, out_channels=256, start_level=0, num_outs=1, seed=1, prefix='neck.'):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for j, i in enumerate(range(start_level, len(in_channels))):
        sd['%slateral_convs.%d.conv.weight' % (prefix, j)] = _xavier_uniform((out_channels, in_channels[i], 1, 1), g)
        _gn(sd, '%slateral_convs.%d.gn' % (prefix, j), out_channels, g)
        if i < start_level + num_outs:
            sd['%sfpn_convs.%d.conv.weight' % (prefix, j)] = _xavier_uniform((out_channels, out_channels, 3, 3), g)
            _gn(sd, '%sfpn_convs.%d.gn' % (prefix, j), out_channels, g)
    return sd


def cpr_head_state_dict(num_classes=1, in_channels=256, feat_channels=256, stacked_convs=4, seed=2,
                        prefix='bbox_head.', std=0.01, num_cls_fcs=0, fc_out_channels=1024, binary_ins=False,
                        ins_tower=False, out_bg_cls=False):
    """Normal(0, std) on Conv2d/Linear, cls_out bias = bias_init_with_prob(0.01)
    (T/mmdet/models/point/dense_heads/cpr_head.py:939-948).  ``std`` larger than the reference's
    0.01 makes the synthetic logits spread out (used by tests to exercise the refine filters).
    ins_tower: ins_share_head_feat=False -- a second tower ``ins_convs`` / ``ins_fcs`` (cpr_head.py:992-1008);
    out_bg_cls: one more classifier output (cpr_head.py:953)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    chn = in_channels
    for i in range(stacked_convs):
        sd['%scls_convs.%d.conv.weight' % (prefix, i)] = torch.randn((feat_channels, chn, 3, 3), generator=g) * 0.01
        _gn(sd, '%scls_convs.%d.gn' % (prefix, i), feat_channels, g)
        chn = feat_channels
    for i in range(num_cls_fcs):          # cpr_head.py:999-1005 (ins_share_head_feat: no separate ins_fcs)
        sd['%scls_fcs.%d.weight' % (prefix, i)] = torch.randn((fc_out_channels, chn), generator=g) * (2.0 / chn) ** 0.5
        sd['%scls_fcs.%d.bias' % (prefix, i)] = torch.randn((fc_out_channels,), generator=g) * 0.1
        chn = fc_out_channels
    if ins_tower:
        c2 = in_channels
        for i in range(stacked_convs):
            sd['%sins_convs.%d.conv.weight' % (prefix, i)] = torch.randn((feat_channels, c2, 3, 3), generator=g) * 0.01
            _gn(sd, '%sins_convs.%d.gn' % (prefix, i), feat_channels, g)
            c2 = feat_channels
        for i in range(num_cls_fcs):
            sd['%sins_fcs.%d.weight' % (prefix, i)] = torch.randn((fc_out_channels, c2), generator=g) * (2.0 / c2) ** 0.5
            sd['%sins_fcs.%d.bias' % (prefix, i)] = torch.randn((fc_out_channels,), generator=g) * 0.1
            c2 = fc_out_channels
    n_cls = num_classes + 1 if out_bg_cls else num_classes
    sd[prefix + 'cls_out.weight'] = torch.randn((n_cls, chn), generator=g) * std
    sd[prefix + 'cls_out.bias'] = torch.full((n_cls,), -math.log((1 - 0.01) / 0.01))
    n_ins = n_cls * 2 if binary_ins else n_cls          # cpr_head.py:1009-1011
    sd[prefix + 'ins_out.weight'] = torch.randn((n_ins, chn), generator=g) * std
    sd[prefix + 'ins_out.bias'] = torch.zeros(n_ins)
    return sd


def p2p_head_state_dict(num_classes=1, num_points=1, in_channels=256, feat_channels=256, stacked_convs=4,
                        seed=3, prefix='bbox_head.', std=0.01):
    """T/mmdet/models/point/dense_heads/p2p_head.py:84-105 layer layout."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for tower in ('cls_convs', 'reg_convs'):
        chn = in_channels
        for i in range(stacked_convs):
            sd['%s%s.%d.conv.weight' % (prefix, tower, i)] = torch.randn((feat_channels, chn, 3, 3), generator=g) * 0.01
            _gn(sd, '%s%s.%d.gn' % (prefix, tower, i), feat_channels, g)
            chn = feat_channels
    sd[prefix + 'cls_out.weight'] = torch.randn((num_classes * num_points, feat_channels, 3, 3), generator=g) * std
    sd[prefix + 'cls_out.bias'] = torch.full((num_classes * num_points,), -math.log((1 - 0.01) / 0.01))
    sd[prefix + 'reg_out.weight'] = torch.randn((num_points * 2, feat_channels, 3, 3), generator=g) * std
    sd[prefix + 'reg_out.bias'] = torch.zeros(num_points * 2)
    return sd


def locator_state_dict(depth=50, num_classes=1, start_level=0, head='cpr', seed=0, head_std=0.01, num_points=1,
                       num_cls_fcs=0, fc_out_channels=1024, binary_ins=False, ins_tower=False, out_bg_cls=False):
    sd = resnet_state_dict(depth, seed)
    sd.update(fpn_state_dict(backbone_out_channels(depth), 256, start_level, 1, seed + 1))
    if head == 'cpr':
        sd.update(cpr_head_state_dict(num_classes, seed=seed + 2, std=head_std, num_cls_fcs=num_cls_fcs,
                                      fc_out_channels=fc_out_channels, binary_ins=binary_ins, ins_tower=ins_tower,
                                      out_bg_cls=out_bg_cls))
    else:
        sd.update(p2p_head_state_dict(num_classes, num_points, seed=seed + 3, std=head_std))
    return sd


def synthetic_batch(batch=2, height=640, width=640, num_gts=32, num_classes=1, seed=0, ragged=False):
    """SURVEY.md §8(d) inputs: seed-fixed normalised images, ``num_gts`` coarse points per tile uniform in
    [8, W-8) x [8, H-8), 16x16 pseudo boxes (the dataset's ``pseuw16h16`` annotations), labels, ann ids.
    ``ragged`` gives each image a different number of gts (num_gts, num_gts//2+1, ...)."""
    g = torch.Generator().manual_seed(seed)
    img = torch.randn((batch, 3, height, width), generator=g)
    g2 = torch.Generator().manual_seed(seed + 1)
    gt_bboxes, gt_labels, gt_anns_id, img_metas = [], [], [], []
    next_id = 0
    for b in range(batch):
        n = num_gts if not ragged else max(1, num_gts // (b + 1) + (b % 2))
        xy = torch.rand((n, 2), generator=g2)
        xy[:, 0] = xy[:, 0] * (width - 16) + 8
        xy[:, 1] = xy[:, 1] * (height - 16) + 8
        gt_bboxes.append(torch.cat([xy - 8, xy + 8], dim=1))
        if num_classes == 1:
            gt_labels.append(torch.zeros(n, dtype=torch.long))
        else:
            gt_labels.append(torch.randint(0, num_classes, (n,), generator=g2))
        gt_anns_id.append(torch.arange(next_id, next_id + n, dtype=torch.long))
        next_id += n
        img_metas.append(dict(img_shape=(height, width, 3), pad_shape=(height, width, 3),
                              ori_shape=(height, width, 3), scale_factor=[1.0, 1.0, 1.0, 1.0],
                              filename='synthetic_%d' % b, flip=False))
    return dict(img=img, img_metas=img_metas, gt_bboxes=gt_bboxes, gt_labels=gt_labels, gt_anns_id=gt_anns_id)


def with_refine_points(batch, num_refine, seed=0, jitter=6.0):
    """num_refine > 1 inputs (cpr_head.py:1240-1246): every gt brings R pseudo boxes, gt-major ((num_gts*R, 4) per image):
    refine 0 is the annotated point, the others are seeded perturbations of it (as a previous refinement round would
    supply), kept inside the image.  Returns a copy of ``batch`` with the wider ``gt_bboxes``."""
    if num_refine == 1:
        return batch
    g = torch.Generator().manual_seed(seed + 77)
    out = dict(batch)
    boxes = []
    for b, bb in enumerate(batch['gt_bboxes']):
        h, w = batch['img_metas'][b]['img_shape'][:2]
        ctr = (bb[:, :2] + bb[:, 2:]) / 2
        pts = [ctr]
        for _ in range(num_refine - 1):
            p = ctr + torch.randn(ctr.shape, generator=g) * jitter
            p[:, 0] = p[:, 0].clamp(1, w - 2)
            p[:, 1] = p[:, 1].clamp(1, h - 2)
            pts.append(p)
        pts = torch.stack(pts, dim=1).reshape(-1, 2)                       # (G, R, 2) -> (G*R, 2)
        boxes.append(torch.cat([pts - 8, pts + 8], dim=1))
    out['gt_bboxes'] = boxes
    return out
