"""Generate tests/golden/*.npz by running the REFERENCE's own classes (build container only).

TEST INFRASTRUCTURE ONLY.  Usage:  python -m oracle.gen_golden
Inputs and weights come from ``pointtinybenchmark_amd.synthetic`` (seeded, regenerated at test
time), so the fixtures only hold the reference's OUTPUTS.  The reference modules are built with
the constructor arguments of T/configs2/TinyPersonV2/coarsepointv2/
coarse_point_refine_r50_fpns4_1x_TinyPersonV2_640.py:6-72 and the synthetic state dict is loaded
with strict=True, which also pins the parameter-name layout.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import ref_loader  # noqa: E402
from pointtinybenchmark_amd import synthetic  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
GN = dict(type='GN', num_groups=32, requires_grad=True)


def build_reference_cpr(R, depth=50, num_classes=1, start_level=0, stride=4, radius=5, head_std=0.01, seed=0,
                        policy='independent_with_gt_bag', num_cls_fcs=0, fc_out_channels=1024):
    chans = synthetic.backbone_out_channels(depth)
    backbone = R.ResNet(depth=depth, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                        norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, style='pytorch')
    neck = R.FPN(in_channels=chans, out_channels=256, start_level=start_level, add_extra_convs='on_input',
                 num_outs=1, norm_cfg=GN)
    alpha = 0.25
    head = R.CPRHead(
        norm_cfg=GN, num_classes=num_classes, in_channels=256, feat_channels=256, stacked_convs=4,
        num_cls_fcs=num_cls_fcs, fc_out_channels=fc_out_channels, strides=[stride], loss_mil=dict(type='MILLoss', binary_ins=False, loss_weight=alpha), loss_type=0,
        loss_cfg=dict(with_neg=True, neg_loss_weight=1 - alpha, refine_bag_policy=policy,
                      random_remove_rate=0.4, with_gt_loss=True, gt_loss_weight=alpha, with_mil_loss=True),
        normal_cfg=dict(prob_cls_type='sigmoid', out_bg_cls=False),
        train_pts_extractor=dict(pos_generator=dict(type='CirclePtFeatGenerator', radius=radius),
                                 neg_generator=dict(type='OutCirclePtFeatGenerator', radius=radius, class_wise=True)),
        refine_pts_extractor=dict(pos_generator=dict(type='CirclePtFeatGenerator', radius=radius),
                                  neg_generator=dict(type='OutCirclePtFeatGenerator', radius=radius, keep_wh=True,
                                                     class_wise=True)),
        point_refiner=dict(merge_th=0.1, refine_th=0.1, classify_filter=True),
        test_cfg=ref_loader.AttrDict(nms_pre=2000, min_bbox_size=0, score_thr=0.05,
                                     nms=dict(type='nms', iou_threshold=0.5), max_per_img=1000))
    sd = synthetic.locator_state_dict(depth, num_classes, start_level, 'cpr', seed, head_std, num_cls_fcs=num_cls_fcs,
                                      fc_out_channels=fc_out_channels)
    backbone.load_state_dict({k[len('backbone.'):]: v for k, v in sd.items() if k.startswith('backbone.')}, strict=True)
    neck.load_state_dict({k[len('neck.'):]: v for k, v in sd.items() if k.startswith('neck.')}, strict=True)
    head.load_state_dict({k[len('bbox_head.'):]: v for k, v in sd.items() if k.startswith('bbox_head.')}, strict=True)
    for m in (backbone, neck, head):
        m.train()
    return backbone, neck, head, sd


def run_reference_cpr(R, cfg):
    torch.manual_seed(0)
    backbone, neck, head, sd = build_reference_cpr(R, cfg['depth'], cfg['num_classes'], cfg['start_level'],
                                                   cfg['stride'], cfg['radius'], cfg['head_std'], cfg['seed'],
                                                   num_cls_fcs=cfg.get('num_cls_fcs', 0),
                                                   fc_out_channels=cfg.get('fc_out_channels', 1024))
    batch = synthetic.synthetic_batch(cfg['batch'], cfg['height'], cfg['width'], cfg['num_gts'], cfg['num_classes'],
                                      cfg['seed'], cfg.get('ragged', False))
    with torch.no_grad():
        c = backbone(batch['img'])
        feats = neck(c)
        cls_feat, ins_feat = head(feats)
        losses = head.loss(cls_feat, ins_feat, batch['gt_bboxes'], batch['gt_labels'], batch['img_metas'])
        # points / masks / logits the loss consumed
        gt_points = head.pseudo_bbox_to_center(batch['gt_bboxes'])
        gt_r = [p.reshape(len(l), -1, 2) for p, l in zip(gt_points, batch['gt_labels'])]
        pos, neg = head.train_pts_extractor(cls_feat, ins_feat, gt_r, batch['gt_labels'], batch['img_metas'], None, True)
        pos.cls_outs, pos.ins_outs = head.get_pts_outs(pos.cls_feats, pos.ins_feats)
        neg.cls_outs = head.get_pts_outs(neg.cls_feats)
        dets = head.get_bboxes(cls_feat, ins_feat, batch['img_metas'], rescale=False, gt_bboxes=batch['gt_bboxes'],
                               gt_labels=batch['gt_labels'], gt_anns_id=batch['gt_anns_id'])
    out = dict(
        c2_sample=c[0][:, ::37, ::5, ::7].numpy(), c5_sample=c[3][:, ::101, ::3, ::3].numpy(),
        fpn_sample=feats[0][:, ::17, ::5, ::7].numpy(),
        cls_feat_sample=cls_feat[0][:, ::13, ::3, ::5].numpy(),
        cls_feat_sum=np.float64(cls_feat[0].double().sum().item()),
        cls_feat_abs_sum=np.float64(cls_feat[0].double().abs().sum().item()),
        pos_pts=pos.pts[0].numpy(), pos_valid=pos.valid[0].numpy(),
        pos_cls_logit=pos.cls_outs[0].numpy(), pos_ins_logit=pos.ins_outs[0].numpy(),
        neg_valid=np.packbits(neg.valid[0].numpy().astype(np.uint8), axis=None),
        neg_valid_shape=np.array(neg.valid[0].shape), neg_valid_count=np.int64(neg.valid[0].sum().item()),
        neg_logit_sample=neg.cls_outs[0][::97].numpy(),
        dets=np.concatenate([d[0].numpy() for d in dets]),
        det_labels=np.concatenate([d[1].numpy() for d in dets]))
    for k, v in losses.items():
        out['loss_' + k] = np.float32(float(v))
    return out


CPR_CASES = {
    # name: config.  Small spatial sizes so the CPU suite runs in seconds; same code path as 640x640.
    'cpr_r50_c1_160': dict(depth=50, num_classes=1, start_level=0, stride=4, radius=5, head_std=0.01, seed=0, batch=2,
                           height=160, width=160, num_gts=6),
    'cpr_r50_c1_160_spread': dict(depth=50, num_classes=1, start_level=0, stride=4, radius=5, head_std=0.3, seed=3,
                                  batch=2, height=160, width=192, num_gts=7, ragged=True),
    'cpr_r18_c3_128': dict(depth=18, num_classes=3, start_level=0, stride=4, radius=5, head_std=0.3, seed=5, batch=2,
                           height=128, width=128, num_gts=9, ragged=True),
    # num_cls_fcs > 0: two shared FC layers between the sampled features and the classifiers
    'cpr_r18_c3_fc2': dict(depth=18, num_classes=3, start_level=0, stride=4, radius=5, head_std=0.05, seed=11, batch=2,
                           height=128, width=160, num_gts=7, ragged=True, num_cls_fcs=2, fc_out_channels=64),
    'cpr_r50_c80_s8_r8': dict(depth=50, num_classes=80, start_level=1, stride=8, radius=8, head_std=0.3, seed=7,
                              batch=1, height=224, width=256, num_gts=8),
}


# round 5: + the configs[2] family (80 classes, start_level 1, stride 8, radius 8) -- the config whose BASELINE line is the
# gradient all-reduce
GRAD_CASES = ('cpr_r18_c3_128', 'cpr_r50_c1_160_spread', 'cpr_r50_c80_s8_r8')


def grad_sample_index(numel, k=256):
    """Deterministic flat indices every consumer of the gradient fixtures re-derives (<= k entries, evenly spread)."""
    return np.unique(np.linspace(0, numel - 1, min(k, numel)).round().astype(np.int64))


def run_reference_cpr_grads(R, cfg):
    """loss.backward() through the REFERENCE's own modules (torch autograd on CPU): the total of every key containing
    'loss' (BaseDetector._parse_losses, base.py:179-212).  Per trainable tensor: L2 norm, sum and a strided sample."""
    torch.manual_seed(0)
    backbone, neck, head, sd = build_reference_cpr(R, cfg['depth'], cfg['num_classes'], cfg['start_level'],
                                                   cfg['stride'], cfg['radius'], cfg['head_std'], cfg['seed'])
    batch = synthetic.synthetic_batch(cfg['batch'], cfg['height'], cfg['width'], cfg['num_gts'], cfg['num_classes'],
                                      cfg['seed'], cfg.get('ragged', False))
    cls_feat, ins_feat = head(neck(backbone(batch['img'])))
    losses = head.loss(cls_feat, ins_feat, batch['gt_bboxes'], batch['gt_labels'], batch['img_metas'])
    total = sum(v for k, v in losses.items() if 'loss' in k)
    total.backward()
    out = {'total_loss': np.float64(float(total))}
    n = 0
    for prefix, mod in (('backbone.', backbone), ('neck.', neck), ('bbox_head.', head)):
        for name, prm in mod.named_parameters():
            if not prm.requires_grad:
                continue
            assert prm.grad is not None, prefix + name
            g = prm.grad.detach().double().flatten()
            key = prefix + name
            out['norm:' + key] = np.float64(float(g.norm()))
            out['sum:' + key] = np.float64(float(g.sum()))
            out['sample:' + key] = g[torch.from_numpy(grad_sample_index(g.numel()))].numpy().astype(np.float32)
            n += 1
    out['num_tensors'] = np.int64(n)
    return out


def gen_cpr_grads(R):
    for name in GRAD_CASES:
        out = run_reference_cpr_grads(R, CPR_CASES[name])
        np.savez_compressed(os.path.join(GOLDEN, 'cpr_grads_' + name + '.npz'), **out)
        print('grads', name, 'tensors', int(out['num_tensors']), 'loss', float(out['total_loss']))


def gen_cpr(R):
    for name, cfg in CPR_CASES.items():
        out = run_reference_cpr(R, cfg)
        np.savez_compressed(os.path.join(GOLDEN, name + '.npz'), **out)
        print(name, {k: (v.shape if hasattr(v, 'shape') and v.shape else v) for k, v in out.items()
                     if k.startswith('loss') or k in ('dets', 'neg_valid_count')})


def assigner_inputs(seed, n_side=40, stride=4, G=7, C=1, spread=1.0):
    """Shared generator for the HungarianAssignerV2 fixtures (also used by the tests)."""
    g = torch.Generator().manual_seed(seed)
    ys, xs = torch.meshgrid(torch.arange(n_side), torch.arange(n_side), indexing='ij')
    anchors = torch.stack([xs.flatten(), ys.flatten()], -1).float() * stride
    pred = anchors + torch.randn(anchors.shape, generator=g) * stride * spread
    logits = torch.randn((anchors.shape[0], C), generator=g) * 2 - 3
    gt = torch.rand((G, 2), generator=g) * (n_side * stride - 16) + 8
    labels = torch.randint(0, C, (G,), generator=g)
    return pred, logits, gt, labels, (n_side * stride, n_side * stride, 3)


def gen_assigners(R):
    out = {}
    # PointAssigner random cases (multi-level points, like RepPoints would feed it)
    for seed in range(6):
        g = torch.Generator().manual_seed(100 + seed)
        pts = []
        for s in (8, 16, 32):
            n = 128 // s * 2
            ys, xs = torch.meshgrid(torch.arange(n), torch.arange(n), indexing='ij')
            pts.append(torch.stack([xs.flatten() * s, ys.flatten() * s, torch.full((n * n,), s)], -1).float())
        pts = torch.cat(pts)
        k = 3 + seed * 2
        xy = torch.rand((k, 2), generator=g) * 200 + 10
        wh = torch.rand((k, 2), generator=g) * 90 + 4
        gtb = torch.cat([xy - wh / 2, xy + wh / 2], 1)
        gl = torch.randint(0, 5, (k,), generator=g)
        res = R.PointAssigner(scale=4, pos_num=3).assign(pts, gtb, None, gl)
        out['pa%d_gt_inds' % seed] = res.gt_inds.numpy()
        out['pa%d_labels' % seed] = res.labels.numpy()
    # HungarianAssignerV2 (P2P config costs)
    for seed, (n_side, G, C, k) in enumerate([(40, 7, 1, 5), (40, 7, 1, 1), (32, 20, 3, 5), (24, 100, 1, 5),
                                               (160, 32, 1, 5)]):
        pred, logits, gt, labels, shp = assigner_inputs(200 + seed, n_side, 4, G, C)
        ha = R.HungarianAssignerV2(cls_costs=dict(type='FocalLossCost', weight=2.0),
                                   reg_costs=dict(type='DisCostV2', weight=0.1, norm_with_img_wh=False), topk_k=k)
        res = ha.assign(pred, logits, gt, labels, dict(img_shape=shp))
        out['ha%d_gt_inds' % seed] = res.gt_inds.numpy().astype(np.int32)
        out['ha%d_labels' % seed] = res.labels.numpy().astype(np.int32)
        out['ha%d_cfg' % seed] = np.array([n_side, G, C, k])
    np.savez_compressed(os.path.join(GOLDEN, 'assigners.npz'), **out)
    print('assigners', len(out))


def build_reference_p2p(R, num_classes=1, head_std=0.01, seed=0):
    head = R.P2PHead(
        norm_cfg=GN, num_classes=num_classes, in_channels=256, feat_channels=256, stacked_convs=4, strides=[4],
        point_anchor=[(0., 0.)],
        loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
        loss_reg=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=0.5), pts_gamma=1, reg_norm=1,
        train_cfg=ref_loader.AttrDict(
            neg_weight=1.0,
            assigner=dict(type='HungarianAssignerV2', cls_costs=dict(type='FocalLossCost', weight=2.0),
                          reg_costs=dict(type='DisCostV2', weight=0.1, norm_with_img_wh=False), topk_k=5),
            sampler=dict(type='PseudoSampler')),
        test_cfg=ref_loader.AttrDict(nms_pre=2000, min_bbox_size=0, score_thr=0.05, pseudo_wh=(16, 16),
                                     nms=dict(type='nms', iou_threshold=0.2), max_per_img=1000))
    sd = synthetic.p2p_head_state_dict(num_classes, 1, seed=seed + 3, std=head_std)
    head.load_state_dict({k[len('bbox_head.'):]: v for k, v in sd.items()}, strict=True)
    head.train()
    return head, sd


def gen_p2p(R):
    out = {}
    for ci, (C, hw, G, std) in enumerate([(1, 48, 6, 0.05), (2, 40, 9, 0.08)]):
        head, sd = build_reference_p2p(R, C, std, seed=ci)
        g = torch.Generator().manual_seed(300 + ci)
        feat = torch.randn((2, 256, hw, hw), generator=g)
        batch = synthetic.synthetic_batch(2, hw * 4, hw * 4, G, C, seed=300 + ci, ragged=True)
        with torch.no_grad():
            cls_outs, pts_outs = head((feat,))
            losses = head.loss(cls_outs, pts_outs, batch['gt_bboxes'], batch['gt_labels'], batch['img_metas'],
                               gt_bboxes_ignore=[torch.zeros((0, 4)) for _ in batch['gt_labels']])
            anchor, pred, vflag, co = head.get_pred_points(cls_outs, pts_outs, batch['img_metas'])
            gtp = head.pseudo_bbox_to_center(batch['gt_bboxes'])
            lab, lw, bgt, pw = head.get_targets(pred[..., :2], vflag, co, gtp, batch['gt_labels'], batch['img_metas'])
            res = head.get_bboxes(cls_outs, pts_outs, batch['img_metas'])
        out['p2p%d_cls_out' % ci] = cls_outs[0].numpy()
        out['p2p%d_pts_out' % ci] = pts_outs[0].numpy()
        out['p2p%d_loss_cls' % ci] = np.float32(float(losses['loss_cls'][0]) if isinstance(losses['loss_cls'], (list, tuple)) else float(losses['loss_cls']))
        out['p2p%d_loss_pts' % ci] = np.float32(float(losses['loss_pts'][0]) if isinstance(losses['loss_pts'], (list, tuple)) else float(losses['loss_pts']))
        out['p2p%d_target_labels' % ci] = torch.stack(lab).numpy().astype(np.int32)
        out['p2p%d_target_pts' % ci] = torch.stack(bgt).numpy()
        for b, (bs, l) in enumerate(res):
            out['p2p%d_det%d' % (ci, b)] = bs.numpy()
            out['p2p%d_detlabel%d' % (ci, b)] = l.numpy()
        out['p2p%d_cfg' % ci] = np.array([C, hw, G, int(std * 1000)])
    np.savez_compressed(os.path.join(GOLDEN, 'p2p.npz'), **out)
    print('p2p', {k: v.shape for k, v in out.items() if 'det0' in k or 'loss' in k})


def gen_p2p_grads(R):
    """loss.backward() through the reference's own P2PHead (towers, output convs, focal + SmoothL1 on its scipy Hungarian
    assignment): gradients of every head parameter and of the input feature map, as norm / sum / strided sample."""
    out = {}
    for ci, (C, hw, G, std) in enumerate([(1, 48, 6, 0.05), (2, 40, 9, 0.08)]):
        head, sd = build_reference_p2p(R, C, std, seed=ci)
        head.train()
        g = torch.Generator().manual_seed(300 + ci)
        feat = torch.randn((2, 256, hw, hw), generator=g).requires_grad_(True)
        batch = synthetic.synthetic_batch(2, hw * 4, hw * 4, G, C, seed=300 + ci, ragged=True)
        cls_outs, pts_outs = head((feat,))
        losses = head.loss(cls_outs, pts_outs, batch['gt_bboxes'], batch['gt_labels'], batch['img_metas'],
                           gt_bboxes_ignore=[torch.zeros((0, 4)) for _ in batch['gt_labels']])
        total = sum(sum(v) if isinstance(v, (list, tuple)) else v for k, v in losses.items() if 'loss' in k)
        total.backward()
        out['p2p%d_total_loss' % ci] = np.float64(float(total.detach()))
        tensors = [('bbox_head.' + n, p.grad) for n, p in head.named_parameters()] + [('feat', feat.grad)]
        for key, gr in tensors:
            gr = gr.detach().double().flatten()
            out['p2p%d_norm:%s' % (ci, key)] = np.float64(float(gr.norm()))
            out['p2p%d_sample:%s' % (ci, key)] = gr[torch.from_numpy(grad_sample_index(gr.numel()))].numpy().astype(np.float32)
        out['p2p%d_cfg' % ci] = np.array([C, hw, G, int(std * 1000)])
    np.savez_compressed(os.path.join(GOLDEN, 'p2p_grads.npz'), **out)
    print('p2p grads', {k: float(v) for k, v in out.items() if 'total' in k})


def main():
    assert ref_loader.available(), 'needs /root/reference'
    os.makedirs(GOLDEN, exist_ok=True)
    torch.set_num_threads(8)
    R = ref_loader.load()
    which = sys.argv[1:] or ['cpr', 'grads', 'assigners', 'p2p', 'p2p_grads']
    if 'cpr' in which:
        gen_cpr(R)
    if 'grads' in which:
        gen_cpr_grads(R)
    if 'assigners' in which:
        gen_assigners(R)
    if 'p2p' in which:
        gen_p2p(R)
    if 'p2p_grads' in which:
        gen_p2p_grads(R)


if __name__ == '__main__':
    main()
