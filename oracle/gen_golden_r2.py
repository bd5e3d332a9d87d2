"""Round-2 fixtures from the REFERENCE's own classes (build container only).  TEST INFRASTRUCTURE ONLY.

  python -m oracle.gen_golden_r2 [refine] [options] [costs] [p2p_edge] [options_r3] [ha_transposed]

  refine   tests/golden/refine.npz        PointRefiner internals (refine points, scores, not_refine, the chosen-point
                                          masks) of every CPR case + the out_geo / cascade_out_fmt / not_refine-input /
                                          rescale output formats of CPRHead.get_bboxes (cpr_head.py:780-864,1231-1283)
  options  tests/golden/cpr_options.npz   the CPRHead options no shipped config uses: num_refine > 1
                                          with the three bag policies / two gt_loss_type values, GridCirclesPtFeatGenerator,
                                          softmax / normed_sigmoid probabilities, binary_ins, AllPosLoss
  costs    tests/golden/assigner_costs.npz  the reference's own fp32 cost matrices of the HungarianAssignerV2 fixtures
                                          (FocalLossCost + DisCostV2 evaluated by the reference classes on THIS host)
  p2p_edge tests/golden/p2p_edge.npz      P2PHead targets / losses on a batch with invalid feature-map cells (an image padded
                                          less than the batch maximum) and with a gt-less image (p2p_head.py:275-328,451-463)
  options_r3    tests/golden/cpr_options_r3.npz  (round 3) return_score_type='max', AnchorPtFeatGenerator(scale_factor) as the
                                          refine-time grid generator, ins_share_head_feat=False (with and without FC layers),
                                          out_bg_cls=True for one class (cpr_head.py:206-232,840-842,953,992-1008,1037,1068)
  ha_transposed tests/golden/assigner_transposed.npz  HungarianAssignerV2, topk_k = 1, fewer proposals than gts
                                          (hungarian_assigner.py:229-240)
Inputs and weights come from ``pointtinybenchmark_amd.synthetic`` (seeded, regenerated at test time)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import ref_loader  # noqa: E402
from oracle.gen_golden import CPR_CASES, GN, GOLDEN, assigner_inputs  # noqa: E402
from pointtinybenchmark_amd import synthetic  # noqa: E402

# Round 3: refine cases beyond CPR_CASES (tests/golden/refine.npz only).  The C = 80 / stride 8 / radius 8 case of CPR_CASES
# leaves every gt 'not_refine' (the 80-way classify filter rejects them all): this one reaches the K = 289, C = 80 merge
# branch of PointRefiner.refine_single with 576 chosen points on 4 of its 13 gts.
REFINE_EXTRA_CASES = {
    'cpr_r50_c80_s8_r8_live': dict(depth=50, num_classes=80, start_level=1, stride=8, radius=8, head_std=1.5, seed=35,
                                   batch=2, height=224, width=256, num_gts=8, ragged=True),
}
REFINE_CASES = dict(CPR_CASES, **REFINE_EXTRA_CASES)

HA_CASES = [(40, 7, 1, 5), (40, 7, 1, 1), (32, 20, 3, 5), (24, 100, 1, 5), (160, 32, 1, 5)]   # = gen_golden.gen_assigners

BASE = dict(depth=18, num_classes=3, start_level=0, stride=4, radius=5, head_std=0.3, seed=41, batch=2, height=128,
            width=160, num_gts=7, ragged=True)
# name: overrides of BASE + head options
OPTION_CASES = {
    'r2_independent': dict(num_refine=2, policy='independent_with_gt_bag', gt_loss_type='gt_refine'),
    'r2_merge_gt': dict(num_refine=2, policy='merge_to_gt_bag', gt_loss_type='gt', seed=42),
    'r3_only_refine': dict(num_refine=3, policy='only_refine_bag', gt_loss_type='gt_refine', seed=43),
    'softmax': dict(prob='softmax', seed=44),
    'normed_sigmoid_p1': dict(prob='normed_sigmoid', norm_p=1, seed=45),
    'normed_sigmoid_p2': dict(prob='normed_sigmoid', norm_p=2, seed=46),
    'binary_ins': dict(binary_ins=True, seed=47),
    'allpos': dict(loss='AllPosLoss', seed=48),
    'grid_circles': dict(pos='GridCirclesPtFeatGenerator', radius=3, seed=49),
    'grid_circles_r2': dict(pos='GridCirclesPtFeatGenerator', radius=3, num_refine=2, max_pos_num=160, seed=50),
    'no_mil_loss': dict(with_mil_loss=False, seed=51),
}


# Round 3 (tests/golden/cpr_options_r3.npz): the options the round-2 head still asserted out
OPTION_CASES_R3 = {
    'score_max': dict(score_type='max', seed=52),                                   # cpr_head.py:840-842
    'anchor_scale': dict(refine_neg='AnchorPtFeatGenerator', anchor_scale=0.5, seed=53),   # cpr_head.py:206-232
    'ins_tower': dict(ins_tower=True, seed=54),                                     # ins_share_head_feat=False, :992-1008,1037,1068
    'ins_tower_fc': dict(ins_tower=True, num_cls_fcs=1, fc_out_channels=64, seed=55),
    'bg_cls': dict(num_classes=1, out_bg_cls=True, seed=56),                        # cpr_head.py:953
    'align_corners': dict(align_corners=True, seed=57),                             # cpr_head.py:73-93,126
    'align_corners_grid': dict(align_corners=True, pos='GridCirclesPtFeatGenerator', radius=3, seed=58),
}


def option_cfg(name):
    cfg = dict(BASE)
    cfg.update(OPTION_CASES[name] if name in OPTION_CASES else OPTION_CASES_R3[name])
    return cfg


def cpr_head_kwargs(cfg):
    """CPRHead constructor arguments of a case -- plain dicts, accepted unchanged by the reference's CPRHead and by
    pointtinybenchmark_amd's (the tests build the HIP head from the same dict)."""
    alpha, r = 0.25, cfg['radius']
    pos = dict(type=cfg.get('pos', 'CirclePtFeatGenerator'), radius=r)
    if 'max_pos_num' in cfg:
        pos['max_pos_num'] = cfg['max_pos_num']
    if cfg.get('align_corners', False):
        pos['align_corners'] = True
    normal = dict(prob_cls_type=cfg.get('prob', 'sigmoid'), out_bg_cls=cfg.get('out_bg_cls', False))
    if 'norm_p' in cfg:
        normal['normed_sigmoid_p'] = cfg['norm_p']
    loss_cfg = dict(with_neg=cfg.get('with_neg', True), neg_loss_weight=1 - alpha, refine_bag_policy=cfg.get('policy', 'independent_with_gt_bag'),
                    random_remove_rate=0.4, with_gt_loss=True, gt_loss_weight=alpha,
                    with_mil_loss=cfg.get('with_mil_loss', True))
    if 'gt_loss_type' in cfg:
        loss_cfg['gt_loss_type'] = cfg['gt_loss_type']
    refine_neg = dict(type='OutCirclePtFeatGenerator', radius=r, keep_wh=True, class_wise=True)
    if cfg.get('refine_neg') == 'AnchorPtFeatGenerator':
        refine_neg = dict(type='AnchorPtFeatGenerator', scale_factor=cfg.get('anchor_scale', 1.0))
    refiner = dict(merge_th=0.1, refine_th=0.1, classify_filter=True)
    if 'score_type' in cfg:
        refiner['return_score_type'] = cfg['score_type']
    return dict(
        ins_share_head_feat=not cfg.get('ins_tower', False),
        norm_cfg=GN, num_classes=cfg['num_classes'], in_channels=256, feat_channels=256, stacked_convs=4,
        num_cls_fcs=cfg.get('num_cls_fcs', 0), fc_out_channels=cfg.get('fc_out_channels', 1024), strides=[cfg['stride']],
        loss_mil=dict(type=cfg.get('loss', 'MILLoss'), binary_ins=cfg.get('binary_ins', False), loss_weight=alpha),
        loss_type=0, loss_cfg=loss_cfg, normal_cfg=normal,
        train_pts_extractor=dict(pos_generator=dict(pos),
                                 neg_generator=dict(type='OutCirclePtFeatGenerator', radius=r, class_wise=True)),
        refine_pts_extractor=dict(pos_generator=dict(pos), neg_generator=refine_neg),
        point_refiner=refiner)


def case_inputs(cfg):
    sd = synthetic.locator_state_dict(cfg['depth'], cfg['num_classes'], cfg['start_level'], 'cpr', cfg['seed'],
                                      cfg['head_std'], num_cls_fcs=cfg.get('num_cls_fcs', 0),
                                      fc_out_channels=cfg.get('fc_out_channels', 1024),
                                      binary_ins=cfg.get('binary_ins', False), ins_tower=cfg.get('ins_tower', False),
                                      out_bg_cls=cfg.get('out_bg_cls', False))
    batch = synthetic.synthetic_batch(cfg['batch'], cfg['height'], cfg['width'], cfg['num_gts'], cfg['num_classes'],
                                      cfg['seed'], cfg.get('ragged', False))
    batch = synthetic.with_refine_points(batch, cfg.get('num_refine', 1), cfg['seed'])
    return sd, batch


def not_refine_input(batch, seed):
    """A seeded not_refine input for get_bboxes (one bool per gt), shared with the tests."""
    g = torch.Generator().manual_seed(seed + 500)
    return [torch.rand(len(l), generator=g) < 0.3 for l in batch['gt_labels']]


def build_reference(R, cfg):
    chans = synthetic.backbone_out_channels(cfg['depth'])
    backbone = R.ResNet(depth=cfg['depth'], num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                        norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, style='pytorch')
    neck = R.FPN(in_channels=chans, out_channels=256, start_level=cfg['start_level'], add_extra_convs='on_input',
                 num_outs=1, norm_cfg=GN)
    head = R.CPRHead(**cpr_head_kwargs(cfg),
                     test_cfg=ref_loader.AttrDict(nms_pre=2000, min_bbox_size=0, score_thr=0.05,
                                                  nms=dict(type='nms', iou_threshold=0.5), max_per_img=1000))
    sd, batch = case_inputs(cfg)
    backbone.load_state_dict({k[len('backbone.'):]: v for k, v in sd.items() if k.startswith('backbone.')}, strict=True)
    neck.load_state_dict({k[len('neck.'):]: v for k, v in sd.items() if k.startswith('neck.')}, strict=True)
    head.load_state_dict({k[len('bbox_head.'):]: v for k, v in sd.items() if k.startswith('bbox_head.')}, strict=True)
    for m in (backbone, neck, head):
        m.train()
    return backbone, neck, head, batch


class RefineCapture:
    """Records what PointRefiner.refine_single returns for every image (cpr_head.py:780-850) and turns its chosen-point
    LISTS back into (num_gts, num_refine*num_chosen) masks by matching them, in order, against the bag points."""

    def __init__(self, head):
        self.rec = []
        pr = head.point_refiner
        orig = pr.refine_single

        def wrapped(bag_data, grid_data, gt_r_points, gt_labels, img_meta, gt_true_bboxes, not_refine=None):
            out = orig(bag_data, grid_data, gt_r_points, gt_labels, img_meta, gt_true_bboxes, not_refine)
            refine_pts, scores, nr, chosen_pts, geos = out
            flat = bag_data.pts[0].reshape(len(gt_labels), -1, 3)
            mask = torch.zeros(flat.shape[:2], dtype=torch.bool)
            for i, cp in enumerate(chosen_pts):
                j = 0
                for q in cp:
                    while not bool((flat[i, j] == q).all()):
                        j += 1
                    mask[i, j] = True
                    j += 1
            self.rec.append(dict(refine_pts=refine_pts.clone(), scores=scores.clone(), not_refine=nr.clone(), chosen=mask,
                                 bag_pts=flat[..., :2].clone(), bag_valid=bag_data.valid[0].reshape(len(gt_labels), -1).clone(),
                                 bag_cls_logit=bag_data.cls_outs[0].reshape(len(gt_labels), flat.shape[1], -1).clone()))
            return out
        pr.refine_single = wrapped

    def collect(self):
        rec, self.rec = self.rec, []
        return {k: torch.cat([r[k] for r in rec]) for k in rec[0]}


def _refine_outputs(head, cls_feat, ins_feat, batch, seed, prefix, out):
    """All get_bboxes output formats of one case into ``out`` (keys prefixed)."""
    cap = RefineCapture(head)
    kw = dict(gt_bboxes=batch['gt_bboxes'], gt_labels=batch['gt_labels'], gt_anns_id=batch['gt_anns_id'])
    metas = batch['img_metas']
    dets = head.get_bboxes(cls_feat, ins_feat, metas, rescale=False, **kw)
    c = cap.collect()
    out[prefix + 'dets'] = np.concatenate([d[0].numpy() for d in dets])
    out[prefix + 'det_labels'] = np.concatenate([d[1].numpy() for d in dets])
    out[prefix + 'refine_pts'] = c['refine_pts'].numpy()
    out[prefix + 'scores'] = c['scores'].numpy()
    out[prefix + 'not_refine'] = c['not_refine'].numpy()
    out[prefix + 'chosen'] = np.packbits(c['chosen'].numpy().astype(np.uint8), axis=None)
    out[prefix + 'chosen_shape'] = np.array(c['chosen'].shape)
    out[prefix + 'bag_pts'] = c['bag_pts'].numpy()
    out[prefix + 'bag_valid'] = np.packbits(c['bag_valid'].numpy().astype(np.uint8), axis=None)
    out[prefix + 'bag_cls_logit'] = c['bag_cls_logit'].numpy()
    # out_geo (cpr_head.py:1268-1269): the rows of different images have different widths -> one array per image
    head.other_info = dict(out_geo=True)
    dg = head.get_bboxes(cls_feat, ins_feat, metas, rescale=False, **kw)
    cap.collect()
    for b, d in enumerate(dg):
        out[prefix + 'dets_geo%d' % b] = d[0].numpy()
    # rescale=True with a non-unit scale factor (boxes and geo points are divided, cpr_head.py:1259-1262,1285-1288)
    metas2 = [dict(m, scale_factor=[1.25, 1.6, 1.25, 1.6]) for m in metas]
    dr = head.get_bboxes(cls_feat, ins_feat, metas2, rescale=True, **kw)
    cap.collect()
    for b, d in enumerate(dr):
        out[prefix + 'dets_geo_rescaled%d' % b] = d[0].numpy()
    head.other_info = dict()
    # not_refine input + cascade_out_fmt (cpr_head.py:839,1273-1274)
    nr_in = not_refine_input(batch, seed)
    dc, nr_out = head.get_bboxes(cls_feat, ins_feat, metas, rescale=False, not_refine=nr_in, cascade_out_fmt=True, **kw)
    cap.collect()
    out[prefix + 'cascade_dets'] = np.concatenate([d[0].numpy() for d in dc])
    out[prefix + 'cascade_not_refine'] = np.concatenate([n.numpy() for n in nr_out])


def gen_refine(R):
    out = {}
    for name, cfg in REFINE_CASES.items():
        torch.manual_seed(0)
        backbone, neck, head, batch = build_reference(R, dict(cfg))
        with torch.no_grad():
            cls_feat, ins_feat = head(neck(backbone(batch['img'])))
            _refine_outputs(head, cls_feat, ins_feat, batch, cfg['seed'], name + ':', out)
        print('refine', name, 'gts', len(out[name + ':scores']), 'chosen', int(np.unpackbits(out[name + ':chosen']).sum()),
              'not_refine', int(out[name + ':not_refine'].sum()))
    np.savez_compressed(os.path.join(GOLDEN, 'refine.npz'), **out)


def gen_options(R, cases=None, fname='cpr_options.npz'):
    cases = OPTION_CASES if cases is None else cases
    out = {}
    for name in cases:
        cfg = option_cfg(name)
        torch.manual_seed(0)
        backbone, neck, head, batch = build_reference(R, cfg)
        if cfg.get('refine_neg') == 'AnchorPtFeatGenerator' and cfg.get('anchor_scale', 1.0) != 1.0:
            # AnchorPtFeatGenerator hands scale_factor to F.interpolate as its SECOND POSITIONAL argument, which is `size`
            # (cpr_head.py:229-231): the reference cannot run a scale_factor other than None / 1.0 -- record what it raises
            try:
                with torch.no_grad():
                    cls_feat, ins_feat = head(neck(backbone(batch['img'])))
                    head.get_bboxes(cls_feat, ins_feat, batch['img_metas'], gt_bboxes=batch['gt_bboxes'],
                                    gt_labels=batch['gt_labels'], gt_anns_id=batch['gt_anns_id'])
                out[name + ':reference_error'] = np.array('none')
            except Exception as e:  # noqa: BLE001
                out[name + ':reference_error'] = np.array('%s: %s' % (type(e).__name__, str(e)[:160]))
            print('options', name, 'reference ->', out[name + ':reference_error'])
            continue
        with torch.no_grad():
            cls_feat, ins_feat = head(neck(backbone(batch['img'])))
            losses = head.loss(cls_feat, ins_feat, batch['gt_bboxes'], batch['gt_labels'], batch['img_metas'])
            gt_points = head.pseudo_bbox_to_center(batch['gt_bboxes'])
            gt_r = [p.reshape(len(l), -1, 2) for p, l in zip(gt_points, batch['gt_labels'])]
            pos, neg = head.train_pts_extractor(cls_feat, ins_feat, gt_r, batch['gt_labels'], batch['img_metas'], None,
                                                head.ins_share_head_feat)
            pos.cls_outs, pos.ins_outs = head.get_pts_outs(pos.cls_feats, pos.ins_feats)
            p = name + ':'
            for k, v in losses.items():
                out[p + 'loss_' + k] = np.float32(float(v.mean()))      # AllPosLoss returns a tensor; _parse_losses means it
            out[p + 'pos_pts'] = pos.pts[0][..., :2].numpy()
            out[p + 'pos_valid'] = pos.valid[0][..., 0].numpy()
            out[p + 'pos_cls_logit'] = pos.cls_outs[0].numpy()
            out[p + 'pos_ins_logit'] = pos.ins_outs[0].numpy()
            out[p + 'neg_valid'] = np.packbits(neg.valid[0].numpy().astype(np.uint8), axis=None)
            out[p + 'neg_valid_count'] = np.int64(neg.valid[0].sum().item())
            try:
                _refine_outputs(head, cls_feat, ins_feat, batch, cfg['seed'], p, out)
            except AssertionError:
                # PointRefiner.refine_single asserts that every refine point of a gt coincides with its first one when the
                # bags come from CirclePtFeatGenerator (cpr_head.py:809): get_bboxes cannot run on such num_refine > 1
                # inputs in the reference -> loss only (the grid generators keep one annotated point per bag and do run)
                assert cfg.get('num_refine', 1) > 1 and 'pos' not in cfg
                for k in [k for k in out if k.startswith(p) and not (k.startswith(p + 'loss_') or k.startswith(p + 'pos_')
                                                                      or k.startswith(p + 'neg_'))]:
                    del out[k]
                out[p + 'refine_asserts_in_reference'] = np.array(True)
        print('options', name, {k[len(p):]: float(v) for k, v in out.items() if k.startswith(p + 'loss_')},
              'pos', out[p + 'pos_pts'].shape, 'chosen', int(np.unpackbits(out[p + 'chosen']).sum()) if p + 'chosen' in out else 'n/a')
    if cases is not OPTION_CASES:
        np.savez_compressed(os.path.join(GOLDEN, fname), **out)
        return
    # the one generator that cannot run in the reference (documented in pointtinybenchmark_amd/dense_heads/cpr_head.py)
    try:
        cfg = dict(BASE, pos='GridEllipsePtFeatGenerator', num_refine=2)
        kw = cpr_head_kwargs(cfg)
        for ex in ('train_pts_extractor', 'refine_pts_extractor'):
            kw[ex]['pos_generator'] = dict(type='GridEllipsePtFeatGenerator', a_minus_c=2.0)
        head = R.CPRHead(**kw)
        sd, batch = case_inputs(cfg)
        feat = torch.randn(2, 256, 32, 40)
        head.loss([feat], [feat], batch['gt_bboxes'], batch['gt_labels'], batch['img_metas'])
        out['grid_ellipse_reference_error'] = np.array('none')
    except Exception as e:  # noqa: BLE001
        out['grid_ellipse_reference_error'] = np.array('%s: %s' % (type(e).__name__, str(e)[:200]))
    print('GridEllipsePtFeatGenerator in the reference ->', out['grid_ellipse_reference_error'])
    np.savez_compressed(os.path.join(GOLDEN, 'cpr_options.npz'), **out)


def gen_assigner_costs(R):
    """cost = sum(cls_costs) + sum(reg_costs) exactly as HungarianAssignerV2.assign forms it (hungarian_assigner.py:222-227),
    from the reference's own FocalLossCost / DisCostV2 objects; float32 bits as computed on the build host."""
    out = {}
    for case, (n_side, G, C, k) in enumerate(HA_CASES):
        pred, logits, gt, labels, shp = assigner_inputs(200 + case, n_side, 4, G, C)
        ha = R.HungarianAssignerV2(cls_costs=dict(type='FocalLossCost', weight=2.0),
                                   reg_costs=dict(type='DisCostV2', weight=0.1, norm_with_img_wh=False), topk_k=k)
        cls_costs = [c(logits, labels) for c in ha.cls_costs]
        reg_costs = [c(pred, gt, dict(img_shape=shp)) for c in ha.reg_costs]
        cost = sum(cls_costs) + sum(reg_costs)
        out['ha%d_cost' % case] = cost.numpy()
        res = ha.assign(pred, logits, gt, labels, dict(img_shape=shp))
        g = np.load(os.path.join(GOLDEN, 'assigners.npz'))
        assert np.array_equal(res.gt_inds.numpy(), g['ha%d_gt_inds' % case]), 'fixture host changed its log bits?'
    np.savez_compressed(os.path.join(GOLDEN, 'assigner_costs.npz'), **out)
    print('assigner costs', {k: v.shape for k, v in out.items()})


def p2p_edge_inputs():
    """Shared with the tests: a 2-image batch whose second image is padded LESS than the batch maximum (its feature-map
    cells beyond ceil(pad_shape / stride) are invalid, p2p_head.py:451-463) and a variant with a gt-less image."""
    hw, C, G = 48, 2, 7
    g = torch.Generator().manual_seed(910)
    feat = torch.randn((2, 256, hw, hw), generator=g)
    batch = synthetic.synthetic_batch(2, hw * 4, hw * 4, G, C, seed=910, ragged=True)
    metas = [dict(m) for m in batch['img_metas']]
    metas[1]['pad_shape'] = (160, 176, 3)             # -> 40 x 44 valid cells of the 48 x 48 map
    metas[1]['img_shape'] = (150, 170, 3)
    keep = [(b[:, 2] < 168) & (b[:, 3] < 148) for b in batch['gt_bboxes']]
    keep[0][:] = True
    gtb = [b[k] for b, k in zip(batch['gt_bboxes'], keep)]
    gtl = [l[k] for l, k in zip(batch['gt_labels'], keep)]
    assert all(len(x) > 0 for x in gtl)
    return feat, metas, gtb, gtl, C, hw


def gen_p2p_edge(R):
    from oracle.gen_golden import build_reference_p2p
    feat, metas, gtb, gtl, C, hw = p2p_edge_inputs()
    head, sd = build_reference_p2p(R, C, 0.08, seed=5)
    out = {}
    with torch.no_grad():
        cls_outs, pts_outs = head((feat,))
        losses = head.loss(cls_outs, pts_outs, gtb, gtl, metas, gt_bboxes_ignore=[torch.zeros((0, 4)) for _ in gtl])
        anchor, pred, vflag, co = head.get_pred_points(cls_outs, pts_outs, metas)
        gtp = head.pseudo_bbox_to_center(gtb)
        lab, lw, bgt, pw = head.get_targets(pred[..., :2], vflag, co, gtp, gtl, metas)
        # an image without gts is legal for get_targets (HungarianAssignerV2's no-gt branch, hungarian_assigner.py:207-219)
        gtp0 = [gtp[0], gtp[1][:0]]
        gtl0 = [gtl[0], gtl[1][:0]]
        lab0, lw0, bgt0, pw0 = head.get_targets(pred[..., :2], vflag, co, gtp0, gtl0, metas)
    out['valid_flag'] = vflag.numpy()
    out['loss_cls'] = np.array([float(v) for v in losses['loss_cls']], dtype=np.float32)
    out['loss_pts'] = np.array([float(v) for v in losses['loss_pts']], dtype=np.float32)
    out['labels'] = torch.stack(lab).numpy().astype(np.int32)
    out['label_weights'] = torch.stack(lw).numpy()
    out['target_pts'] = torch.stack(bgt).numpy()
    out['target_weights'] = torch.stack(pw).numpy()
    out['nogt_labels'] = torch.stack(lab0).numpy().astype(np.int32)
    out['nogt_label_weights'] = torch.stack(lw0).numpy()
    np.savez_compressed(os.path.join(GOLDEN, 'p2p_edge.npz'), **out)
    print('p2p_edge: invalid cells', int((~vflag).sum()), 'losses', out['loss_cls'], out['loss_pts'],
          'positives', int((out['labels'] < C).sum()), 'no-gt image positives', int((out['nogt_labels'][1] < C).sum()))


HA_T_CASES = [(3, 11, 1), (4, 30, 3), (2, 5, 2)]     # (n_side, G, C): n_side^2 proposals < G gts, topk_k = 1


def gen_assigner_transposed(R):
    """HungarianAssignerV2 with topk_k = 1 and FEWER proposals than gts (hungarian_assigner.py:229-240): scipy then solves the
    problem with the proposals as rows and every proposal gets a distinct gt."""
    out = {}
    for case, (n_side, G, C) in enumerate(HA_T_CASES):
        pred, logits, gt, labels, shp = assigner_inputs(300 + case, n_side, 4, G, C)
        ha = R.HungarianAssignerV2(cls_costs=dict(type='FocalLossCost', weight=2.0),
                                   reg_costs=dict(type='DisCostV2', weight=0.1, norm_with_img_wh=False), topk_k=1)
        res = ha.assign(pred, logits, gt, labels, dict(img_shape=shp))
        out['hat%d_gt_inds' % case] = res.gt_inds.numpy().astype(np.int32)
        out['hat%d_labels' % case] = res.labels.numpy().astype(np.int32)
        assert int((res.gt_inds > 0).sum()) == n_side * n_side
    np.savez_compressed(os.path.join(GOLDEN, 'assigner_transposed.npz'), **out)
    print('assigner transposed', {k: v.tolist() for k, v in out.items() if 'inds' in k})


def main():
    assert ref_loader.available(), 'needs /root/reference'
    torch.set_num_threads(8)
    R = ref_loader.load()
    which = sys.argv[1:] or ['refine', 'options', 'costs', 'p2p_edge', 'options_r3', 'ha_transposed']
    if 'refine' in which:
        gen_refine(R)
    if 'options' in which:
        gen_options(R)
    if 'costs' in which:
        gen_assigner_costs(R)
    if 'p2p_edge' in which:
        gen_p2p_edge(R)
    if 'options_r3' in which:
        gen_options(R, OPTION_CASES_R3, 'cpr_options_r3.npz')
    if 'ha_transposed' in which:
        gen_assigner_transposed(R)


if __name__ == '__main__':
    main()
