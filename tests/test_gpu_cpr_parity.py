"""-m gpu: the whole CPR path on HIP against (a) the CPU oracle on the same seeded inputs and (b) the golden
fixtures generated from the reference's own classes.  Bars: neg mask / bag validity / labels bit-exact, bag point
coordinates bit-exact, head logits within 1e-4, losses within 1e-4 relative."""
import os

import numpy as np
import pytest
import torch

from oracle import cpr_oracle as O
from oracle.gen_golden import CPR_CASES
from pointtinybenchmark_amd import synthetic

pytestmark = pytest.mark.gpu
GN = dict(type='GN', num_groups=32, requires_grad=True)


def build_hip_locator(cfg):
    import pointtinybenchmark_amd as P
    alpha = 0.25
    r = cfg['radius']
    model = dict(
        type='BasicLocator',
        backbone=dict(type='ResNet', depth=cfg['depth'], num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                      norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, style='pytorch'),
        neck=dict(type='FPN', in_channels=synthetic.backbone_out_channels(cfg['depth']), out_channels=256,
                  start_level=cfg['start_level'], add_extra_convs='on_input', num_outs=1, norm_cfg=GN),
        bbox_head=dict(
            type='CPRHead', norm_cfg=GN, num_classes=cfg['num_classes'], in_channels=256, feat_channels=256,
            stacked_convs=4, num_cls_fcs=cfg.get('num_cls_fcs', 0), fc_out_channels=cfg.get('fc_out_channels', 1024),
            strides=[cfg['stride']],
            loss_mil=dict(type='MILLoss', binary_ins=False, loss_weight=alpha), loss_type=0,
            loss_cfg=dict(with_neg=True, neg_loss_weight=1 - alpha, refine_bag_policy='independent_with_gt_bag',
                          random_remove_rate=0.4, with_gt_loss=True, gt_loss_weight=alpha, with_mil_loss=True),
            normal_cfg=dict(prob_cls_type='sigmoid', out_bg_cls=False),
            train_pts_extractor=dict(pos_generator=dict(type='CirclePtFeatGenerator', radius=r),
                                     neg_generator=dict(type='OutCirclePtFeatGenerator', radius=r, class_wise=True)),
            refine_pts_extractor=dict(pos_generator=dict(type='CirclePtFeatGenerator', radius=r),
                                      neg_generator=dict(type='OutCirclePtFeatGenerator', radius=r, keep_wh=True,
                                                         class_wise=True)),
            point_refiner=dict(merge_th=0.1, refine_th=0.1, classify_filter=True)))
    m = P.build_detector(model).cuda()
    sd = synthetic.locator_state_dict(cfg['depth'], cfg['num_classes'], cfg['start_level'], 'cpr', cfg['seed'],
                                      cfg['head_std'], num_cls_fcs=cfg.get('num_cls_fcs', 0),
                                      fc_out_channels=cfg.get('fc_out_channels', 1024))
    m.load_state_dict(sd, strict=True)
    m.train()
    return m, sd


def to_cuda(batch):
    return dict(img=batch['img'].cuda(), img_metas=batch['img_metas'], gt_bboxes=[b.cuda() for b in batch['gt_bboxes']],
                gt_labels=[l.cuda() for l in batch['gt_labels']], gt_anns_id=[a.cuda() for a in batch['gt_anns_id']])


def run_hip(cfg):
    from pointtinybenchmark_amd import ops
    m, sd = build_hip_locator(cfg)
    batch = synthetic.synthetic_batch(cfg['batch'], cfg['height'], cfg['width'], cfg['num_gts'], cfg['num_classes'],
                                      cfg['seed'], cfg.get('ragged', False))
    cb = to_cuda(batch)
    with torch.no_grad():
        c = m.backbone(cb['img'])
        feats = m.neck(c)
        cls_feat, ins_feat = m.bbox_head(feats)
        losses = m.bbox_head.loss(cls_feat, ins_feat, cb['gt_bboxes'], cb['gt_labels'], cb['img_metas'])
        dets = m.bbox_head.get_bboxes(cls_feat, ins_feat, cb['img_metas'], rescale=False, gt_bboxes=cb['gt_bboxes'],
                                      gt_labels=cb['gt_labels'], gt_anns_id=cb['gt_anns_id'])
        # internals for the parity checks
        head = m.bbox_head
        feat = ops.from_nchw(cls_feat[0])
        lmap = head._logit_map(feat)
        gts = head._gt_tensors(cb['gt_bboxes'], cb['gt_labels'], cb['img_metas'], feat.device)
        ex = head.train_pts_extractor
        pts, valid, bag, _ = head._bags(ex, feat, lmap, gts, cfg['stride'])
        mask, _ = ops.neg_mask_loss(lmap, gts.points, gts.pt_labels, gts.pt_start, gts.pad_hw, cfg['num_classes'],
                                    cfg['stride'], head._d2_threshold(cfg['stride'], ex.neg_radius), 1e-6,
                                    ex.neg_class_wise)
        refined = head.refine_points(cls_feat, cb['img_metas'], cb['gt_bboxes'], cb['gt_labels'])
        torch.cuda.synchronize()
    return dict(model=m, sd=sd, batch=batch, c=[t.cpu() for t in c], fpn=feats[0].cpu(), cls_feat=cls_feat[0].cpu(),
                losses={k: float(v) for k, v in losses.items()}, dets=[(d.cpu(), l.cpu()) for d, l in dets],
                lmap=lmap.cpu(), pts=pts.cpu(), valid=valid.cpu(), bag=bag.cpu(), mask=mask.cpu(),
                refined=[t.cpu() for t in refined[2:]])


def _relerr(a, b):
    return abs(a - b) / max(abs(b), 1e-12)


@pytest.mark.parametrize('name', list(CPR_CASES))
def test_cpr_path_vs_reference_golden(golden_dir, name):
    cfg = CPR_CASES[name]
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    r = run_hip(cfg)
    C = cfg['num_classes']
    scale = lambda ref: max(1.0, float(np.abs(ref).max()))
    # feature maps: fp32 MFMA accumulation order differs from oneDNN -> small relative error, stated here
    for key, got in (('c2_sample', r['c'][0][:, ::37, ::5, ::7]), ('c5_sample', r['c'][3][:, ::101, ::3, ::3]),
                     ('fpn_sample', r['fpn'][:, ::17, ::5, ::7]), ('cls_feat_sample', r['cls_feat'][:, ::13, ::3, ::5])):
        err = float(np.abs(got.numpy() - g[key]).max())
        assert err <= 2e-4 * scale(g[key]), '%s: max abs err %.3e (scale %.2e)' % (key, err, scale(g[key]))
    assert np.array_equal(r['pts'].numpy(), g['pos_pts'][:, 0, :, :2]), 'bag point coordinates must be bit-exact'
    assert np.array_equal(r['valid'].numpy().astype(bool), g['pos_valid'][:, 0, :, 0]), 'bag validity must be bit-exact'
    err_c = float(np.abs(r['bag'][..., :C].numpy() - g['pos_cls_logit'][:, 0]).max())
    err_i = float(np.abs(r['bag'][..., C:].numpy() - g['pos_ins_logit'][:, 0]).max())
    assert err_c <= 1e-4 and err_i <= 1e-4, 'head logits: cls %.3e ins %.3e (bar 1e-4)' % (err_c, err_i)
    gv = np.unpackbits(g['neg_valid'])[:r['mask'].numel()].reshape(r['mask'].shape).astype(bool)
    nbad = int((r['mask'].numpy().astype(bool) != gv).sum())
    assert nbad == 0, 'negative mask differs from the reference in %d of %d entries' % (nbad, gv.size)
    lm = r['lmap'].reshape(-1, r['lmap'].shape[-1])[::97, :C].numpy()
    assert float(np.abs(lm - g['neg_logit_sample']).max()) <= 1e-4
    for k in ('gt_loss', 'pos_loss', 'neg_loss', 'bag_acc'):
        assert _relerr(r['losses'][k], float(g['loss_' + k])) <= 2e-4, '%s: hip %.8g ref %.8g' % (
            k, r['losses'][k], float(g['loss_' + k]))
    dets = torch.cat([d for d, _ in r['dets']]).numpy()
    assert np.array_equal(torch.cat([l for _, l in r['dets']]).numpy(), g['det_labels'])
    assert np.array_equal(dets[:, 5], g['dets'][:, 5])
    np.testing.assert_allclose(dets[:, :5], g['dets'][:, :5], rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize('seed,hw,G,C', [(11, 160, 6, 1), (12, 192, 10, 1), (13, 128, 8, 4)])
def test_cpr_path_vs_oracle_more_seeds(seed, hw, G, C):
    """Same checks against the CPU oracle (no fixture): extra seeds / gt counts / classes."""
    cfg = dict(depth=18, num_classes=C, start_level=0, stride=4, radius=5, head_std=0.3, seed=seed, batch=2,
               height=hw, width=hw + 32, num_gts=G, ragged=True)
    r = run_hip(cfg)
    sd, batch = r['sd'], r['batch']
    torch.set_num_threads(min(16, os.cpu_count() or 1))     # 256 oversubscribed threads on the GPU box take 30 s here
    with torch.no_grad():
        losses, cls_feat, per = O.locator_forward_train(sd, batch, cfg['depth'], 0, 4, 5, C)
        ref = O.cpr_refine(sd, cls_feat, batch['gt_bboxes'], batch['gt_labels'], batch['gt_anns_id'],
                           batch['img_metas'], 4, 5, C)
    sc = max(1.0, float(cls_feat.abs().max()))
    err = float((r['cls_feat'] - cls_feat).abs().max())
    assert err <= 2e-4 * sc, 'cls_feat max abs err %.3e (scale %.2e)' % (err, sc)
    assert torch.equal(r['pts'], torch.cat([p['pts'] for p in per]))
    assert torch.equal(r['valid'].bool(), torch.cat([p['valid'] for p in per]))
    nv = torch.cat([p['neg_valid'] for p in per])
    assert torch.equal(r['mask'].bool(), nv), 'neg mask mismatch: %d' % int((r['mask'].bool() != nv).sum())
    lc = torch.cat([p['cls_logit'] for p in per])
    li = torch.cat([p['ins_logit'] for p in per])
    assert float((r['bag'][..., :C] - lc).abs().max()) <= 1e-4 and float((r['bag'][..., C:] - li).abs().max()) <= 1e-4
    for k in ('gt_loss', 'pos_loss', 'neg_loss', 'bag_acc'):
        assert _relerr(r['losses'][k], float(losses[k])) <= 2e-4, (k, r['losses'][k], float(losses[k]))
    dets = torch.cat([d for d, _ in r['dets']])
    rd = torch.cat([x['dets'] for x in ref])
    np.testing.assert_allclose(dets.numpy(), rd.numpy(), rtol=1e-4, atol=2e-3)


def test_fused_forward_train_matches_module_by_module_path():
    """BasicLocator.forward_train (lazy GroupNorm hand-off neck -> head -> projection, optional sub-batch streams) gives
    the same losses as running backbone / neck / head / loss module by module."""
    cfg = dict(depth=18, num_classes=2, start_level=0, stride=4, radius=5, head_std=0.3, seed=31, batch=4, height=128,
               width=128, num_gts=5, ragged=True)
    m, sd = build_hip_locator(cfg)
    batch = synthetic.synthetic_batch(4, 128, 128, 5, 2, 31, True)
    cb = to_cuda(batch)
    with torch.no_grad():
        feats = m.neck(m.backbone(cb['img']))
        ref = m.bbox_head.loss(*m.bbox_head(feats), cb['gt_bboxes'], cb['gt_labels'], cb['img_metas'])
        fused = m.forward_train(cb['img'], cb['img_metas'], cb['gt_bboxes'], cb['gt_labels'])
        m.num_streams = 2
        streamed = m.forward_train(cb['img'], cb['img_metas'], cb['gt_bboxes'], cb['gt_labels'])
        torch.cuda.synchronize()
    for k in ref:
        assert _relerr(float(fused[k]), float(ref[k])) <= 1e-5, (k, float(fused[k]), float(ref[k]))
        assert _relerr(float(streamed[k]), float(ref[k])) <= 1e-5, (k, float(streamed[k]), float(ref[k]))


@pytest.mark.parametrize('mode', ['fp32', 'bf16'])
def test_fused_stem_kernels_and_the_two_kernel_fallback_agree_through_the_network(mode):
    """Round 4: ResNet.stem runs conv + BN + ReLU + max-pool as one kernel (csrc/stem_f32.hip; csrc/stem_bf16.hip in the bf16
    compute mode); ``CPR_F32_STEM=0`` / ``CPR_BF16_STEM=0`` / ``CPR_BF16_STEM_POOL=0`` (module flags here) keep the implicit-GEMM
    stem + maxpool3x3s2.  Both paths must give the same network: stage outputs and losses within fp32 summation-order distance
    (fp32 mode) resp. the bf16 mode's own rounding (one bf16 rounding of the stem inputs), and the bf16 fused pool the SAME bits as
    its unfused pair."""
    from pointtinybenchmark_amd.backbones import resnet as R
    cfg = dict(depth=18, num_classes=2, start_level=0, stride=4, radius=5, head_std=0.3, seed=33, batch=2, height=160,
               width=192, num_gts=5, ragged=True)
    m, _ = build_hip_locator(cfg)
    if mode == 'bf16':
        m.set_compute_dtype('bf16')
    cb = to_cuda(synthetic.synthetic_batch(2, 160, 192, 5, 2, 33, True))
    flags = (R.F32_STEM, R.BF16_STEM, R.BF16_STEM_POOL)
    outs = {}
    try:
        for name, vals in (('fused', (True, True, True)), ('pair', (True, True, False)), ('fallback', (False, False, False))):
            for f, v in zip(flags, vals):
                f[0] = v
            with torch.no_grad():
                c2 = m.backbone(cb['img'])[0].float()
                losses = m.forward_train(cb['img'], cb['img_metas'], cb['gt_bboxes'], cb['gt_labels'])
            torch.cuda.synchronize()
            outs[name] = (c2, {k: float(v) for k, v in losses.items()})
    finally:
        for f in flags:
            f[0] = True
    mx = float(outs['fallback'][0].abs().max())
    tol = 2e-5 if mode == 'fp32' else 6e-2
    d = float((outs['fused'][0] - outs['fallback'][0]).abs().max())
    assert d <= tol * mx, (d, mx)
    if mode == 'bf16':
        assert torch.equal(outs['fused'][0], outs['pair'][0]), 'the fused bf16 stem + pool must equal conv -> pool bit for bit'
    for k, v in outs['fallback'][1].items():
        assert _relerr(outs['fused'][1][k], v) <= (1e-4 if mode == 'fp32' else 5e-2), (k, outs['fused'][1][k], v)

