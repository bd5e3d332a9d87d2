"""-m gpu: (a) PointRefiner selections held bit-exact -- the kernel on the REFERENCE's own bag logits must reproduce the
reference's chosen-point masks / not_refine flags exactly, and end to end (HIP conv stack -> logits -> refine) the masks
must agree except where a probability sits within the logit tolerance of a threshold; (b) every get_bboxes output format
(out_geo, rescale, not_refine input, cascade_out_fmt); (c) the CPRHead options no shipped config uses (num_refine > 1 bag
policies, GridCirclesPtFeatGenerator, softmax / normed_sigmoid, binary_ins, AllPosLoss) against fixtures produced by the
reference's own classes (oracle/gen_golden_r2.py)."""
import os

import numpy as np
import pytest
import torch

from oracle.gen_golden import CPR_CASES
from oracle.gen_golden_r2 import (OPTION_CASES, OPTION_CASES_R3, REFINE_CASES, case_inputs, cpr_head_kwargs, not_refine_input,
                                  option_cfg)
from pointtinybenchmark_amd import synthetic

pytestmark = pytest.mark.gpu
GN = dict(type='GN', num_groups=32, requires_grad=True)


def build_hip(cfg):
    import pointtinybenchmark_amd as P
    model = dict(
        type='BasicLocator',
        backbone=dict(type='ResNet', depth=cfg['depth'], num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                      norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, style='pytorch'),
        neck=dict(type='FPN', in_channels=synthetic.backbone_out_channels(cfg['depth']), out_channels=256,
                  start_level=cfg['start_level'], add_extra_convs='on_input', num_outs=1, norm_cfg=GN),
        bbox_head=dict(type='CPRHead', **cpr_head_kwargs(cfg)))
    m = P.build_detector(model).cuda()
    sd, batch = case_inputs(cfg)
    m.load_state_dict(sd, strict=True)
    m.train()
    return m, batch


def cuda_batch(batch):
    return dict(img=batch['img'].cuda(), img_metas=batch['img_metas'], gt_bboxes=[b.cuda() for b in batch['gt_bboxes']],
                gt_labels=[l.cuda() for l in batch['gt_labels']], gt_anns_id=[a.cuda() for a in batch['gt_anns_id']])


def _mask(g, key, shape_key=None, shape=None):
    shape = tuple(g[shape_key]) if shape is None else shape
    return np.unpackbits(g[key])[:int(np.prod(shape))].reshape(shape).astype(bool)


def _kernel_on_reference_logits(g, p, cfg, batch, head):
    """ops.refine fed with the reference's bag logits / points / validity: selections must be bit-exact."""
    from pointtinybenchmark_amd import ops
    cb = cuda_batch(batch)
    gts = head._gt_tensors(cb['gt_bboxes'], cb['gt_labels'], cb['img_metas'], torch.device('cuda'))
    chosen_ref = _mask(g, p + 'chosen', p + 'chosen_shape')
    G, Kt = chosen_ref.shape
    logits = torch.from_numpy(g[p + 'bag_cls_logit']).cuda().contiguous()
    pts = torch.from_numpy(g[p + 'bag_pts']).cuda().contiguous()
    valid = torch.from_numpy(_mask(g, p + 'bag_valid', shape=(G, Kt)).astype(np.uint8)).cuda()
    pr = head.point_refiner
    rp, sc, nr, chosen = ops.refine(logits, pts, valid, gts.points, gts.labels, gts.gt_img, gts.gt_start, gts.img_hw,
                                    head.num_cls_out, pr['gt_alpha'], pr['merge_th'], pr['refine_th'],
                                    pr['nearest_filter'], pr['classify_filter'], None, sub_bags=1, ctr_stride=gts.R,
                                    prob_type=head.prob_type, norm_p=head.norm_p,
                                    score_max=pr['return_score_type'] == 'max')
    nbad = int((chosen.cpu().numpy().astype(bool) != chosen_ref).sum())
    assert nbad == 0, '%d of %d chosen flags differ from the reference on identical logits' % (nbad, chosen_ref.size)
    assert np.array_equal(nr.cpu().numpy().astype(bool), g[p + 'not_refine'])
    np.testing.assert_allclose(rp.cpu().numpy(), g[p + 'refine_pts'], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(sc.cpu().numpy(), g[p + 'scores'], rtol=1e-5, atol=1e-7)
    return chosen_ref


def _logit_bar(g, p):
    """1e-4 on the head logits (north star) for logits of realistic size; the synthetic high-variance classifier of the
    '_live' case produces |logit| up to 61, where 1e-4 is 1.6e-6 relative -- the bar scales with the magnitude past 16."""
    return 1e-4 * max(1.0, float(np.abs(g[p + 'bag_cls_logit']).max()) / 16.0)


def _near_threshold(g, p, head, labels):
    """Entries whose class probability lies within the logit tolerance (1e-4 on the logit -> < 3e-5 on the probability) of
    a PointRefiner threshold: the only places where the end-to-end selection may legitimately differ."""
    lg = torch.from_numpy(g[p + 'bag_cls_logit'])                                  # (G, Kt, C)
    if head.prob_type == 'sigmoid':
        prob = lg.sigmoid()
    elif head.prob_type == 'softmax':
        prob = lg.softmax(-1)
    else:
        prob = torch.nn.functional.normalize(lg.sigmoid(), p=head.norm_p, dim=-1)
    G = lg.shape[0]
    pl = prob[torch.arange(G), :, labels]
    gate = pl[:, -1:] * head.point_refiner['gt_alpha']
    top2 = prob.topk(min(2, prob.shape[-1]), dim=-1)[0]
    margin = (top2[..., 0] - top2[..., -1]) if prob.shape[-1] > 1 else torch.ones_like(pl)
    tol = 0.5 * _logit_bar(g, p)
    return ((pl - head.point_refiner['merge_th']).abs() < tol) | ((pl - gate).abs() < tol) | (margin < tol)


def _end_to_end_refine(g, p, cfg, m, batch, seed):
    head = m.bbox_head
    cb = cuda_batch(batch)
    kw = dict(gt_bboxes=cb['gt_bboxes'], gt_labels=cb['gt_labels'], gt_anns_id=cb['gt_anns_id'])
    with torch.no_grad():
        cls_feat, ins_feat = head(m.neck(m.backbone(cb['img'])))
        gts, pts, rp, sc, nr, chosen = head.refine_points(cls_feat, cb['img_metas'], cb['gt_bboxes'], cb['gt_labels'])
        dets = head.get_bboxes(cls_feat, ins_feat, cb['img_metas'], rescale=False, **kw)
        head.other_info = dict(out_geo=True)
        dg = head.get_bboxes(cls_feat, ins_feat, cb['img_metas'], rescale=False, **kw)
        metas2 = [dict(mm, scale_factor=[1.25, 1.6, 1.25, 1.6]) for mm in cb['img_metas']]
        dr = head.get_bboxes(cls_feat, ins_feat, metas2, rescale=True, **kw)
        head.other_info = dict()
        nr_in = [t.cuda() for t in not_refine_input(batch, seed)]
        dc, nr_out = head.get_bboxes(cls_feat, ins_feat, cb['img_metas'], rescale=False, not_refine=nr_in,
                                     cascade_out_fmt=True, **kw)
        torch.cuda.synchronize()
    chosen_ref = _mask(g, p + 'chosen', p + 'chosen_shape')
    assert np.array_equal(pts.cpu().numpy(), g[p + 'bag_pts']), 'bag points must be bit-exact'
    err = float(np.abs(head_logits(head, cls_feat, gts, cfg)[..., :head.num_cls_out] - g[p + 'bag_cls_logit']).max())
    assert err <= _logit_bar(g, p), 'bag logits %.3e (bar %.1e)' % (err, _logit_bar(g, p))
    diff = chosen.cpu().numpy().astype(bool) != chosen_ref
    labels = torch.cat(batch['gt_labels'])
    near = _near_threshold(g, p, head, labels).numpy()
    # a gate / classify decision of the annotated point's own probability moves a whole row: rows where the gate itself is
    # near a threshold are excused as a whole
    assert not (diff & ~near).any(), '%d chosen flags differ away from any threshold (of %d; %d near-threshold entries)' % (
        int((diff & ~near).sum()), diff.size, int(near.sum()))
    same_rows = ~diff.any(axis=1)
    nr_ref = g[p + 'not_refine']
    assert np.array_equal(nr.cpu().numpy().astype(bool)[same_rows], nr_ref[same_rows])
    np.testing.assert_allclose(rp.cpu().numpy()[same_rows], g[p + 'refine_pts'][same_rows], rtol=1e-4, atol=2e-3)
    np.testing.assert_allclose(sc.cpu().numpy()[same_rows], g[p + 'scores'][same_rows], rtol=1e-4, atol=1e-5)
    d = torch.cat([x for x, _ in dets]).cpu().numpy()
    assert np.array_equal(torch.cat([l for _, l in dets]).cpu().numpy(), g[p + 'det_labels'])
    assert np.array_equal(d[:, 5], g[p + 'dets'][:, 5])                                # ann ids
    np.testing.assert_allclose(d[same_rows, :5], g[p + 'dets'][same_rows, :5], rtol=1e-4, atol=2e-3)
    s = 0
    for b, (x, _) in enumerate(dg):                       # out_geo rows: [box, score, ann_id, refined pt, chosen pts.., -1 pad]
        x, ref = x.cpu().numpy(), g[p + 'dets_geo%d' % b]
        rows = same_rows[s:s + len(ref)]
        if rows.all():
            assert x.shape == ref.shape, (x.shape, ref.shape)
            assert np.array_equal(x[:, 8:], ref[:, 8:]), 'chosen-point columns of the geo output must be bit-exact'
            np.testing.assert_allclose(x[:, :8], ref[:, :8], rtol=1e-4, atol=2e-3)
            xr, rr = dr[b][0].cpu().numpy(), g[p + 'dets_geo_rescaled%d' % b]
            assert xr.shape == rr.shape
            np.testing.assert_allclose(xr, rr, rtol=1e-4, atol=2e-3)
            assert np.array_equal(xr[:, 8:] == -1, rr[:, 8:] == -1)
        s += len(ref)
    nro = torch.cat(nr_out).cpu().numpy()
    assert np.array_equal(nro[same_rows], g[p + 'cascade_not_refine'][same_rows])
    dcc = torch.cat([x for x, _ in dc]).cpu().numpy()
    np.testing.assert_allclose(dcc[same_rows, :5], g[p + 'cascade_dets'][same_rows, :5], rtol=1e-4, atol=2e-3)
    return int(diff.sum()), int(near.sum())


def head_logits(head, cls_feat, gts, cfg):
    from pointtinybenchmark_amd import ops
    feat = ops.from_nchw(cls_feat[0])
    part = None if head.ins_share_head_feat else 'cls'
    lmap = head._logit_map(feat, part=part)
    return head._bags(head.refine_pts_extractor, feat, lmap, gts, cfg['stride'], part=part)[2].cpu().numpy()


@pytest.mark.parametrize('name', list(REFINE_CASES))
def test_refine_selections_vs_reference(golden_dir, name, record_property):
    from tests.test_gpu_cpr_parity import build_hip_locator
    cfg = dict(REFINE_CASES[name])
    g = np.load(os.path.join(golden_dir, 'refine.npz'))
    p = name + ':'
    m, _ = build_hip_locator(cfg)
    _, batch = case_inputs(cfg)
    if cfg.get('num_cls_fcs', 0) == 0:
        _kernel_on_reference_logits(g, p, cfg, batch, m.bbox_head)
    ndiff, nnear = _end_to_end_refine(g, p, cfg, m, batch, cfg['seed'])
    record_property('chosen_flags_differing_near_threshold', ndiff)
    record_property('near_threshold_entries', nnear)


@pytest.mark.parametrize('name', list(OPTION_CASES) + list(OPTION_CASES_R3))
def test_cpr_options_vs_reference(golden_dir, name, record_property):
    """(round 3: + return_score_type='max', ins_share_head_feat=False with and without FC layers, out_bg_cls for one class;
    AnchorPtFeatGenerator(scale_factor != 1) raises in the reference -- the fixture holds its TypeError -- and is refused.)"""
    cfg = option_cfg(name)
    g = np.load(os.path.join(golden_dir, 'cpr_options.npz' if name in OPTION_CASES else 'cpr_options_r3.npz'))
    p = name + ':'
    if (p + 'reference_error') in g.files:
        assert 'TypeError' in str(g[p + 'reference_error'])
        with pytest.raises(NotImplementedError):
            build_hip(cfg)
        return
    m, batch = build_hip(cfg)
    head = m.bbox_head
    C = head.num_cls_out                    # classifier outputs: num_classes (+ 1 with out_bg_cls)
    cb = cuda_batch(batch)
    from pointtinybenchmark_amd import ops
    with torch.no_grad():
        cls_feat, ins_feat = head(m.neck(m.backbone(cb['img'])))
        losses = head.loss(cls_feat, ins_feat, cb['gt_bboxes'], cb['gt_labels'], cb['img_metas'])
        feat = ops.from_nchw(cls_feat[0])
        ifeat = None if head.ins_share_head_feat else ops.from_nchw(ins_feat[0])
        lmap = head._lmap_all(feat, None, ifeat, None)
        gts = head._gt_tensors(cb['gt_bboxes'], cb['gt_labels'], cb['img_metas'], feat.device)
        ex = head.train_pts_extractor
        pts, valid, bag, view = head._bags(ex, feat, lmap, gts, cfg['stride'], ifeat=ifeat)
        mask, _ = ops.neg_mask_loss(lmap, gts.points, gts.pt_labels, gts.pt_start, gts.pad_hw, C, cfg['stride'],
                                    head._d2_threshold(cfg['stride'], ex.neg_radius), 1e-6, ex.neg_class_wise,
                                    head.prob_type, head.norm_p, mask_classes=head.num_classes)
        torch.cuda.synchronize()
    if head.out_bg_cls:                     # the reference's (.., 1) validity broadcasts over [class, background]
        assert bool((mask[:, 0] == mask[:, 1]).all())
        mask = mask[:, :1]
    ref_pts = g[p + 'pos_pts']                                  # (G, R|1, K, 2)
    G = ref_pts.shape[0]
    assert np.array_equal(pts.cpu().numpy().reshape(ref_pts.shape), ref_pts), 'bag points must be bit-exact'
    assert np.array_equal(valid.cpu().numpy().astype(bool).reshape(g[p + 'pos_valid'].shape), g[p + 'pos_valid'])
    bagn = bag.cpu().numpy()
    ec = float(np.abs(bagn[..., :C].reshape(g[p + 'pos_cls_logit'].shape) - g[p + 'pos_cls_logit']).max())
    ei = float(np.abs(bagn[..., C:].reshape(g[p + 'pos_ins_logit'].shape) - g[p + 'pos_ins_logit']).max())
    assert ec <= 1e-4 and ei <= 1e-4, 'bag logits: cls %.3e ins %.3e (bar 1e-4)' % (ec, ei)
    gv = np.unpackbits(g[p + 'neg_valid'])[:mask.numel()].reshape(mask.shape).astype(bool)
    assert int((mask.cpu().numpy().astype(bool) != gv).sum()) == 0, 'negative mask must be bit-exact'
    for k, v in losses.items():
        ref = float(g[p + 'loss_' + k])
        assert abs(float(v) - ref) <= 3e-4 * max(abs(ref), 1e-6), '%s: hip %.8g ref %.8g' % (k, float(v), ref)
    assert set('loss_' + k for k in losses) == set(k[len(p):] for k in g.files if k.startswith(p + 'loss_'))
    if (p + 'refine_asserts_in_reference') in g.files:
        with pytest.raises(AssertionError):                     # same contract as cpr_head.py:809
            head.get_bboxes(cls_feat, ins_feat, cb['img_metas'], gt_bboxes=cb['gt_bboxes'], gt_labels=cb['gt_labels'],
                            gt_anns_id=cb['gt_anns_id'])
        return
    _kernel_on_reference_logits(g, p, cfg, batch, head)
    ndiff, nnear = _end_to_end_refine(g, p, cfg, m, batch, cfg['seed'])
    record_property('chosen_flags_differing_near_threshold', ndiff)
    record_property('near_threshold_entries', nnear)


from oracle.gen_golden_r5 import OPTION_GRAD_BARS, OPTION_GRAD_CASES, grad_option_cfg      # name -> (option case, overrides)


@pytest.mark.parametrize('name', list(OPTION_GRAD_CASES))
def test_option_backward_vs_reference_autograd(golden_dir, name):
    """The CPRHead options that gained a hand-written backward in round 5 -- num_refine = 2 inputs under the default bag policy
    (cpr_head.py:1159-1211), a separate instance tower (ins_share_head_feat=False, :992-1008,1037-1040,1061-1070), the same with an
    FC layer between the sampled features and the classifiers, and two FC layers on shared features (num_cls_fcs > 0,
    :999-1005,1055-1059), and the loss options behind the general loss-backward kernels (softmax / normed_sigmoid class
    probabilities, binary_ins, AllPosLoss, merge_to_gt_bag / only_refine_bag with gt_loss_type='gt', out_bg_cls,
    with_mil_loss=False, with_neg=False) and the point-list gather (grid bags, align_corners=True; on the logit map and, with an FC
    layer, on the sampled features) -- against loss.backward() through the REFERENCE's own modules (tests/golden/cpr_option_grads.npz,
    oracle/gen_golden_r5.py): total loss 1e-4, per-tensor norm 2e-3, strided samples 2e-3 of the tensor's max.
    'ins_tower_fc_boundary' is a sample on which an activation of the sparse-gradient instance tower sits within fp32 conv
    rounding of a ReLU boundary (the reference's own gradient moves 3e-3 .. 7e-3 when that boundary is shifted by 1e-5:
    oracle/gen_golden_r5.py) -- same comparison under a 2e-2 bar; 'r3_only_refine' and 'bg_cls' are samples WITHOUT such an
    event, held to 2e-4 on every tensor (measured 5e-6), and the classifier tensors next to the loss to 1e-4 in every case.  Then loss.backward() through the autograd bridge on a fresh
    model must equal the native trainer BIT for bit, as for the shipped options."""
    from oracle.gen_golden import grad_sample_index
    from pointtinybenchmark_amd import autograd_bridge
    from pointtinybenchmark_amd.training import CprTrainer
    cfg = grad_option_cfg(name)
    bar = OPTION_GRAD_BARS.get(name, 2e-3)
    g = np.load(os.path.join(golden_dir, 'cpr_option_grads.npz'))
    p = name + ':'
    m, batch = build_hip(cfg)
    cb = cuda_batch(batch)
    data = dict(img=cb['img'], img_metas=cb['img_metas'], gt_bboxes=cb['gt_bboxes'], gt_labels=cb['gt_labels'])
    assert autograd_bridge.unsupported_reason(m, cb['gt_bboxes'], cb['gt_labels']) is None
    tr = CprTrainer(m)
    losses = tr.forward_backward(**data)
    torch.cuda.synchronize()
    total = float(sum(v for k, v in losses.items() if 'loss' in k))
    ref_total = float(g[p + 'total_loss'])
    assert abs(total - ref_total) <= 1e-4 * max(1.0, abs(ref_total)), (total, ref_total)
    params = dict(m.named_parameters())
    keys = [k[len(p + 'norm:'):] for k in g.files if k.startswith(p + 'norm:')]
    trainable = [k for k, q in params.items() if q.requires_grad]
    unused = sorted(set(trainable) - set(keys))              # parameters the reference's graph never reaches (grad None there)
    assert not set(keys) - set(trainable) and unused == sorted(k[len(p + 'unused:'):] for k in g.files if k.startswith(p + 'unused:')), (unused, set(keys) ^ set(trainable))
    for k in unused:                                         # with_mil_loss=False: ins_out -- ours leaves an exactly zero gradient
        assert params[k].grad is None or not bool(params[k].grad.any()), k
    gmax = max(float(g[p + 'norm:' + k]) for k in keys)
    want = {}
    for k in keys:
        gr = params[k].grad.detach().double().flatten().cpu()
        want[k] = params[k].grad.detach().clone()
        ref_n = float(g[p + 'norm:' + k])
        # the classifier tensors sit next to the loss (no ReLU between them and the loss-backward kernels): 1e-4 in every case
        kbar = min(bar, 1e-4) if k.startswith(('bbox_head.cls_out.', 'bbox_head.ins_out.')) else bar
        assert abs(float(gr.norm()) - ref_n) <= kbar * ref_n + 1e-6 * gmax, (k, float(gr.norm()), ref_n)
        smp = gr[torch.from_numpy(grad_sample_index(gr.numel()))].numpy()
        ref = g[p + 'sample:' + k].astype(np.float64)
        assert np.abs(smp - ref).max() <= kbar * max(np.abs(ref).max(), 1e-5 * gmax), k
    del tr, m
    m2, _ = build_hip(cfg)
    out = m2.train_step(dict(data), optimizer=None)
    assert out['loss'].grad_fn is not None
    out['loss'].backward()
    torch.cuda.synchronize()
    for k, q in m2.named_parameters():
        if q.requires_grad and k in want:
            assert q.grad is not None and torch.equal(q.grad, want[k]), k
        elif q.requires_grad:
            assert q.grad is None or not bool(q.grad.any()), k


@pytest.mark.parametrize('name', ['r2_merge_gt', 'softmax', 'ins_tower', 'grid_circles', 'align_corners'])
def test_option_backward_in_the_mixed_precision_mode(name):
    """The option backward rules in the bf16 compute mode (mixed precision: bf16 recorded maps, fp32 logit map / loss backward /
    gradients): the trainer's gradient against torch autograd over the fp32 options oracle at the mixed-precision bar of
    tests/test_gpu_train_step.py (global cosine >= 0.99, total loss 3e-2 -- the mode's arithmetic is not the reference's), and
    loss.backward() through the bridge BIT-equal to the mixed-precision trainer (bf16 maps behind the fp32 carriers)."""
    from oracle import cpr_oracle as O, cpr_options_oracle as OO
    from pointtinybenchmark_amd.training import CprTrainer
    cfg = grad_option_cfg(name)
    m, batch = build_hip(cfg)
    m.set_compute_dtype('bf16')
    cb = cuda_batch(batch)
    data = dict(img=cb['img'], img_metas=cb['img_metas'], gt_bboxes=cb['gt_bboxes'], gt_labels=cb['gt_labels'])
    tr = CprTrainer(m)
    losses = tr.forward_backward(**data)
    torch.cuda.synchronize()
    got = {k: q.grad.detach().clone() for k, q in m.named_parameters() if q.requires_grad}
    total = float(sum(v for k, v in losses.items() if 'loss' in k))
    sd, _ = case_inputs(cfg)
    sd = {k: v.clone() for k, v in sd.items()}
    for k in got:
        sd[k].requires_grad_(True)
    feats = O.fpn_forward(sd, O.resnet_forward(sd, batch['img'], cfg['depth']), cfg['start_level'], 1)
    cf, _ = O.cpr_head_forward(sd, feats)
    inf = OO.ins_tower_forward(sd, feats)[0] if cfg.get('ins_tower') else None
    lo, _ = OO.cpr_loss(sd, cf[0], batch['gt_bboxes'], batch['gt_labels'], batch['img_metas'], cfg, ins_feat=inf)
    ref_total = sum(v for k, v in lo.items() if 'loss' in k)
    ref_total.backward()
    ref_val = float(ref_total.detach())
    assert abs(total - ref_val) <= 3e-2 * max(1.0, abs(ref_val)), (total, ref_val)
    a = torch.cat([got[k].double().flatten().cpu() for k in got])
    b = torch.cat([(sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])).double().flatten() for k in got])
    cos = float((a * b).sum() / (a.norm() * b.norm()))
    assert cos >= 0.99, 'mixed-precision gradient cosine %.5f' % cos
    del tr, m
    m2, _ = build_hip(cfg)
    m2.set_compute_dtype('bf16')
    out = m2.train_step(dict(data), optimizer=None)
    assert out['loss'].grad_fn is not None
    out['loss'].backward()
    torch.cuda.synchronize()
    for k, q in m2.named_parameters():
        if q.requires_grad:
            assert q.grad is not None and torch.equal(q.grad, got[k]), k


def test_grid_ellipse_generator_raises_like_the_reference(golden_dir):
    """GridEllipsePtFeatGenerator cannot run in the reference (fixture records its RuntimeError); ours refuses at build."""
    from pointtinybenchmark_amd.registry import build_head
    g = np.load(os.path.join(golden_dir, 'cpr_options.npz'))
    assert 'RuntimeError' in str(g['grid_ellipse_reference_error'])
    kw = cpr_head_kwargs(option_cfg('softmax'))
    kw['train_pts_extractor']['pos_generator'] = dict(type='GridEllipsePtFeatGenerator', a_minus_c=2.0)
    with pytest.raises(NotImplementedError):
        build_head(dict(type='CPRHead', **kw))


def test_mil_loss_reference_signature_matches_formula():
    """MILLoss.forward / AllPosLoss.forward with the reference's (prob, ins logits, labels, valid) signature against the
    formulas of multi_instance_learning_loss.py:153-243 evaluated in torch on the CPU."""
    import torch.nn.functional as F
    from pointtinybenchmark_amd.losses.mil_loss import AllPosLoss, MILLoss
    g = torch.Generator().manual_seed(3)
    B, N, C = 9, 37, 4
    prob = torch.rand((B, N, C), generator=g)
    labels = torch.randint(0, C, (B,), generator=g)
    valid = (torch.rand((B, N, 1), generator=g) > 0.3).float()
    valid[2] = 0                                            # a bag without any valid point
    for binary in (False, True):
        ins = torch.randn((B, N, C * (2 if binary else 1)), generator=g)
        loss, acc, ns = MILLoss(binary_ins=binary, loss_weight=0.7)(prob.cuda(), ins.cuda(), labels.cuda(), valid.cuda())
        pi = ins.reshape(B, N, C, -1).softmax(dim=1) * valid.unsqueeze(-1)
        pi = F.normalize(pi, dim=1, p=1)
        pb = (prob.unsqueeze(-1) * pi).sum(dim=1)                       # (B, C, 1|2)
        lw = (valid.sum(dim=1) > 0).float()
        onehot = F.one_hot(labels, C).float()
        num = max(float((lw.sum(-1) > 0).sum()), 1.0)
        gf = lambda p_, q_, w_: -(((p_ - q_) ** 2) * (q_ * (p_ + 1e-6).log() + (1 - q_) * (1 - p_ + 1e-6).log()) * w_).sum(-1)
        ref = gf(pb[..., 0], onehot, lw).sum()
        if binary:
            ref = ref + gf(pb[..., 1], torch.zeros_like(onehot), lw).sum()
        ref = ref / num * 0.7
        assert abs(float(loss) - float(ref)) <= 1e-5 * abs(float(ref)), (binary, float(loss), float(ref))
        assert float(ns) == num
        racc = float((pb[..., 0].argmax(-1) == labels).float().mean() * 100)
        assert abs(float(acc) - racc) < 1e-3
    ins = torch.randn((B, N, C), generator=g)
    loss, acc, ns = AllPosLoss(loss_weight=0.3)(prob.cuda(), ins.cuda(), labels.cuda(), valid.cuda())
    pr, lab, v = prob.reshape(B * N, C), labels.unsqueeze(-1).repeat(1, N).flatten(), valid.reshape(B * N, 1)
    num = max(float((v.sum(-1) > 0).sum()), 1.0)
    ref = gf(pr, F.one_hot(lab, C).float(), v).sum() / num * 0.3
    assert abs(float(loss) - float(ref)) <= 1e-5 * abs(float(ref)), (float(loss), float(ref))
    assert float(ns) == num
    assert abs(float(acc) - float((pr.argmax(-1) == lab).float().mean() * 100)) < 1e-3


@pytest.mark.parametrize('over', [dict(num_refine=2, policy='merge_to_gt_bag', gt_loss_type='gt', prob='normed_sigmoid', norm_p=2),
                                  dict(pos='GridCirclesPtFeatGenerator', radius=5, binary_ins=True)],
                         ids=['r2_merge_normed_sigmoid', 'grid_circles_binary_ins'])
def test_cpr_options_full_size_vs_oracle(over):
    """The option paths at the configs[1] size (ResNet-50, 640x640, 32 gts, 3 classes) against the CPU options oracle
    (oracle/cpr_options_oracle.py, itself pinned by the reference fixtures): bag points / validity / negative mask bit-exact,
    losses 5e-4."""
    from oracle import cpr_options_oracle as OO
    from oracle import cpr_oracle as O
    from pointtinybenchmark_amd import ops
    cfg = dict(depth=50, num_classes=3, start_level=0, stride=4, radius=5, head_std=0.3, seed=81, batch=1, height=640,
               width=640, num_gts=32)
    cfg.update(over)
    m, batch = build_hip(cfg)
    sd, _ = case_inputs(cfg)
    head, C = m.bbox_head, 3
    cb = cuda_batch(batch)
    with torch.no_grad():
        cls_feat, ins_feat = head(m.neck(m.backbone(cb['img'])))
        losses = head.loss(cls_feat, ins_feat, cb['gt_bboxes'], cb['gt_labels'], cb['img_metas'])
        feat = ops.from_nchw(cls_feat[0])
        lmap = head._logit_map(feat)
        gts = head._gt_tensors(cb['gt_bboxes'], cb['gt_labels'], cb['img_metas'], feat.device)
        ex = head.train_pts_extractor
        pts, valid, bag, view = head._bags(ex, feat, lmap, gts, 4)
        mask, _ = ops.neg_mask_loss(lmap, gts.points, gts.pt_labels, gts.pt_start, gts.pad_hw, C, 4,
                                    head._d2_threshold(4, ex.neg_radius), 1e-6, ex.neg_class_wise, head.prob_type, head.norm_p)
        torch.cuda.synchronize()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        feats = O.fpn_forward(sd, O.resnet_forward(sd, batch['img'], 50), 0, 1)
        ref_feat, _ = O.cpr_head_forward(sd, feats)
        ref_losses, per = OO.cpr_loss(sd, ref_feat[0], batch['gt_bboxes'], batch['gt_labels'], batch['img_metas'], cfg)
    rp = torch.cat([q['pts'] for q in per])
    assert torch.equal(pts.cpu().reshape(rp.shape), rp)
    assert torch.equal(valid.cpu().bool().reshape(rp.shape[:-1]), torch.cat([q['valid'] for q in per]))
    nv = torch.cat([q['neg_valid'] for q in per])
    assert torch.equal(mask.cpu().bool(), nv), 'negative mask: %d entries differ' % int((mask.cpu().bool() != nv).sum())
    for k, v in ref_losses.items():
        assert abs(float(losses[k]) - float(v)) <= 5e-4 * max(abs(float(v)), 1e-6), (k, float(losses[k]), float(v))


SWEEP_CASES = ['binary_ins', 'normed_sigmoid_p2', 'r2_merge_gt', 'r3_only_refine', 'combo_tower_binary_normed']


@pytest.mark.parametrize('name', SWEEP_CASES)
def test_general_loss_backward_seed_sweep_vs_oracle_autograd(name):
    """Round 6 (advisor): the reference fixtures of test_option_backward_vs_reference_autograd sit on seeds that were PICKED (smallest
    device-vs-oracle error among candidates), so a small error of the general loss-backward kernels could hide behind the selection.
    Here: seeds nobody selected (fixture seed + 1000 + {1, 2, 3}), the same device step against torch autograd over the options
    oracle (itself pinned to the reference's loss.backward() at 1e-3, tests/test_oracle_golden.py).
      * The classifier tensors sit right behind the loss-backward kernels (no ReLU between them and the loss): their device gradients
        against the oracle's loss differentiated ON THE DEVICE'S OWN head features -- identical inputs, so nothing but the
        loss-backward kernels and the classifier projection's backward is compared -- 1e-4 on EVERY swept seed where the oracle's
        own gradient is smooth (probed with a 1e-7 relative perturbation of the features: binary_ins / merged bags / only_refine move
        ~1e-6 and the device is 1e-6 .. 4e-6 off; the normed_sigmoid cases sit on saturated logits -- head_std 0.3, neg_loss ~ 1600 --
        where single elements cross a clamp of the loss and the oracle's own gradient jumps 4e-5 .. 1e-3 under that perturbation: such
        seeds are reported and held to ten times the oracle's jump).  (Against the
        oracle's own features the same tensors are up to 3.9e-4 off on combo_tower_binary_normed: the normalised probabilities of
        normed_sigmoid amplify feature noise -- a 1e-5 relative perturbation of the oracle's features moves cls_out.weight by 3.0e-4
        on seed 1179, 5.9e-5 on seed 1178 -- which is why the comparison is made on identical features.)
      * Every other tensor against the full oracle graph: 1e-2 of its norm (a ReLU boundary within fp32 conv rounding moves a tower
        tensor by 3e-3 .. 7e-3 in the reference's own graph: oracle/gen_golden_r5.py); numerically nil tensors on the absolute bar
        of the fixture test; total loss 1e-4."""
    from oracle import cpr_options_oracle as OO
    from oracle import cpr_oracle as O
    from pointtinybenchmark_amd.training import CprTrainer
    worst_cls, worst_other, events, n_smooth = 0.0, (0.0, None, None), [], 0
    for ds in (1, 2, 3):
        cfg = dict(grad_option_cfg(name))
        cfg['seed'] = cfg['seed'] + 1000 + ds
        m, batch = build_hip(cfg)
        cb = cuda_batch(batch)
        tr = CprTrainer(m)
        losses = tr.forward_backward(img=cb['img'], img_metas=cb['img_metas'], gt_bboxes=cb['gt_bboxes'], gt_labels=cb['gt_labels'])
        torch.cuda.synchronize()
        total = float(sum(v for k, v in losses.items() if 'loss' in k))
        got = {k: q.grad.detach().double().cpu() for k, q in m.named_parameters() if q.requires_grad and q.grad is not None}
        with torch.no_grad():
            dcf, dif = m.bbox_head(m.neck(m.backbone(cb['img'])))
        dcf = dcf[0].float().cpu().contiguous()
        dif = None if not cfg.get('ins_tower') else (dif[0] if isinstance(dif, (list, tuple)) else dif).float().cpu().contiguous()
        del tr, m
        sd, _ = case_inputs(cfg)
        sd = {k: v.clone() for k, v in sd.items()}
        for k in got:
            sd[k].requires_grad_(True)
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        # (1) the classifier tensors on identical features
        clsk = [k for k in got if k.startswith(('bbox_head.cls_out.', 'bbox_head.ins_out.'))]
        dl, _ = OO.cpr_loss(sd, dcf, batch['gt_bboxes'], batch['gt_labels'], batch['img_metas'], cfg, ins_feat=dif)
        dt = sum(v for k, v in dl.items() if 'loss' in k)
        assert abs(total - float(dt.detach())) <= 5e-5 * max(1.0, abs(float(dt.detach()))), (ds, total, float(dt.detach()))
        dg = torch.autograd.grad(dt, [sd[k] for k in clsk], allow_unused=True)
        cmax = max(float(g_.double().norm()) for g_ in dg if g_ is not None)
        # is the oracle's gradient SMOOTH at this point?  The same features times (1 + 1e-7 noise): on a smooth point the gradient
        # moves by ~1e-6; where a probability sits on a clamp of the loss (saturated logits: head_std = 0.3 gives neg_loss ~ 1600 on
        # combo_tower_binary_normed seed 1179) one element's gradient switches on or off and cls_out.weight jumps by 6.5e-4 under
        # that 1e-7 perturbation -- exactly what the device is off by there.  Such a point has no 1e-4 answer: bar = 2 x the jump.
        gen = torch.Generator().manual_seed(7)
        pcf = dcf * (1 + 1e-7 * torch.randn(dcf.shape, generator=gen))
        pif = None if dif is None else dif * (1 + 1e-7 * torch.randn(dif.shape, generator=gen))
        pl, _ = OO.cpr_loss(sd, pcf, batch['gt_bboxes'], batch['gt_labels'], batch['img_metas'], cfg, ins_feat=pif)
        pg = torch.autograd.grad(sum(v for k, v in pl.items() if 'loss' in k), [sd[k] for k in clsk], allow_unused=True)
        jump = {k: 0.0 if (a_ is None or b_ is None or float(a_.double().norm()) <= 1e-6 * cmax) else
                float((a_.double() - b_.double()).norm()) / float(a_.double().norm()) for k, a_, b_ in zip(clsk, dg, pg)}
        smooth = max(jump.values()) <= 2e-5
        n_smooth += int(smooth)
        for k, g_ in zip(clsk, dg):
            if g_ is None:
                assert not bool(got[k].any()), k
                continue
            ref = g_.detach().double()
            diff = float((got[k] - ref).norm())
            if float(ref.norm()) <= 1e-6 * cmax:        # numerically nil (ins_out.bias under normed_sigmoid: its true gradient cancels)
                assert diff <= 1e-6 * cmax, (name, cfg['seed'], k, diff, cmax)
                continue
            err = diff / float(ref.norm())
            if smooth:
                worst_cls = max(worst_cls, err)
                assert err <= 1e-4, (name, cfg['seed'], k, err, jump[k])
            else:           # a clamp event at this seed: reported, held to a sanity bound (ten times the oracle's own jump) only
                events.append((cfg['seed'], k, '%.2e off, the oracle itself jumps %.2e under a 1e-7 perturbation' % (err, jump[k])))
                assert err <= max(1e-4, 10 * max(jump.values())), (name, cfg['seed'], k, err, jump)
        # (2) everything else against the full oracle graph
        feats = O.fpn_forward(sd, O.resnet_forward(sd, batch['img'], cfg['depth']), cfg['start_level'], 1)
        cls_feat, _ = O.cpr_head_forward(sd, feats)
        ins_feat = OO.ins_tower_forward(sd, feats)[0] if cfg.get('ins_tower') else None
        ol, _ = OO.cpr_loss(sd, cls_feat[0], batch['gt_bboxes'], batch['gt_labels'], batch['img_metas'], cfg, ins_feat=ins_feat)
        ot = sum(v for k, v in ol.items() if 'loss' in k)
        assert abs(total - float(ot.detach())) <= 1e-4 * max(1.0, abs(float(ot.detach()))), (ds, total, float(ot.detach()))
        ot.backward()
        gmax = max(float(sd[k].grad.double().norm()) for k in got if sd[k].grad is not None)
        for k, gr in got.items():
            if k in clsk:
                continue
            if sd[k].grad is None:
                assert not bool(gr.any()), k
                continue
            ref = sd[k].grad.detach().double()
            if float(ref.norm()) <= 1e-6 * gmax:
                assert float((gr - ref).norm()) <= 1e-6 * gmax, (name, cfg['seed'], k, float((gr - ref).norm()), gmax)
                continue
            err = float((gr - ref).norm()) / float(ref.norm())
            if err > worst_other[0]:
                worst_other = (err, k, cfg['seed'])
            assert err <= 1e-2, (name, cfg['seed'], k, err)
    print('%s: %d of 3 seeds smooth; classifier tensors on identical features <= %.2e on smooth points, other tensors vs the full oracle graph <= %.2e (%s, seed %s)%s'
          % ((name, n_smooth, worst_cls) + worst_other + ('; non-smooth points: %r' % events if events else '',)))
