"""P2PHead / Hungarian-assigner option surface (round 6) against fixtures produced by the reference's own classes
(tests/golden/p2p_options.npz, oracle/gen_golden_r6.py): losses, assignments (bit-identical indices), reference autograd gradients,
detections; HungarianAssigner (v1) incl. the reference's own property tests as data (T/tests/test_utils/test_assigner.py:382-425)."""
import os

import numpy as np
import pytest
import torch

from oracle.gen_golden import GN, assigner_inputs, grad_sample_index
from oracle.gen_golden_r6 import COST_CASES, HEAD_CASES, TEST_CFG, head_inputs, head_state_dict, v1_inputs

pytestmark = pytest.mark.gpu


def build_head(cfg):
    import pointtinybenchmark_amd as P
    head = P.build_head(dict(type='P2PHead', norm_cfg=GN, num_classes=cfg['C'], in_channels=256, feat_channels=256, stacked_convs=4,
                             strides=cfg['strides'], point_anchor=cfg['anchors'], loss_cls=cfg['loss_cls'], loss_reg=cfg['loss_reg'],
                             pts_gamma=1, reg_norm=1,
                             train_cfg=dict(neg_weight=1.0, assigner=cfg['assigner'], sampler=dict(type='PseudoSampler')),
                             test_cfg=dict(TEST_CFG))).cuda()
    head.load_state_dict({k[len('bbox_head.'):]: v for k, v in head_state_dict(cfg).items()}, strict=True)
    return head


def same_up_to_equal_score_runs(gd, gl, rd, rl):
    """The same detections (coordinates in the leading columns, score last, label) in the same order, except that detections whose
    reference scores agree to 2e-6 may swap places."""
    assert gd.shape == rd.shape and gl.shape == rl.shape
    nc = rd.shape[1] - 1
    used = np.zeros(len(rd), bool)
    for i in range(len(gd)):
        d = np.abs(rd[:, :nc] - gd[i, :nc]).max(1) + (rl != gl[i]) * 1e3 + used * 1e3
        j = int(d.argmin())
        assert d[j] <= 2e-3 and abs(rd[j, nc] - gd[i, nc]) <= 2e-6, (i, float(d[j]))
        used[j] = True
        assert i == j or abs(rd[i, nc] - rd[j, nc]) <= 2e-6, 'order differs outside a run of equal scores: %d vs %d' % (i, j)


@pytest.mark.parametrize('name', list(HEAD_CASES))
def test_p2p_head_options_vs_reference_fixture(golden_dir, name):
    g = np.load(os.path.join(golden_dir, 'p2p_options.npz'))
    cfg = HEAD_CASES[name]
    head = build_head(cfg)
    feats, batch = head_inputs(cfg)
    gtb, gtl = [b.cuda() for b in batch['gt_bboxes']], [l.cuda() for l in batch['gt_labels']]
    with torch.no_grad():
        cls_outs, pts_outs = head(tuple(f.cuda() for f in feats))
        for lvl in range(len(cfg['strides'])):
            for got, key in ((cls_outs[lvl], 'cls_out'), (pts_outs[lvl], 'pts_out')):
                ref = torch.from_numpy(g['%s:%s%d' % (name, key, lvl)])
                assert float((got.cpu() - ref).abs().max()) <= 2e-4 * max(1.0, float(ref.abs().max())), (key, lvl)
        losses = head.loss(cls_outs, pts_outs, gtb, gtl, batch['img_metas'])
        anchor, pred, vflag, co = head.get_pred_points(cls_outs, pts_outs, batch['img_metas'])
        lab, lw, tgt, pw = head.get_targets(pred[..., :2].contiguous(), vflag, co, head.pseudo_bbox_to_center(gtb), gtl, batch['img_metas'])
        torch.cuda.synchronize()
    # the assignment: bit-identical labels / targets (the indices behind them are the assigner's)
    assert np.array_equal(torch.stack(lab).cpu().numpy().astype(np.int32), g[name + ':target_labels']), 'assignment differs from the reference'
    np.testing.assert_allclose(torch.stack(tgt).cpu().numpy(), g[name + ':target_pts'], rtol=0, atol=1e-4)
    for key in ('loss_cls', 'loss_pts'):
        got = np.array([float(v) for v in losses[key]])
        np.testing.assert_allclose(got, g['%s:%s' % (name, key)], rtol=3e-4, atol=1e-7, err_msg=key)
    with torch.no_grad():
        res = head.get_bboxes(cls_outs, pts_outs, batch['img_metas'])
        for b, (bs, l) in enumerate(res):
            rd, rl = g['%s:det%d' % (name, b)], g['%s:detlabel%d' % (name, b)]
            assert bs.shape[0] == rd.shape[0], (b, bs.shape, rd.shape)
            if head.use_sigmoid_cls:
                assert np.array_equal(l.cpu().numpy(), rl)
                np.testing.assert_allclose(bs.cpu().numpy(), rd, rtol=0, atol=2e-3)
            else:
                # softmax scores come from the device's softmax, not torch's CPU kernel (no bit-exact restatement: the row sum's order is the
                # host's vector width), so detections whose scores agree to 1e-6 may swap places: same detections, order within such runs free
                same_up_to_equal_score_runs(bs.cpu().numpy(), l.cpu().numpy(), rd, rl)
        if not head.use_sigmoid_cls:
            d, l = head._get_bboxes_single(pred[0][..., :2], vflag[0], co[0], batch['img_metas'][0]['img_shape'],
                                           batch['img_metas'][0]['scale_factor'], None, False, with_nms=False)
            same_up_to_equal_score_runs(d.cpu().numpy(), l.cpu().numpy(), g[name + ':nonms_det'], g[name + ':nonms_label'])


@pytest.mark.parametrize('name', [n for n, c in HEAD_CASES.items() if c['grads']])
def test_p2p_head_option_backward_vs_reference_autograd(golden_dir, name):
    """loss.backward() through the reference's P2PHead with these losses / costs against the HIP backward (head-only trainer rules)."""
    from pointtinybenchmark_amd.training import P2PTrainer
    g = np.load(os.path.join(golden_dir, 'p2p_options.npz'))
    cfg = HEAD_CASES[name]
    head = build_head(cfg)

    class HeadOnly(P2PTrainer):
        def __init__(self, head):
            self.side = None
            for p in head.parameters():
                p.grad = torch.zeros_like(p)

        def _done(self, p):
            pass
    tr = HeadOnly(head)
    feats, batch = head_inputs(cfg)
    raw = feats[0].permute(0, 2, 3, 1).contiguous().cuda()
    ones, zeros = torch.ones((2, 256), device='cuda'), torch.zeros((2, 256), device='cuda')
    losses, saved = tr._forward_head(head, [(raw, (ones, zeros))], batch['img_metas'], [b.cuda() for b in batch['gt_bboxes']],
                                     [l.cuda() for l in batch['gt_labels']], None, None)
    dfeat = tr._backward_head(head, saved)
    torch.cuda.synchronize()
    total = sum(float(v) for k, vs in losses.items() for v in vs)
    ref_total = float(g[name + ':total_loss'])
    assert abs(total - ref_total) <= 3e-4 * max(1.0, abs(ref_total)), (total, ref_total)
    got = {'bbox_head.' + n: p.grad for n, p in head.named_parameters()}
    got['feat'] = dfeat.permute(0, 3, 1, 2).contiguous()
    keys = [k.split(':', 2)[2] for k in g.files if k.startswith(name + ':norm:')]
    assert sorted(keys) == sorted(got)
    gmax = max(float(g['%s:norm:%s' % (name, k)]) for k in keys)
    for k in keys:
        gr = got[k].detach().double().flatten().cpu()
        ref_n = float(g['%s:norm:%s' % (name, k)])
        smp = gr[torch.from_numpy(grad_sample_index(gr.numel()))].numpy()
        ref = g['%s:sample:%s' % (name, k)].astype(np.float64)
        rel = np.linalg.norm(smp - ref) / max(np.linalg.norm(ref), 1e-5 * gmax)
        # (bars as tests/test_gpu_p2p.py::test_p2p_head_backward_vs_reference_autograd: the regression tower's gradient lives on a few dozen
        # positives, one fp32 ReLU flip moves a whole tensor by ~1e-2)
        bar = 2e-3 if ('cls_' in k or 'reg_out' in k or 'reg_convs.3' in k) else 3e-2
        assert rel <= bar, (k, rel)
        assert abs(float(gr.norm()) - ref_n) <= bar * ref_n + 1e-6 * gmax, (k, float(gr.norm()), ref_n)


@pytest.mark.parametrize('ci', range(len(COST_CASES)))
def test_hungarian_v2_cost_lists_vs_reference(golden_dir, ci):
    """Any list of the registered costs through the general cost kernel: indices bit-identical to the reference's scipy assignment on its
    own cost, the cost itself within 2 ulp-class tolerance (sigmoid / focal / distance terms carry the CPU bits; the softmax row sum can
    differ in the last bit: csrc/assign.hip, match_cost_kernel)."""
    import pointtinybenchmark_amd as P
    g = np.load(os.path.join(golden_dir, 'p2p_options.npz'))
    seed, n_side, G, C, cc, rc, k = COST_CASES[ci]
    pred, logits, gt, labels, shape = assigner_inputs(seed, n_side, 4, G, C)
    a = P.build_assigner(dict(type='HungarianAssignerV2', cls_costs=cc, reg_costs=rc, topk_k=k))
    assert not a.fused
    costT = a.cost_t(pred.cuda(), logits.cuda(), gt.cuda(), labels.cuda(), dict(img_shape=shape))
    ref_cost = g['cost%d:cost' % ci]
    np.testing.assert_allclose(costT.t().cpu().numpy(), ref_cost, rtol=3e-7, atol=3e-7)
    res = a.assign(pred.cuda(), logits.cuda(), gt.cuda(), labels.cuda(), dict(img_shape=shape))
    assert np.array_equal(res.gt_inds.cpu().numpy().astype(np.int32), g['cost%d:gt_inds' % ci])
    assert np.array_equal(res.labels.cpu().numpy().astype(np.int32), g['cost%d:labels' % ci])


def test_hungarian_v2_errors_are_the_references(golden_dir):
    import pointtinybenchmark_amd as P
    g = np.load(os.path.join(golden_dir, 'p2p_options.npz'))
    with pytest.raises(TypeError) as e:
        P.build_assigner(dict(type='HungarianAssignerV2'))                       # its default reg_costs carry a keyword BBoxL1Cost does not take
    assert 'norm_with_img_size' in str(e.value) and 'norm_with_img_size' in str(g['v2_default_error'])
    a = P.build_assigner(dict(type='HungarianAssignerV2', cls_costs=dict(type='FocalLossCost'), reg_costs=dict(type='BBoxL1Cost'), topk_k=1))
    pred, logits, gt, labels, _ = assigner_inputs(31, 8, 4, 3, 1)
    with pytest.raises(TypeError) as e:
        a.assign(pred.cuda(), logits.cuda(), gt.cuda(), labels.cuda(), dict(img_shape=(32, 32, 3)))
    assert str(e.value) == str(g['v2_bboxl1_error'])


def test_hungarian_assigner_v1(golden_dir):
    """HungarianAssigner (DETR form): the reference's own property tests as data + indices equal to the reference on random problems."""
    import pointtinybenchmark_amd as P
    g = np.load(os.path.join(golden_dir, 'p2p_options.npz'))
    for vi, seed in enumerate((41, 42, 43)):
        bp, lg, gts, lbl, meta = v1_inputs(seed, C=4 if vi < 2 else 1)
        kw = {} if vi != 1 else dict(iou_cost=dict(type='IoUCost', iou_mode='iou', weight=2.0), reg_cost=dict(type='BBoxL1Cost', weight=5.0))
        a = P.build_assigner(dict(type='HungarianAssigner', **kw))
        res = a.assign(bp.cuda(), lg.cuda(), gts.cuda(), lbl.cuda(), meta)
        assert np.array_equal(res.gt_inds.cpu().numpy().astype(np.int32), g['v1_%d:gt_inds' % vi]), vi
        assert np.array_equal(res.labels.cpu().numpy().astype(np.int32), g['v1_%d:labels' % vi]), vi
    # T/tests/test_utils/test_assigner.py:382-425
    a = P.build_assigner(dict(type='HungarianAssigner'))
    bbox_pred, cls_pred = torch.rand((10, 4), generator=torch.Generator().manual_seed(1)), torch.rand((10, 81), generator=torch.Generator().manual_seed(2))
    r0 = a.assign(bbox_pred.cuda(), cls_pred.cuda(), torch.empty((0, 4)).float().cuda(), torch.empty((0,)).long().cuda(), dict(img_shape=(10, 8, 3)))
    assert torch.all(r0.gt_inds == 0) and torch.all(r0.labels == -1) and bool(g['v1_prop:no_gt_all_zero'])
    gtb, gtl = torch.FloatTensor([[0, 0, 5, 7], [3, 5, 7, 8]]), torch.LongTensor([1, 20])
    for tag, kw in (('default', {}), ('iou', dict(iou_cost=dict(type='IoUCost', iou_mode='iou', weight=1.0))),
                    ('focal', dict(cls_cost=dict(type='FocalLossCost', weight=1.)))):
        r = P.build_assigner(dict(type='HungarianAssigner', **kw)).assign(bbox_pred.cuda(), cls_pred.cuda(), gtb.cuda(), gtl.cuda(), dict(img_shape=(10, 8, 3)))
        assert int((r.gt_inds > 0).sum()) == gtb.size(0) and int((r.labels > -1).sum()) == gtb.size(0)      # the reference's assertions
        assert np.array_equal(r.gt_inds.cpu().numpy().astype(np.int32), g['v1_prop:%s:gt_inds' % tag]), tag
        assert np.array_equal(r.labels.cpu().numpy().astype(np.int32), g['v1_prop:%s:labels' % tag]), tag
