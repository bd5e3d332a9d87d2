"""TEST INFRASTRUCTURE (see oracle/__init__.py): CPU restatement of the P2PHead / Hungarian-assigner OPTION surface (round 6) -- the
losses, match costs and post-processing branches the shipped P2P config does not use.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this; the product path never does.

Each function cites the reference lines it restates (T = /root/reference/TOV_mmdetection).  Pinned by tests/golden/p2p_options.npz,
which oracle/gen_golden_r6.py produces by running the REFERENCE's own classes (tests/test_oracle_golden.py compares).
The NMS inside p2p_get_bboxes_single is the restated mmcv batched_nms of oracle/cpr_oracle.py: parity unpinned, as everywhere."""
import torch
import torch.nn.functional as F

from . import cpr_oracle as O


# ------------------------------------------------------------------------------------------------ match costs (T/mmdet/core/bbox/match_costs/match_cost.py)
def cls_cost(cfg, cls_pred, gt_labels):
    t, w = cfg['type'], cfg.get('weight', 1.0)
    if t == 'FocalLossCost':                                  # :54-100
        return O.focal_loss_cost(cls_pred, gt_labels, w, cfg.get('alpha', 0.25), cfg.get('gamma', 2), cfg.get('eps', 1e-12))
    if t == 'ClassificationCost' or (t == 'ClassificationCostV2' and not cfg.get('use_sigmoid', False)):    # :102-132, 229-245
        return -cls_pred.softmax(-1)[:, gt_labels] * w
    if t == 'ClassificationCostV2':
        return -cls_pred.sigmoid()[:, gt_labels] * w
    if t == 'ZeroCost':                                       # :222-226
        return 0
    raise KeyError(t)


def bbox_overlaps(b1, b2, mode='iou', eps=1e-6):
    """T/mmdet/core/bbox/iou_calculators/iou2d_calculator.py, is_aligned=False."""
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    lt = torch.max(b1[:, None, :2], b2[None, :, :2])
    rb = torch.min(b1[:, None, 2:], b2[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    ov = wh[..., 0] * wh[..., 1]
    union = a1[:, None] + a2[None] - ov
    union = torch.max(union, union.new_tensor([eps]))
    iou = ov / union
    if mode == 'iou':
        return iou
    elt = torch.min(b1[:, None, :2], b2[None, :, :2])
    erb = torch.max(b1[:, None, 2:], b2[None, :, 2:])
    ewh = (erb - elt).clamp(min=0)
    ea = torch.max(ewh[..., 0] * ewh[..., 1], union.new_tensor([eps]))
    return iou - (ea - union) / ea


def reg_cost_v2(cfg, pred, gt, img_shape):
    """A regression cost as HungarianAssignerV2 calls it: reg_cost(bbox_pred, gt_bboxes, img_meta) (hungarian_assigner.py:227)."""
    t, w = cfg['type'], cfg.get('weight', 1.0)
    if t == 'DisCostV2':                                      # :190-214
        return O.dis_cost_v2(pred, gt, img_shape, w, cfg.get('norm_with_img_wh', True), cfg.get('p', 1))
    if t == 'IoUCostV2':                                      # :217-220 -> IoUCost :134-166
        assert pred.size(-1) == 4
        return -bbox_overlaps(pred, gt, cfg.get('iou_mode', 'giou')) * w
    raise TypeError('%s.__call__() takes 3 positional arguments but 4 were given' % t)


def hungarian_assign_v2(cls_costs, reg_costs, topk_k, pred, cls_pred, gt, gt_labels, img_shape):
    """HungarianAssignerV2.assign (hungarian_assigner.py:166-270) with any cost lists -> (gt_inds, labels, cost)."""
    M, G = pred.shape[0], gt.shape[0]
    labels = torch.full((M,), -1, dtype=torch.long)
    if G == 0 or M == 0:
        return torch.full((M,), 0 if G == 0 else -1, dtype=torch.long), labels, None
    cost = sum(cls_cost(c, cls_pred, gt_labels) for c in cls_costs) + sum(reg_cost_v2(c, pred, gt, img_shape) for c in reg_costs)
    inds = O.lsa_topk(cost.detach(), topk_k)
    pos = inds > 0
    labels[pos] = gt_labels[inds[pos] - 1]
    return inds, labels, cost


def hungarian_assign_v1(bbox_pred, cls_pred, gt_bboxes, gt_labels, img_shape, cls_cfg=None, reg_w=1.0, iou_w=1.0, iou_mode='giou'):
    """HungarianAssigner.assign (hungarian_assigner.py:53-145): bbox_pred normalised (cx, cy, w, h), gt xyxy unnormalised."""
    from scipy.optimize import linear_sum_assignment
    M, G = bbox_pred.shape[0], gt_bboxes.shape[0]
    inds = torch.full((M,), -1, dtype=torch.long)
    labels = torch.full((M,), -1, dtype=torch.long)
    if G == 0 or M == 0:
        if G == 0:
            inds[:] = 0
        return inds, labels
    h, w = img_shape[:2]
    factor = gt_bboxes.new_tensor([w, h, w, h]).unsqueeze(0)
    cx, cy, bw, bh = bbox_pred.unbind(-1)
    xyxy = torch.stack([cx - 0.5 * bw, cy - 0.5 * bh, cx + 0.5 * bw, cy + 0.5 * bh], -1)      # bbox_cxcywh_to_xyxy
    cost = cls_cost(cls_cfg or dict(type='ClassificationCost'), cls_pred, gt_labels) + \
        torch.cdist(xyxy, gt_bboxes / factor, p=1) * reg_w + -bbox_overlaps(xyxy * factor, gt_bboxes, iou_mode) * iou_w
    r, c = linear_sum_assignment(cost.detach())
    inds[:] = 0
    inds[torch.from_numpy(r)] = torch.from_numpy(c) + 1
    labels[torch.from_numpy(r)] = gt_labels[torch.from_numpy(c)]
    return inds, labels


# ------------------------------------------------------------------------------------------------ P2PHead options (T/mmdet/models/point/dense_heads/p2p_head.py)
def get_pred_points(cls_outs, pts_outs, strides, point_anchor, pts_gamma, num_cls_out):
    """:125-170 for any number of levels -> pred (B, M, 3) = (x, y, stride), cls (B, M, num_cls_out), levels in order."""
    k = len(point_anchor)
    preds, clss = [], []
    for co, po, s in zip(cls_outs, pts_outs, strides):
        B, _, H, W = co.shape
        anchor = O.p2p_grid_points(H, W, s)[:, None, :2] + torch.tensor(point_anchor, dtype=torch.float32)[None] * s
        reg = po.permute(0, 2, 3, 1).reshape(B, H * W, k, 2)
        pred = (anchor[None] + reg * pts_gamma * s).reshape(B, H * W * k, 2)
        preds.append(torch.cat([pred, torch.full((B, H * W * k, 1), float(s))], -1))
        clss.append(co.permute(0, 2, 3, 1).reshape(B, H * W * k, num_cls_out))
    return torch.cat(preds, 1), torch.cat(clss, 1)


def p2p_loss(cls_outs, pts_outs, gt_bboxes, gt_labels, img_shape, strides, num_classes, loss_cls, loss_reg, assigner,
             pts_gamma=1.0, reg_norm=1.0, point_anchor=((0., 0.),), pos_weight=1.0, neg_weight=1.0):
    """P2PHead.loss (:172-248) for the loss / assigner options: loss_cls FocalLoss | CrossEntropyLoss(use_sigmoid) | CrossEntropyLoss
    (cross_entropy_loss.py:9-99; avg_factor = all proposals for CrossEntropyLoss, positives for FocalLoss: :220-224), loss_reg
    SmoothL1Loss | MSELoss | L1Loss (avg_factor = positives).  Differentiable.  -> ({'loss_cls': [...], 'loss_pts': [...]}, gt_inds list)."""
    use_sigmoid = loss_cls.get('use_sigmoid', False)
    nco = num_classes if use_sigmoid else num_classes + 1
    pred3, cls = get_pred_points(cls_outs, pts_outs, strides, point_anchor, pts_gamma, nco)
    pred, stride = pred3[..., :2], pred3[..., 2:]
    B, M = cls.shape[:2]
    cc = assigner['cls_costs'] if isinstance(assigner['cls_costs'], (list, tuple)) else [assigner['cls_costs']]
    rc = assigner['reg_costs'] if isinstance(assigner['reg_costs'], (list, tuple)) else [assigner['reg_costs']]
    labels, lw, tgt, pw, inds_all = [], [], [], [], []
    for b in range(B):
        ctr = (gt_bboxes[b][:, :2] + gt_bboxes[b][:, 2:]) / 2
        inds, _, _ = hungarian_assign_v2(cc, rc, assigner.get('topk_k', 1), pred[b].detach(), cls[b].detach(), ctr, gt_labels[b], img_shape)
        pos = inds > 0
        lab = torch.full((M,), num_classes, dtype=torch.long)
        lab[pos] = gt_labels[b][inds[pos] - 1]
        w = torch.full((M,), 1.0 if neg_weight <= 0 else neg_weight)
        w[pos] = pos_weight
        t = torch.zeros((M, 2))
        t[pos] = ctr[inds[pos] - 1]
        ww = torch.zeros((M, 2))
        ww[pos] = 1.0
        labels.append(lab), lw.append(w), tgt.append(t), pw.append(ww), inds_all.append(inds)
    num_pos = sum(int((w[:, 0] > 0).sum()) for w in pw)
    num_total = B * M
    w_cls, w_reg = loss_cls.get('loss_weight', 1.0), loss_reg.get('loss_weight', 1.0)
    loss_c, loss_p = [], []
    for b in range(B):
        if loss_cls['type'] == 'FocalLoss':
            alpha, gamma = loss_cls.get('alpha', 0.25), loss_cls.get('gamma', 2.0)
            target = F.one_hot(labels[b], num_classes=num_classes + 1)[:, :num_classes].type_as(cls)
            ps = cls[b].sigmoid()
            pt = (1 - ps) * target + ps * (1 - target)
            fw = (alpha * target + (1 - alpha) * (1 - target)) * pt.pow(gamma)
            l = F.binary_cross_entropy_with_logits(cls[b], target, reduction='none') * fw * lw[b].view(-1, 1)
            loss_c.append(w_cls * l.sum() / num_pos)
        elif use_sigmoid:
            target = F.one_hot(labels[b], num_classes=num_classes + 1)[:, :num_classes].type_as(cls)        # _expand_onehot_labels :43-56
            l = F.binary_cross_entropy_with_logits(cls[b], target, reduction='none') * lw[b].view(-1, 1)
            loss_c.append(w_cls * l.sum() / num_total)
        else:
            l = F.cross_entropy(cls[b], labels[b], reduction='none') * lw[b]
            loss_c.append(w_cls * l.sum() / num_total)
        e = pred[b] / stride[b] / reg_norm - tgt[b] / stride[b] / reg_norm
        if loss_reg['type'] == 'SmoothL1Loss':
            beta = loss_reg.get('beta', 1.0)
            d = e.abs()
            r = torch.where(d < beta, 0.5 * d * d / beta, d - 0.5 * beta)
        elif loss_reg['type'] == 'MSELoss':
            r = e * e
        else:
            r = e.abs()
        loss_p.append(w_reg * (r * pw[b]).sum() / num_pos)
    return {'loss_cls': loss_c, 'loss_pts': loss_p}, inds_all


def p2p_get_bboxes_single(cls, pred_pts, img_shape, num_levels, use_sigmoid, num_cls_out, nms_pre=2000, score_thr=0.05, iou_thr=0.2,
                          max_per_img=1000, pseudo_wh=(16, 16), with_nms=True):
    """P2PHead._get_bboxes_single (:345-423): the proposals are cut into ``num_levels`` EQUAL chunks by a reshape (:357-358), top
    nms_pre per chunk, sigmoid (+ padded background column) or softmax scores, then multiclass NMS on 16x16 pseudo boxes or -- with_nms
    False, softmax heads only, see the product's note -- every (point, class) pair above score_thr, top max_per_img."""
    pts_l, sc_l = [], []
    for logits, pp in zip(cls.reshape(num_levels, -1, num_cls_out), pred_pts.reshape(num_levels, -1, 2)):
        scores = logits.sigmoid() if use_sigmoid else logits.softmax(-1)
        if 0 < nms_pre < scores.shape[0]:
            mx = scores.max(dim=1)[0] if use_sigmoid else scores[:, :-1].max(dim=1)[0]
            _, ti = mx.topk(nms_pre)
            scores, pp = scores[ti], pp[ti]
        pts_l.append(torch.stack([pp[:, 0].clamp(min=0, max=img_shape[1]), pp[:, 1].clamp(min=0, max=img_shape[0])], -1))
        sc_l.append(scores)
    pts, scores = torch.cat(pts_l), torch.cat(sc_l)
    if use_sigmoid:
        scores = torch.cat([scores, scores.new_zeros(scores.shape[0], 1)], 1)
    if not with_nms:
        n = pts.shape[0]
        mp = pts[:, None].expand(n, num_cls_out, 2).reshape(-1, 2)
        labels = torch.arange(num_cls_out).view(1, -1).expand_as(scores).reshape(-1)
        ms = scores.reshape(-1)
        inds = (ms > score_thr).nonzero(as_tuple=False).squeeze(1)
        dets, labels = torch.cat([mp[inds], ms[inds][:, None]], -1), labels[inds]
        if 0 < max_per_img < len(inds):
            _, idx = ms[inds].topk(max_per_img)
            dets, labels = dets[idx], labels[idx]
        return dets, labels
    wh = pts.new_tensor(pseudo_wh)
    boxes = torch.cat([pts - wh / 2, pts + wh / 2], -1)
    dets, labels, _ = O.multiclass_nms(boxes, scores, score_thr, iou_thr, max_per_img)
    out = torch.stack([(dets[:, 0] + dets[:, 2]) / 2, (dets[:, 1] + dets[:, 3]) / 2, dets[:, 4]], -1) if len(dets) else torch.zeros((0, 3))
    return out, labels
