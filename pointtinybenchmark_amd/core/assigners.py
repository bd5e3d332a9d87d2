"""Assigners, samplers and match costs of the P2P / point path, same registry names as the reference.

  PointAssigner        T/mmdet/core/bbox/assigners/point_assigner.py:8-133
  HungarianAssignerV2  T/mmdet/core/bbox/assigners/hungarian_assigner.py:149-270
  HungarianAssigner    T/mmdet/core/bbox/assigners/hungarian_assigner.py:15-145 (DETR form; round 6)
  FocalLossCost        T/mmdet/core/bbox/match_costs/match_cost.py:54-100
  DisCostV2            T/mmdet/core/bbox/match_costs/match_cost.py:190-214
  ClassificationCost(V2), ZeroCost, BBoxL1Cost, IoUCost(V2)   match_cost.py:9-52,102-188,217-245 (round 6: the general cost kernel)
  PseudoSampler        T/mmdet/core/bbox/samplers/pseudo_sampler.py:9-41
  AssignResult         T/mmdet/core/bbox/assigners/assign_result.py

The Hungarian step runs on the device (cost kernel + shortest-augmenting-path LSA, one workgroup per image):
the reference's ``cost.cpu()`` -> scipy -> ``.to(device)`` round trip (hungarian_assigner.py:230-261) is gone."""
import torch

from .. import ops
from ..registry import BBOX_ASSIGNERS, BBOX_SAMPLERS, MATCH_COST, build_match_cost


class AssignResult:
    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels

    @property
    def num_preds(self):
        return len(self.gt_inds)


class SamplingResult:
    def __init__(self, pos_inds, neg_inds, bboxes, gt_bboxes, assign_result, gt_flags):
        self.pos_inds, self.neg_inds = pos_inds, neg_inds
        self.pos_bboxes, self.neg_bboxes = bboxes[pos_inds], bboxes[neg_inds]
        self.pos_is_gt = gt_flags[pos_inds]
        self.num_gts = gt_bboxes.shape[0]
        self.pos_assigned_gt_inds = assign_result.gt_inds[pos_inds] - 1
        if gt_bboxes.numel() == 0:
            self.pos_gt_bboxes = torch.empty_like(gt_bboxes).view(-1, gt_bboxes.shape[-1] if gt_bboxes.dim() > 1 else 4)
        else:
            self.pos_gt_bboxes = gt_bboxes[self.pos_assigned_gt_inds, :]
        self.pos_gt_labels = assign_result.labels[pos_inds] if assign_result.labels is not None else None


@BBOX_SAMPLERS.register_module()
class PseudoSampler:
    def __init__(self, **kwargs):
        pass

    def sample(self, assign_result, bboxes, gt_bboxes, **kwargs):
        pos_inds = torch.nonzero(assign_result.gt_inds > 0, as_tuple=False).squeeze(-1).unique()
        neg_inds = torch.nonzero(assign_result.gt_inds == 0, as_tuple=False).squeeze(-1).unique()
        gt_flags = bboxes.new_zeros(bboxes.shape[0], dtype=torch.uint8)
        return SamplingResult(pos_inds, neg_inds, bboxes, gt_bboxes, assign_result, gt_flags)


@MATCH_COST.register_module()
class FocalLossCost:
    def __init__(self, weight=1., alpha=0.25, gamma=2, eps=1e-12):
        self.weight, self.alpha, self.gamma, self.eps = weight, alpha, gamma, eps


@MATCH_COST.register_module()
class DisCostV2:
    def __init__(self, weight=1., norm_with_img_wh=True, p=1):
        assert p in (1, 2), 'DisCostV2: p = 1 (the shipped P2P config) or 2'
        self.weight, self.norm_with_img_wh, self.p = weight, norm_with_img_wh, p

    def term(self, img_meta):
        fx = fy = 1.0
        if self.norm_with_img_wh:
            h, w = img_meta['img_shape'][:2]
            fx, fy = float(w), float(h)
        return (0, self.weight, float(self.p), fx, fy)


# Round 6: the other costs of match_cost.py.  The classes only carry their parameters (the arithmetic is csrc/assign.hip,
# match_cost_kernel); ``term()`` is the (type, weight, a, b, c) tuple of include/cpr_hip.h, cpr_match_cost.  Signatures as in the
# reference, INCLUDING the ones that make a config unusable there: the V2 assigner calls every regression cost with
# (pred, gt, img_meta) (hungarian_assigner.py:227), which BBoxL1Cost / IoUCost / ZeroCost do not accept -- the same TypeError is
# raised here -- and IoUCostV2 on 2-column points fails in bbox_overlaps' shape assertion.
@MATCH_COST.register_module()
class ClassificationCost:
    def __init__(self, weight=1.):
        self.weight = weight

    def term(self):
        return (1, self.weight)


@MATCH_COST.register_module()
class ClassificationCostV2:
    def __init__(self, weight=1., use_sigmoid=False):
        self.weight, self.use_sigmoid = weight, use_sigmoid

    def term(self):
        return (2 if self.use_sigmoid else 1, self.weight)


@MATCH_COST.register_module()
class ZeroCost:
    def term(self):
        return (3, 0.0)


@MATCH_COST.register_module()
class BBoxL1Cost:
    def __init__(self, weight=1., box_format='xyxy', same_fmt=False):
        assert box_format in ['xyxy', 'xywh']
        self.weight, self.box_format, self.same_fmt = weight, box_format, same_fmt


@MATCH_COST.register_module()
class IoUCost:
    def __init__(self, iou_mode='giou', weight=1.):
        assert iou_mode in ('iou', 'giou'), iou_mode
        self.weight, self.iou_mode = weight, iou_mode

    def term(self):
        return (3 if self.iou_mode == 'giou' else 2, self.weight)


@MATCH_COST.register_module()
class IoUCostV2(IoUCost):
    pass


def _cls_term(c):
    if isinstance(c, FocalLossCost):
        return (0, c.weight, c.alpha, float(c.gamma), c.eps)
    if isinstance(c, (ClassificationCost, ClassificationCostV2, ZeroCost)):
        return c.term()
    raise TypeError('%s is not a classification cost' % type(c).__name__)


@BBOX_ASSIGNERS.register_module()
class PointAssigner:
    def __init__(self, scale=4, pos_num=3):
        self.scale, self.pos_num = scale, pos_num

    def assign(self, points, gt_bboxes, gt_bboxes_ignore=None, gt_labels=None):
        n, k = points.shape[0], gt_bboxes.shape[0]
        if k == 0 or n == 0:
            inds = points.new_full((n,), 0, dtype=torch.long)
            labels = None if gt_labels is None else points.new_full((n,), -1, dtype=torch.long)
            return AssignResult(k, inds, None, labels=labels)
        inds = ops.point_assign(points.float().contiguous(), gt_bboxes.float().contiguous(), self.scale, self.pos_num)
        labels = None
        if gt_labels is not None:
            labels = inds.new_full((n,), -1)
            pos = inds > 0
            labels[pos] = gt_labels[inds[pos] - 1]
        return AssignResult(k, inds, None, labels=labels)


@BBOX_ASSIGNERS.register_module()
class HungarianAssignerV2:
    def __init__(self, cls_costs=[dict(type='ClassificationCost', weight=1.)],
                 reg_costs=[dict(type='BBoxL1Cost', weight=1.0, norm_with_img_size=True),
                            dict(type='IoUCost', iou_mode='giou', weight=1.0)], topk_k=1):
        cls_costs = cls_costs if isinstance(cls_costs, (tuple, list)) else [cls_costs]
        reg_costs = reg_costs if isinstance(reg_costs, (tuple, list)) else [reg_costs]
        self.cls_costs = [build_match_cost(c) for c in cls_costs]
        self.reg_costs = [build_match_cost(c) for c in reg_costs]
        assert len(self.cls_costs) <= 4 and len(self.reg_costs) <= 4, 'the cost kernel sums at most four costs of a kind'
        # the shipped P2P pair keeps its fused kernel (bit-exact cost, round 1); every other list goes through the general one
        self.fused = len(self.cls_costs) == 1 and type(self.cls_costs[0]) is FocalLossCost and \
            len(self.reg_costs) == 1 and type(self.reg_costs[0]) is DisCostV2
        self.topk_k = topk_k

    def cost_t(self, bbox_pred, cls_pred, gt_bboxes, gt_labels, img_meta):
        """cost^T (G, M): the reference's (M, G) cost matrix, stored gt-major for coalesced LSA scans."""
        if not self.fused:
            if bbox_pred.shape[1] == 3:          # (x, y, stride) rows of P2PHead.get_pred_points: the costs see the point
                bbox_pred = bbox_pred[:, :2]
            cls_terms = [_cls_term(c) for c in self.cls_costs]
            reg_terms = []
            for rc in self.reg_costs:
                if isinstance(rc, DisCostV2):
                    reg_terms.append(rc.term(img_meta))
                elif isinstance(rc, IoUCostV2):
                    assert bbox_pred.size(-1) == 4 or bbox_pred.size(0) == 0       # bbox_overlaps' own assertion (iou2d_calculator.py)
                    reg_terms.append(rc.term())
                else:
                    # reg_cost(bbox_pred, gt_bboxes, img_meta), hungarian_assigner.py:227: BBoxL1Cost / IoUCost / ZeroCost take two
                    raise TypeError('%s.__call__() takes 3 positional arguments but 4 were given' % type(rc).__name__)
            return ops.match_cost(bbox_pred.float().contiguous(), cls_pred.float().contiguous(), gt_bboxes.float().contiguous(),
                                  gt_labels.to(torch.int32).contiguous(), cls_terms, reg_terms)
        cc, rc = self.cls_costs[0], self.reg_costs[0]
        fx = fy = 1.0
        if rc.norm_with_img_wh:
            h, w = img_meta['img_shape'][:2]
            fx, fy = float(w), float(h)
        return ops.hungarian_cost(bbox_pred.float().contiguous(), cls_pred.float().contiguous(),
                                  gt_bboxes.float().contiguous(), gt_labels.to(torch.int32).contiguous(), cc.weight,
                                  cc.alpha, float(cc.gamma), cc.eps, rc.weight, fx, fy, rc.p)

    @staticmethod
    def transposed_inds(costT_list):
        """gt_inds of problems with FEWER proposals than gts (topk_k == 1).  costT (G, M), M < G: the device solver wants
        rows <= columns, so it gets the (M, G) matrix -- the proposals in the role of its rows, exactly scipy's orientation
        for this shape -- and answers per gt; the answer is inverted into per-proposal indices (j + 1 = gt j)."""
        per_gt, _ = ops.lsa_topk([c.t().contiguous() for c in costT_list], 1)      # (G,): 1 + proposal, 0 = none
        out = []
        for c, pg in zip(costT_list, per_gt):
            inds = pg.new_zeros((c.shape[1],))
            g_of = torch.nonzero(pg > 0, as_tuple=False).squeeze(1)
            inds[pg[g_of] - 1] = g_of + 1
            out.append(inds)
        return out

    def assign(self, bbox_pred, cls_pred, gt_bboxes, gt_labels, img_meta, gt_bboxes_ignore=None, eps=1e-7):
        assert gt_bboxes_ignore is None, 'Only case when gt_bboxes_ignore is None is supported.'
        num_gts, num_bboxes = gt_bboxes.size(0), bbox_pred.size(0)
        inds = bbox_pred.new_full((num_bboxes,), -1, dtype=torch.long)
        labels = bbox_pred.new_full((num_bboxes,), -1, dtype=torch.long)
        if num_gts == 0 or num_bboxes == 0:
            if num_gts == 0:
                inds[:] = 0
            return AssignResult(num_gts, inds, None, labels=labels)
        costT = None
        if num_bboxes <= num_gts:
            if self.topk_k != 1 and num_bboxes < num_gts:
                inds[:] = 0  # the reference's loop condition fails immediately (hungarian_assigner.py:251)
                return AssignResult(num_gts, inds, None, labels=labels)
            # topk_k == 1 (hungarian_assigner.py:229-240): scipy keeps the FEWER proposals as rows, each gets a distinct gt.
            # As many proposals as gts (any topk_k: one round assigns everything): scipy does not transpose a square matrix
            # either -- the proposals stay its rows, which decides WHICH of several exactly tied optima comes out
            costT = self.cost_t(bbox_pred, cls_pred, gt_bboxes, gt_labels, img_meta)
            inds = self.transposed_inds([costT])[0]
        else:
            costT = self.cost_t(bbox_pred, cls_pred, gt_bboxes, gt_labels, img_meta)
            (inds,), status = ops.lsa_topk([costT], self.topk_k)
        pos = inds > 0
        labels[pos] = gt_labels[inds[pos] - 1]
        return AssignResult(num_gts, inds, None, labels=labels)


@BBOX_ASSIGNERS.register_module()
class HungarianAssigner:
    """The DETR-form assigner (hungarian_assigner.py:15-145): bbox_pred normalised (cx, cy, w, h), gt_bboxes unnormalised xyxy;
    cost = cls_cost + L1 cost on normalised xyxy + IoU cost on unnormalised xyxy, one assignment round.  Off the point path (no
    shipped config uses it); registered because it is the Hungarian assigner the reference's own tests exercise
    (T/tests/test_utils/test_assigner.py:382-425).  Same device pipeline as V2: general cost kernel + LSA kernel."""

    def __init__(self, cls_cost=dict(type='ClassificationCost', weight=1.), reg_cost=dict(type='BBoxL1Cost', weight=1.0),
                 iou_cost=dict(type='IoUCost', iou_mode='giou', weight=1.0)):
        self.cls_cost, self.reg_cost, self.iou_cost = build_match_cost(cls_cost), build_match_cost(reg_cost), build_match_cost(iou_cost)
        assert isinstance(self.reg_cost, BBoxL1Cost) and isinstance(self.iou_cost, IoUCost)

    def assign(self, bbox_pred, cls_pred, gt_bboxes, gt_labels, img_meta, gt_bboxes_ignore=None, eps=1e-7):
        assert gt_bboxes_ignore is None, 'Only case when gt_bboxes_ignore is None is supported.'
        num_gts, num_bboxes = gt_bboxes.size(0), bbox_pred.size(0)
        inds = bbox_pred.new_full((num_bboxes,), -1, dtype=torch.long)
        labels = bbox_pred.new_full((num_bboxes,), -1, dtype=torch.long)
        if num_gts == 0 or num_bboxes == 0:
            if num_gts == 0:
                inds[:] = 0
            return AssignResult(num_gts, inds, None, labels=labels)
        img_h, img_w = img_meta['img_shape'][:2]
        factor = gt_bboxes.new_tensor([img_w, img_h, img_w, img_h]).unsqueeze(0)
        cx, cy, w, h = bbox_pred.float().unbind(-1)                       # bbox_cxcywh_to_xyxy (transforms.py)
        xyxy = torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)
        gt = gt_bboxes.float()
        rc = self.reg_cost
        if rc.same_fmt:
            pa, ga = bbox_pred.float(), gt / factor
        elif rc.box_format == 'xywh':
            g = gt / factor
            pa, ga = bbox_pred.float(), torch.stack([(g[:, 0] + g[:, 2]) / 2, (g[:, 1] + g[:, 3]) / 2, g[:, 2] - g[:, 0], g[:, 3] - g[:, 1]], -1)
        else:
            pa, ga = xyxy, gt / factor
        lab = gt_labels.to(torch.int32).contiguous()
        logits = cls_pred.float().contiguous()
        # cost = cls_cost + reg_cost + iou_cost (hungarian_assigner.py:118-126): two launches of the general kernel, summed in that order
        c1 = ops.match_cost(pa.contiguous(), logits, ga.contiguous(), lab, [_cls_term(self.cls_cost)], [(1, rc.weight)])
        c2 = ops.match_cost((xyxy * factor).contiguous(), logits, gt.contiguous(), lab, [], [self.iou_cost.term()])
        costT = (c1 + c2).contiguous()
        if num_bboxes < num_gts:
            gi = HungarianAssignerV2.transposed_inds([costT])[0]
        elif num_bboxes == num_gts:
            gi = HungarianAssignerV2.transposed_inds([costT])[0]
        else:
            (gi,), _ = ops.lsa_topk([costT], 1)
        inds[:] = gi
        pos = inds > 0
        labels[pos] = gt_labels[inds[pos] - 1]
        return AssignResult(num_gts, inds, None, labels=labels)
