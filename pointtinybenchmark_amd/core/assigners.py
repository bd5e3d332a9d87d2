"""Assigners, samplers and match costs of the P2P / point path, same registry names as the reference.

  PointAssigner        T/mmdet/core/bbox/assigners/point_assigner.py:8-133
  HungarianAssignerV2  T/mmdet/core/bbox/assigners/hungarian_assigner.py:149-270
  FocalLossCost        T/mmdet/core/bbox/match_costs/match_cost.py:54-100
  DisCostV2            T/mmdet/core/bbox/match_costs/match_cost.py:190-214
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
        assert len(self.cls_costs) == 1 and isinstance(self.cls_costs[0], FocalLossCost) and \
            len(self.reg_costs) == 1 and isinstance(self.reg_costs[0], DisCostV2), \
            'the fused cost kernel implements the P2P config: FocalLossCost + DisCostV2'
        self.topk_k = topk_k

    def cost_t(self, bbox_pred, cls_pred, gt_bboxes, gt_labels, img_meta):
        """cost^T (G, M): the reference's (M, G) cost matrix, stored gt-major for coalesced LSA scans."""
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
