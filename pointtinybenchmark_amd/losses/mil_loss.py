"""MILLoss / AllPosLoss (T/mmdet/models/losses/multi_instance_learning_loss.py:121-243).

MILLoss: softmax-over-bag instance weights x class probabilities -> bag probability -> gfocal loss (optionally with the
``binary_ins`` second instance branch, whose bag labels are all zero).  AllPosLoss: every bag point is a positive sample.
One wave per bag in HIP (csrc/cpr_points.hip: mil_bag_kernel); no host sync -- ``num_sample`` stays on the device."""
import torch
import torch.nn as nn

from .. import ops
from ..registry import LOSSES


@LOSSES.register_module()
class MILLoss(nn.Module):
    allpos = False

    def __init__(self, binary_ins=False, loss_weight=1.0, eps=1e-6, loss_type='gfocal_loss'):
        super().__init__()
        if loss_type != 'gfocal_loss':
            raise NotImplementedError('loss_type=%r: only gfocal_loss is built (binary_cross_entropy is not used by any '
                                      'config of the reference)' % loss_type)
        self.binary_ins, self.loss_weight, self.eps, self.loss_type = binary_ins, loss_weight, eps, loss_type

    def forward_logits(self, bag_logits, ins_off, valid_u8, labels_i32, num_classes, weight=None, neg_partial=None,
                       w_gt=0.0, w_neg=0.0, want_bag_ws=False, **geometry):
        """Fused entry used by CPRHead: bag_logits (B,N,J) raw cls logits in [0,C) and ins logits from ins_off on.
        ``geometry``: bags / centres / prob_type / norm_p / neg_from_gt of ops.mil_loss.
        Returns the 5-vector {gt_loss, pos_loss, bag_acc, neg_loss, num_sample}."""
        out, bag_ws = ops.mil_loss(bag_logits, ins_off, valid_u8, labels_i32, num_classes, neg_partial,
                                   self.loss_weight, w_gt, w_neg, gt_weight=weight, eps=self.eps,
                                   binary_ins=self.binary_ins, allpos=self.allpos, **geometry)
        return (out, bag_ws) if want_bag_ws else out   # bag_ws (B,5): per-bag terms the backward kernel re-reads

    def forward(self, bag_cls_prob, bag_ins_outs, labels, valid, weight=None):
        """Reference signature: (B,N,C) class PROBABILITIES, (B,N,C or 2C) instance logits, (B,) labels, (B,N,1) validity
        (x weight) -> (loss, acc, num_sample)."""
        assert weight is None, 'the reference never passes `weight` (cpr_head.py:1213)'
        B, N, C = bag_cls_prob.shape
        assert bag_ins_outs.shape[-1] == C * (2 if self.binary_ins else 1)
        logits = torch.cat([bag_cls_prob.float(), bag_ins_outs.float()], dim=-1).contiguous()
        v = valid.reshape(B, N, -1)[..., 0].float()
        w = v.max(dim=1)[0]                                       # per-bag weight (valid carries valid * gt_weight)
        vu8 = (v > 0).to(torch.uint8).contiguous()
        out = self.forward_logits(logits, C, vu8, labels.to(torch.int32).contiguous(), C,
                                  weight=torch.where(w > 0, w, torch.ones_like(w)).contiguous(), centres=(0, 1, 0, 1),
                                  prob_type='identity')
        return out[1], out[2], out[4]

    def gfocal_loss(self, p, q, w=1.0):
        l1 = (p - q) ** 2
        l2 = q * (p + self.eps).log() + (1 - q) * (1 - p + self.eps).log()
        return -(l1 * l2 * w).sum(dim=-1)


@LOSSES.register_module()
class AllPosLoss(MILLoss):
    """multi_instance_learning_loss.py:206-243.  The reference returns ``loss + bag_ins_outs * 0`` (a (B,N,C) tensor whose
    every element is the loss, so that the unused instance branch still receives a zero gradient); its mean -- what
    BaseDetector._parse_losses logs and back-propagates -- is the scalar returned here."""
    allpos = True

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        assert not self.binary_ins, 'AllPosLoss ignores the instance branch'
