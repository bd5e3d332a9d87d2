"""MILLoss (T/mmdet/models/losses/multi_instance_learning_loss.py:121-203): softmax-over-bag instance
weights x sigmoid class scores -> bag probability -> gfocal loss.  One wave per bag in HIP
(csrc/cpr_points.hip: mil_bag_kernel); no host sync -- ``num_sample`` stays on the device."""
import torch
import torch.nn as nn

from .. import ops
from ..registry import LOSSES


@LOSSES.register_module()
class MILLoss(nn.Module):
    def __init__(self, binary_ins=False, loss_weight=1.0, eps=1e-6, loss_type='gfocal_loss'):
        super().__init__()
        assert not binary_ins and loss_type == 'gfocal_loss', \
            'binary_ins / binary_cross_entropy are not used by any shipped config (SURVEY.md §8f rank 4)'
        self.binary_ins, self.loss_weight, self.eps, self.loss_type = binary_ins, loss_weight, eps, loss_type

    def forward_logits(self, bag_logits, ins_off, valid_u8, labels_i32, num_classes, weight=None, neg_partial=None,
                       w_gt=0.0, w_neg=0.0, want_bag_ws=False):
        """Fused entry used by CPRHead: bag_logits (B,N,J) raw cls logits in [0,C) and ins logits in
        [ins_off, ins_off+C).  Returns the 5-vector {gt_loss, pos_loss, bag_acc, neg_loss, num_sample}."""
        out, bag_ws = ops.mil_loss(bag_logits, ins_off, valid_u8, labels_i32, num_classes, neg_partial,
                                   self.loss_weight, w_gt, w_neg, gt_weight=weight, eps=self.eps)
        return (out, bag_ws) if want_bag_ws else out   # bag_ws (B,5): per-bag terms the backward kernel re-reads

    def forward(self, bag_cls_prob, bag_ins_outs, labels, valid, weight=None):
        """Reference signature: (B,N,C) probabilities, (B,N,C) instance logits, (B,) labels, (B,N,1) validity
        -> (loss, acc, num_sample).  Probabilities are mapped back to logits for the fused kernel
        (sigmoid is the only prob_cls_type on this path)."""
        B, N, C = bag_cls_prob.shape
        p = bag_cls_prob.float().clamp(1e-30, 1 - 1e-7)
        logits = torch.cat([torch.log(p) - torch.log1p(-p), bag_ins_outs.float()], dim=-1).contiguous()
        v = (valid.reshape(B, N) > 0).to(torch.uint8).contiguous()
        out = self.forward_logits(logits, C, v, labels.to(torch.int32).contiguous(), C)
        return out[1], out[2], out[4]

    def gfocal_loss(self, p, q, w=1.0):
        l1 = (p - q) ** 2
        l2 = q * (p + self.eps).log() + (1 - q) * (1 - p + self.eps).log()
        return -(l1 * l2 * w).sum(dim=-1)
