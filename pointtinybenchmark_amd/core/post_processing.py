"""multiclass_nms (T/mmdet/core/post_processing/bbox_nms.py:7-94) as two device stages: the candidate list -- every
(proposal, class) pair above ``score_thr`` -- comes out of one ordered-compaction kernel, the class-aware greedy NMS (what
mmcv.ops.nms.batched_nms does in the un-vendored mmcv-full) out of the 64x64 bitmask kernels; both in csrc/postproc.hip."""
import torch

from .. import ops


def multiclass_nms(multi_bboxes, multi_scores, score_thr, nms_cfg, max_num=-1, score_factors=None, return_inds=False):
    """Same signature and outputs as the reference: (dets (k,5), labels (k,)[, inds (k,)]) in descending score order."""
    assert nms_cfg.get('type', 'nms') == 'nms' and not nms_cfg.get('class_agnostic', False), nms_cfg
    iou_thr = nms_cfg.get('iou_threshold', nms_cfg.get('iou_thr'))
    factors = None if score_factors is None else score_factors.float().contiguous()
    boxes, scores, labels, inds = ops.nms_candidates(multi_bboxes.float().contiguous(), multi_scores.float().contiguous(),
                                                     score_thr, factors)
    if len(scores) > 0:
        keep = ops.nms(boxes, scores, labels, iou_thr)
        if max_num > 0:
            keep = keep[:max_num]
        boxes, scores, labels, inds = boxes[keep], scores[keep], labels[keep], inds[keep]
    dets = torch.cat([boxes, scores[:, None]], dim=-1)
    labels = labels.long()
    return (dets, labels, inds) if return_inds else (dets, labels)
