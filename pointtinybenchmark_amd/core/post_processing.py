"""multiclass_nms (T/mmdet/core/post_processing/bbox_nms.py:7-94) on the HIP bitmask NMS kernel
(replaces mmcv.ops.nms.batched_nms, a C++/CUDA extension of the un-vendored mmcv-full).
The candidate expansion / score filter is index plumbing on a few thousand rows; sort + IoU + suppression
run in csrc/postproc.hip."""
import torch

from .. import ops


def multiclass_nms(multi_bboxes, multi_scores, score_thr, nms_cfg, max_num=-1, score_factors=None, return_inds=False):
    num_classes = multi_scores.size(1) - 1
    if multi_bboxes.shape[1] > 4:
        bboxes = multi_bboxes.view(multi_scores.size(0), -1, 4)
    else:
        bboxes = multi_bboxes[:, None].expand(multi_scores.size(0), num_classes, 4)
    scores = multi_scores[:, :-1]
    labels = torch.arange(num_classes, dtype=torch.long, device=scores.device).view(1, -1).expand_as(scores)
    bboxes, scores, labels = bboxes.reshape(-1, 4), scores.reshape(-1), labels.reshape(-1)
    valid_mask = scores > score_thr
    if score_factors is not None:
        score_factors = score_factors.view(-1, 1).expand(multi_scores.size(0), num_classes).reshape(-1)
        scores = scores * score_factors
    inds = valid_mask.nonzero(as_tuple=False).squeeze(1)
    bboxes, scores, labels = bboxes[inds], scores[inds], labels[inds]
    if inds.numel() == 0:
        dets = torch.cat([bboxes, scores[:, None]], -1)
        return (dets, labels, inds) if return_inds else (dets, labels)
    thr = nms_cfg.get('iou_threshold', nms_cfg.get('iou_thr'))
    assert nms_cfg.get('type', 'nms') == 'nms' and not nms_cfg.get('class_agnostic', False)
    keep = ops.nms(bboxes.float().contiguous(), scores.float().contiguous(), labels.to(torch.int32).contiguous(), thr)
    if max_num > 0:
        keep = keep[:max_num]
    dets = torch.cat([bboxes[keep], scores[keep, None]], -1)
    if return_inds:
        return dets, labels[keep], inds[keep]
    return dets, labels[keep]
