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


def multiclass_nms_batched(multi_bboxes, multi_scores, score_thr, nms_cfg, max_num=-1):
    """multiclass_nms for all images of a batch -- boxes (B, n, 4) or (B, n, 4C), scores (B, n, C+1) -> list of B
    (dets (k_b, 5), labels (k_b,)) -- as three batched launches and ONE host read (the per-image candidate and keep counts
    together), where the per-image loop of the reference (p2p_head.py:330-343 -> bbox_nms.py:7-94) synchronises the host
    twice per image.  Same candidates, same order, same keep decisions as ``multiclass_nms`` image by image."""
    assert nms_cfg.get('type', 'nms') == 'nms' and not nms_cfg.get('class_agnostic', False), nms_cfg
    iou_thr = nms_cfg.get('iou_threshold', nms_cfg.get('iou_thr'))
    B = multi_scores.shape[0]
    boxes, scores, labels, _, cnt = ops.nms_candidates_batched(multi_bboxes.float().contiguous(),
                                                               multi_scores.float().contiguous(), score_thr)
    keep, num = ops.nms_batched(boxes, scores, labels, cnt, iou_thr)
    cap = scores.shape[1]
    # gather in keep order on the whole slabs (entries past num_keep[b] are garbage indices: clamp, they are sliced away below)
    kc = keep.clamp_(0, max(cap - 1, 0))
    dets = torch.cat([torch.gather(boxes, 1, kc[..., None].expand(-1, -1, 4)), torch.gather(scores, 1, kc)[..., None]], dim=-1)
    labs = torch.gather(labels, 1, kc).long()
    counts = torch.stack([cnt, num]).cpu()                   # the one host read of the batch
    out = []
    for b in range(B):
        k = int(counts[1, b]) if int(counts[0, b]) > 0 else 0
        if max_num > 0:
            k = min(k, max_num)
        out.append((dets[b, :k], labs[b, :k]))
    return out
