"""multiclass_nms (T/mmdet/core/post_processing/bbox_nms.py:7-94) as two device stages: the candidate list -- every
(proposal, class) pair above ``score_thr`` -- comes out of one ordered-compaction kernel, the class-aware greedy NMS (what
mmcv.ops.nms.batched_nms does in the un-vendored mmcv-full) out of the 64x64 bitmask kernels; both in csrc/postproc.hip."""
import torch

from .. import _lib, ops

NMS_SLAB_MAX = 16384                 # candidates per image the BATCHED NMS launches take (whole pair mask per image); above: image by image
NMS_MASK_BYTES_MAX = 256 << 20       # worst-case pair-mask workspace above which the slabs are sized by the actual counts


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
    cap = scores.shape[1]
    if cap > NMS_SLAB_MAX or B * cap * ((cap + 63) // 64) * 8 > NMS_MASK_BYTES_MAX:
        # many classes (DOTA: 15 x nms_pre 2000 = 30 000 slots, COCO: 80 x 1000 = 80 000): the slab CAPACITY n * C is far above
        # what the NMS kernels take (16 384 candidates, the limit of the per-image path too) and its worst-case pair mask would
        # be hundreds of MB per image, while the candidates that pass score_thr are a few thousand.  One extra host read of the
        # candidate counts sizes the NMS slabs by the largest actual count instead.
        kmax = int(cnt.max().item())
        if kmax > NMS_SLAB_MAX:
            # one image with more candidates than a whole-mask slab holds (80 classes x nms_pre 1000 on a noisy early-training image):
            # image by image through the banded form of cpr_nms, which has no ceiling below 262 144 (round 6; the reference has none)
            counts = cnt.cpu()
            out = []
            for b in range(B):
                k = int(counts[b])
                bb, ss, ll = boxes[b, :k].contiguous(), scores[b, :k].contiguous(), labels[b, :k].contiguous()
                if k > 0:
                    keep = ops.nms(bb, ss, ll, iou_thr)
                    if max_num > 0:
                        keep = keep[:max_num]
                    bb, ss, ll = bb[keep], ss[keep], ll[keep]
                out.append((torch.cat([bb, ss[:, None]], dim=-1), ll.long()))
            return out
        kk = max(kmax, 1)
        boxes, scores, labels = boxes[:, :kk].contiguous(), scores[:, :kk].contiguous(), labels[:, :kk].contiguous()
        cap = kk
    keep, num = ops.nms_batched(boxes, scores, labels, cnt, iou_thr)
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
