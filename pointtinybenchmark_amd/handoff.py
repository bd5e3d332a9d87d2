"""Refine -> annotation hand-off: what happens to ``CPRHead.get_bboxes`` output after the hot path.

    simple_test output  list[(dets (G, 6+), labels (G,))]      dets = x1,y1,x2,y2,score,ann_id[,geo...]
      -> bbox2result      per-class list of arrays              T/mmdet/core/bbox/transforms.py:124-141
      -> det2json         COCO-style result dicts + ann_id/geo  T/mmdet/datasets/coco.py:213-235 (the fork's edits :229-234)
      -> result2ann       refined points written back into the training annotation file that P2PNet trains on
                          T/exp/tools/result2ann.py:1-92 (CLI flags kept), which leans on pycocotools' COCO.loadRes
                          (third party, not vendored in the reference; its bbox branch is restated in ``load_res_bbox``).
Host-side glue: plain Python over dicts / numpy, no device work, no third-party imports."""
import argparse
import copy
import json

import numpy as np


def bbox2result(bboxes, labels, num_classes):
    """(n, 5+) detections + (n,) labels -> list over classes of (k, 5+) float arrays (transforms.py:124-141)."""
    if bboxes.shape[0] == 0:
        return [np.zeros((0, 5), dtype=np.float32) for _ in range(num_classes)]
    if hasattr(bboxes, 'detach'):
        bboxes = bboxes.detach().cpu().numpy()
        labels = labels.detach().cpu().numpy()
    return [bboxes[labels == i, :] for i in range(num_classes)]


def xyxy2xywh(bbox):
    b = bbox.tolist()
    return [b[0], b[1], b[2] - b[0], b[3] - b[1]]


def det2json(results, img_ids, cat_ids):
    """CocoDataset._det2json with the fork's ann_id / geo columns (coco.py:213-235).  results[idx][label] = (k, 5+)."""
    out = []
    for idx, img_id in enumerate(img_ids):
        result = results[idx]
        for label in range(len(result)):
            bboxes = result[label]
            for i in range(bboxes.shape[0]):
                data = dict(image_id=img_id, bbox=xyxy2xywh(bboxes[i]), score=float(bboxes[i][4]),
                            category_id=cat_ids[label])
                if len(bboxes[i]) >= 6:
                    data['ann_id'] = int(bboxes[i][5])
                if len(bboxes[i]) >= 7:
                    # rounded in the array's own dtype (np.float32 from the detector), then widened: what the reference's
                    # round(np.float32, 1) + mmcv's json handler (.item()) write to the result file
                    geos = [round(e, 1).item() for e in bboxes[i][6:] if e >= 0]
                    assert len(geos) % 2 == 0
                    data['geo'] = geos
                out.append(data)
    return out


def refine_to_json(refine_out, img_ids, cat_ids):
    """``BasicLocator.simple_test`` output (one (dets, labels) per image) -> result dicts (single_stage.py:99-104 + det2json)."""
    num_classes = len(cat_ids)
    return det2json([bbox2result(d, l, num_classes) for d, l in refine_out], img_ids, cat_ids)


def load_res_bbox(dataset, anns):
    """pycocotools COCO.loadRes, bbox branch: completes each result IN PLACE (segmentation, area, id, iscrowd) and returns
    the result dataset.  Image ids must exist in ``dataset``."""
    assert isinstance(anns, list), 'results is not an array of objects'
    known = {im['id'] for im in dataset['images']}
    assert {a['image_id'] for a in anns} <= known, 'Results do not correspond to current coco set'
    res = dict(images=[im for im in dataset['images']])
    assert anns and 'bbox' in anns[0] and anns[0]['bbox'] != []
    res['categories'] = copy.deepcopy(dataset['categories'])
    for i, ann in enumerate(anns):
        bb = ann['bbox']
        x1, x2, y1, y2 = bb[0], bb[0] + bb[2], bb[1], bb[1] + bb[3]
        if 'segmentation' not in ann:
            ann['segmentation'] = [[x1, y1, x1, y2, x2, y2, x2, y1]]
        ann['area'] = bb[2] * bb[3]
        ann['id'] = i + 1
        ann['iscrowd'] = 0
    res['annotations'] = anns
    return res


def _centre(xywh):
    return xywh[0] + xywh[2] / 2, xywh[1] + xywh[3] / 2


def turn_bbox_wh(bbox, new_wh):
    """Re-size a box around its centre when both target sides are positive (result2ann.py:44-54); the centre must survive
    to the pixel."""
    if not (new_wh[0] > 0 and new_wh[1] > 0):
        return bbox
    cx, cy = _centre(bbox)
    out = [cx - new_wh[0] / 2, cy - new_wh[1] / 2, new_wh[0], new_wh[1]]
    ox, oy = _centre(out)
    assert round(ox) == round(cx) and round(oy) == round(cy), (bbox, out)
    return out


def result2ann(ori_dataset, det_results, wh=-1):
    """The body of result2ann.py: returns ``ori_dataset`` (modified in place, like the tool's ``coco.dataset``) with the
    refined boxes, segmentation, area and geo written into the annotations the results point at through ``ann_id``."""
    if isinstance(wh, (int, float)):
        wh = (wh, wh)
    by_id = {a['id']: a for a in ori_dataset['annotations']}
    res = load_res_bbox(ori_dataset, det_results)
    res_by_img, img_with_anns = {}, []
    for a in res['annotations']:
        res_by_img.setdefault(a['image_id'], []).append(a)
    for a in ori_dataset['annotations']:            # COCO.imgToAnns iteration order: first appearance of each image
        if a['image_id'] not in img_with_anns:
            img_with_anns.append(a['image_id'])
    for im_id in img_with_anns:
        for ann_res in res_by_img.get(im_id, []):
            ori = by_id[ann_res['ann_id']]
            for key in ('image_id', 'category_id', 'iscrowd'):
                assert ori[key] == ann_res[key], key
            ori['bbox'] = turn_bbox_wh(ann_res['bbox'], wh)
            for key in ('segmentation', 'area'):
                ori[key] = ann_res[key]
            if 'geo' in ann_res:
                ori['geo'] = ann_res['geo']
    # the tool's check(): every written annotation keeps its centre (to the pixel) and carries the result's shape fields
    for im_id in img_with_anns:
        for ann_res in res_by_img.get(im_id, []):
            ori = by_id[ann_res['ann_id']]
            oc, nc = _centre(ori['bbox']), _centre(ann_res['bbox'])
            assert round(oc[0]) == round(nc[0]) and round(oc[1]) == round(nc[1])
            for key in ('segmentation', 'area'):
                assert ori[key] == ann_res[key], key
    return ori_dataset


def main(argv=None):
    ap = argparse.ArgumentParser(description='refined results -> annotation file (flags of exp/tools/result2ann.py)')
    ap.add_argument('--ori_ann')
    ap.add_argument('--det_file')
    ap.add_argument('--save_ann')
    ap.add_argument('--wh', default=-1, type=int)
    args = ap.parse_args(argv)
    out = result2ann(json.load(open(args.ori_ann)), json.load(open(args.det_file)), args.wh)
    json.dump(out, open(args.save_ann, 'w'))


if __name__ == '__main__':
    main()
