"""TEST INFRASTRUCTURE ONLY.  Oracle for the data side feeding the path.

* Annotation parsing: the REFERENCE's own method bodies (`CocoFmtDataset._parse_ann_info`, `._filter_imgs`,
  `CocoDataset._filter_imgs`, `RandomFlip.bbox_flip`) are cut out of the source files under /root/reference and executed
  on a synthetic COCO-format dataset (the modules themselves import mmcv / huicv / terminaltables, none installed).
  `python -m oracle.data_oracle` writes tests/golden/data_side.json.
* Image tail (flip -> normalise -> pad -> channels-last): numpy restatement of mmcv.imflip / mmcv.imnormalize_ /
  mmcv.impad / DefaultFormatBundle.  mmcv and cv2 are third-party, un-vendored and not installed: PARITY UNPINNED for the
  float rounding of OpenCV's subtract/multiply (assumed: fp32 arithmetic with the scalars converted to float, which is
  what OpenCV documents for CV_32F sources)."""
import json
import os
import re
import textwrap

import numpy as np

T = '/root/reference/TOV_mmdetection/mmdet/'
GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'data_side.json')


def _method(path, cls_marker, name):
    src = open(path).read()
    start = src.index(cls_marker)
    m = re.search(r'\n    def %s\(.*?(?=\n    def |\n    @|\nclass |\Z)' % name, src[start:], flags=re.S)
    return textwrap.dedent(m.group(0))


def synthetic_dataset(seed=0):
    rng = np.random.RandomState(seed)
    images, anns, aid = [], [], 1
    for i in range(7):
        w, h = (640, 640) if i != 5 else (24, 640)        # image 5 is below min_size 32
        images.append(dict(id=100 + i, width=w, height=h, file_name='t%d.jpg' % i))
        n = 0 if i == 3 else int(rng.randint(1, 6))       # image 3 has no annotation
        for _ in range(n):
            cx, cy = rng.uniform(-10, 650, 2)              # some boxes fall outside the image
            bw = bh = 16.0 if i != 6 else 1.5               # image 6: only boxes under min_gt_size
            a = dict(id=aid, image_id=100 + i, category_id=int(rng.choice([1, 2, 7])), iscrowd=int(rng.rand() < 0.15),
                     ignore=int(rng.rand() < 0.2), bbox=[float(cx - bw / 2), float(cy - bh / 2), bw, bh],
                     area=float(bw * bh) if rng.rand() > 0.1 else 0.0, segmentation=[[1.0, 2.0]])
            if i % 2 == 0:
                a['true_bbox'] = [float(cx - 9), float(cy - 20), 18.0, 40.0]
            anns.append(a)
            aid += 1
    return dict(images=images, annotations=anns,
                categories=[dict(id=1, name='person'), dict(id=2, name='rider'), dict(id=7, name='other')])


class _Coco:
    def __init__(self, ds):
        self.dataset = ds
        self.anns = {a['id']: a for a in ds['annotations']}
        self.imgToAnns, self.cat_img_map = {}, {}
        for im in ds['images']:
            self.imgToAnns[im['id']] = []
        for a in ds['annotations']:
            self.imgToAnns[a['image_id']].append(a)
            self.cat_img_map.setdefault(a['category_id'], []).append(a['image_id'])
        for c in ds['categories']:
            self.cat_img_map.setdefault(c['id'], [])


def reference_parse(ds, classes, min_gt_size, train_ignore_as_bg=True):
    """Runs the reference method bodies; returns (valid image ids after filtering, per-image ann_info)."""
    ns = {'np': np, 'print': lambda *a, **k: None}
    base = 'class Base:\n' + textwrap.indent(_method(T + 'datasets/coco.py', 'class CocoDataset', '_filter_imgs'), '    ')
    sub = 'class Sub(Base):\n' + textwrap.indent(
        _method(T + 'datasets/cocofmt.py', 'class CocoFmtDataset', '_filter_imgs').replace(
            'super(CocoFmtDataset, self)', 'super(Sub, self)'), '    ') + '\n' + textwrap.indent(
        _method(T + 'datasets/cocofmt.py', 'class CocoFmtDataset', '_parse_ann_info'), '    ')
    exec(base + '\n' + sub, ns)
    d = ns['Sub']()
    d.coco = _Coco(ds)
    # self.coco.get_cat_ids(cat_names=CLASSES) = pycocotools getCatIds: the json's categories filtered by name, JSON order
    d.cat_ids = [c['id'] for c in ds['categories'] if c['name'] in classes]
    d.cat2label = {c: i for i, c in enumerate(d.cat_ids)}
    d.img_ids = [im['id'] for im in ds['images']]
    d.data_infos = [dict(im, filename=im['file_name']) for im in ds['images']]
    d.filter_empty_gt, d.min_gt_size, d.train_ignore_as_bg = True, min_gt_size, train_ignore_as_bg
    valid = d._filter_imgs()
    infos = [d.data_infos[i] for i in valid]
    parsed = [d._parse_ann_info(info, d.coco.imgToAnns[info['id']]) for info in infos]
    return d.img_ids, parsed


def reference_bbox_flip(boxes, img_shape):
    ns = {'np': np}
    exec('class F:\n' + textwrap.indent(_method(T + 'datasets/pipelines/transforms.py', 'class RandomFlip', 'bbox_flip'),
                                        '    '), ns)
    return ns['F']().bbox_flip(boxes, img_shape, 'horizontal')


def reference_resize_bboxes(boxes, img_shape, scale_factor=(1.0, 1.0, 1.0, 1.0)):
    """Resize._resize_bboxes (transforms.py:241-249) executed from the reference source, bbox_clip_border=True."""
    ns = {'np': np}
    exec('class R:\n    bbox_clip_border = True\n' + textwrap.indent(
        _method(T + 'datasets/pipelines/transforms.py', 'class Resize', '_resize_bboxes'), '    '), ns)
    res = dict(bbox_fields=['b'], b=boxes.copy(), scale_factor=np.array(scale_factor, dtype=np.float32), img_shape=img_shape)
    ns['R']()._resize_bboxes(res)
    return res['b']


def resize_clip_bboxes(boxes, img_shape):
    """Resize._resize_bboxes at scale 1 with bbox_clip_border=True, restated (pinned by tests/golden/data_side.json)."""
    b = boxes * np.array([1.0, 1.0, 1.0, 1.0], dtype=np.float32)
    b[:, 0::2] = np.clip(b[:, 0::2], 0, img_shape[1])
    b[:, 1::2] = np.clip(b[:, 1::2], 0, img_shape[0])
    return b


def bbox_flip(boxes, img_shape):
    """RandomFlip.bbox_flip horizontal, restated (pinned against the reference body by tests/golden/data_side.json);
    usable where /root/reference does not exist (the GPU box)."""
    f = boxes.copy()
    w = img_shape[1]
    f[..., 0::4] = w - boxes[..., 2::4]
    f[..., 2::4] = w - boxes[..., 0::4]
    return f


def image_tail(img_u8_bgr, flip, mean, std, to_rgb=True, size_divisor=32):
    """mmcv.imflip -> mmcv.imnormalize -> mmcv.impad(pad_val=0) -> HWC float32 (restated; see the header)."""
    img = img_u8_bgr[:, ::-1] if flip else img_u8_bgr
    img = img.astype(np.float32)
    mean = np.float64(np.asarray(mean, np.float32).reshape(1, -1))
    stdinv = 1 / np.float64(np.asarray(std, np.float32).reshape(1, -1))
    if to_rgb:
        img = img[:, :, ::-1]
    img = (img - mean.astype(np.float32)).astype(np.float32)
    img = (img * stdinv.astype(np.float32)).astype(np.float32)
    h, w = img.shape[:2]
    ph, pw = (h + size_divisor - 1) // size_divisor * size_divisor, (w + size_divisor - 1) // size_divisor * size_divisor
    out = np.zeros((ph, pw, 3), np.float32)
    out[:h, :w] = img
    return out


def reference_group_sampler(flags, samples_per_gpu, num_replicas, rank, seed, epoch):
    """The reference's DistributedGroupSampler class (samplers/group_sampler.py:51-148, pure torch / numpy) executed from
    its source on a dataset stub that only has ``flag``; mmcv's get_dist_info is not needed when rank / num_replicas are given."""
    import math
    import torch
    from torch.utils.data import Sampler
    src = open(T + 'datasets/samplers/group_sampler.py').read()
    cls = src[src.index('class DistributedGroupSampler'):]
    ns = dict(math=math, np=np, torch=torch, Sampler=Sampler, get_dist_info=lambda: (0, 1))
    exec(cls, ns)
    ds = type('D', (), {})()
    ds.flag = np.asarray(flags, dtype=np.uint8)
    smp = ns['DistributedGroupSampler'](ds, samples_per_gpu, num_replicas, rank, seed)
    smp.set_epoch(epoch)
    return list(iter(smp)), len(smp)


def reference_load_annotations(ann_info):
    """LoadAnnotations._load_bboxes / ._load_labels (pipelines/loading.py:246-278) executed from the reference source:
    which ann_info fields become which pipeline keys (gt_true_bboxes falls back to bboxes; gt_anns_id)."""
    ns = {'np': np}
    path = T + 'datasets/pipelines/loading.py'
    exec('class L:\n' + textwrap.indent(_method(path, 'class LoadAnnotations', '_load_bboxes'), '    ') + '\n' +
         textwrap.indent(_method(path, 'class LoadAnnotations', '_load_labels'), '    '), ns)
    res = dict(ann_info=ann_info, bbox_fields=[])
    ld = ns['L']()
    ld._load_bboxes(res)
    ld._load_labels(res)
    return {k: v for k, v in res.items() if k != 'ann_info'}


def _jsonable(o):
    if isinstance(o, dict):
        return {k: _jsonable(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_jsonable(v) for v in o]
    if isinstance(o, np.ndarray):
        return dict(dtype=str(o.dtype), shape=list(o.shape), data=o.tolist())
    if isinstance(o, np.generic):
        return o.item()
    return o


def main():
    out = {}
    for name, classes, mgs, seed in (('all3_min2', ['person', 'rider', 'other'], 2, 0), ('person_only', ['person'], 2, 1),
                                     ('no_min', ['person', 'rider'], None, 2),
                                     # CLASSES in another order than the json's categories: labels follow the JSON order
                                     ('permuted_classes', ['other', 'person'], 2, 3)):
        ds = synthetic_dataset(seed)
        ids, parsed = reference_parse(ds, classes, mgs)
        out[name] = dict(seed=seed, classes=classes, min_gt_size=mgs, img_ids=ids, parsed=_jsonable(parsed))
    rng = np.random.RandomState(0)
    b = rng.uniform(0, 600, (9, 4)).astype(np.float32)
    out['bbox_flip'] = dict(boxes=b.tolist(), width=633, flipped=reference_bbox_flip(b, (480, 633)).tolist())
    # pseudo boxes overhanging the image border (points closer than pseudo_wh/2 to an edge): Resize clips them before the flip
    c = rng.uniform(-12, 652, (40, 2)).astype(np.float32)
    ob = np.concatenate([c - 8, c + 8], axis=1).astype(np.float32)
    clipped = reference_resize_bboxes(ob, (480, 633, 3))
    out['resize_clip'] = dict(boxes=ob.tolist(), img_shape=[480, 633, 3], clipped=clipped.tolist(),
                              clipped_then_flipped=reference_bbox_flip(clipped, (480, 633)).tolist())
    # DistributedGroupSampler index streams: two aspect-ratio groups, 2 ranks, two epochs
    flags = [int(v) for v in (np.random.RandomState(5).rand(23) < 0.35)]
    out['group_sampler'] = dict(flags=flags, cases=[])
    for spg, world, seed, epoch in ((2, 2, 0, 0), (2, 2, 0, 3), (3, 2, 7, 1), (2, 1, 0, 0), (4, 3, 1, 2)):
        for rank in range(world):
            idx, n = reference_group_sampler(flags, spg, world, rank, seed, epoch)
            out['group_sampler']['cases'].append(dict(samples_per_gpu=spg, num_replicas=world, rank=rank, seed=seed, epoch=epoch,
                                                      indices=[int(i) for i in idx], length=n))
    # LoadAnnotations key mapping, with and without true_bboxes
    ds = synthetic_dataset(0)
    _, parsed = reference_parse(ds, ['person', 'rider', 'other'], 2)
    out['load_annotations'] = [_jsonable(reference_load_annotations(p)) for p in parsed[:2]]
    json.dump(out, open(GOLDEN, 'w'))
    print('wrote', GOLDEN, sorted(out))


if __name__ == '__main__':
    main()
