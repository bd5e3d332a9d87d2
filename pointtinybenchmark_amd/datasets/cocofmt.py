"""CocoFmtDataset annotation side (T/mmdet/datasets/cocofmt.py:63-225 on top of CocoDataset, coco.py:40-118): COCO-format
json -> per-image ``ann_info`` with the fork's extra fields (``true_bboxes``, ``anns_id``) that become the pipeline keys
``gt_true_bboxes`` / ``gt_anns_id`` (pipelines/loading.py:246-278, formating.py:210).  Host-side, pure Python + numpy; the
pycocotools index (third party, not vendored) is replaced by three dict look-ups.

``corner_kwargs`` (640x640 tile generation with overlap, cocofmt.py:22-43) is served by ``datasets/tiles.py`` (restated from the
documented parameters; the generator itself lives in the un-vendored ``huicv`` package -- parity unpinned); tile entries carry
``corner = [l, u, r, b]`` and ``load_sample`` crops the decoded image like LoadImageFromFile does (loading.py:63-68).
Not built: ``noise_kwargs`` (pseudo-box synthesis around noisy points, also ``huicv``; it only rewrites annotation FILES before
training and the shipped CPR configs point ``ann_file`` at the generated files)."""
import json
import os

import numpy as np

from ..registry import Registry

DATASETS = Registry('dataset')


@DATASETS.register_module()
class CocoFmtDataset:
    CLASSES = None

    def __init__(self, ann_file, pipeline=None, classes=None, data_root=None, img_prefix='', test_mode=False,
                 filter_empty_gt=True, corner_kwargs=None, train_ignore_as_bg=True, noise_kwargs=None,
                 merge_after_infer_kwargs=None, min_gt_size=None, image_loader=None):
        assert noise_kwargs is None, 'pseudo-box synthesis lives in huicv (not vendored); point ann_file at the generated json'
        if data_root is not None and not os.path.isabs(ann_file):
            ann_file = os.path.join(data_root, ann_file)
        if corner_kwargs is not None:          # cocofmt.py:77-81: train on the tile file, generating it when missing
            from .tiles import corner_file_name, generate_corner_dataset
            tile_file = corner_file_name(ann_file, corner_kwargs['max_tile_size'], corner_kwargs['tile_overlap'])
            if not os.path.exists(tile_file):
                generate_corner_dataset(ann_file, save_path=tile_file, **corner_kwargs)
            ann_file = tile_file
        if data_root is not None and img_prefix and not os.path.isabs(img_prefix):
            img_prefix = os.path.join(data_root, img_prefix)
        self.ann_file, self.img_prefix, self.test_mode = ann_file, img_prefix, test_mode
        self.filter_empty_gt, self.train_ignore_as_bg, self.min_gt_size = filter_empty_gt, train_ignore_as_bg, min_gt_size
        self.merge_after_infer_kwargs = merge_after_infer_kwargs
        self.pipeline, self.image_loader = pipeline, image_loader
        if classes is not None:
            self.CLASSES = list(classes)
        self.data_infos = self.load_annotations(ann_file)
        if not test_mode:
            valid = self._filter_imgs()
            self.data_infos = [self.data_infos[i] for i in valid]

    # ------------------------------------------------------------------ cocofmt.py:104-133
    def load_annotations(self, ann_file):
        ds = json.load(open(ann_file)) if isinstance(ann_file, str) else ann_file
        self.dataset = ds
        self.anns = {a['id']: a for a in ds.get('annotations', [])}
        assert len(self.anns) == len(ds.get('annotations', [])), "Annotation ids in '%s' are not unique!" % (ann_file,)
        self.img_to_anns = {}
        for a in ds.get('annotations', []):
            self.img_to_anns.setdefault(a['image_id'], []).append(a)
        if self.CLASSES is None:
            self.CLASSES = [c['name'] for c in ds['categories']]
        # COCO.getCatIds(catNms=CLASSES) filters the json's category list: ids come in the JSON's order, "will not change
        # with the order of the CLASSES" (cocofmt.py:117-119; pycocotools is un-vendored, published algorithm)
        self.cat_ids = [c['id'] for c in ds['categories'] if c['name'] in self.CLASSES]
        self.cat2label = {cid: i for i, cid in enumerate(self.cat_ids)}
        self.img_ids = [im['id'] for im in ds['images']]
        infos = []
        for im in ds['images']:
            info = dict(im)
            info['filename'] = info['file_name']
            infos.append(info)
        return infos

    # ------------------------------------------------------------------ coco.py:96-118 + cocofmt.py:135-155
    def _filter_imgs(self, min_size=32):
        ids_with_ann = {a['image_id'] for a in self.anns.values()}
        ids_in_cat = {a['image_id'] for a in self.anns.values() if a.get('category_id') in self.cat_ids}
        ids_in_cat &= ids_with_ann
        valid_inds, valid_img_ids = [], []
        for i, info in enumerate(self.data_infos):
            img_id = self.img_ids[i]
            if self.filter_empty_gt and img_id not in ids_in_cat:
                continue
            if min(info['width'], info['height']) >= min_size:
                valid_inds.append(i)
                valid_img_ids.append(img_id)
        self.img_ids = valid_img_ids
        if self.min_gt_size:
            new_inds, new_ids = [], []
            for i, img_id in enumerate(self.img_ids):
                ok = False
                for ann in self.img_to_anns.get(img_id, []):
                    if 'ignore' in ann and ann['ignore']:
                        continue
                    if ann['bbox'][-1] > self.min_gt_size and ann['bbox'][-2] > self.min_gt_size:
                        ok = True
                if ok:
                    new_inds.append(valid_inds[i])
                    new_ids.append(img_id)
            self.img_ids, valid_inds = new_ids, new_inds
        return valid_inds

    def __len__(self):
        return len(self.data_infos)

    def get_ann_info(self, idx):
        info = self.data_infos[idx]
        return self._parse_ann_info(info, self.img_to_anns.get(info['id'], []))

    # ------------------------------------------------------------------ cocofmt.py:157-225
    def _parse_ann_info(self, img_info, ann_info):
        """One image's annotations -> the ``ann_info`` dict of the reference (same keys, dtypes and the same quirk: ``anns_id``
        stays a plain list unless some annotation carries a ``true_bbox``)."""
        W, H = img_info['width'], img_info['height']

        def usable(a):
            if self.train_ignore_as_bg and a.get('ignore', False):
                return False
            x, y, w, h = a['bbox']
            clipped_w = max(0, min(x + w, W) - max(x, 0))
            clipped_h = max(0, min(y + h, H) - max(y, 0))
            return clipped_w * clipped_h != 0 and a['area'] > 0 and w >= 1 and h >= 1 and a['category_id'] in self.cat2label

        def xyxy(box):
            x, y, w, h = box
            return [x, y, x + w, y + h]
        kept = [a for a in ann_info if usable(a)]
        crowd = [a for a in kept if a.get('iscrowd', False)]
        real = [a for a in kept if not a.get('iscrowd', False)]
        true_boxes = [xyxy(a['true_bbox']) for a in real if 'true_bbox' in a]
        ids = [a['id'] for a in real]

        def boxes(rows):
            return np.array(rows, dtype=np.float32) if rows else np.zeros((0, 4), dtype=np.float32)
        ann = dict(bboxes=boxes([xyxy(a['bbox']) for a in real]),
                   labels=np.array([self.cat2label[a['category_id']] for a in real], dtype=np.int64),
                   anns_id=np.array(ids, dtype=np.int64) if true_boxes else ids,
                   bboxes_ignore=boxes([xyxy(a['bbox']) for a in crowd]),
                   masks=[a.get('segmentation', None) for a in real],
                   seg_map=img_info['filename'].replace('jpg', 'png'))
        if true_boxes:
            ann['true_bboxes'] = np.array(true_boxes, dtype=np.float32)
        return ann

    # ------------------------------------------------------------------ sample assembly (custom.py:185-215 + LoadAnnotations)
    def load_sample(self, idx):
        """One un-augmented sample: decoded uint8 BGR image (H,W,3) + the LoadAnnotations fields
        (pipelines/loading.py:246-278).  ``image_loader(path) -> uint8 HxWx3 BGR`` is supplied by the caller (the
        reference decodes with cv2 through mmcv; neither is a dependency here)."""
        info = self.data_infos[idx]
        ann = self.get_ann_info(idx)
        path = os.path.join(self.img_prefix, info['filename']) if self.img_prefix else info['filename']
        assert self.image_loader is not None, 'pass image_loader= (e.g. datasets.pipeline.pil_bgr_loader)'
        img = self.image_loader(path)
        if 'corner' in info:                   # LoadImageFromFile, loading.py:63-68: a tile of the source image
            l, u, r, b = info['corner']
            img = np.ascontiguousarray(img[u:b, l:r])
            assert img.shape[0] * img.shape[1] > 0
        s = dict(img=img, filename=path, ori_filename=info['filename'], ori_shape=tuple(img.shape),
                 gt_bboxes=ann['bboxes'].copy(), gt_labels=ann['labels'].copy(),
                 gt_bboxes_ignore=ann['bboxes_ignore'].copy(),
                 gt_true_bboxes=(ann['true_bboxes'] if 'true_bboxes' in ann else ann['bboxes']).copy())
        s['gt_anns_id'] = np.asarray(ann['anns_id'], dtype=np.int64).copy()      # loading.py:274-275
        if 'corner' in info:
            s['corner'] = info['corner']
        return s

    def load_batch(self, indices, rng=None):
        samples = [self.load_sample(i) for i in indices]
        return self.pipeline(samples, rng) if self.pipeline is not None else samples
