"""TEST INFRASTRUCTURE ONLY.  Golden vectors for the refine -> annotation hand-off.

Runs the REFERENCE's own tool (`/root/reference/TOV_mmdetection/exp/tools/result2ann.py`, executed where it lies with
runpy) and the reference's own `CocoDataset._det2json` body.  The tool imports pycocotools, which is neither vendored in
the reference nor installed here: a minimal `pycocotools.coco.COCO` (createIndex / loadRes bbox branch / loadAnns /
imgToAnns, restated from pycocotools 2.0's published source) is injected into sys.modules first.  The tool's own logic is
therefore pinned by executing it; the loadRes restatement is "parity unpinned" (no copy of pycocotools to check against).

    python -m oracle.handoff_oracle        # regenerates tests/golden/handoff.json"""
import copy
import json
import os
import runpy
import sys
import tempfile
import types
from collections import defaultdict

import numpy as np

REF_TOOL = '/root/reference/TOV_mmdetection/exp/tools/result2ann.py'
GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'handoff.json')


class COCO:
    """pycocotools.coco.COCO, the subset result2ann.py touches."""

    def __init__(self, annotation_file=None):
        self.dataset, self.anns, self.imgs, self.cats = {}, {}, {}, {}
        self.imgToAnns, self.catToImgs = defaultdict(list), defaultdict(list)
        if annotation_file is not None:
            self.dataset = json.load(open(annotation_file))
            self.createIndex()

    def createIndex(self):
        anns, cats, imgs = {}, {}, {}
        imgToAnns, catToImgs = defaultdict(list), defaultdict(list)
        for ann in self.dataset.get('annotations', []):
            imgToAnns[ann['image_id']].append(ann)
            anns[ann['id']] = ann
        for img in self.dataset.get('images', []):
            imgs[img['id']] = img
        for cat in self.dataset.get('categories', []):
            cats[cat['id']] = cat
        for ann in self.dataset.get('annotations', []):
            if 'category_id' in ann:
                catToImgs[ann['category_id']].append(ann['image_id'])
        self.anns, self.imgToAnns, self.catToImgs, self.imgs, self.cats = anns, imgToAnns, catToImgs, imgs, cats

    def getImgIds(self):
        return list(self.imgs.keys())

    def loadAnns(self, ids):
        return [self.anns[i] for i in ids] if isinstance(ids, (list, tuple)) else [self.anns[ids]]

    def loadRes(self, resFile):
        res = COCO()
        res.dataset['images'] = [img for img in self.dataset['images']]
        anns = json.load(open(resFile)) if isinstance(resFile, str) else resFile
        assert type(anns) == list, 'results in not an array of objects'
        annsImgIds = [ann['image_id'] for ann in anns]
        assert set(annsImgIds) == (set(annsImgIds) & set(self.getImgIds())), \
            'Results do not correspond to current coco set'
        assert 'bbox' in anns[0] and not anns[0]['bbox'] == []
        res.dataset['categories'] = copy.deepcopy(self.dataset['categories'])
        for id, ann in enumerate(anns):
            bb = ann['bbox']
            x1, x2, y1, y2 = [bb[0], bb[0] + bb[2], bb[1], bb[1] + bb[3]]
            if 'segmentation' not in ann:
                ann['segmentation'] = [[x1, y1, x1, y2, x2, y2, x2, y1]]
            ann['area'] = bb[2] * bb[3]
            ann['id'] = id + 1
            ann['iscrowd'] = 0
        res.dataset['annotations'] = anns
        res.createIndex()
        return res


def synthetic_case(seed=0, n_img=3, wh=16, with_geo=True):
    """A small coarse-point annotation file (16x16 pseudo boxes, like the TinyPerson `pseuw16h16` files) and refine
    results for most of its annotations (one image has none, one annotation is left un-refined)."""
    rng = np.random.RandomState(seed)
    images = [dict(id=10 + i, width=640, height=640, file_name='tile_%d.jpg' % i) for i in range(n_img + 1)]
    anns, aid = [], 100
    for i in range(n_img):
        for _ in range(3 + i):
            cx, cy = rng.uniform(20, 620, 2)
            anns.append(dict(id=aid, image_id=10 + i, category_id=1 + (aid % 2), iscrowd=0, ignore=0,
                             bbox=[float(cx - wh / 2), float(cy - wh / 2), float(wh), float(wh)], area=float(wh * wh),
                             segmentation=[[0.0] * 8], true_bbox=[float(cx - 9), float(cy - 20), 18.0, 40.0]))
            aid += 1
    dataset = dict(images=images, annotations=anns, categories=[dict(id=1, name='person'), dict(id=2, name='rider')],
                   info=dict(description='synthetic'), licenses=[])
    # refine output per image: x1,y1,x2,y2,score,ann_id[,geo pairs padded with -1]
    results, img_ids = [], []
    for i in range(n_img):
        per_cls = [[], []]
        for a in [a for a in anns if a['image_id'] == 10 + i][: None if i else -1]:
            c = np.array(a['bbox'][:2]) + wh / 2 + rng.uniform(-6, 6, 2)
            row = [c[0] - 8, c[1] - 8, c[0] + 8, c[1] + 8, rng.uniform(0.1, 0.9), a['id']]
            if with_geo:
                k = int(rng.randint(1, 4))
                geo = np.concatenate([c, rng.uniform(0, 640, 2 * (k - 1))])
                row += list(np.round(geo, 3)) + [-1.0] * (2 * (3 - k))
            per_cls[a['category_id'] - 1].append(row)
        ncol = 6 + (6 if with_geo else 0)
        results.append([np.array(r, dtype=np.float32).reshape(-1, ncol) for r in per_cls])
        img_ids.append(10 + i)
    return dataset, results, img_ids


def reference_det2json(results, img_ids, cat_ids):
    """CocoDataset._det2json executed from the reference source (method body only; `self` supplies img_ids / cat_ids)."""
    src = open('/root/reference/TOV_mmdetection/mmdet/datasets/coco.py').read()
    start = src.index('    def xyxy2xywh(self, bbox):')
    end = src.index('    def _segm2json(self, results):')
    ns = {}
    exec('class _D:\n' + src[start:end].replace('    def _proposal2json', '    def _proposal2json'), ns)
    d = ns['_D']()
    d.img_ids, d.cat_ids = img_ids, cat_ids
    d.__class__.__len__ = lambda self: len(self.img_ids)
    return d._det2json(results)


def run_reference_tool(dataset, det_json, wh):
    mod = types.ModuleType('pycocotools')
    sub = types.ModuleType('pycocotools.coco')
    sub.COCO = COCO
    mod.coco = sub
    sys.modules['pycocotools'], sys.modules['pycocotools.coco'] = mod, sub
    with tempfile.TemporaryDirectory() as td:
        a, b, c = (os.path.join(td, n) for n in ('ori.json', 'det.json', 'out.json'))
        json.dump(dataset, open(a, 'w'))
        # mmcv.dump's json handler turns numpy scalars into Python numbers with .item() (mmcv/fileio/handlers/json_handler.py)
        json.dump(det_json, open(b, 'w'), default=lambda o: o.item())
        argv = sys.argv
        sys.argv = ['result2ann.py', '--ori_ann', a, '--det_file', b, '--save_ann', c, '--wh', str(wh)]
        try:
            runpy.run_path(REF_TOOL, run_name='__main__')
        finally:
            sys.argv = argv
        return json.load(open(c))


def main():
    cases = {}
    for name, kw, wh in (('geo_wh-1', dict(seed=0, with_geo=True), -1), ('plain_wh16', dict(seed=1, with_geo=False), 16),
                         ('geo_wh32', dict(seed=2, with_geo=True, n_img=2), 32)):
        dataset, results, img_ids = synthetic_case(**kw)
        det_json = reference_det2json(results, img_ids, [1, 2])
        out = run_reference_tool(copy.deepcopy(dataset), copy.deepcopy(det_json), wh)
        det_json = json.loads(json.dumps(det_json, default=lambda o: o.item()))
        cases[name] = dict(kw=kw, wh=wh, det_json=det_json, out=out)
    json.dump(cases, open(GOLDEN, 'w'))
    print('wrote', GOLDEN, {k: len(v['out']['annotations']) for k, v in cases.items()})


if __name__ == '__main__':
    main()
