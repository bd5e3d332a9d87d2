"""Data side feeding the path.
CPU: annotation parsing / filtering against golden vectors produced by executing the reference's own method bodies
(oracle/data_oracle.py).  GPU: the fused flip+normalise+pad+layout kernel and the box flip, bit-exact against the numpy
restatement of the mmcv image tail; the pipeline's output drives forward_train to the same losses as the float NCHW image."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import data_oracle as DO

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'data_side.json')))


def _arr(o):
    return np.array(o['data'], dtype=o['dtype']).reshape(o['shape']) if isinstance(o, dict) and 'dtype' in o else o


@pytest.mark.parametrize('name', ['all3_min2', 'person_only', 'no_min', 'permuted_classes'])
def test_cocofmt_parse_matches_reference_methods(name):
    from pointtinybenchmark_amd.datasets import CocoFmtDataset
    g = GOLD[name]
    ds = CocoFmtDataset(DO.synthetic_dataset(g['seed']), classes=g['classes'], min_gt_size=g['min_gt_size'])
    assert ds.img_ids == g['img_ids'] and len(ds) == len(g['parsed'])
    for i, ref in enumerate(g['parsed']):
        got = ds.get_ann_info(i)
        assert set(got) == set(ref), (set(got), set(ref))
        for k in ref:
            r = _arr(ref[k])
            if isinstance(r, np.ndarray):
                assert isinstance(got[k], np.ndarray) and got[k].dtype == r.dtype and got[k].shape == r.shape, (k, got[k], r)
                assert np.array_equal(got[k], r), k
            else:
                assert got[k] == r, (k, got[k], r)


def test_box_flip_oracle_is_the_reference_formula():
    g = GOLD['bbox_flip']
    b = np.array(g['boxes'], np.float32)
    assert np.array_equal(DO.bbox_flip(b, (480, g['width'])), np.array(g['flipped'], np.float32))


def test_resize_clip_oracle_is_the_reference_formula():
    """Resize._resize_bboxes at scale 1 (bbox_clip_border=True) clips boxes that overhang the image BEFORE the flip
    (fixture: the reference's own method body on border-overhanging 16x16 pseudo boxes)."""
    g = GOLD['resize_clip']
    b = np.array(g['boxes'], np.float32)
    shp = tuple(g['img_shape'])
    clipped = DO.resize_clip_bboxes(b, shp)
    assert np.array_equal(clipped, np.array(g['clipped'], np.float32))
    assert (clipped != b).any(), 'the fixture must contain overhanging boxes'
    assert np.array_equal(DO.bbox_flip(clipped, shp[:2]), np.array(g['clipped_then_flipped'], np.float32))


def _samples(n, shapes, seed=0):
    rng = np.random.RandomState(seed)
    out = []
    for i in range(n):
        h, w = shapes[i % len(shapes)]
        k = 3 + i
        xy = rng.uniform(8, min(h, w) - 8, (k, 2)).astype(np.float32)
        boxes = np.concatenate([xy - 8, xy + 8], 1)
        out.append(dict(img=rng.randint(0, 256, (h, w, 3)).astype(np.uint8), filename='f%d.jpg' % i,
                        gt_bboxes=boxes, gt_labels=np.zeros(k, np.int64), gt_bboxes_ignore=np.zeros((0, 4), np.float32),
                        gt_true_bboxes=boxes + 1, gt_anns_id=np.arange(k, dtype=np.int64) + 10 * i))
    return out


class _Rng:
    def __init__(self, vals):
        self.vals = list(vals)

    def rand(self):
        return self.vals.pop(0)


@pytest.mark.gpu
@pytest.mark.parametrize('shapes', [[(96, 128)], [(100, 90), (64, 127), (97, 33)]], ids=['uniform', 'ragged'])
def test_gpu_pipeline_bit_exact(shapes):
    from pointtinybenchmark_amd.datasets import GpuImagePipeline
    mean, std = (123.675, 116.28, 103.53), (58.395, 57.12, 57.375)
    n = 4
    samples = _samples(n, shapes)
    pipe = GpuImagePipeline(mean, std, True, 32, flip_ratio=0.5,
                            keys=('img', 'gt_bboxes', 'gt_labels', 'gt_bboxes_ignore', 'gt_true_bboxes', 'gt_anns_id'))
    draws = [0.1, 0.9, 0.3, 0.7]                       # < 0.5 flips
    batch = pipe(samples, _Rng(draws))
    img = batch['img']
    assert img.shape[1] == 4 and img.stride(1) == 1
    nhwc = img.permute(0, 2, 3, 1).cpu().numpy()
    Hp, Wp = nhwc.shape[1:3]
    assert Hp % 32 == 0 and Wp % 32 == 0
    for i, s in enumerate(samples):
        flip = draws[i] < 0.5
        ref = DO.image_tail(s['img'], flip, mean, std, True, 32)
        h, w = ref.shape[:2]
        assert np.array_equal(nhwc[i, :h, :w, :3], ref), 'image %d differs' % i
        assert not nhwc[i, :, :, 3].any() and not nhwc[i, h:].any() and not nhwc[i, :, w:].any()
        m = batch['img_metas'][i]
        assert m['flip'] == flip and m['img_shape'] == s['img'].shape and m['pad_shape'] == (h, w, 3)
        for key in ('gt_bboxes', 'gt_true_bboxes'):
            want = DO.resize_clip_bboxes(s[key], s['img'].shape)        # Resize (scale 1) clips to the image first
            want = DO.bbox_flip(want, s['img'].shape[:2]) if flip else want
            assert np.array_equal(batch[key][i].cpu().numpy(), want), key
        assert torch.equal(batch['gt_anns_id'][i].cpu(), torch.from_numpy(s['gt_anns_id']))
        assert batch['gt_bboxes_ignore'][i].shape == (0, 4)


@pytest.mark.gpu
def test_gpu_pipeline_clips_overhanging_boxes_like_resize():
    """The reference fixture (border-overhanging pseudo boxes through Resize._resize_bboxes, then bbox_flip) through the
    device pipeline, flipped and not flipped."""
    from pointtinybenchmark_amd.datasets import GpuImagePipeline
    g = GOLD['resize_clip']
    h, w, _ = g['img_shape']
    boxes = np.array(g['boxes'], np.float32)
    sample = dict(img=np.zeros((h, w, 3), np.uint8), gt_bboxes=boxes, gt_labels=np.zeros(len(boxes), np.int64),
                  gt_bboxes_ignore=np.zeros((0, 4), np.float32), gt_true_bboxes=boxes.copy())
    pipe = GpuImagePipeline(flip_ratio=0.5)
    for draw, key in ((0.9, 'clipped'), (0.1, 'clipped_then_flipped')):
        batch = pipe([sample], _Rng([draw]))
        assert np.array_equal(batch['gt_bboxes'][0].cpu().numpy(), np.array(g[key], np.float32)), key
        assert np.array_equal(batch['gt_true_bboxes'][0].cpu().numpy(), np.array(g[key], np.float32)), key


@pytest.mark.gpu
def test_gpu_pipeline_output_feeds_forward_train():
    """The 4-channel channels-last batch is consumed by ResNet.forward as is: same losses as the float NCHW image."""
    from oracle.gen_golden import CPR_CASES
    from pointtinybenchmark_amd.datasets import GpuImagePipeline
    from tests.test_gpu_cpr_parity import build_hip_locator
    cfg = CPR_CASES['cpr_r18_c3_128']
    m, _ = build_hip_locator(cfg)
    samples = _samples(2, [(128, 128)], seed=3)
    for s in samples:
        s['gt_labels'] = np.arange(len(s['gt_labels']), dtype=np.int64) % 3
    pipe = GpuImagePipeline(flip_ratio=0.0)
    batch = pipe(samples)
    with torch.no_grad():
        a = m.forward_train(batch['img'], batch['img_metas'], batch['gt_bboxes'], batch['gt_labels'])
        nchw = batch['img'][:, :3].contiguous()                     # the reference's input format
        b = m.forward_train(nchw, batch['img_metas'], batch['gt_bboxes'], batch['gt_labels'])
    assert {k: float(v) for k, v in a.items()} == {k: float(v) for k, v in b.items()}
