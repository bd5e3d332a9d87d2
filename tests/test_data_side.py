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


def test_group_sampler_matches_reference_class():
    """DistributedGroupSampler: index streams of the reference class executed from its source (two aspect-ratio groups,
    several (samples_per_gpu, world, seed, epoch) settings, every rank)."""
    from pointtinybenchmark_amd.datasets import DistributedGroupSampler
    g = GOLD['group_sampler']
    ds = type('D', (), {})()
    ds.flag = np.array(g['flags'], dtype=np.uint8)
    for c in g['cases']:
        s = DistributedGroupSampler(ds, c['samples_per_gpu'], c['num_replicas'], c['rank'], c['seed'])
        s.set_epoch(c['epoch'])
        assert len(s) == c['length']
        assert [int(i) for i in s] == c['indices'], c
    # the ranks of one setting partition ONE shared stream: same multiset of chunks, no chunk on two ranks
    a = [c for c in g['cases'] if (c['samples_per_gpu'], c['num_replicas'], c['epoch']) == (2, 2, 0)]
    assert len(a) == 2 and len(set(map(tuple, np.array(a[0]['indices']).reshape(-1, 2).tolist())) &
                               set(map(tuple, np.array(a[1]['indices']).reshape(-1, 2).tolist()))) <= 1


def test_load_sample_fields_match_reference_load_annotations(tmp_path):
    """CocoFmtDataset.load_sample against LoadAnnotations._load_bboxes / _load_labels executed from the reference source
    (which ann_info field feeds which pipeline key), plus the corner crop of LoadImageFromFile (loading.py:63-68)."""
    from PIL import Image
    from pointtinybenchmark_amd.datasets import CocoFmtDataset
    from pointtinybenchmark_amd.datasets.pipeline import pil_bgr_loader
    ds = DO.synthetic_dataset(0)
    rng = np.random.RandomState(0)
    for im in ds['images']:
        Image.fromarray(rng.randint(0, 256, (im['height'], im['width'], 3)).astype(np.uint8)).save(
            str(tmp_path / im['file_name'].replace('.jpg', '.png')))
        im['file_name'] = im['file_name'].replace('.jpg', '.png')
    d = CocoFmtDataset(ds, classes=['person', 'rider', 'other'], img_prefix=str(tmp_path), min_gt_size=2,
                       image_loader=pil_bgr_loader)
    for i, ref in enumerate(GOLD['load_annotations']):
        s = d.load_sample(i)
        for key in ('gt_bboxes', 'gt_bboxes_ignore', 'gt_true_bboxes', 'gt_labels', 'gt_anns_id'):
            if isinstance(ref[key], list):      # anns_id stays a plain list without true_bbox; DefaultFormatBundle's to_tensor
                want = np.array(ref[key], dtype=np.int64)       # (formating.py:210) turns it into the same int64 tensor
            else:
                want = np.array(ref[key]['data'], dtype=ref[key]['dtype']).reshape(ref[key]['shape'])
            assert np.array_equal(np.asarray(s[key]), want) and np.asarray(s[key]).dtype == want.dtype, key
        assert s['img'].dtype == np.uint8 and s['img'].shape == (640, 640, 3)
    # a tile entry: the decoded image is cropped to its corner, annotations are already in tile coordinates
    full = pil_bgr_loader(os.path.join(str(tmp_path), d.data_infos[0]['filename']))
    d.data_infos[0] = dict(d.data_infos[0], corner=[100, 40, 420, 300])
    s = d.load_sample(0)
    assert np.array_equal(s['img'], full[40:300, 100:420]) and s['corner'] == [100, 40, 420, 300]


def test_tile_generation_properties(tmp_path):
    """640x640 tiles with 100 px overlap (TinyPersonV2.md:7-36; generator un-vendored -> restated, parity unpinned):
    full coverage, full-size tiles, overlap >= 100 between neighbours, every annotation re-appears in tile coordinates in
    every tile that holds its centre, and CocoFmtDataset(corner_kwargs=...) trains on the generated file."""
    from pointtinybenchmark_amd.datasets import CocoFmtDataset, generate_corner_dataset, image_tiles
    from pointtinybenchmark_amd.datasets.tiles import corner_file_name, tile_origins
    for L in (300, 640, 641, 1180, 1181, 1920, 2000, 5000):
        xs = tile_origins(L, 640, 100)
        assert xs[0] == 0 and xs[-1] + min(640, L) == L and xs == sorted(set(xs))
        # neighbours overlap by >= 100 px, less at most MIN_LAST_TILE_SHIFT - 1 px where a flush last tile replaced a
        # predecessor a few pixels away (L = 1181: [0, 541] instead of [0, 540, 541], which duplicated a whole band)
        from pointtinybenchmark_amd.datasets.tiles import MIN_LAST_TILE_SHIFT
        assert all(b - a <= 540 for a, b in zip(xs[:-1], xs[1:-1])), (L, xs)
        assert len(xs) < 2 or xs[-1] - xs[-2] <= 540 + MIN_LAST_TILE_SHIFT - 1, (L, xs)
        assert len(xs) < 3 or xs[-1] - xs[-2] >= MIN_LAST_TILE_SHIFT, (L, xs)   # no near-duplicate last tile
    tiles = image_tiles(1920, 1080)
    cover = np.zeros((1080, 1920), dtype=np.int32)
    for l, u, r, b in tiles:
        assert (r - l, b - u) == (640, 640)
        cover[u:b, l:r] += 1
    assert cover.min() >= 1
    rng = np.random.RandomState(1)
    anns = []
    for i in range(200):
        cx, cy = rng.uniform(0, 1920), rng.uniform(0, 1080)
        anns.append(dict(id=i + 1, image_id=7, category_id=1, bbox=[cx - 8, cy - 8, 16.0, 16.0], area=256.0, iscrowd=0,
                         true_bbox=[cx - 5, cy - 11, 10.0, 22.0]))
    src = dict(images=[dict(id=7, width=1920, height=1080, file_name='a.jpg')], annotations=anns,
               categories=[dict(id=1, name='person')])
    path = tmp_path / 'rgb_train.json'
    import json
    json.dump(src, open(path, 'w'))
    out = generate_corner_dataset(str(path))
    assert len(out['images']) == 8 and len({a['id'] for a in out['annotations']}) == len(out['annotations'])
    corner = {im['id']: im['corner'] for im in out['images']}
    seen = {}
    for a in out['annotations']:
        l, u, r, b = corner[a['image_id']]
        o = anns[a['ori_id'] - 1]
        assert a['bbox'][0] + l == o['bbox'][0] and a['bbox'][1] + u == o['bbox'][1] and a['bbox'][2:] == o['bbox'][2:]
        assert a['true_bbox'][0] + l == o['true_bbox'][0] and a['true_bbox'][1] + u == o['true_bbox'][1]
        cx, cy = a['bbox'][0] + 8, a['bbox'][1] + 8
        assert 0 <= cx < r - l and 0 <= cy < b - u                        # centre inside its tile
        seen[a['ori_id']] = seen.get(a['ori_id'], 0) + 1
    assert set(seen) == {a['id'] for a in anns}                           # nothing lost
    want = {a['id']: sum(1 for (l, u, r, b) in tiles if l <= a['bbox'][0] + 8 < r and u <= a['bbox'][1] + 8 < b) for a in anns}
    assert seen == want and max(seen.values()) > 1                        # points in an overlap strip live in several tiles
    d = CocoFmtDataset(str(path), classes=['person'], corner_kwargs=dict(max_tile_size=(640, 640), tile_overlap=(100, 100)))
    assert os.path.exists(corner_file_name(str(path), (640, 640), (100, 100))) and len(d) == 8
    assert all('corner' in info for info in d.data_infos)


@pytest.mark.gpu
def test_batch_loader_feeds_forward_train(tmp_path):
    """Source image + point annotations -> 640x640 tiles (100 px overlap) -> CocoFmtDataset -> DistributedGroupSampler ->
    GpuImagePipeline -> BasicLocator.forward_train: the whole data side in front of the hot path, two 'ranks'."""
    import json
    from PIL import Image
    from oracle.gen_golden import CPR_CASES
    from pointtinybenchmark_amd.datasets import BatchLoader, CocoFmtDataset, GpuImagePipeline
    from pointtinybenchmark_amd.datasets.pipeline import pil_bgr_loader
    from tests.test_gpu_cpr_parity import build_hip_locator
    rng = np.random.RandomState(2)
    W, H = 1300, 700
    Image.fromarray(rng.randint(0, 256, (H, W, 3)).astype(np.uint8)).save(str(tmp_path / 'a.png'))
    anns = []
    for i in range(60):
        cx, cy = rng.uniform(8, W - 8), rng.uniform(8, H - 8)
        anns.append(dict(id=i + 1, image_id=1, category_id=1 + i % 3, bbox=[cx - 8, cy - 8, 16.0, 16.0], area=256.0, iscrowd=0))
    src = dict(images=[dict(id=1, width=W, height=H, file_name='a.png')], annotations=anns,
               categories=[dict(id=1, name='person'), dict(id=2, name='rider'), dict(id=3, name='other')])
    json.dump(src, open(tmp_path / 'ann.json', 'w'))
    d = CocoFmtDataset(str(tmp_path / 'ann.json'), classes=['person', 'rider', 'other'], img_prefix=str(tmp_path),
                       corner_kwargs=dict(max_tile_size=(640, 640), tile_overlap=(100, 100)), image_loader=pil_bgr_loader)
    assert len(d) == 6                                         # 3 x 2 tiles
    m, _ = build_hip_locator(CPR_CASES['cpr_r18_c3_128'])
    pipe = GpuImagePipeline(flip_ratio=0.5)
    seen = []
    for rank in range(2):
        loader = BatchLoader(d, pipe, samples_per_gpu=2, num_replicas=2, rank=rank, seed=0)
        assert len(loader) == 2
        for batch in loader:
            assert tuple(batch['img'].shape) == (2, 4, 640, 640) and len(batch['gt_bboxes']) == 2
            assert all(mt['img_shape'] == (640, 640, 3) for mt in batch['img_metas'])
            with torch.no_grad():
                losses = m.forward_train(batch['img'], batch['img_metas'], batch['gt_bboxes'], batch['gt_labels'])
            assert all(bool(torch.isfinite(v)) for v in losses.values())
            seen += [mt['ori_filename'] for mt in batch['img_metas']]
    assert len(seen) == 8                                      # 6 tiles padded to 2 ranks x 2 batches x 2 images
