"""640x640 tile ("corner") datasets with 100 px overlap -- the annotation files every TinyPersonV2 config trains on
(T/configs2/TinyPersonV2/TinyPersonV2.md:7-36: ``--max_tile_w 640 --max_tile_h 640 --tile_overlap_w 100 --tile_overlap_h 100``;
entry point T/mmdet/datasets/cocofmt.py:22-43 ``corner_kwargs``).

PARITY UNPINNED: the generator itself (``huicv.corner_dataset.corner_dataset_util.generate_corner_dataset``) is not in the
reference tree and not installed; what IS in the tree and fixes the format is the consumer: an image entry carries
``corner = [l, u, r, b]`` and LoadImageFromFile crops ``img[u:b, l:r]`` (pipelines/loading.py:63-68), so tile annotations live
in tile coordinates.  Restated from that contract and the documented parameters:

  * tile origins along an axis of length L: 0, s, 2s, ... with stride s = tile - overlap, the last tile shifted back so that it
    ends flush with the image border (every tile has the full size when the image is at least one tile large; consecutive
    tiles overlap by >= ``overlap``); an image smaller than a tile is one tile;
  * an annotation goes to every tile that contains its box centre, translated by (-l, -u), not clipped (Resize clips boxes to
    the tile afterwards, transforms.py:241-249); ``true_bbox`` travels the same way; annotation ids are re-numbered,
    ``ori_id`` / ``ori_image_id`` keep the link to the source file;
  * tiles without annotations are kept (CocoDataset._filter_imgs drops them when filter_empty_gt is set).
"""
import copy
import json
import os
import warnings

MIN_LAST_TILE_SHIFT = 8     # px: a flush last tile closer than this to its predecessor replaces it


def tile_origins(length, tile, overlap):
    assert tile > overlap >= 0
    if length <= tile:
        return [0]
    stride = tile - overlap
    xs = list(range(0, length - tile, stride))
    last = length - tile
    # a flush last tile a few pixels past the previous origin (L = 1181: 0, 540, 541) would emit every object of that band
    # twice: the flush tile REPLACES that predecessor (0, 541: still every pixel covered; the overlap with the tile before
    # shrinks by less than MIN_LAST_TILE_SHIFT pixels).  A single predecessor at 0 cannot be replaced (L = 641: 0, 1).
    if len(xs) > 1 and last - xs[-1] < MIN_LAST_TILE_SHIFT:
        xs[-1] = last
    else:
        xs.append(last)
    return xs


def image_tiles(width, height, max_tile_size=(640, 640), tile_overlap=(100, 100)):
    """[l, u, r, b] of every tile of a width x height image, row-major."""
    tw, th = max_tile_size
    ow, oh = tile_overlap
    return [[l, u, min(l + tw, width), min(u + th, height)]
            for u in tile_origins(height, th, oh) for l in tile_origins(width, tw, ow)]


def generate_corner_dataset(ann, save_path=None, max_tile_size=(640, 640), tile_overlap=(100, 100), **unused):
    """COCO-format dict (or json path) -> tile-level COCO-format dict; written to ``save_path`` when given."""
    if save_path:
        warnings.warn('generating the tile annotation file %s with pointtinybenchmark_amd.datasets.tiles -- a restatement of '
                      "huicv's generator (not vendored, parity unpinned); AP parity with the reference needs the "
                      'huicv-generated file (the reference itself generates the file and exits)' % save_path, stacklevel=2)
    ds = json.load(open(ann)) if isinstance(ann, str) else ann
    by_img = {}
    for a in ds.get('annotations', []):
        by_img.setdefault(a['image_id'], []).append(a)
    images, anns = [], []
    next_img, next_ann = 1, 1
    for im in ds['images']:
        for (l, u, r, b) in image_tiles(im['width'], im['height'], max_tile_size, tile_overlap):
            tile = dict(im)
            tile.update(id=next_img, ori_id=im['id'], corner=[l, u, r, b], width=r - l, height=b - u)
            images.append(tile)
            for a in by_img.get(im['id'], []):
                x, y, w, h = a['bbox']
                cx, cy = x + w / 2.0, y + h / 2.0
                if not (l <= cx < r and u <= cy < b):
                    continue
                t = copy.deepcopy(a)
                t.update(id=next_ann, ori_id=a['id'], image_id=next_img, ori_image_id=im['id'], bbox=[x - l, y - u, w, h])
                if 'true_bbox' in a:
                    tx, ty, tw_, th_ = a['true_bbox']
                    t['true_bbox'] = [tx - l, ty - u, tw_, th_]
                if 'point' in a:
                    t['point'] = [a['point'][0] - l, a['point'][1] - u]
                anns.append(t)
                next_ann += 1
            next_img += 1
    out = {k: v for k, v in ds.items() if k not in ('images', 'annotations')}
    out.update(images=images, annotations=anns)
    if save_path:
        os.makedirs(os.path.dirname(os.path.abspath(save_path)), exist_ok=True)
        json.dump(out, open(save_path, 'w'))
    return out


def corner_file_name(ann_file, max_tile_size, tile_overlap):
    """The path generate_corner_json_file_if_not_exist derives (cocofmt.py:31-36)."""
    base = '%s_corner_w%dh%dow%doh%d.json' % (ann_file[:-5], max_tile_size[0], max_tile_size[1], tile_overlap[0],
                                             tile_overlap[1])
    d, name = os.path.split(base)
    return os.path.join(d, 'corner', name)
