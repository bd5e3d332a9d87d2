"""-m gpu: BASELINE.json's full size (ResNet-50 + FPN, 640x640, 32 gts/image).  Direct oracle parity where the CPU
oracle finishes in seconds, plus size-independent properties: run-to-run determinism, batch-order equivariance
(bit-exact), conv linearity."""
import os

import numpy as np
import pytest
import torch

from oracle import cpr_oracle as O
from pointtinybenchmark_amd import synthetic
from tests.test_gpu_cpr_parity import build_hip_locator, to_cuda

pytestmark = pytest.mark.gpu
CFG = dict(depth=50, num_classes=1, start_level=0, stride=4, radius=5, head_std=0.3, seed=0, batch=2, height=640,
           width=640, num_gts=32)


@pytest.fixture(scope='module')
def full():
    m, sd = build_hip_locator(CFG)
    batch = synthetic.synthetic_batch(2, 640, 640, 32, 1, 0)
    return m, sd, batch, to_cuda(batch)


def _run(m, cb):
    with torch.no_grad():
        feats = m.neck(m.backbone(cb['img']))
        cls_feat, ins_feat = m.bbox_head(feats)
        losses = m.bbox_head.loss(cls_feat, ins_feat, cb['gt_bboxes'], cb['gt_labels'], cb['img_metas'])
        dets = m.bbox_head.get_bboxes(cls_feat, ins_feat, cb['img_metas'], gt_bboxes=cb['gt_bboxes'],
                                      gt_labels=cb['gt_labels'], gt_anns_id=cb['gt_anns_id'])
        torch.cuda.synchronize()
    return cls_feat[0], {k: v.clone() for k, v in losses.items()}, dets


def test_full_size_parity_with_oracle(full):
    """The whole 640x640 step against the CPU oracle (a few seconds of host time at B=2)."""
    m, sd, batch, cb = full
    cls_feat, losses, dets = _run(m, cb)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        ref_losses, ref_feat, per = O.locator_forward_train(sd, batch, 50, 0, 4, 5, 1)
    scale = max(1.0, float(ref_feat.abs().max()))
    err = float((cls_feat.cpu() - ref_feat).abs().max())
    assert err <= 2e-4 * scale, 'cls_feat max abs err %.3e (scale %.2e)' % (err, scale)
    for k in ('gt_loss', 'pos_loss', 'neg_loss', 'bag_acc'):
        a, b = float(losses[k]), float(ref_losses[k])
        assert abs(a - b) <= 2e-4 * max(abs(b), 1e-6), (k, a, b)


@pytest.mark.parametrize('G', [8, 32, 128])
def test_full_size_masks_and_bag_points_bit_exact(G):
    """Negative mask (25 600 grid points x G gts) and bag geometry at full size, bit-exact vs the oracle."""
    from pointtinybenchmark_amd import ops
    from pointtinybenchmark_amd.dense_heads.cpr_head import circle_offsets, sqrt_threshold
    batch = synthetic.synthetic_batch(2, 640, 640, G, 3, seed=G)
    boxes = torch.cat(batch['gt_bboxes']).cuda()
    labels = torch.cat(batch['gt_labels']).to(torch.int32).cuda()
    centers = ops.box_centers(boxes)
    gt_start = torch.tensor([0, G, 2 * G], dtype=torch.int32).cuda()
    gt_img = torch.arange(2, dtype=torch.int32).repeat_interleave(G).cuda()
    pad_hw = torch.tensor([640, 640, 640, 640], dtype=torch.int32).cuda()
    lmap = torch.randn((2, 160, 160, 6), generator=torch.Generator().manual_seed(1)).cuda()
    mask, _ = ops.neg_mask_loss(lmap, centers, labels, gt_start, pad_hw, 3, 4, sqrt_threshold(20.0), 1e-6, True)
    pts, valid, _ = ops.bag_sample(lmap, centers, gt_img, pad_hw, circle_offsets(5, 4).cuda(), 4)
    ref_mask, ref_pts, ref_valid = [], [], []
    for b in range(2):
        c = (batch['gt_bboxes'][b][:, :2] + batch['gt_bboxes'][b][:, 2:]) / 2
        ref_mask.append(O.neg_valid_mask(160, 160, 4, 5, c, batch['gt_labels'][b], 3, 640, 640)[1])
        p = O.bag_points(c, 5, 4)
        ref_pts.append(p)
        ref_valid.append(O.inside(p, 640, 640))
    assert torch.equal(mask.cpu().bool(), torch.cat(ref_mask))
    assert torch.equal(pts.cpu(), torch.cat(ref_pts)) and torch.equal(valid.cpu().bool(), torch.cat(ref_valid))


def test_full_size_run_to_run_determinism(full):
    m, sd, batch, cb = full
    f1, l1, d1 = _run(m, cb)
    f1 = f1.clone()
    f2, l2, d2 = _run(m, cb)
    assert torch.equal(f1, f2), 'feature maps differ between two runs on the same input'
    for k in l1:
        assert torch.equal(l1[k], l2[k]), k
    for (a, _), (b, _) in zip(d1, d2):
        assert torch.equal(a, b)


def test_full_size_batch_order_equivariance(full):
    """Swapping the two images swaps the per-image outputs BIT-exactly (tile / slot indexing is per image)."""
    m, sd, batch, cb = full
    f1, l1, d1 = _run(m, cb)
    f1 = f1.clone()
    sw = dict(img=cb['img'].flip(0).contiguous(), img_metas=cb['img_metas'][::-1],
              gt_bboxes=cb['gt_bboxes'][::-1], gt_labels=cb['gt_labels'][::-1], gt_anns_id=cb['gt_anns_id'][::-1])
    f2, l2, d2 = _run(m, sw)
    assert torch.equal(f1.flip(0), f2)
    assert torch.equal(d1[0][0], d2[1][0]) and torch.equal(d1[1][0], d2[0][0])


def _logit_map_vs_oracle(m, sd, cls_feat, ref_feat, C):
    """head._logit_map over the WHOLE map against F.linear of the oracle's features (cpr_head.py:1045-1099: cls_out / ins_out
    on every location): max abs error of the [cls ++ ins] logits."""
    import torch.nn.functional as F
    from pointtinybenchmark_amd import ops
    lmap = m.bbox_head._logit_map(ops.from_nchw(cls_feat)).cpu()                        # (N, H, W, 2C)
    f = ref_feat.permute(0, 2, 3, 1)
    ref = torch.cat([F.linear(f, sd['bbox_head.cls_out.weight'], sd['bbox_head.cls_out.bias']),
                     F.linear(f, sd['bbox_head.ins_out.weight'], sd['bbox_head.ins_out.bias'])], dim=-1)
    assert lmap.shape == ref.shape and lmap.shape[-1] == 2 * C
    return float((lmap - ref).abs().max()), float(ref.abs().max())


def test_full_size_logit_map_vs_oracle(full):
    """north star: 'within 1e-4 on the head logits' -- asserted directly on the full 160x160x2 logit map of the 640x640
    configuration (R50, B=2), not only on the sampled bag logits."""
    m, sd, batch, cb = full
    cls_feat, _, _ = _run(m, cb)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        _, ref_feat, per = O.locator_forward_train(sd, batch, 50, 0, 4, 5, 1)
        err, mag = _logit_map_vs_oracle(m, sd, cls_feat, ref_feat, 1)
    assert err <= 1e-4, 'logit map max abs err %.3e (|logit| max %.2f, bar 1e-4)' % (err, mag)
    # the oracle's own per-location classification logits are the same numbers (neg_logit = F.linear over the grid)
    # (two CPU evaluations of one dot product: the host BLAS picks its summation order by shape, so equal to rounding only)
    assert torch.allclose(per[0]['neg_logit'].reshape(160, 160),
                          torch.nn.functional.linear(ref_feat[0].permute(1, 2, 0), sd['bbox_head.cls_out.weight'],
                                                     sd['bbox_head.cls_out.bias'])[..., 0], rtol=0, atol=2e-5)


def test_headline_batch_64_bit_equal_to_single_image_runs():
    """The configuration bench.py quotes (R50, 640x640, B=64: 1.68 GB maps, 78 % of the kernels' 32-bit byte-offset range).
    Images 0, 31 and 63 of the 64-batch must come out BIT-equal to the same images run alone (tile / region / GroupNorm-slot
    indexing is per image and no kernel's arithmetic depends on the batch size), then the oracle pins the last two images:
    features 2e-4 * scale, the full logit map 1e-4, and the losses of that 2-image sub-batch."""
    from pointtinybenchmark_amd import ops
    m, sd = build_hip_locator(CFG)
    B = 64
    batch = synthetic.synthetic_batch(B, 640, 640, 32, 1, 0)
    cb = to_cuda(batch)
    head = m.bbox_head
    with torch.no_grad():
        cls64, ins64 = head(m.neck(m.backbone(cb['img'])))
        lmap64 = head._logit_map(ops.from_nchw(cls64[0]))
        losses64 = head.loss(cls64, ins64, cb['gt_bboxes'], cb['gt_labels'], cb['img_metas'])
        torch.cuda.synchronize()
        assert all(bool(torch.isfinite(v).all()) for v in losses64.values())
        for i in (0, 31, 63):
            one, _ = head(m.neck(m.backbone(cb['img'][i:i + 1].contiguous())))
            assert torch.equal(one[0][0], cls64[0][i]), 'image %d of the 64-batch differs from its single-image run' % i
            assert torch.equal(head._logit_map(ops.from_nchw(one[0]))[0], lmap64[i])
        sub = {k: v[62:64] for k, v in batch.items()}
        sc = to_cuda(sub)
        l2 = head.loss([cls64[0][62:64]], [ins64[0][62:64]], sc['gt_bboxes'], sc['gt_labels'], sc['img_metas'])
        torch.cuda.synchronize()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        ref_losses, ref_feat, _ = O.locator_forward_train(sd, sub, 50, 0, 4, 5, 1)
        err_l, mag = _logit_map_vs_oracle(m, sd, cls64[0][62:64], ref_feat, 1)
    scale = max(1.0, float(ref_feat.abs().max()))
    err = float((cls64[0][62:64].cpu() - ref_feat).abs().max())
    assert err <= 2e-4 * scale, 'images 62-63 of the 64-batch: cls_feat max abs err %.3e (scale %.2e)' % (err, scale)
    assert err_l <= 1e-4, 'images 62-63 of the 64-batch: logit map max abs err %.3e (bar 1e-4)' % err_l
    for k in ('gt_loss', 'pos_loss', 'neg_loss', 'bag_acc'):
        a, b = float(l2[k]), float(ref_losses[k])
        assert abs(a - b) <= 2e-4 * max(abs(b), 1e-6), (k, a, b)
    del cls64, ins64, lmap64
    torch.cuda.empty_cache()


@pytest.mark.parametrize('depth,size,B', [(50, 640, 96), (101, 1024, 32)])
def test_batches_past_2gib_run_in_chunks_bit_identically(depth, size, B):
    """B=96 at 640x640 (2.5 GB per 160x160x256 map) and B=32 at R101 1024x1024 (2.1 GB): the conv launchers walk such a batch
    in chunks of whole images below the 2 GiB range of their 32-bit buffer offsets (csrc/common.h cpr_images_per_launch).
    Every image of the big batch must equal its single-image run bit for bit -- first / last image and both sides of the
    chunk boundary -- and the loss of the big batch must be finite."""
    cfg = dict(CFG, depth=depth, height=size, width=size, seed=70 + depth)
    m, sd = build_hip_locator(cfg)
    batch = synthetic.synthetic_batch(B, size, size, 8, 1, cfg['seed'])
    cb = to_cuda(batch)
    head = m.bbox_head
    with torch.no_grad():
        big, ins = head(m.neck(m.backbone(cb['img'])))
        losses = m.forward_train(cb['img'], cb['img_metas'], cb['gt_bboxes'], cb['gt_labels'])
        torch.cuda.synchronize()
        assert all(bool(torch.isfinite(v).all()) for v in losses.values())
        for i in (0, B // 2 - 1, B // 2, B - 1):
            one, _ = head(m.neck(m.backbone(cb['img'][i:i + 1].contiguous())))
            assert torch.equal(one[0][0], big[0][i]), 'image %d of the %d-batch differs from its single-image run' % (i, B)
    del big, ins
    torch.cuda.empty_cache()


def test_split_batch_conv_entry_points_match_unsplit_halves():
    """The C-ABI conv entry points on tensors past 2 GiB against the same call on each half (below the range): direct kernel
    with residual + GroupNorm partials, Winograd kernel with fused affine input + partials (bit-equal), weight gradients
    (direct and Winograd: the halves accumulate, equal up to fp32 summation order)."""
    from pointtinybenchmark_amd import ops
    g = torch.Generator().manual_seed(9)
    N, H, W, C = 84, 160, 160, 256                                    # 84 * 160 * 160 * 256 * 4 = 2.2 GB
    x = torch.randn((N, H, W, C), generator=g).cuda()
    h = N // 2
    # direct 1x1 conv 256 -> 256 with residual, ReLU and GroupNorm partials
    w1 = (torch.randn((C, C, 1, 1), generator=g) * 0.05).cuda()
    pc1 = ops.PackedConv(w1, 1, 0)
    res = torch.randn((N, H, W, C), generator=g).cuda()
    y, part = ops.conv2d(x, pc1, residual=res, relu=True, gn_part=True)
    for sl in (slice(0, h), slice(h, N)):
        yy, pp = ops.conv2d(x[sl], pc1, residual=res[sl], relu=True, gn_part=True)
        assert torch.equal(yy, y[sl])
        per = part.shape[0] // N
        assert torch.equal(pp, part[sl.start * per:sl.stop * per])
    del res, y, part, yy, pp
    # Winograd 3x3 with the fused producer affine
    w3 = (torch.randn((C, C, 3, 3), generator=g) * 0.02).cuda()
    pc3 = ops.PackedConv(w3, 1, 1)
    a = (torch.rand((N, C), generator=g) + 0.5).cuda()
    b = torch.randn((N, C), generator=g).cuda()
    y, part = ops.conv2d(x, pc3, in_ab=(a, b), in_relu=True, gn_part=True)
    for sl in (slice(0, h), slice(h, N)):
        yy, pp = ops.conv2d(x[sl], pc3, in_ab=(a[sl].contiguous(), b[sl].contiguous()), in_relu=True, gn_part=True)
        assert torch.equal(yy, y[sl])
        per = part.shape[0] // N
        assert torch.equal(pp, part[sl.start * per:sl.stop * per])
    # weight gradients: whole batch in one call vs the two halves accumulated
    dy = y

    def three(fn):
        whole, g0, g1 = fn(dy, x), fn(dy[:h], x[:h]), fn(dy[h:], x[h:])
        ref = g0 + g1
        return float((whole - ref).abs().max()) / float(ref.abs().max())
    err = three(lambda d, v: ops.conv3x3_wino_wgrad(d, v, w3.shape))
    assert err <= 2e-5, ('winograd weight gradient', err)
    ops.WINOGRAD[0] = False
    try:
        err = three(lambda d, v: ops.conv2d_wgrad(d, v, w3.shape, 1, 1))
    finally:
        ops.WINOGRAD[0] = True
    assert err <= 2e-5, ('direct weight gradient', err)


def test_full_size_conv_linearity():
    """conv(a*x + y) == a*conv(x) + conv(y) on the head's 3x3 256->256 layer at 160x160 (fp32 rounding only)."""
    from pointtinybenchmark_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn((2, 160, 160, 256), generator=g).cuda()
    y = torch.randn((2, 160, 160, 256), generator=g).cuda()
    w = (torch.randn((256, 256, 3, 3), generator=g) * 0.02).cuda()
    pc = ops.PackedConv(w, 1, 1)
    lhs = ops.conv2d((1.5 * x + y).contiguous(), pc)
    rhs = 1.5 * ops.conv2d(x, pc) + ops.conv2d(y, pc)
    err = float((lhs - rhs).abs().max())
    assert err <= 2e-5 * float(rhs.abs().max()) + 1e-5, err


@pytest.mark.parametrize('name,cfg', [
    # BASELINE.json configs[2]: COCO-style 1333x800 padded to /32 -> 800x1344, 80 classes, stride 8, radius 8
    # (T/configs2/COCO/coarsepointv2/coarse_point_refine_r50_fpn_1x_coco400.py:20,51,75-96).  100x168 map: NOT 128-pixel
    # aligned, so this also covers the unfused GroupNorm-statistics path at full size.
    ('coco_style_800x1344', dict(depth=50, num_classes=80, start_level=1, stride=8, radius=8, head_std=0.3, seed=41,
                                 batch=1, height=800, width=1344, num_gts=24)),
    # BASELINE.json configs[0] at ITS OWN shape on the GPU: ResNet-18 (BasicBlock, T/mmdet/models/backbones/resnet.py:13-93,
    # 360-366) + FPN(in_channels 64..512), 640x640, samples_per_gpu = 2, 32 gts per image
    ('r18_640_b2', dict(depth=18, num_classes=1, start_level=0, stride=4, radius=5, head_std=0.3, seed=47, batch=2,
                        height=640, width=640, num_gts=32)),
    # configs[4] backbone at a small size (fast) ...
    ('r101_384', dict(depth=101, num_classes=1, start_level=0, stride=4, radius=5, head_std=0.3, seed=43, batch=1,
                      height=384, width=384, num_gts=12)),
    # ... and BASELINE.json configs[4] at ITS OWN size: ResNet-101 + FPN, 1024x1024, one image, in the fp32 parity mode
    # (the bf16 compute mode of the same config: tests/test_gpu_bf16.py::test_bf16_path_vs_fp32_oracle[r101_1024])
    ('r101_1024', dict(depth=101, num_classes=1, start_level=0, stride=4, radius=5, head_std=0.3, seed=45, batch=1,
                       height=1024, width=1024, num_gts=32)),
])
def test_other_baseline_configs_parity_with_oracle(name, cfg):
    m, sd = build_hip_locator(cfg)
    batch = synthetic.synthetic_batch(cfg['batch'], cfg['height'], cfg['width'], cfg['num_gts'], cfg['num_classes'],
                                      cfg['seed'])
    cb = to_cuda(batch)
    cls_feat, losses, dets = _run(m, cb)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        ref_losses, ref_feat, per = O.locator_forward_train(sd, batch, cfg['depth'], cfg['start_level'], cfg['stride'],
                                                            cfg['radius'], cfg['num_classes'])
        ref = O.cpr_refine(sd, ref_feat, batch['gt_bboxes'], batch['gt_labels'], batch['gt_anns_id'],
                           batch['img_metas'], cfg['stride'], cfg['radius'], cfg['num_classes'])
    scale = max(1.0, float(ref_feat.abs().max()))
    err = float((cls_feat.cpu() - ref_feat).abs().max())
    assert err <= 3e-4 * scale, '%s: cls_feat max abs err %.3e (scale %.2e)' % (name, err, scale)
    for k in ('gt_loss', 'pos_loss', 'neg_loss', 'bag_acc'):
        a, b = float(losses[k]), float(ref_losses[k])
        assert abs(a - b) <= 5e-4 * max(abs(b), 1e-6), (name, k, a, b)
    np.testing.assert_allclose(torch.cat([d for d, _ in dets]).cpu().numpy(),
                               torch.cat([r['dets'] for r in ref]).numpy(), rtol=1e-4, atol=5e-3)
    # the full [cls ++ ins] logit map (2C channels: 160 for the COCO-style case) against the oracle, 1e-4 abs
    with torch.no_grad():
        err_l, mag = _logit_map_vs_oracle(m, sd, cls_feat, ref_feat, cfg['num_classes'])
    assert err_l <= 1e-4 * max(1.0, mag / 16.0), '%s: logit map max abs err %.3e (|logit| max %.2f)' % (name, err_l, mag)


def test_coco_style_800x1344_batch_8_vs_single_images_and_oracle():
    """BASELINE.json configs[2] at ITS OWN batch: R50, 800x1344 (1333x800 padded to /32), 80 classes, stride 8 (start_level 1),
    radius 8, samples_per_gpu = 8 (T/configs2/COCO/coarsepointv2/coarse_point_refine_r50_fpn_1x_coco400.py:20,51,75-96).  The
    100x168 head map is not 128-pixel aligned: GroupNorm statistics take the separate pass and the producer affine is applied by
    gn_apply, at a real batch.  Images 0, 3 and 7 of the 8-batch must equal their single-image runs bit for bit; the oracle
    then pins the last two images (features, the full 160-channel logit map, the losses of that sub-batch) and the whole
    batch's losses must equal forward_train's lazy path."""
    from pointtinybenchmark_amd import ops
    cfg = dict(depth=50, num_classes=80, start_level=1, stride=8, radius=8, head_std=0.3, seed=49, batch=8, height=800,
               width=1344, num_gts=24)
    m, sd = build_hip_locator(cfg)
    B = cfg['batch']
    batch = synthetic.synthetic_batch(B, 800, 1344, 24, 80, cfg['seed'], ragged=True)
    cb = to_cuda(batch)
    head = m.bbox_head
    with torch.no_grad():
        cls8, ins8 = head(m.neck(m.backbone(cb['img'])))
        assert tuple(cls8[0].shape) == (B, 256, 100, 168)
        lmap8 = head._logit_map(ops.from_nchw(cls8[0]))
        losses8 = head.loss(cls8, ins8, cb['gt_bboxes'], cb['gt_labels'], cb['img_metas'])
        lazy8 = m.forward_train(cb['img'], cb['img_metas'], cb['gt_bboxes'], cb['gt_labels'])
        torch.cuda.synchronize()
        for k, v in losses8.items():
            assert bool(torch.isfinite(v).all()), k
            # the lazy path never materialises the normalised maps: same values up to the rounding of the fused affine
            assert abs(float(v) - float(lazy8[k])) <= 1e-4 * max(abs(float(v)), 1e-6), (k, float(v), float(lazy8[k]))
        for i in (0, 3, 7):
            one, _ = head(m.neck(m.backbone(cb['img'][i:i + 1].contiguous())))
            assert torch.equal(one[0][0], cls8[0][i]), 'image %d of the 8-batch differs from its single-image run' % i
            assert torch.equal(head._logit_map(ops.from_nchw(one[0]))[0], lmap8[i])
        sub = {k: v[6:8] for k, v in batch.items()}
        sc = to_cuda(sub)
        l2 = head.loss([cls8[0][6:8]], [ins8[0][6:8]], sc['gt_bboxes'], sc['gt_labels'], sc['img_metas'])
        torch.cuda.synchronize()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        ref_losses, ref_feat, _ = O.locator_forward_train(sd, sub, 50, 1, 8, 8, 80)
        err_l, mag = _logit_map_vs_oracle(m, sd, cls8[0][6:8], ref_feat, 80)
    scale = max(1.0, float(ref_feat.abs().max()))
    err = float((cls8[0][6:8].cpu() - ref_feat).abs().max())
    assert err <= 3e-4 * scale, 'images 6-7 of the 8-batch: cls_feat max abs err %.3e (scale %.2e)' % (err, scale)
    assert err_l <= 1e-4 * max(1.0, mag / 16.0), 'images 6-7: logit map max abs err %.3e (|logit| max %.2f)' % (err_l, mag)
    for k in ('gt_loss', 'pos_loss', 'neg_loss', 'bag_acc'):
        a, b = float(l2[k]), float(ref_losses[k])
        assert abs(a - b) <= 5e-4 * max(abs(b), 1e-6), (k, a, b)
    del cls8, ins8, lmap8
    torch.cuda.empty_cache()


def test_p2p_r50_640_full_network_vs_oracle():
    """BASELINE.json configs[3] at its own size, through the WHOLE network (ResNet-50 -> FPN -> P2PHead towers -> Hungarian
    targets -> losses, and top-k + pseudo-box NMS), one 640x640 image with 32 gts: tower outputs against the oracle on the
    same weights (1e-4 logits), assignment bit-exact on the device's own predictions, detections against the oracle."""
    import pointtinybenchmark_amd as P
    from bench import p2p_model_cfg
    m = P.build_detector(p2p_model_cfg(50)).cuda()
    sd = synthetic.locator_state_dict(50, 1, 0, 'p2p', 61, head_std=0.05)
    m.load_state_dict(sd, strict=True)
    m.train()
    batch = synthetic.synthetic_batch(1, 640, 640, 32, 1, seed=61)
    cb = to_cuda(batch)
    head = m.bbox_head
    with torch.no_grad():
        cls_outs, pts_outs = head(m.neck(m.backbone(cb['img'])))
        losses = head.loss(cls_outs, pts_outs, cb['gt_bboxes'], cb['gt_labels'], cb['img_metas'])
        anchor, pred, valid, cls = head.get_pred_points(cls_outs, pts_outs, cb['img_metas'])
        gtp = head.pseudo_bbox_to_center(cb['gt_bboxes'])
        gt_inds = head.assign_batch(pred, cls, gtp, cb['gt_labels'], cb['img_metas'])
        res = head.get_bboxes(cls_outs, pts_outs, cb['img_metas'])
        torch.cuda.synchronize()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        feats = O.fpn_forward(sd, O.resnet_forward(sd, batch['img'], 50), 0, 1)
        rc, rp = O.p2p_head_forward(sd, feats)
    ec = float((cls_outs[0].cpu() - rc[0]).abs().max())
    ep = float((pts_outs[0].cpu() - rp[0]).abs().max())
    assert ec <= 1e-4 and ep <= 1e-4, 'P2P tower outputs at 640x640: cls %.3e pts %.3e (bar 1e-4)' % (ec, ep)
    inds, lab, cost = O.hungarian_assign_v2(pred[0, :, :2].cpu(), cls[0].cpu(), gtp[0].cpu(), cb['gt_labels'][0].cpu(),
                                            (640, 640, 3), topk_k=5, log_mode='cr')
    got = gt_inds[0].cpu()
    assert int((got > 0).sum()) == 160 and int((got != inds).sum()) == 0, 'Hungarian indices differ: %d' % int((got != inds).sum())
    dets, labels, topk_inds, keep = O.p2p_get_points_single(cls[0].cpu(), pred[0, :, :2].cpu(), None, (640, 640, 3))
    got_d = res[0][0].cpu()
    assert got_d.shape[0] == dets.shape[0], (got_d.shape, dets.shape)
    ref_boxes = torch.cat([dets[:, :2] - 8, dets[:, :2] + 8, dets[:, 2:]], dim=1)

    def canon(d):     # detections arrive sorted by score; the order INSIDE a run of bit-equal scores is unspecified in the
        d = d.numpy()  # reference (an unstable sort inside mmcv's nms) -> compare with ties ordered by (x1, y1)
        return d[np.lexsort((d[:, 1], d[:, 0], -d[:, 4]))]
    assert np.array_equal(got_d.numpy()[:, 4], ref_boxes.numpy()[:, 4]), 'scores (bit-exact sigmoid) must agree in order'
    np.testing.assert_allclose(canon(got_d), canon(ref_boxes), rtol=1e-5, atol=1e-4)
    assert all(bool(torch.isfinite(v[0] if isinstance(v, (list, tuple)) else v).all()) for v in losses.values())
