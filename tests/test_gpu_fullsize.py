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
