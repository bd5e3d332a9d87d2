"""-m gpu: the bf16 compute mode (BASELINE.json configs[4]: bf16 MFMA conv path, fp32 accumulate).  The reference has no
bf16 path, so the tolerance is stated here: conv outputs within bf16 rounding of an fp32 conv on the same bf16-rounded
operands (2^-8 relative + accumulation noise), end-to-end feature maps within 5e-2 of the fp32 oracle after GroupNorm,
losses within 3 %, integer outputs (bag validity, negative mask) still bit-exact (they do not depend on the features)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cpr_oracle as O
from pointtinybenchmark_amd import synthetic
from tests.test_gpu_cpr_parity import build_hip_locator, to_cuda

pytestmark = pytest.mark.gpu

CASES = [
    (2, 64, 40, 40, 64, 3, 1, 1, ''), (2, 256, 24, 20, 64, 1, 1, 0, 'bn relu'), (1, 128, 33, 29, 128, 3, 2, 1, 'bn relu'),
    (2, 256, 20, 20, 512, 1, 2, 0, 'bn'), (2, 64, 25, 21, 256, 1, 1, 0, 'bn res relu'), (1, 256, 32, 32, 256, 3, 1, 1, 'bias gn'),
    (3, 512, 7, 9, 2048, 1, 1, 0, 'bn res relu'), (2, 256, 16, 16, 2, 1, 1, 0, 'bias f32out'), (1, 2048, 5, 5, 256, 1, 1, 0, ''),
    (2, 256, 64, 64, 256, 3, 1, 1, 'gn'), (1, 64, 16, 16, 3, 1, 1, 0, 'bias'),
]


@pytest.mark.parametrize('case', CASES, ids=lambda c: 'n%d_c%d_%dx%d_o%d_k%d_s%d_%s' % (c[:7] + (c[8].replace(' ', '-'),)))
def test_conv2d_bf16_vs_torch(case):
    from pointtinybenchmark_amd import ops
    N, Cin, H, W, Cout, k, stride, pad, flags = case
    g = torch.Generator().manual_seed(abs(hash(case[:8])) % 1000)
    x = torch.randn((N, Cin, H, W), generator=g).bfloat16()
    w = (torch.randn((Cout, Cin, k, k), generator=g) / (Cin * k * k) ** 0.5).bfloat16()
    ref = F.conv2d(x.float(), w.float(), None, stride, pad)
    scale = bias = res = None
    if 'bn' in flags:
        scale = torch.rand(Cout, generator=g) + 0.5
        bias = torch.randn(Cout, generator=g)
        ref = ref * scale[None, :, None, None] + bias[None, :, None, None]
    elif 'bias' in flags:
        bias = torch.randn(Cout, generator=g)
        ref = ref + bias[None, :, None, None]
    if 'res' in flags:
        res = torch.randn(ref.shape, generator=g).bfloat16()
        ref = ref + res.float()
    if 'relu' in flags:
        ref = F.relu(ref)
    pc = ops.PackedConv(w.float().cuda(), stride, pad, torch.bfloat16)
    xin = x.permute(0, 2, 3, 1).contiguous().cuda()
    f32out = 'f32out' in flags
    out = ops.conv2d(xin, pc, scale=None if scale is None else scale.cuda(), bias=None if bias is None else bias.cuda(),
                     residual=None if res is None else res.permute(0, 2, 3, 1).contiguous().cuda(), relu='relu' in flags,
                     gn_part='gn' in flags, out_dtype=torch.float32 if f32out else None)
    part = None
    if 'gn' in flags:
        out, part = out
    torch.cuda.synchronize()
    assert out.dtype == (torch.float32 if f32out else torch.bfloat16)
    got = out.float().permute(0, 3, 1, 2).cpu()
    tol = (1e-4 if f32out else 2.0 ** -7) * ref.abs() + 2e-3
    bad = (got - ref).abs() > tol
    assert not bool(bad.any()), 'max abs err %.3e, %d/%d over tol' % (float((got - ref).abs().max()), int(bad.sum()), bad.numel())
    if part is not None:       # statistics come from the fp32 accumulators, not the rounded outputs
        s = part.reshape(N, -1, Cout, 2).sum(1).cpu()
        np.testing.assert_allclose(s[..., 0].numpy(), ref.sum(dim=(2, 3)).numpy(), rtol=1e-3, atol=2e-2)
        np.testing.assert_allclose(s[..., 1].numpy(), (ref * ref).sum(dim=(2, 3)).numpy(), rtol=1e-3, atol=2e-2)


# shapes that reach the LDS-DMA staged kernels (csrc/conv_bf16_dma.hip; dispatch rule: bf16_dma_shape in conv_mfma_bf16.hip).
# 256 x 256 tiles (interleaved cout layout, whole-line pair stores): every layer with >= 384 such tiles; 128 x 128 tiles, two
# workgroups per CU, output through LDS (round 4): the rest with Cout % 128 == 0 and >= 256 tiles
# the variant words conv_bf16_dma_launch reports.  The 256 x 256 tile has three instances: PP = two wave groups in ping-pong (round 6,
# csrc/conv_bf16_pp.hip: every layer with >= 4 K chunks), BIG = the weights loaded straight into registers from the fragment-order
# image (round 5, conv_bf16_dma_kernel<4, 2, 4, true>: the short-K layers), BIG_LDS = both operands through LDS-DMA in lock step
# (ops.WFRAG[0] = False) -- identical MFMA sequence per accumulator on identical operands, so all three must agree BIT for bit
BIG, BIG_LDS, SMALL = 3000000 + 256 * 1000 + 256, 256 * 1000 + 256, 1000000 + 128 * 1000 + 128
PP = 5000000 + 256 * 1000 + 256
DMA_CASES = [
    (8, 64, 128, 128, 256, 3, 1, 1, 'gn', PP),                  # 3x3, borders on every side, GroupNorm statistics (two slots per tile)
    (8, 128, 128, 128, 256, 3, 1, 1, 'bn res relu', PP),        # K = 1152 + residual (4-byte pair loads)
    (10, 128, 121, 119, 256, 3, 1, 1, 'bias relu', PP),         # ragged M: the last tile is partial
    (6, 128, 256, 192, 512, 3, 2, 1, 'bn', PP),                 # stride 2, two cout tiles
    (8, 1024, 128, 128, 256, 1, 1, 0, 'bias f32out', PP),       # plain GEMM path, fp32 output (8-byte pair stores)
    (8, 128, 128, 128, 256, 1, 1, 0, 'bn res relu', BIG),        # bottleneck conv3 form, two K chunks
    (8, 64, 128, 128, 256, 1, 1, 0, 'bn res relu', BIG),         # a single K chunk (layer1 conv3: 64 -> 256)
    (8, 64, 96, 128, 512, 3, 1, 1, 'bias relu gn', PP),
    (13, 256, 80, 96, 256, 3, 1, 1, 'gn', PP),                    # K = 2304: 36 chunks (the head layer's K), 390 tiles
    (8, 256, 128, 128, 512, 1, 1, 0, 'bn res relu', PP),          # exactly 4 chunks: the shortest K the ping-pong instance takes
    (8, 320, 128, 128, 256, 1, 1, 0, 'bn relu', PP),              # 5 chunks: an odd count through the 3-stage / 2-stage rings         # statistics behind bias + ReLU, two cout tiles
    (8, 256, 64, 64, 256, 3, 1, 1, 'bn relu', SMALL),            # R101 layer3 conv2 at 1024^2 B = 8: 128 big tiles would idle half the CUs
    (8, 1024, 64, 64, 256, 1, 1, 0, 'bn res relu', SMALL),       # layer3 conv1 form, K = 1024
    (3, 64, 121, 119, 256, 3, 1, 1, 'bias res relu', SMALL),     # ragged M + residual: rows past M neither read nor written
    (8, 128, 128, 128, 128, 3, 1, 1, 'bn relu', SMALL),          # one cout tile (layer2 conv2)
    (4, 128, 256, 192, 128, 3, 2, 1, 'bn', SMALL),               # stride 2
    (8, 64, 128, 128, 128, 1, 1, 0, 'bn res relu', SMALL),       # a single K chunk
    (8, 256, 100, 100, 128, 1, 1, 0, 'bias f32out', SMALL),      # fp32 output: the direct epilogue, ragged M
    (8, 512, 32, 32, 2048, 1, 1, 0, 'bn res relu', SMALL),       # layer4 conv3: 16 cout tiles
]


@pytest.mark.parametrize('case', DMA_CASES, ids=lambda c: 'n%d_c%d_%dx%d_o%d_k%d_s%d_%s' % (c[:7] + (c[8].replace(' ', '-'),)))
def test_conv2d_bf16_dma_kernel_vs_torch(case):
    """The same bar as test_conv2d_bf16_vs_torch on shapes that dispatch to the LDS-DMA staged kernel (the launched template
    instance is asserted through the variant word)."""
    from pointtinybenchmark_amd import ops
    N, Cin, H, W, Cout, k, stride, pad, flags = case[:9]
    want = case[9] if len(case) > 9 else BIG
    g = torch.Generator().manual_seed(sum(case[:8]))
    x = torch.randn((N, Cin, H, W), generator=g).bfloat16()
    w = (torch.randn((Cout, Cin, k, k), generator=g) / (Cin * k * k) ** 0.5).bfloat16()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ref = F.conv2d(x.float(), w.float(), None, stride, pad)
    scale = bias = res = None
    if 'bn' in flags:
        scale = torch.rand(Cout, generator=g) + 0.5
        bias = torch.randn(Cout, generator=g)
        ref = ref * scale[None, :, None, None] + bias[None, :, None, None]
    elif 'bias' in flags:
        bias = torch.randn(Cout, generator=g)
        ref = ref + bias[None, :, None, None]
    if 'res' in flags:
        res = torch.randn(ref.shape, generator=g).bfloat16()
        ref = ref + res.float()
    if 'relu' in flags:
        ref = F.relu(ref)
    pc = ops.PackedConv(w.float().cuda(), stride, pad, torch.bfloat16)
    xin = x.permute(0, 2, 3, 1).contiguous().cuda()
    f32out = 'f32out' in flags
    ops.TRACE_CONV_VARIANT[0] = True
    try:
        out = ops.conv2d(xin, pc, scale=None if scale is None else scale.cuda(), bias=None if bias is None else bias.cuda(),
                         residual=None if res is None else res.permute(0, 2, 3, 1).contiguous().cuda(), relu='relu' in flags,
                         gn_part='gn' in flags, out_dtype=torch.float32 if f32out else None)
        variant = ops.TRACE_CONV_VARIANT[1]
    finally:
        ops.TRACE_CONV_VARIANT[0] = False
    assert variant == ('bf16', want), 'this shape must run the LDS-DMA staged instance %d, got %r' % (want, variant)
    if want in (BIG, PP):   # the same launch with both operands staged through LDS in lock step: bit-equal outputs and statistics
        ops.WFRAG[0], ops.TRACE_CONV_VARIANT[0] = False, True
        try:
            out2 = ops.conv2d(xin, pc, scale=None if scale is None else scale.cuda(), bias=None if bias is None else bias.cuda(),
                              residual=None if res is None else res.permute(0, 2, 3, 1).contiguous().cuda(), relu='relu' in flags,
                              gn_part='gn' in flags, out_dtype=torch.float32 if f32out else None)
            variant2 = ops.TRACE_CONV_VARIANT[1]
        finally:
            ops.WFRAG[0], ops.TRACE_CONV_VARIANT[0] = True, False
        assert variant2 == ('bf16', BIG_LDS), variant2
        for u, v in zip(out if isinstance(out, tuple) else (out,), out2 if isinstance(out2, tuple) else (out2,)):
            assert torch.equal(u, v), 'instance %d differs from the lock-step LDS-staged one: max abs %%.3e' % want % float(
                (u.float() - v.float()).abs().max())
    part = None
    if 'gn' in flags:
        out, part = out
    torch.cuda.synchronize()
    got = out.float().permute(0, 3, 1, 2).cpu()
    tol = (1e-4 if f32out else 2.0 ** -7) * ref.abs() + 2e-3
    bad = (got - ref).abs() > tol
    assert not bool(bad.any()), 'max abs err %.3e, %d/%d over tol' % (float((got - ref).abs().max()), int(bad.sum()), bad.numel())
    if part is not None:
        s_ = part.reshape(N, -1, Cout, 2).sum(1).cpu()
        np.testing.assert_allclose(s_[..., 0].numpy(), ref.sum(dim=(2, 3)).numpy(), rtol=1e-3, atol=5e-2)
        np.testing.assert_allclose(s_[..., 1].numpy(), (ref * ref).sum(dim=(2, 3)).numpy(), rtol=1e-3, atol=5e-2)


@pytest.mark.parametrize('N,H,W,seed', [(2, 64, 96, 0), (1, 61, 75, 1), (2, 224, 224, 2), (1, 7, 5, 3), (1, 130, 258, 4)])
def test_stem_bf16_vs_torch(N, H, W, seed):
    """csrc/stem_bf16.hip (round 4): conv 7x7 / 2 / pad 3, 3 -> 64 + folded BN + ReLU on the bf16 matrix cores against F.conv2d in
    fp32 on the SAME bf16-rounded image and weights: |err| <= 2^-7 |ref| + 2e-3 (one bf16 rounding of the output; sums of 147
    products in fp32), odd sizes, partial tiles and maps smaller than a tile included; and against the fp32-kernel stem that the
    mode used before (same bar on its own bf16 output)."""
    from pointtinybenchmark_amd import ops
    g = torch.Generator().manual_seed(seed)
    img = torch.randn((N, 3, H, W), generator=g) * 1.2
    w = torch.randn((64, 3, 7, 7), generator=g) * 0.08
    scale = torch.rand(64, generator=g) + 0.5
    bias = torch.randn(64, generator=g) * 0.3
    ref = F.conv2d(img.bfloat16().float(), w.bfloat16().float(), None, 2, 3)
    ref = F.relu(ref * scale[None, :, None, None] + bias[None, :, None, None])
    x = ops.nchw_to_nhwc(img.cuda())
    assert x.shape[-1] == 4
    out = ops.stem7x7s2_bf16(x, ops.stem_weight_bf16(w.cuda()), scale.cuda(), bias.cuda(), relu=True)
    torch.cuda.synchronize()
    assert out.dtype == torch.bfloat16 and tuple(out.shape) == (N, ref.shape[2], ref.shape[3], 64)
    got = out.float().permute(0, 3, 1, 2).cpu()
    tol = 2.0 ** -7 * ref.abs() + 2e-3
    bad = (got - ref).abs() > tol
    assert not bool(bad.any()), 'max abs err %.3e, %d/%d over tol' % (float((got - ref).abs().max()), int(bad.sum()), bad.numel())
    old = ops.conv2d(x, ops.PackedConv(w.cuda(), 2, 3), scale=scale.cuda(), bias=bias.cuda(), relu=True, out_dtype=torch.bfloat16)
    ref32 = F.relu(F.conv2d(img, w, None, 2, 3) * scale[None, :, None, None] + bias[None, :, None, None])
    d = (got - old.float().permute(0, 3, 1, 2).cpu()).abs()
    assert float(d.max()) <= 0.03 * float(ref32.abs().max()), float(d.max())      # bf16 inputs vs fp32 inputs: 8-bit operands
    # the fused conv + BN + ReLU + max-pool kernel: the SAME bits as the two kernels one after the other
    fused = ops.stem7x7s2_pool_bf16(x, ops.stem_weight_bf16(w.cuda()), scale.cuda(), bias.cuda())
    pair = ops.maxpool3x3s2(out)
    torch.cuda.synchronize()
    assert fused.shape == pair.shape and torch.equal(fused, pair), 'fused stem + pool differs from conv -> pool: %d entries' % int(
        (fused != pair).sum())
    # the (N,3,H,W) network input read plane by plane: the same bits from both kernels
    wq = ops.stem_weight_bf16(w.cuda())
    assert torch.equal(ops.stem7x7s2_pool_bf16(img.cuda().contiguous(), wq, scale.cuda(), bias.cuda(), planar=True), fused)
    assert torch.equal(ops.stem7x7s2_bf16(img.cuda().contiguous(), wq, scale.cuda(), bias.cuda(), relu=True, planar=True), out)


def test_bf16_aux_kernels():
    from pointtinybenchmark_amd import ops
    g = torch.Generator().manual_seed(2)
    x = torch.randn((2, 256, 20, 28), generator=g).bfloat16()
    u = torch.randn((2, 256, 10, 14), generator=g).bfloat16()
    gam, bet = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g)
    ref = F.relu(F.group_norm(x.float(), 32, gam, bet, 1e-5)) + F.interpolate(u.float(), size=(20, 28), mode='nearest')
    xn = x.permute(0, 2, 3, 1).contiguous().cuda()
    part = ops.gn_stats(xn)
    a, b = ops.gn_finalize(part, gam.cuda(), bet.cuda(), 2, 20 * 28, 32, 1e-5)
    out = ops.gn_apply(xn, a, b, relu=True, up=u.permute(0, 2, 3, 1).contiguous().cuda())
    assert out.dtype == torch.bfloat16
    err = (out.float().permute(0, 3, 1, 2).cpu() - ref).abs()
    assert float(err.max()) <= 2.0 ** -7 * float(ref.abs().max()) + 1e-3
    p = torch.randn((2, 64, 33, 31), generator=g).bfloat16()
    mp = ops.maxpool3x3s2(p.permute(0, 2, 3, 1).contiguous().cuda())
    assert torch.equal(mp.float().permute(0, 3, 1, 2).cpu(), F.max_pool2d(p.float(), 3, 2, 1))


@pytest.mark.parametrize('cfg', [
    dict(depth=50, num_classes=1, start_level=0, stride=4, radius=5, head_std=0.3, seed=51, batch=2, height=256, width=256,
         num_gts=8),
    dict(depth=101, num_classes=3, start_level=0, stride=4, radius=5, head_std=0.3, seed=53, batch=1, height=384,
         width=320, num_gts=10),
    # BASELINE.json configs[4] at its own size: ResNet-101 + FPN, 1024x1024, bf16 compute mode, one image
    dict(depth=101, num_classes=1, start_level=0, stride=4, radius=5, head_std=0.3, seed=45, batch=1, height=1024,
         width=1024, num_gts=32),
], ids=['r50_256', 'r101_384x320', 'r101_1024'])
def test_bf16_path_vs_fp32_oracle(cfg):
    from pointtinybenchmark_amd import ops
    m, sd = build_hip_locator(cfg)
    m.set_compute_dtype('bf16')
    batch = synthetic.synthetic_batch(cfg['batch'], cfg['height'], cfg['width'], cfg['num_gts'], cfg['num_classes'],
                                      cfg['seed'])
    cb = to_cuda(batch)
    with torch.no_grad():
        feats = m.neck(m.backbone(cb['img']))
        assert feats[0].dtype == torch.bfloat16
        cls_feat, ins_feat = m.bbox_head(feats)
        losses = m.bbox_head.loss(cls_feat, ins_feat, cb['gt_bboxes'], cb['gt_labels'], cb['img_metas'])
        fused = m.forward_train(cb['img'], cb['img_metas'], cb['gt_bboxes'], cb['gt_labels'])
        dets = m.bbox_head.get_bboxes(cls_feat, ins_feat, cb['img_metas'], gt_bboxes=cb['gt_bboxes'],
                                      gt_labels=cb['gt_labels'], gt_anns_id=cb['gt_anns_id'])
        torch.cuda.synchronize()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        ref_losses, ref_feat, per = O.locator_forward_train(sd, batch, cfg['depth'], 0, 4, 5, cfg['num_classes'])
    scale = max(1.0, float(ref_feat.abs().max()))
    err = (cls_feat[0].float().cpu() - ref_feat).abs()
    assert float(err.max()) <= 8e-2 * scale and float(err.mean()) <= 1e-2 * scale, \
        'bf16 cls_feat: max %.3e mean %.3e (scale %.2e)' % (float(err.max()), float(err.mean()), scale)
    for k in ('gt_loss', 'pos_loss', 'neg_loss'):
        a, b = float(losses[k]), float(ref_losses[k])
        assert abs(a - b) <= 3e-2 * max(abs(b), 1e-3), (k, a, b)
        assert abs(float(fused[k]) - a) <= 1e-5 * max(abs(a), 1e-6)
    assert all(bool(torch.isfinite(d).all()) for d, _ in dets)


WGRAD_BF16_CASES = [
    # N, H, W, Cin, Cout, k, x dtype
    (2, 16, 16, 256, 256, 3, 'bf16'),      # the head / FPN shape in small: 9 taps, three kw copies
    (3, 13, 21, 256, 128, 3, 'bf16'),      # ragged map (Wp = 24), fewer gradient rows than a tile
    (1, 8, 40, 512, 64, 3, 'f32'),         # two cin tiles, fp32 x rounded on the way in
    (2, 24, 24, 256, 512, 1, 'bf16'),      # 1x1 (a lateral): one tap, no border
    (4, 10, 10, 256, 320, 3, 'bf16'),      # Cout not a multiple of the 256-row tile
]


@pytest.mark.parametrize('case', WGRAD_BF16_CASES, ids=lambda c: 'n%d_%dx%d_c%d_o%d_k%d_%s' % c)
def test_conv_wgrad_bf16_vs_fp64(case):
    """csrc/conv_wgrad_bf16.hip (mixed-precision step): channel-major zero-bordered bf16 copies of both maps + one NT GEMM per tap on
    the LDS-DMA kernel, split over the pixels, fp32 partial sums.  Against torch's fp64 weight gradient of the SAME bf16-rounded
    operands only the fp32 summation order differs: 2e-4 of the gradient's max; against the unrounded operands bf16's 8 bits show
    (3e-2 relative L2).  Accumulating into an existing gradient adds exactly."""
    from pointtinybenchmark_amd import ops
    N, H, W, Cin, Cout, k, xdt = case
    g = torch.Generator().manual_seed(sum(case[:6]))
    x = torch.randn((N, Cin, H, W), generator=g)
    dy = torch.randn((N, Cout, H, W), generator=g) * 0.1
    assert ops.conv_wgrad_bf16_supported((N, H, W, Cin), (Cout, Cin, k, k), 1, k // 2)
    xr, dyr = x.bfloat16().double(), dy.bfloat16().double()
    wz = torch.zeros((Cout, Cin, k, k), dtype=torch.float64, requires_grad=True)
    F.conv2d(xr, wz, padding=k // 2).backward(dyr)
    ref = wz.grad
    xc = x.permute(0, 2, 3, 1).contiguous().cuda()
    if xdt == 'bf16':
        xc = xc.bfloat16()
    dyc = dy.permute(0, 2, 3, 1).contiguous().cuda()
    got = ops.conv_wgrad_bf16(dyc, xc, (Cout, Cin, k, k))
    torch.cuda.synchronize()
    err = float((got.cpu().double() - ref).abs().max() / ref.abs().max())
    assert err <= 2e-4, 'bf16 weight gradient vs fp64 on the rounded operands: %.3e of the max' % err
    wf = torch.zeros((Cout, Cin, k, k), dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wf, padding=k // 2).backward(dy.double())
    rel = float((got.cpu().double() - wf.grad).norm() / wf.grad.norm())
    assert rel <= 3e-2, rel
    acc = got.clone()
    ops.conv_wgrad_bf16(dyc, xc, (Cout, Cin, k, k), out=acc, accumulate=True)
    torch.cuda.synchronize()
    assert torch.equal(acc, got + got)


def test_conv_wgrad_bf16_head_layer_shape_vs_fp32_kernel():
    """The head / FPN layer at its real map size (160 x 160, 256 -> 256, 3x3; B=16: 112 pixel splits x 9 taps): the bf16 weight
    gradient against the fp32 Winograd weight gradient of the same maps -- bf16 operand rounding only (relative L2 <= 1e-2)."""
    from pointtinybenchmark_amd import ops
    g = torch.Generator().manual_seed(3)
    B, H, W, C = 16, 160, 160, 256
    x = torch.randn((B, H, W, C), generator=g).cuda()
    dy = (torch.randn((B, H, W, C), generator=g) * 0.05).cuda()
    ref = ops.conv2d_wgrad(dy, x, (C, C, 3, 3), 1, 1)
    got = ops.conv_wgrad_bf16(dy, x.bfloat16(), (C, C, 3, 3))
    torch.cuda.synchronize()
    rel = float((got - ref).norm() / ref.norm())
    assert rel <= 1e-2, rel
    # round 6: the pixel-major kernel on the same maps in bf16 (its splits and chunk boundaries differ from the rewriting path's, the
    # products are the same): against the rewriting path on the SAME bf16 operands only fp32 summation order differs
    dy16, x16 = dy.bfloat16(), x.bfloat16()
    old = ops.wgrad_tn(True)
    try:
        tn = ops.conv_wgrad_bf16(dy16, x16, (C, C, 3, 3))
        ops.wgrad_tn(False)
        nt = ops.conv_wgrad_bf16(dy16, x16, (C, C, 3, 3))
        torch.cuda.synchronize()
    finally:
        ops.wgrad_tn(old)
    assert float((tn - nt).abs().max()) <= 1e-4 * float(nt.abs().max()), float((tn - nt).abs().max() / nt.abs().max())


@pytest.mark.parametrize('shape', [(256, 256, 3), (512, 128, 1), (64, 64, 3), (1024, 256, 1)])
def test_bf16_weight_pack_kernel_equals_the_torch_expression(shape):
    """csrc/pack.hip cpr_pack_weights_bf16 (round 5: one launch per layer instead of three torch launches; the mixed-precision step
    re-packs every bf16 layer after each optimizer update): the [Cout][K] image, the fragment-order image and the data-gradient
    pack (channels swapped, taps flipped, folded-BN scale multiplied in fp32 before the rounding) must be the torch expressions'
    bits."""
    from pointtinybenchmark_amd import ops
    Cout, Cin, k = shape
    g = torch.Generator().manual_seed(5)
    w = (torch.randn((Cout, Cin, k, k), generator=g) * 0.05).cuda()
    scale = (torch.rand(Cout, generator=g) + 0.5).cuda()
    assert ops.PACK_BF16_KERNEL[0]
    pc = ops.PackedConv(w, 1, k // 2, torch.bfloat16)
    ref = w.permute(0, 2, 3, 1).reshape(Cout, k * k * Cin).to(torch.bfloat16).contiguous()
    assert pc.w.dtype == torch.bfloat16 and torch.equal(pc.w, ref)
    if Cout % 256 == 0:
        G, KS = Cout // 64, k * k * Cin // 16
        assert pc.wfrag is not None and torch.equal(pc.frag_image().reshape(-1),
                                                    ref.view(G, 32, 2, KS, 2, 8).permute(0, 3, 2, 4, 1, 5).reshape(-1))
    else:
        assert pc.frag_image() is None
    for sc in (None, scale):
        pd = ops.PackedConv.for_dgrad_bf16(w, k // 2, scale=sc)
        ws = w if sc is None else w * sc[:, None, None, None]
        wt = ws.flip(2, 3).permute(1, 0, 2, 3)
        ops.PACK_BF16_KERNEL[0] = False
        try:
            old = ops.PackedConv(wt, 1, k - 1 - k // 2, torch.bfloat16)
        finally:
            ops.PACK_BF16_KERNEL[0] = True
        assert (pd.Cout, pd.Cin, pd.KH, pd.Kpad, pd.padding) == (old.Cout, old.Cin, old.KH, old.Kpad, old.padding)
        assert torch.equal(pd.w, old.w)
        fo = old.frag_image()
        assert (fo is None) == (pd.frag_image() is None) and (fo is None or torch.equal(pd.frag_image().reshape(-1), fo.reshape(-1)))


MASK_CASES = [
    # N, Cin, H, W, Cout, k, out fp32, instance: every epilogue that knows the mode, ragged pixel counts included
    (8, 128, 64, 64, 128, 3, False, SMALL),       # conv2 of layer2 at 512^2 B=8: 128 x 128 tile, output through LDS, one slot per tile
    (14, 512, 50, 50, 128, 1, False, SMALL),      # conv3's data gradient (1x1, 4 planes -> planes), M = 35 000: ragged last tile
    (8, 128, 100, 99, 128, 3, True, SMALL),       # the same tile's direct epilogue (fp32 out): a slot per wave row, ragged
    (13, 256, 80, 97, 256, 3, False, PP),         # ping-pong instance (>= 4 K chunks), pair epilogue with the mask prefetched, ragged
    (13, 128, 80, 95, 512, 1, False, BIG),        # weights-direct instance (2 K chunks), M = 98 800: ragged
]


@pytest.mark.parametrize('case', MASK_CASES, ids=lambda c: 'n%d_c%d_%dx%d_o%d_k%d_%s_%d' % (c[:6] + ('f32' if c[6] else 'bf16', c[7])))
def test_conv2d_bf16_mask_mode_equals_conv_then_streaming_pass(case):
    """Round 6: the bf16 data gradient with the ReLU backward, the bf16 rounding and the column sums in its epilogue (relu == 2 of
    cpr_conv2d_fwd_bf16) against the launch + streaming pass it replaces (fp32 result, ops.relu_bwd_colsum with want16): the map must
    be BIT-equal (same accumulators, same single rounding), the column sums equal up to the order of the fp32 additions."""
    from pointtinybenchmark_amd import ops
    N, Cin, H, W, Cout, k, f32out, want = case
    g = torch.Generator().manual_seed(sum(case[:6]))
    x = torch.randn((N, H, W, Cin), generator=g).bfloat16().cuda()
    w = (torch.randn((Cout, Cin, k, k), generator=g) / (Cin * k * k) ** 0.5)
    mask = torch.randn((N, H, W, Cout), generator=g).clamp_min(0).bfloat16().cuda()          # a ReLU output: half of it zeros
    pc = ops.PackedConv(w.cuda(), 1, k // 2, torch.bfloat16)
    odt = torch.float32 if f32out else torch.bfloat16
    slots = ops.conv2d_bf16_mask_slots(x.shape, pc, odt)
    assert slots > 0
    ops.TRACE_CONV_VARIANT[0] = True
    try:
        out, part = ops.conv2d(x, pc, residual=mask, res_mask=True, colsum=True, out_dtype=odt)
        variant = ops.TRACE_CONV_VARIANT[1]
    finally:
        ops.TRACE_CONV_VARIANT[0] = False
    assert variant == ('bf16', want), variant
    assert part.tiles == slots and out.dtype == odt
    cs = part.reduce()
    plain = ops.conv2d(x, pc, out_dtype=torch.float32)
    g32, cs_ref, g16 = ops.relu_bwd_colsum(plain, mask, want16=True)
    torch.cuda.synchronize()
    ref = g32 if f32out else g16
    assert torch.equal(out, ref), 'mask-mode map differs in %d entries' % int((out != ref).sum())
    scale = float(g32.abs().sum(dim=(0, 1, 2)).max())
    assert float((cs - cs_ref).abs().max()) <= 1e-5 * scale, (float((cs - cs_ref).abs().max()), scale)
    # the column sums against fp64 of the map itself
    np.testing.assert_allclose(cs.cpu().numpy(), g32.double().sum(dim=(0, 1, 2)).cpu().numpy(), rtol=0, atol=2e-5 * scale)


def test_conv2d_bf16_mask_mode_says_which_shapes_it_takes():
    """Shapes whose kernel is the register-staged one (too few tiles for an LDS-DMA instance) report 0 slots, and asking for the
    mode anyway is refused by the library (CPR_ERR_UNSUPPORTED), not computed some other way."""
    from pointtinybenchmark_amd import ops
    from pointtinybenchmark_amd._lib import CprHipError
    w = torch.randn((128, 128, 3, 3)).cuda() * 0.03
    pc = ops.PackedConv(w, 1, 1, torch.bfloat16)
    assert ops.conv2d_bf16_mask_slots((2, 16, 16, 128), pc) == 0
    assert ops.conv2d_bf16_mask_slots((8, 64, 64, 128), pc) == 8 * 64 * 64 // 128
    x = torch.randn((2, 16, 16, 128)).bfloat16().cuda()
    with pytest.raises(AssertionError):
        ops.conv2d(x, pc, residual=torch.ones_like(x), res_mask=True, colsum=True)
    import ctypes
    from pointtinybenchmark_amd import _lib
    out = torch.empty_like(x)
    part = torch.empty((16, 128, 2), device='cuda')
    with pytest.raises(CprHipError, match='unsupported'):
        _lib.call('cpr_conv2d_fwd_bf16', x.data_ptr(), pc.w.data_ptr(), None, out.data_ptr(), None, None, x.data_ptr(), part.data_ptr(),
                  2, 16, 16, 128, 128, 3, 3, 1, 1, pc.Kpad, 2, 0, None, None)


FUSED_CASES = [
    # N, Cin, H, W, Cout, instance: conv1's data gradient of a bottleneck (1x1, planes -> 4 planes) on the two pair-epilogue instances
    (13, 128, 80, 95, 512, BIG),          # layer2: two K chunks -> weights direct to registers; M = 98 800: ragged last tile
    (16, 256, 40, 39, 1024, PP),          # layer3: four K chunks -> ping-pong; M = 24 960: ragged
    (13, 512, 32, 32, 2048, PP),          # layer4: eight chunks
]


@pytest.mark.parametrize('case', FUSED_CASES, ids=lambda c: 'n%d_c%d_%dx%d_o%d_%d' % c)
def test_conv2d_dgrad_bf16_fused_equals_conv_then_streaming_pass(case):
    """Round 6: the block-boundary data gradient in one launch (cpr_conv2d_dgrad_bf16_fused: shortcut sum + ReLU mask + fp32 and
    bf16 outputs + column sums in the epilogue of the 256 x 256 tile's TR instances) against what it replaces (fp32 conv output, then
    ops.relu_bwd_colsum with add and want16): both maps BIT-equal, column sums equal up to the order of the fp32 additions."""
    from pointtinybenchmark_amd import ops
    N, Cin, H, W, Cout, want = case
    g = torch.Generator().manual_seed(sum(case[:5]))
    x = torch.randn((N, H, W, Cin), generator=g).bfloat16().cuda()
    w = torch.randn((Cout, Cin, 1, 1), generator=g) / Cin ** 0.5
    mask = torch.randn((N, H, W, Cout), generator=g).clamp_min(0).bfloat16().cuda()
    add = torch.randn((N, H, W, Cout), generator=g).cuda()
    pc = ops.PackedConv(w.cuda(), 1, 0, torch.bfloat16)
    assert ops.conv2d_bf16_mask_slots(x.shape, pc, fused_add=True) == (N * H * W + 255) // 256 * 2
    ops.TRACE_CONV_VARIANT[0] = True
    try:
        g32, g16, part = ops.conv2d_dgrad_bf16_fused(x, pc, mask, add)
        variant = ops.TRACE_CONV_VARIANT[1]
    finally:
        ops.TRACE_CONV_VARIANT[0] = False
    assert variant == ('bf16', want), variant
    cs = part.reduce()
    r32, cs_ref, r16 = ops.relu_bwd_colsum(ops.conv2d(x, pc, out_dtype=torch.float32), mask, want16=True, add=add)
    torch.cuda.synchronize()
    assert torch.equal(g32, r32), 'fp32 map differs in %d entries' % int((g32 != r32).sum())
    assert torch.equal(g16, r16), 'bf16 map differs in %d entries' % int((g16 != r16).sum())
    scale = float(r32.abs().sum(dim=(0, 1, 2)).max())
    assert float((cs - cs_ref).abs().max()) <= 1e-5 * scale
    np.testing.assert_allclose(cs.cpu().numpy(), r32.double().sum(dim=(0, 1, 2)).cpu().numpy(), rtol=0, atol=2e-5 * scale)
    # shapes of the 128-pixel tile have no such instance: the query says so
    small = ops.PackedConv(torch.randn((128, 512, 1, 1)).cuda() * 0.04, 1, 0, torch.bfloat16)
    assert ops.conv2d_bf16_mask_slots((14, 50, 50, 512), small, fused_add=True) == 0


WGRAD_TN_CASES = [
    # N, H, W, Cin, Cout, k, stride: the pixel-major kernel reads both NHWC maps as they are (bf16)
    (2, 24, 24, 256, 512, 1, 1),      # 1x1 (a lateral)
    (3, 17, 23, 256, 256, 3, 1),      # 3x3, ragged pixel count (1173 = 18 chunks + 21), W < 64: a chunk spans rows
    (2, 9, 7, 512, 320, 3, 1),        # tiny map: a chunk spans more than an image's rows; Cout not a multiple of the 256-row tile
    (1, 40, 72, 256, 64, 3, 1),       # W > 64, one partial cout tile
    (4, 32, 32, 1024, 256, 1, 1),     # four cin tiles
    (2, 16, 16, 128, 256, 3, 1),      # Cin = 128: half a cin tile (only this kernel takes the shape)
    (3, 20, 20, 512, 128, 1, 1),      # 1x1 with Cout < 256 (the rewriting path declines it: its rewrites cost what its GEMM saves)
    (3, 34, 30, 128, 128, 3, 2),      # conv2 of a stage's first block: 3x3 / stride 2 (17 x 15 outputs)
    (2, 33, 31, 256, 512, 1, 2),      # its projection shortcut: 1x1 / stride 2, odd map (17 x 16 outputs)
    (5, 9, 11, 256, 256, 3, 2),       # stride 2 on a tiny map: a chunk spans images
]


@pytest.mark.parametrize('case', WGRAD_TN_CASES, ids=lambda c: 'n%d_%dx%d_c%d_o%d_k%d_s%d' % c)
def test_conv_wgrad_bf16_pixel_major_kernel_vs_fp64(case):
    """Round 6: csrc/conv_wgrad_bf16_tn.hip -- the bf16 weight gradient straight from the NHWC maps (LDS-DMA of pixel rows,
    ds_read_b64_tr_b16 fragments, no channel-major rewrites).  Same bars as the rewriting path: against torch's fp64 weight gradient of
    the SAME bf16 operands only the fp32 summation order differs (2e-4 of the gradient's max); and against the rewriting path itself
    (different split boundaries, same products) 2e-4 as well."""
    from pointtinybenchmark_amd import _lib, ops
    N, H, W, Cin, Cout, k, stride = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn((N, Cin, H, W), generator=g).bfloat16()
    OH, OW = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    dy = (torch.randn((N, Cout, OH, OW), generator=g) * 0.1).bfloat16()
    wz = torch.zeros((Cout, Cin, k, k), dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wz, stride=stride, padding=k // 2).backward(dy.double())
    ref = wz.grad
    xc = x.permute(0, 2, 3, 1).contiguous().cuda()
    dyc = dy.permute(0, 2, 3, 1).contiguous().cuda()
    old = ops.wgrad_tn(True)
    try:
        got = ops.conv_wgrad_bf16(dyc, xc, (Cout, Cin, k, k), stride=stride)
        acc = got.clone()
        ops.conv_wgrad_bf16(dyc, xc, (Cout, Cin, k, k), out=acc, accumulate=True, stride=stride)
        torch.cuda.synchronize()
    finally:
        ops.wgrad_tn(old)
    err = float((got.cpu().double() - ref).abs().max() / ref.abs().max())
    assert err <= 2e-4, 'pixel-major bf16 weight gradient vs fp64 on the same operands: %.3e of the max' % err
    assert torch.equal(acc, got + got)
    assert ops.conv_wgrad_bf16_supported((N, H, W, Cin), (Cout, Cin, k, k), stride, k // 2, maps_bf16=True)
    assert not ops.conv_wgrad_bf16_supported((N, H, W, Cin), (Cout, Cin, k, k), stride, k // 2) or (stride == 1 and Cin % 256 == 0)
    if Cin % 256 == 0 and stride == 1:
        old = ops.wgrad_tn(False)
        try:
            nt = ops.conv_wgrad_bf16(dyc, xc, (Cout, Cin, k, k))
            torch.cuda.synchronize()
        finally:
            ops.wgrad_tn(old)
        assert float((got - nt).abs().max()) <= 2e-4 * float(ref.abs().max())
