"""-m gpu: each HIP kernel through the C-ABI against a torch-CPU fp32 reference of the same op."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ops():
    from pointtinybenchmark_amd import ops
    return ops


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().cuda()


def _cmp(name, got, ref, atol, rtol=1e-5):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    assert bool((err <= tol).all()), '%s: max abs err %.3e (ref max %.3e), %d/%d over tol' % (
        name, float(err.max()), float(ref.abs().max()), int((err > tol).sum()), err.numel())


CONV_CASES = [
    # N, Cin, H, W, Cout, k, stride, pad, flags
    (2, 64, 40, 40, 64, 3, 1, 1, ''),
    (2, 256, 24, 20, 64, 1, 1, 0, 'bn relu'),
    (1, 128, 33, 29, 128, 3, 2, 1, 'bn relu'),
    (2, 256, 20, 20, 512, 1, 2, 0, 'bn'),
    (2, 3, 64, 96, 64, 7, 2, 3, 'bn relu'),
    (2, 64, 25, 21, 256, 1, 1, 0, 'bn res relu'),
    (1, 256, 32, 32, 256, 3, 1, 1, 'bias'),
    (3, 512, 7, 9, 2048, 1, 1, 0, 'bn res relu'),
    (2, 256, 16, 16, 2, 1, 1, 0, 'bias'),
    (1, 2048, 5, 5, 256, 1, 1, 0, ''),
]


@pytest.mark.parametrize('case', CONV_CASES, ids=lambda c: 'n%d_c%d_%dx%d_o%d_k%d_s%d_%s' % (c[:7] + (c[8].replace(' ', '-'),)))
def test_conv2d_vs_torch(case):
    ops = _ops()
    N, Cin, H, W, Cout, k, stride, pad, flags = case
    g = torch.Generator().manual_seed(hash(case[:8]) % 1000)
    x = torch.randn((N, Cin, H, W), generator=g)
    w = torch.randn((Cout, Cin, k, k), generator=g) / (Cin * k * k) ** 0.5
    ref = F.conv2d(x, w, None, stride, pad)
    scale = bias = res = None
    if 'bn' in flags:
        scale = torch.rand(Cout, generator=g) + 0.5
        bias = torch.randn(Cout, generator=g)
        ref = ref * scale[None, :, None, None] + bias[None, :, None, None]
    elif 'bias' in flags:
        bias = torch.randn(Cout, generator=g)
        ref = ref + bias[None, :, None, None]
    if 'res' in flags:
        res = torch.randn(ref.shape, generator=g)
        ref = ref + res
    if 'relu' in flags:
        ref = F.relu(ref)
    pc = ops.PackedConv(w.cuda(), stride, pad)
    xin = ops.nchw_to_nhwc(x.cuda()) if Cin <= 4 else _nhwc(x)
    out = ops.conv2d(xin, pc, scale=None if scale is None else scale.cuda(), bias=None if bias is None else bias.cuda(),
                     residual=None if res is None else _nhwc(res), relu='relu' in flags)
    torch.cuda.synchronize()
    _cmp('conv', out.permute(0, 3, 1, 2), ref, atol=2e-5, rtol=2e-5)


DUAL_CASES = [
    # N, H2, W2, Cin2 (shortcut input), stride2, Cin (main 1x1 input), Cout
    (2, 40, 40, 64, 1, 64, 256),        # layer1 block 0
    (2, 40, 36, 256, 2, 128, 512),      # layer2 block 0 (stride-2 shortcut)
    (3, 21, 19, 64, 2, 32, 96),         # ragged M, Cout not a tile multiple, odd map
    (1, 64, 64, 512, 2, 256, 1024),     # long K
    (64, 32, 32, 512, 1, 256, 1024),    # enough tiles for the 128x128 instance
]


@pytest.mark.parametrize('case', DUAL_CASES, ids=lambda c: 'n%d_%dx%d_c%d_s%d_c%d_o%d' % c)
def test_conv2d_dual_bit_equal_to_two_launches(case):
    """conv3 + bn3 + projection shortcut + ReLU in one launch (ops.conv2d_dual) == the two-launch form BIT for bit, and both
    agree with torch (the first block of every ResNet stage, resnet.py:262-302)."""
    ops = _ops()
    N, H2, W2, Cin2, s2, Cin, Cout = case
    g = torch.Generator().manual_seed(sum(case))
    OH, OW = (H2 - 1) // s2 + 1, (W2 - 1) // s2 + 1
    x2 = torch.randn((N, Cin2, H2, W2), generator=g)
    x = torch.randn((N, Cin, OH, OW), generator=g)
    w = torch.randn((Cout, Cin, 1, 1), generator=g) / Cin ** 0.5
    w2 = torch.randn((Cout, Cin2, 1, 1), generator=g) / Cin2 ** 0.5
    sc, bi, sc2, bi2 = [(torch.rand(Cout, generator=g) + 0.5) if i % 2 == 0 else torch.randn(Cout, generator=g) for i in range(4)]
    pc, pc2 = ops.PackedConv(w.cuda(), 1, 0), ops.PackedConv(w2.cuda(), s2, 0)
    xc, x2c = _nhwc(x), _nhwc(x2)
    ident = ops.conv2d(x2c, pc2, scale=sc2.cuda(), bias=bi2.cuda())
    two = ops.conv2d(xc, pc, scale=sc.cuda(), bias=bi.cuda(), residual=ident, relu=True)
    one = ops.conv2d_dual(xc, pc, x2c, pc2, scale=sc.cuda(), bias=bi.cuda(), scale2=sc2.cuda(), bias2=bi2.cuda(), relu=True)
    torch.cuda.synchronize()
    assert torch.equal(one, two), 'fused shortcut differs from the two-launch form: max %.3e' % float((one - two).abs().max())
    ref = F.relu(F.conv2d(x, w) * sc[None, :, None, None] + bi[None, :, None, None] +
                 F.conv2d(x2, w2, None, s2) * sc2[None, :, None, None] + bi2[None, :, None, None])
    _cmp('dual conv', one.permute(0, 3, 1, 2), ref, atol=3e-5, rtol=3e-5)


STREAM_CASES = [
    # N, H, W, Cin, Cout, residual mode (0 none, 1 add, 2 ReLU mask), relu
    (2, 32, 32, 64, 256, 1, True),       # conv3 + shortcut add + ReLU of layer1 (resnet.py:262-302)
    (1, 16, 24, 64, 256, 0, False),      # 3 pixel tiles: some workgroups idle
    (3, 32, 32, 64, 512, 2, False),      # two cout panels; data gradient with the ReLU mask epilogue
    (2, 16, 32, 128, 512, 1, True),      # layer2: four cout panels of 128
    (5, 8, 8, 128, 128, 0, True),        # one panel, 5 tiles of 64 pixels
    (64, 16, 16, 64, 256, 1, True),      # 128 tiles over 128 workgroups ... and enough rows to wrap the double buffer
    (9, 32, 32, 128, 256, 1, False),     # odd tile counts per workgroup
    (2, 32, 32, 256, 64, 0, True),       # K = 256: four k chunks stream through per 128-pixel tile (conv1 of a 256-channel stage)
    (3, 16, 16, 256, 1024, 1, True),     # conv3 of layer3: 16 cout panels, shortcut add
    (1, 16, 24, 256, 128, 2, False),     # three tiles, two panels, mask epilogue
    (5, 32, 32, 256, 256, 1, False),     # 40 tiles x 4 panels: several tiles per workgroup, the chunk stream crosses tiles
]


@pytest.mark.parametrize('case', STREAM_CASES, ids=lambda c: 'n%d_%dx%d_c%d_o%d_r%d_%s' % c)
def test_streamed_1x1_bit_equal_to_tiled_kernel(case):
    """csrc/conv1x1_stream.hip (persistent workgroups, weight panel resident in LDS, pixel tiles by LDS-DMA) == the tiled kernel
    BIT for bit (same accumulation order, same epilogue) -- these launches are small enough that ops.conv2d itself still takes the
    tiled kernel -- and both agree with torch."""
    ops = _ops()
    N, H, W, Cin, Cout, rmode, relu = case
    g = torch.Generator().manual_seed(sum(case[:6]))
    x = torch.randn((N, Cin, H, W), generator=g)
    w = torch.randn((Cout, Cin, 1, 1), generator=g) / Cin ** 0.5
    sc, bi = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    res = torch.randn((N, Cout, H, W), generator=g) if rmode else None
    pc = ops.PackedConv(w.cuda(), 1, 0)
    xc = _nhwc(x)
    rc = _nhwc(res) if res is not None else None
    ops.TRACE_CONV_VARIANT[0] = True
    try:
        tiled = ops.conv2d(xc, pc, scale=sc.cuda(), bias=bi.cuda(), residual=rc, relu=relu, res_mask=rmode == 2)
        assert ops.TRACE_CONV_VARIANT[1][1] % 10 != 3, 'this launch was meant to run the tiled kernel'
    finally:
        ops.TRACE_CONV_VARIANT[0] = False
    stream = ops.conv1x1_stream(xc, pc, scale=sc.cuda(), bias=bi.cuda(), residual=rc, relu=relu, res_mask=rmode == 2)
    torch.cuda.synchronize()
    assert torch.equal(stream, tiled), 'streamed 1x1 differs from the tiled kernel: max %.3e' % float((stream - tiled).abs().max())
    ref = F.conv2d(x, w) * sc[None, :, None, None] + bi[None, :, None, None]
    if rmode == 1:
        ref = ref + res
    elif rmode == 2:
        ref = torch.where(res > 0, ref, torch.zeros_like(ref))
    if relu:
        ref = F.relu(ref)
    _cmp('streamed 1x1', stream.permute(0, 3, 1, 2), ref, atol=3e-5, rtol=3e-5)


def test_conv2d_takes_the_streamed_kernel_for_large_1x1_launches():
    """A launch with >= 1024 tiles of a streamed shape goes to conv1x1_stream_kernel through ops.conv2d (variant code ...3) and
    every image of the batch equals the same image run alone through the tiled kernel."""
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    N, H, W, Cin, Cout = 16, 96, 96, 64, 256       # 1152 tiles of 128 pixels
    x = _nhwc(torch.randn((N, Cin, H, W), generator=g))
    w = torch.randn((Cout, Cin, 1, 1), generator=g) / 8
    res = _nhwc(torch.randn((N, Cout, H, W), generator=g))
    sc, bi = (torch.rand(Cout, generator=g) + 0.5).cuda(), torch.randn(Cout, generator=g).cuda()
    pc = ops.PackedConv(w.cuda(), 1, 0)
    ops.TRACE_CONV_VARIANT[0] = True
    try:
        big = ops.conv2d(x, pc, scale=sc, bias=bi, residual=res, relu=True)
        assert ops.TRACE_CONV_VARIANT[1][1] % 10 == 3, 'expected the streamed kernel, got variant %r' % (ops.TRACE_CONV_VARIANT[1],)
        one = ops.conv2d(x[5:6].contiguous(), pc, scale=sc, bias=bi, residual=res[5:6].contiguous(), relu=True)
        assert ops.TRACE_CONV_VARIANT[1][1] % 10 != 3
    finally:
        ops.TRACE_CONV_VARIANT[0] = False
    torch.cuda.synchronize()
    assert torch.equal(big[5:6], one)


def test_conv_fused_groupnorm_chain():
    """conv -> GN stats in the epilogue -> finalize -> next conv applies GN+ReLU on load, vs the unfused torch ops."""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    x = torch.randn((2, 256, 16, 24), generator=g)
    w1 = torch.randn((256, 256, 3, 3), generator=g) * 0.02
    w2 = torch.randn((256, 256, 3, 3), generator=g) * 0.02
    gam = torch.rand(256, generator=g) + 0.5
    bet = torch.randn(256, generator=g) * 0.1
    y1 = F.conv2d(x, w1, None, 1, 1)
    n1 = F.relu(F.group_norm(y1, 32, gam, bet, 1e-5))
    y2 = F.conv2d(n1, w2, None, 1, 1)
    pc1, pc2 = ops.PackedConv(w1.cuda(), 1, 1), ops.PackedConv(w2.cuda(), 1, 1)
    raw, part = ops.conv2d(_nhwc(x), pc1, gn_part=True)
    a, b, mean, rstd = ops.gn_finalize(part, gam.cuda(), bet.cuda(), 2, 16 * 24, 32, 1e-5, want_stats=True)
    out2 = ops.conv2d(raw, pc2, in_ab=(a, b), in_relu=True)
    torch.cuda.synchronize()
    _cmp('conv1 raw', raw.permute(0, 3, 1, 2), y1, 2e-5)
    yg = y1.reshape(2, 32, -1)
    _cmp('gn mean', mean, yg.mean(-1), 1e-5)
    _cmp('gn rstd', rstd, 1.0 / torch.sqrt(yg.var(-1, unbiased=False) + 1e-5), 1e-4, 1e-5)
    _cmp('conv2 on fused GN input', out2.permute(0, 3, 1, 2), y2, 5e-5, 5e-5)
    # unfused statistics kernel agrees with the fused partials
    part2 = ops.gn_stats(raw)
    a2, b2 = ops.gn_finalize(part2, gam.cuda(), bet.cuda(), 2, 16 * 24, 32, 1e-5)
    _cmp('stats kernel a', a2, a, 1e-6, 1e-5)
    _cmp('stats kernel b', b2, b, 1e-5, 1e-5)
    mat = ops.gn_apply(raw, a, b, relu=True)
    _cmp('gn apply', mat.permute(0, 3, 1, 2), n1, 2e-5, 2e-5)


@pytest.mark.parametrize('shape', [(2, 64, 32, 48), (1, 64, 33, 31)])
def test_maxpool(shape):
    ops = _ops()
    x = torch.randn(shape, generator=torch.Generator().manual_seed(1))
    out = ops.maxpool3x3s2(_nhwc(x))
    assert torch.equal(out.permute(0, 3, 1, 2).cpu(), F.max_pool2d(x, 3, 2, 1))


@pytest.mark.parametrize('N,H,W,seed', [(2, 64, 96, 0), (1, 61, 75, 1), (2, 224, 224, 2), (1, 7, 5, 3), (1, 130, 258, 4), (3, 33, 129, 5)])
def test_fused_stem_vs_torch_and_the_two_kernel_path(N, H, W, seed):
    """csrc/stem_f32.hip (round 4): conv 7x7 / 2 / pad 3 (3 -> 64) + folded BN + ReLU + max-pool 3x3 / 2 / pad 1 in one exact-fp32
    kernel (resnet.py:630-637).  Against torch in fp64 (conv -> affine -> ReLU -> max_pool2d): <= 2e-6 of the map's max + 1e-6
    (fp32 sums of 147 products); against the path it replaces (implicit-GEMM stem + maxpool3x3s2_kernel): <= 4e-6 of the max
    (two fp32 summation orders).  Odd sizes, partial tiles, maps smaller than one tile, negative-heavy outputs (ReLU zeros next to the
    pool's padding)."""
    ops = _ops()
    g = torch.Generator().manual_seed(seed)
    img = torch.randn((N, 3, H, W), generator=g) * 1.2
    w = torch.randn((64, 3, 7, 7), generator=g) * 0.08
    scale = torch.rand(64, generator=g) + 0.5
    bias = torch.randn(64, generator=g) * 0.6 - 0.3
    ref = F.conv2d(img.double(), w.double(), None, 2, 3)
    ref = F.max_pool2d(F.relu(ref * scale.double()[None, :, None, None] + bias.double()[None, :, None, None]), 3, 2, 1).float()
    x = ops.nchw_to_nhwc(img.cuda())
    out = ops.stem7x7s2_pool_f32(x, ops.stem_weight_f32(w.cuda()), scale.cuda(), bias.cuda())
    old = ops.maxpool3x3s2(ops.conv2d(x, ops.PackedConv(w.cuda(), 2, 3), scale=scale.cuda(), bias=bias.cuda(), relu=True))
    torch.cuda.synchronize()
    assert tuple(out.shape) == (N, ref.shape[2], ref.shape[3], 64) and out.shape == old.shape
    got = out.permute(0, 3, 1, 2).cpu()
    mx = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 2e-6 * mx + 1e-6, float((got - ref).abs().max())
    assert float((out - old).abs().max()) <= 4e-6 * mx + 1e-6, float((out - old).abs().max())
    assert float((got == 0).float().mean()) > 0.001       # the ReLU really clips here
    # the (N,3,H,W) network input read plane by plane (no nchw_to_nhwc4 pass): the same bits
    planar = ops.stem7x7s2_pool_f32(img.cuda().contiguous(), ops.stem_weight_f32(w.cuda()), scale.cuda(), bias.cuda(), planar=True)
    assert torch.equal(planar, out)


@pytest.mark.parametrize('shape,up', [((2, 256, 20, 28), (10, 14)), ((1, 256, 25, 21), (13, 11))])
def test_gn_apply_with_nearest_upsample_add(shape, up):
    ops = _ops()
    g = torch.Generator().manual_seed(2)
    x = torch.randn(shape, generator=g)
    u = torch.randn((shape[0], shape[1]) + up, generator=g)
    gam, bet = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g)
    ref = F.group_norm(x, 32, gam, bet, 1e-5) + F.interpolate(u, size=shape[2:], mode='nearest')
    xn = _nhwc(x)
    part = ops.gn_stats(xn)
    a, b = ops.gn_finalize(part, gam.cuda(), bet.cuda(), shape[0], shape[2] * shape[3], 32, 1e-5)
    out = ops.gn_apply(xn, a, b, relu=False, up=_nhwc(u))
    _cmp('gn+up', out.permute(0, 3, 1, 2), ref, 2e-5, 2e-5)


def test_layout_kernels():
    ops = _ops()
    x = torch.randn((2, 3, 17, 23), generator=torch.Generator().manual_seed(3))
    y = ops.nchw_to_nhwc(x.cuda()).cpu()
    assert torch.equal(y[..., :3], x.permute(0, 2, 3, 1)) and bool((y[..., 3] == 0).all())
    z = torch.randn((2, 19, 21, 70), generator=torch.Generator().manual_seed(4))
    assert torch.equal(ops.nhwc_to_nchw_dense(z.cuda()).cpu(), z.permute(0, 3, 1, 2).contiguous())


def test_missing_gpu_tensor_is_an_error():
    ops = _ops()
    from pointtinybenchmark_amd._lib import CprHipError
    with pytest.raises(CprHipError):
        ops.maxpool3x3s2(torch.zeros((1, 4, 4, 4)))
