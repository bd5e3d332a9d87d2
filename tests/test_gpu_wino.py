"""Fused Winograd F(2x2,3x3) conv (csrc/conv_wino.hip) against an fp64 torch convolution of the same operands, next to the
direct implicit-GEMM kernel on the same inputs.  Stated bound: |wino - fp64| <= 1e-5 * max|fp64| per layer and <= 3x the direct
kernel's own error + 1e-6 (both are printed); the end-to-end 1e-4 logit bar is in test_gpu_fullsize.py."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=['64', '32'])
def wino_tile(request):
    """Every test of this file runs on both fused Winograd forward kernels: '64' = csrc/conv_wino.hip (one 8-wave workgroup per
    CU, 16 x 16 pixel regions), '32' = csrc/conv_wino32.hip (round 4: two 4-wave workgroups per CU, 8 x 16 regions, 4-channel
    chunks).  ops.WINO_TILE forces the instance wherever it can run."""
    from pointtinybenchmark_amd import ops
    keep = ops.WINO_TILE[0]
    ops.WINO_TILE[0] = request.param
    yield request.param
    ops.WINO_TILE[0] = keep


CASES = [  # N, H, W, Cin, Cout, scale/bias, relu
    (2, 32, 32, 64, 64, True, True),
    (1, 48, 40, 256, 128, False, False),     # partial regions on the right edge (40 = 2.5 x 16)
    (2, 16, 16, 32, 64, True, False),        # four chunks only
    (1, 40, 40, 256, 256, True, True),       # layer3 3x3 shape
    (3, 18, 34, 32, 64, False, True),        # one spare row / two spare columns
    (1, 19, 21, 64, 64, True, True),         # odd height and width
    (1, 160, 160, 256, 256, False, False),   # head tower shape
]


def _ref(x, w, scale, bias, relu):
    y = F.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w.double().cpu(), padding=1)
    if scale is not None:
        y = y * scale.double().cpu()[None, :, None, None] + bias.double().cpu()[None, :, None, None]
    if relu:
        y = y.relu()
    return y.permute(0, 2, 3, 1)


@pytest.mark.parametrize('case', CASES)
def test_wino_matches_fp64_conv(case):
    from pointtinybenchmark_amd import ops
    N, H, W, Cin, Cout, affine, relu = case
    g = torch.Generator().manual_seed(H * 1000 + Cin)
    x = torch.randn((N, H, W, Cin), generator=g).cuda()
    w = (torch.randn((Cout, Cin, 3, 3), generator=g) * (2.0 / (9 * Cin)) ** 0.5).cuda()
    scale = (torch.rand(Cout, generator=g) + 0.5).cuda() if affine else None
    bias = torch.randn(Cout, generator=g).cuda() if affine else None
    pc = ops.PackedConv(w, 1, 1)
    assert ops.wino_eligible(pc, H, W) or H * W < 0.6 * ((H + 15) // 16 * 16) * ((W + 15) // 16 * 16)
    y = ops.conv3x3_wino(x, pc, scale, bias, relu)
    ops.WINOGRAD[0] = False
    try:
        yd = ops.conv2d(x, pc, scale, bias, relu=relu)
    finally:
        ops.WINOGRAD[0] = True
    ref = _ref(x, w, scale, bias, relu)
    m = ref.abs().max().item()
    ew = (y.double().cpu() - ref).abs().max().item() / m
    ed = (yd.double().cpu() - ref).abs().max().item() / m
    print('wino %.2e direct %.2e of max' % (ew, ed))
    assert ed <= 5e-6, ed
    assert ew <= 1e-5 and ew <= 3 * ed + 1e-6, (ew, ed)


def test_wino_gn_partials():
    from pointtinybenchmark_amd import ops
    g = torch.Generator().manual_seed(5)
    for (N, H, W, C) in ((2, 32, 48, 64), (1, 40, 24, 128)):
        x = torch.randn((N, H, W, C), generator=g).cuda()
        w = (torch.randn((C, C, 3, 3), generator=g) * 0.05).cuda()
        b = torch.randn(C, generator=g).cuda()
        pc = ops.PackedConv(w, 1, 1)
        y, part = ops.conv3x3_wino(x, pc, None, b, False, gn_part=True)
        P = ops.wino_gn_slots(pc, H, W, False)          # 16 x 16 or 8 x 16 pixel regions, by the kernel that runs
        assert P == (((H + 15) // 16) if ops.WINO_TILE[0] == '64' else ((H + 7) // 8)) * ((W + 15) // 16)
        assert part.shape == (N * P, C, 2)
        s = part.view(N, P, C, 2).double().sum(1)
        yy = y.double()
        torch.testing.assert_close(s[..., 0], yy.sum((1, 2)), rtol=1e-5, atol=1e-3)
        torch.testing.assert_close(s[..., 1], (yy * yy).sum((1, 2)), rtol=1e-5, atol=1e-3)
        # through the dispatcher + gn_finalize == GroupNorm of the output
        gam, bet = torch.rand(C, generator=g).cuda() + 0.5, torch.randn(C, generator=g).cuda()
        y2, part2 = ops.conv2d(x, pc, bias=b, gn_part=True)
        assert torch.equal(y2, y) and part2.shape == part.shape
        a, bb = ops.gn_finalize(part2, gam, bet, N, H * W, 32, 1e-5)
        out = ops.gn_apply(y2, a, bb, relu=False)
        ref = F.group_norm(y.permute(0, 3, 1, 2), 32, gam, bet, 1e-5).permute(0, 2, 3, 1)
        torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)


def test_wino_dispatch_rules():
    from pointtinybenchmark_amd import ops
    w = torch.zeros((64, 64, 3, 3)).cuda()
    assert ops.wino_eligible(ops.PackedConv(w, 1, 1), 160, 160)
    assert ops.wino_eligible(ops.PackedConv(w, 1, 1), 40, 40)
    assert not ops.wino_eligible(ops.PackedConv(w, 1, 1), 20, 20)          # 39 % of its regions
    assert not ops.wino_eligible(ops.PackedConv(w, 2, 1), 160, 160)        # stride 2 stays direct
    assert not ops.wino_eligible(ops.PackedConv(torch.zeros((32, 64, 3, 3)).cuda(), 1, 1), 160, 160)


def _to_b8(x):
    N, H, W, C = x.shape
    return x.view(N, H, W, C // 8, 8).permute(0, 3, 1, 2, 4).contiguous()


def _from_b8(x):
    N, C8, H, W, _ = x.shape
    return x.permute(0, 2, 3, 1, 4).reshape(N, H, W, C8 * 8).contiguous()


@pytest.mark.parametrize('case', [(2, 32, 48, 64, 128), (1, 40, 24, 256, 64), (2, 16, 16, 512, 64)])
def test_wino_blocked_layout_and_fused_affine(case, wino_tile):
    """Channel-blocked input / output and the fused producer-GroupNorm affine (+ReLU) give the SAME bits as the NHWC
    kernel on the materialised input (the arithmetic is identical, only the addressing / where the affine runs differ)."""
    from pointtinybenchmark_amd import ops
    N, H, W, Cin, Cout = case
    if wino_tile == '32' and Cin > 256:
        pytest.skip('the two-workgroups-per-CU kernel keeps the affine table of <= 256 input channels (512-channel layers with a '
                    'fused affine run the other kernel)')
    g = torch.Generator().manual_seed(Cin + W)
    x = torch.randn((N, H, W, Cin), generator=g).cuda()
    w = (torch.randn((Cout, Cin, 3, 3), generator=g) * 0.05).cuda()
    bias = torch.randn(Cout, generator=g).cuda()
    a = (torch.rand((N, Cin), generator=g) + 0.5).cuda()
    b = torch.randn((N, Cin), generator=g).cuda()
    pc = ops.PackedConv(w, 1, 1)
    ref = ops.conv3x3_wino(x, pc, None, bias, True)
    y1 = ops.conv3x3_wino(_to_b8(x), pc, None, bias, True)
    assert torch.equal(y1, ref)
    y2, part2 = ops.conv3x3_wino(_to_b8(x), pc, None, bias, True, gn_part=True, out_b8=True)
    assert ops.is_b8(y2) and torch.equal(_from_b8(y2), ref)
    _, part = ops.conv3x3_wino(x, pc, None, bias, True, gn_part=True)
    assert torch.equal(part, part2)
    assert torch.equal(ops.gn_apply_b8(y2), ref)
    torch.testing.assert_close(ops.gn_apply_b8(y2, a[:, :1].expand(N, Cout).contiguous(), b[:, :1].expand(N, Cout).contiguous(), relu=True),
                               (ref * a[:, :1, None, None].permute(0, 2, 3, 1) + b[:, :1, None, None].permute(0, 2, 3, 1)).relu(),
                               rtol=1e-6, atol=1e-6)
    for relu in (False, True):
        xm = ops.gn_apply(x, a, b, relu=relu)                       # the materialised input: same fma, same max
        refx = ops.conv3x3_wino(xm, pc, None, bias, False)
        for xin in (x, _to_b8(x)):
            y = ops.conv3x3_wino(xin, pc, None, bias, False, in_ab=(a, b), in_relu=relu)
            assert torch.equal(y, refx), (relu, ops.is_b8(xin), float((y - refx).abs().max()))


@pytest.mark.parametrize('case', [(2, 16, 16, 64, 64, False), (1, 20, 40, 128, 64, True), (3, 18, 34, 64, 128, True), (1, 19, 21, 64, 64, True),
                                  (2, 32, 32, 256, 256, False), (5, 16, 16, 256, 256, True), (7, 24, 40, 128, 128, True)])
def test_wino_wgrad_matches_autograd(case):
    """Winograd weight gradient (csrc/conv_wino_wgrad.hip) against fp64 autograd of the same conv, next to the direct kernel.
    Stated bound: 2e-5 of the gradient's largest entry (sums of N*H*W products per entry; measured values are printed).
    The last two cases have K slices that run across image boundaries (the fused affine table changes inside a slice)."""
    from pointtinybenchmark_amd import ops
    N, H, W, Cin, Cout, xf = case
    g = torch.Generator().manual_seed(N * 100 + W)
    x = torch.randn((N, H, W, Cin), generator=g).cuda()
    dy = torch.randn((N, H, W, Cout), generator=g).cuda()
    ab = ((torch.rand((N, Cin), generator=g) + 0.5).cuda(), torch.randn((N, Cin), generator=g).cuda()) if xf else None
    shape = (Cout, Cin, 3, 3)
    gw = ops.conv3x3_wino_wgrad(dy, x, shape, in_ab=ab, in_relu=True)
    ops.WINOGRAD[0] = False
    try:
        gd = ops.conv2d_wgrad(dy, x, shape, 1, 1, in_ab=ab, in_relu=True)
    finally:
        ops.WINOGRAD[0] = True
    xin = x.double().cpu()
    if xf:
        xin = (xin * ab[0].double().cpu()[:, None, None, :] + ab[1].double().cpu()[:, None, None, :]).relu()
    w = torch.zeros(shape, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(xin.permute(0, 3, 1, 2), w, padding=1)
    (y * dy.double().cpu().permute(0, 3, 1, 2)).sum().backward()
    m = w.grad.abs().max().item()
    ew = (gw.double().cpu() - w.grad).abs().max().item() / m
    ed = (gd.double().cpu() - w.grad).abs().max().item() / m
    print('wino wgrad %.2e direct %.2e of max' % (ew, ed))
    assert ed <= 2e-5 and ew <= 2e-5, (ew, ed)
    # accumulate=True adds into an existing gradient
    base = torch.randn(shape, generator=g).cuda()
    acc = base.clone()
    ops.conv3x3_wino_wgrad(dy, x, shape, in_ab=ab, in_relu=True, grad=acc)
    torch.testing.assert_close(acc - base, gw, rtol=1e-4, atol=1e-4 * m)


def test_lds_dma_staged_kernels_are_repeatable_under_memory_pressure():
    """The Winograd forward stages its weight image, and the bf16 256x256 kernel both operands, by LDS-DMA issued from inline
    asm with hand-counted vmcnt waits (csrc/conv_wino.hip, csrc/conv_bf16_dma.hip).  A wrong count or a missing barrier would
    show as a timing-dependent difference, not on every launch: 60 launches of each such kernel, while a second stream saturates
    HBM with copies (latencies of the DMA requests vary by several x), must all be BIT-equal to the first quiet launch."""
    from pointtinybenchmark_amd import ops
    g = torch.Generator().manual_seed(11)
    x = torch.randn((6, 96, 112, 128), generator=g).cuda()
    w = (torch.randn((256, 128, 3, 3), generator=g) * 0.03).cuda()
    a = (torch.rand((6, 128), generator=g) + 0.5).cuda()
    b = torch.randn((6, 128), generator=g).cuda()
    pc = ops.PackedConv(w, 1, 1, torch.float32)
    xb8 = _to_b8(x)
    xh = torch.randn((8, 128, 128, 64), generator=g).bfloat16().cuda()
    wh = (torch.randn((256, 64, 3, 3), generator=g) * 0.04).cuda()
    pch = ops.PackedConv(wh, 1, 1, torch.bfloat16)
    # the streamed 1x1 kernels (csrc/conv1x1_stream.hip: operands AND residual tile by LDS-DMA) and the NT mode of the bf16 kernel
    # under the bf16 weight gradient (csrc/conv_wgrad_bf16.hip) count the same way
    xs = {K: torch.randn((6, 32, 32, K), generator=g).cuda() for K in (64, 128, 256)}
    ws = {K: ops.PackedConv((torch.randn((512, K, 1, 1), generator=g) / K ** 0.5).cuda(), 1, 0) for K in (64, 128, 256)}
    rs = torch.randn((6, 32, 32, 512), generator=g).cuda()
    dyw = (torch.randn((4, 24, 24, 128), generator=g) * 0.1).cuda()
    xw = torch.randn((4, 24, 24, 256), generator=g).bfloat16().cuda()
    runs = {
        'wino plain': lambda: ops.conv3x3_wino(x, pc, gn_part=True),
        'wino blocked + fused affine': lambda: ops.conv3x3_wino(xb8, pc, gn_part=True, in_ab=(a, b), in_relu=True, out_b8=True),
        'bf16 dma': lambda: ops.conv2d(xh, pch, gn_part=True),
        'streamed 1x1 K=64': lambda: (ops.conv1x1_stream(xs[64], ws[64], residual=rs, relu=True),),
        'streamed 1x1 K=128': lambda: (ops.conv1x1_stream(xs[128], ws[128], residual=rs, relu=True),),
        'streamed 1x1 K=256': lambda: (ops.conv1x1_stream(xs[256], ws[256], residual=rs, relu=True),),
        'bf16 weight gradient': lambda: (ops.conv_wgrad_bf16(dyw, xw, (128, 256, 3, 3)),),
    }
    ref = {k: [t.clone() for t in f()] for k, f in runs.items()}
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    src = torch.empty((256 << 20,), dtype=torch.uint8, device='cuda')
    dst = torch.empty_like(src)
    for k, f in runs.items():
        for it in range(60):
            if it % 2 == 0:                      # every other launch runs beside a 256 MB device copy
                with torch.cuda.stream(side):
                    dst.copy_(src)
            out = f()
            torch.cuda.synchronize()
            assert all(torch.equal(o, r) for o, r in zip(out, ref[k])), '%s: launch %d differs from the first one' % (k, it)
