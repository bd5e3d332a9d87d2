"""-m gpu: backward / optimizer kernels through the C-ABI against torch autograd (CPU, fp32) of the same op.
The oracle's functions are plain differentiable torch, so autograd over them is the reference for the gradients the
reference framework would compute (torch autograd over the same modules)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ops():
    from pointtinybenchmark_amd import ops
    return ops


def _nhwc(x):
    return x.detach().permute(0, 2, 3, 1).contiguous().cuda()


def _nchw(x):
    return x.detach().cpu().permute(0, 3, 1, 2)


def _cmp(name, got, ref, atol, rtol=1e-4):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    assert bool((err <= tol).all()), '%s: max abs err %.3e (ref max %.3e), %d/%d over tol' % (
        name, float(err.max()), float(ref.abs().max()), int((err > tol).sum()), err.numel())


GRAD_CASES = [
    # N, Cin, H, W, Cout, k, stride, pad
    (2, 64, 24, 20, 64, 3, 1, 1),
    (2, 256, 16, 16, 256, 3, 1, 1),
    (3, 256, 13, 11, 64, 1, 1, 0),
    (2, 128, 17, 15, 128, 3, 2, 1),
    (2, 256, 14, 14, 512, 1, 2, 0),
    (1, 512, 9, 7, 2048, 1, 1, 0),
    (2, 256, 16, 16, 4, 1, 1, 0),
    (2, 64, 32, 32, 160, 1, 1, 0),
]


@pytest.mark.parametrize('case', GRAD_CASES, ids=lambda c: 'n%d_c%d_%dx%d_o%d_k%d_s%d_p%d' % c)
def test_conv_wgrad_dgrad_vs_autograd(case):
    ops = _ops()
    N, Cin, H, W, Cout, k, stride, pad = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn((N, Cin, H, W), generator=g, requires_grad=True)
    w = (torch.randn((Cout, Cin, k, k), generator=g) / (Cin * k * k) ** 0.5).requires_grad_(True)
    y = F.conv2d(x, w, None, stride, pad)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    gw = ops.conv2d_wgrad(_nhwc(dy), _nhwc(x), w.shape, stride, pad)
    _cmp('wgrad', gw, w.grad, atol=2e-4 * float(w.grad.abs().max()))
    # accumulate flag
    gw2 = ops.conv2d_wgrad(_nhwc(dy), _nhwc(x), w.shape, stride, pad, grad=gw.clone())
    _cmp('wgrad-acc', gw2, 2 * w.grad, atol=4e-4 * float(w.grad.abs().max()))
    if Cout % 32 == 0 or Cout <= 4:
        pt = ops.dgrad_pack(w.detach().cuda(), stride, pad)
        dx = ops.conv2d_dgrad(_nhwc(dy), pt, (H, W), stride)
        _cmp('dgrad', _nchw(dx), x.grad, atol=2e-4 * float(x.grad.abs().max()))


def test_conv_wgrad_fused_gn_input():
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    N, C, H, W = 2, 256, 16, 16
    x = torch.randn((N, C, H, W), generator=g)
    a, b = torch.rand((N, C), generator=g) + 0.5, torch.randn((N, C), generator=g)
    for relu in (True, False):
        z = x * a[:, :, None, None] + b[:, :, None, None]
        if relu:
            z = z.relu()
        w = (torch.randn((256, C, 3, 3), generator=g) / 48).requires_grad_(True)
        y = F.conv2d(z, w, None, 1, 1)
        dy = torch.randn(y.shape, generator=g)
        y.backward(dy)
        gw = ops.conv2d_wgrad(_nhwc(dy), _nhwc(x), w.shape, 1, 1, in_ab=(a.cuda(), b.cuda()), in_relu=relu)
        _cmp('wgrad-xf relu=%s' % relu, gw, w.grad, atol=2e-4 * float(w.grad.abs().max()))


@pytest.mark.parametrize('relu', [True, False])
@pytest.mark.parametrize('shape', [(2, 256, 16, 16), (3, 256, 13, 9), (1, 256, 40, 40)])
def test_group_norm_backward(shape, relu):
    ops = _ops()
    N, C, H, W = shape
    g = torch.Generator().manual_seed(N * H + W)
    x = (torch.randn(shape, generator=g) * 2 + 0.3).requires_grad_(True)
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    beta = torch.randn(C, generator=g).requires_grad_(True)
    y = F.group_norm(x, 32, gamma, beta, 1e-5)
    z = y.relu() if relu else y
    dz = torch.randn(shape, generator=g)
    z.backward(dz)
    xd = _nhwc(x)
    part = ops.gn_stats(xd)
    a, b, mean, rstd = ops.gn_finalize(part, gamma.detach().cuda(), beta.detach().cuda(), N, H * W, 32, 1e-5,
                                       want_stats=True)
    dx, dg, db = ops.gn_bwd(xd, _nhwc(dz), a, b, mean, rstd, gamma.detach().cuda(), relu)
    _cmp('dx', _nchw(dx), x.grad, atol=2e-5 * float(x.grad.abs().max()), rtol=2e-4)
    _cmp('dgamma', dg, gamma.grad, atol=1e-4 * float(gamma.grad.abs().max()))
    _cmp('dbeta', db, beta.grad, atol=1e-4 * float(beta.grad.abs().max()))
    dx2, dg2, db2 = ops.gn_bwd(xd, _nhwc(dz), a, b, mean, rstd, gamma.detach().cuda(), relu, dgamma=dg.clone(),
                               dbeta=db.clone())
    _cmp('dgamma-acc', dg2, 2 * gamma.grad, atol=2e-4 * float(gamma.grad.abs().max()))


@pytest.mark.parametrize('hw', [((8, 8), (16, 16)), ((5, 7), (10, 13)), ((4, 4), (7, 8))])
def test_upsample_add_backward(hw):
    ops = _ops()
    (UH, UW), (H, W) = hw
    g = torch.Generator().manual_seed(UH * W)
    coarse = torch.randn((2, 64, UH, UW), generator=g, requires_grad=True)
    fine = F.interpolate(coarse, size=(H, W), mode='nearest')
    d = torch.randn(fine.shape, generator=g)
    fine.backward(d)
    dc = ops.upsample_add_bwd(_nhwc(d), (2, UH, UW, 64))
    _cmp('dcoarse', _nchw(dc), coarse.grad, atol=1e-5)
    dc2 = ops.upsample_add_bwd(_nhwc(d), dc.clone())
    _cmp('dcoarse-acc', _nchw(dc2), 2 * coarse.grad, atol=2e-5)


def test_bn_fold_relu_backward():
    """Bottleneck tail: y = relu(bn_eval(conv(x)) + identity): g, shortcut grad, dW, dgamma, dbeta."""
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    N, Cin, Cout, H, W = 2, 64, 128, 12, 10
    x = torch.randn((N, Cin, H, W), generator=g)
    idt = torch.randn((N, Cout, H, W), generator=g, requires_grad=True)
    w = (torch.randn((Cout, Cin, 1, 1), generator=g) / 8).requires_grad_(True)
    gamma = (torch.rand(Cout, generator=g) + 0.5).requires_grad_(True)
    gamma.data[3] = 0.0     # zero_init_residual-style channel: no division by the scale anywhere
    beta = torch.randn(Cout, generator=g).requires_grad_(True)
    rm, rv = torch.randn(Cout, generator=g), torch.rand(Cout, generator=g) + 0.5
    y = F.batch_norm(F.conv2d(x, w), rm, rv, gamma, beta, False, 0.0, 1e-5) + idt
    out = y.relu()
    d = torch.randn(out.shape, generator=g)
    out.backward(d)
    inv_sigma = (1.0 / torch.sqrt(rv + 1e-5))
    scale = (gamma.detach() * inv_sigma)
    gten, colsum = ops.relu_bwd_colsum(_nhwc(d), _nhwc(out))
    _cmp('shortcut grad', _nchw(gten), idt.grad, atol=1e-6)
    Gw = ops.conv2d_wgrad(gten, _nhwc(x), w.shape, 1, 0)
    dg, db = ops.bn_fold_bwd(Gw, w.detach().cuda(), scale.cuda(), rm.cuda(), inv_sigma.cuda(), colsum)
    _cmp('dW', Gw, w.grad, atol=2e-4 * float(w.grad.abs().max()))
    _cmp('dgamma', dg, gamma.grad, atol=2e-4 * float(gamma.grad.abs().max()))
    _cmp('dbeta', db, beta.grad, atol=2e-4 * float(beta.grad.abs().max()))
    # round 6: the column sums handed over as a conv epilogue's partials [tiles][C][2] -- bn_fold_bwd adds the column up itself
    parts = torch.randn((37, Cout, 2), generator=g).cuda()
    Gw2 = ops.conv2d_wgrad(gten, _nhwc(x), w.shape, 1, 0)
    Gw3 = Gw2.clone()
    tp = ops.TilePartials(parts, 37, Cout)
    dg2, db2 = ops.bn_fold_bwd(Gw2, w.detach().cuda(), scale.cuda(), rm.cuda(), inv_sigma.cuda(), tp)
    dg3, db3 = ops.bn_fold_bwd(Gw3, w.detach().cuda(), scale.cuda(), rm.cuda(), inv_sigma.cuda(), tp.reduce())
    assert torch.equal(Gw2, Gw3)
    _cmp('dbeta from partials', db2, db3, atol=1e-5 * float(db3.abs().max()))
    _cmp('dgamma from partials', dg2, dg3, atol=1e-5 * float(dg3.abs().max()))
    # colsum without a mask / without writing g, odd channel count for the thread layout (C/4 = 40)
    for C in (160, 2048, 1028):
        t = torch.randn((3, 7, 5, C), generator=g)
        _, cs = ops.relu_bwd_colsum(t.cuda(), None, want_g=False)
        _cmp('colsum C=%d' % C, cs, t.sum((0, 1, 2)), atol=1e-4)
        gt, cs = ops.relu_bwd_colsum(t.cuda(), (t * 0 + torch.randn(t.shape, generator=g)).cuda())
        assert float((gt != 0).float().mean()) > 0.3
        # round 6: a second gradient of the same tensor (the shortcut's) joins before the mask; fp32 or bf16 mask source; the bf16
        # rounding of the result written by the same pass -- bit-equal to the torch expression
        a2, y2 = torch.randn(t.shape, generator=g).cuda(), torch.randn(t.shape, generator=g).clamp_min(0)
        ref = (t.cuda() + a2) * (y2.cuda() > 0)
        for ymap in (y2.cuda(), y2.bfloat16().cuda()):
            gt, cs, g16 = ops.relu_bwd_colsum(t.cuda(), ymap, want16=True, add=a2)
            assert torch.equal(gt, ref) and torch.equal(g16, ref.to(torch.bfloat16))
            _cmp('colsum with add C=%d' % C, cs, ref.double().sum((0, 1, 2)).float(), atol=1e-4)
        gt, cs = ops.relu_bwd_colsum(t.cuda(), None, add=a2)
        assert torch.equal(gt, t.cuda() + a2)


def test_sgd_clip_vs_torch():
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    ps = [torch.randn(n, generator=g) for n in (1000, 77777, 5)]
    ref = [p.clone().requires_grad_(True) for p in ps]
    opt = torch.optim.SGD(ref, lr=0.01, momentum=0.9, weight_decay=1e-4)
    dev = [p.clone().cuda() for p in ps]
    bufs = [torch.zeros_like(p) for p in dev]
    norm2 = torch.zeros(1, dtype=torch.float64, device='cuda')
    ws = torch.empty(1024, dtype=torch.float64, device='cuda')
    for step in range(3):
        grads = [torch.randn(p.shape, generator=g) * (10.0 if step == 1 else 0.01) for p in ps]
        for r, gr in zip(ref, grads):
            r.grad = gr.clone()
        tn = torch.nn.utils.clip_grad_norm_(ref, max_norm=35, norm_type=2)
        opt.step()
        gd = [gr.cuda() for gr in grads]
        for i, gr in enumerate(gd):
            ops.grad_sumsq(gr, norm2, ws, accumulate=i > 0)
        assert abs(float(norm2.sqrt()) - float(tn)) <= 1e-5 * float(tn)
        for p, gr, b in zip(dev, gd, bufs):
            ops.sgd_step(p, gr, b, norm2, 0.01, 0.9, 1e-4, 35.0, 1.0, first=(step == 0))
        for p, r in zip(dev, ref):
            _cmp('sgd step %d' % step, p, r, atol=1e-6, rtol=1e-5)


@pytest.mark.parametrize('case', [(2, 128, 17, 15, 64, 3, 1, 1), (2, 256, 14, 14, 512, 1, 2, 0), (2, 128, 16, 16, 128, 3, 2, 1),
                                  (4, 64, 40, 40, 256, 1, 1, 0)], ids=lambda c: 'n%d_c%d_%dx%d_o%d_k%d_s%d_p%d' % c)
def test_dgrad_fused_relu_mask_add_colsum(case):
    """The data-gradient conv with the consumer-side ReLU backward, a second gradient and the column sums in its epilogue."""
    ops = _ops()
    N, Cin, H, W, Cout, k, stride, pad = case
    g = torch.Generator().manual_seed(sum(case) + 1)
    pre = torch.randn((N, Cin, H, W), generator=g, requires_grad=True)
    x = pre.relu()
    w = (torch.randn((Cout, Cin, k, k), generator=g) / (Cin * k * k) ** 0.5)
    y = F.conv2d(x, w, None, stride, pad)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    pt = ops.dgrad_pack(w.cuda(), stride, pad)
    dpre, cs = ops.conv2d_dgrad(_nhwc(dy), pt, (H, W), stride, mask=_nhwc(x), colsum=True)
    sc = float(pre.grad.abs().max())
    _cmp('masked dgrad', _nchw(dpre), pre.grad, atol=2e-4 * sc)
    cs = cs.reduce() if hasattr(cs, 'reduce') else cs        # strided layers return the reduced vector directly
    _cmp('colsum', cs, pre.grad.sum((0, 2, 3)), atol=2e-4 * sc * (N * H * W) ** 0.5)
    other = torch.randn((N, Cin, H, W), generator=g)
    x2 = x.detach().clone().requires_grad_(True)
    F.conv2d(x2, w, None, stride, pad).backward(dy)
    dsum = ops.conv2d_dgrad(_nhwc(dy), pt, (H, W), stride, add=_nhwc(other))
    _cmp('dgrad + add', _nchw(dsum), x2.grad + other, atol=2e-4 * sc)


@pytest.mark.parametrize('case', [(2, 128, 17, 15, 64, 3, 2, 1), (2, 256, 14, 14, 512, 1, 2, 0), (1, 64, 9, 12, 32, 3, 2, 1)],
                         ids=lambda c: 'n%d_c%d_%dx%d_o%d_k%d_s%d_p%d' % c)
def test_strided_dgrad_phased_equals_zero_insertion(case):
    """The two forms of the stride-2 data gradient (per-parity sub-convolutions vs conv over the dilated gradient)."""
    ops = _ops()
    N, Cin, H, W, Cout, k, stride, pad = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn((N, Cin, H, W), generator=g, requires_grad=True)
    w = torch.randn((Cout, Cin, k, k), generator=g) / (Cin * k * k) ** 0.5
    scale = torch.rand(Cout, generator=g) + 0.5
    y = F.conv2d(x, w * scale[:, None, None, None], None, stride, pad)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    outs = []
    for phased in (True, False):
        ops._PHASED[0] = phased
        try:
            pt = ops.dgrad_pack(w.cuda(), stride, pad, scale=scale.cuda())
            assert isinstance(pt, ops.PhasedDgrad) == phased
            outs.append(ops.conv2d_dgrad(_nhwc(dy), pt, (H, W), stride))
        finally:
            ops._PHASED[0] = True
    sc = float(x.grad.abs().max())
    _cmp('phased', _nchw(outs[0]), x.grad, atol=2e-4 * sc)
    _cmp('zero-insert', _nchw(outs[1]), x.grad, atol=2e-4 * sc)


@pytest.mark.parametrize('case', [(2, 128, 34, 30, 128, 3, 2, 1), (2, 256, 28, 28, 512, 1, 2, 0), (3, 64, 18, 24, 64, 3, 2, 1)],
                         ids=lambda c: 'n%d_c%d_%dx%d_o%d_k%d_s%d_p%d' % c)
def test_strided_dgrad_phased_bf16_vs_fp64(case):
    """Round 6 (mixed-precision step): the four parity sub-convolutions of the stride-2 data gradient on the bf16 matrix pipe
    (PhasedDgrad(dtype=bf16): bf16 gradient map, bf16 sub-kernel packs with the folded-BN scale multiplied in before the rounding,
    fp32 accumulators and fp32 scatter).  Against torch's fp64 data gradient of the SAME bf16 operands only the fp32 summation
    order differs (2e-4 of the max); against the unrounded operands bf16's 8 bits show (3e-2 relative L2).  With ``add``."""
    ops = _ops()
    N, Cin, H, W, Cout, k, stride, pad = case
    g = torch.Generator().manual_seed(sum(case))
    w = torch.randn((Cout, Cin, k, k), generator=g) / (Cin * k * k) ** 0.5
    scale = torch.rand(Cout, generator=g) + 0.5
    ws = (w * scale[:, None, None, None])
    x = torch.zeros((N, Cin, H, W), dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x, ws.bfloat16().double(), None, stride, pad)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.bfloat16().double())
    ref = x.grad
    pt = ops.dgrad_pack(w.cuda(), stride, pad, scale=scale.cuda(), dtype=torch.bfloat16)
    assert isinstance(pt, ops.PhasedDgrad) and pt.dtype == torch.bfloat16
    other = torch.randn((N, H, W, Cin), generator=g).cuda()
    got = pt(_nhwc(dy).bfloat16(), (H, W))
    got2 = pt(_nhwc(dy).bfloat16(), (H, W), add=other)
    torch.cuda.synchronize()
    sc = float(ref.abs().max())
    _cmp('bf16 phased vs fp64 on the rounded operands', _nchw(got).double(), ref, atol=2e-4 * sc)
    assert float((got2 - other - got).abs().max()) <= 1e-5 * sc
    x2 = torch.zeros((N, Cin, H, W), dtype=torch.float64, requires_grad=True)
    F.conv2d(x2, ws.double(), None, stride, pad).backward(dy.double())
    rel = float((_nchw(got).double().cpu() - x2.grad).norm() / x2.grad.norm())
    assert rel <= 3e-2, rel
