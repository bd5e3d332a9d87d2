"""-m gpu: the streaming logit projection (csrc/project.hip) against the MFMA conv path and torch, and the hipGraph replay
of BasicLocator.forward_train against the eager path."""
import pytest
import torch

from pointtinybenchmark_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('J,hw,relu,affine', [(2, (160, 160), True, True), (1, (40, 56), True, True), (6, (64, 48), True, True),
                                              (8, (32, 32), False, True), (4, (24, 40), True, False), (3, (5, 7), True, True)])
def test_logit_project_vs_torch(J, hw, relu, affine):
    from pointtinybenchmark_amd import ops
    g = torch.Generator().manual_seed(J)
    N, (H, W), C = 3, hw, 256
    x = torch.randn((N, H, W, C), generator=g)
    w = torch.randn((J, C), generator=g) * 0.05
    b = torch.randn((J,), generator=g)
    a_, b_ = torch.rand((N, C), generator=g) + 0.5, torch.randn((N, C), generator=g) * 0.3
    out = ops.logit_project(x.cuda(), w.cuda(), b.cuda(), (a_.cuda(), b_.cuda()) if affine else None, in_relu=relu)
    assert out is not None and out.shape == (N, H, W, J)
    t = x.double()
    if affine:
        t = t * a_.double()[:, None, None, :] + b_.double()[:, None, None, :]
    if relu and affine:
        t = t.clamp_min(0)
    ref = t @ w.double().t() + b.double()
    err = float((out.cpu().double() - ref).abs().max())
    assert err <= 2e-5 * max(1.0, float(ref.abs().max())), err
    # the MFMA conv path it replaces gives the same map to fp32 rounding
    pc = ops.PackedConv(w.cuda()[:, :, None, None], 1, 0)
    if H * W % 128 == 0 or not affine:
        conv = ops.conv2d(x.cuda(), pc, bias=b.cuda(), in_ab=(a_.cuda(), b_.cuda()) if affine else None, in_relu=relu)
        assert float((conv - out).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))


def test_logit_project_declines_what_it_does_not_cover():
    from pointtinybenchmark_amd import ops
    x = torch.randn((1, 8, 8, 256)).cuda()
    assert ops.logit_project(x, torch.randn((160, 256)).cuda(), torch.zeros(160).cuda()) is None       # C = 80: MFMA path
    assert ops.logit_project(torch.randn((1, 8, 8, 96)).cuda(), torch.randn((2, 96)).cuda(), torch.zeros(2).cuda()) is None


def test_cat_rows_rejoins_split_views_without_a_copy():
    from pointtinybenchmark_amd.dense_heads.cpr_head import cat_rows
    t = torch.arange(40, dtype=torch.float32).reshape(10, 4).cuda()
    parts = list(torch.split(t, [3, 1, 4, 2]))
    j = cat_rows(parts)
    assert j.data_ptr() == t.data_ptr() and torch.equal(j, t)             # a view of the original storage
    mixed = [parts[0], parts[2]]                                           # not adjacent -> a real cat
    assert torch.equal(cat_rows(mixed), torch.cat(mixed)) and cat_rows(mixed).data_ptr() != t.data_ptr()
    lab = list(torch.split(torch.arange(9).cuda(), [2, 0, 7]))            # an empty image in the middle
    assert torch.equal(cat_rows(lab), torch.arange(9).cuda())
    assert torch.equal(cat_rows([t[:2].clone(), t[2:5].clone()]), t[:5])
