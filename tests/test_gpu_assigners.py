"""-m gpu: assigner indices bit-exact against the reference fixtures, the reference's own known-answer
vectors (T/tests/test_utils/test_assigner.py:155-194) and the CPU oracle on extra seeds."""
import os

import numpy as np
import pytest
import torch

from oracle import cpr_oracle as O
from oracle.gen_golden import assigner_inputs
from tests.test_oracle_golden import _pa_inputs

pytestmark = pytest.mark.gpu


def test_point_assigner_reference_known_answers():
    from pointtinybenchmark_amd import PointAssigner
    a = PointAssigner()
    pts = torch.FloatTensor([[0, 0, 1], [10, 10, 1], [5, 5, 1], [32, 32, 1]]).cuda()
    gtb = torch.FloatTensor([[0, 0, 10, 9], [0, 10, 10, 19]]).cuda()
    res = a.assign(pts, gtb)
    assert res.gt_inds.cpu().tolist() == [1, 2, 1, 0]
    res = a.assign(pts, torch.zeros((0, 4)).cuda())          # empty gt -> all background
    assert res.gt_inds.cpu().tolist() == [0, 0, 0, 0]
    res = a.assign(torch.zeros((0, 3)).cuda(), torch.zeros((0, 4)).cuda())
    assert len(res.gt_inds) == 0


@pytest.mark.parametrize('seed', range(6))
def test_point_assigner_vs_reference_fixture(golden_dir, seed):
    from pointtinybenchmark_amd import PointAssigner
    g = np.load(os.path.join(golden_dir, 'assigners.npz'))
    pts, gtb, gl = _pa_inputs(seed)
    res = PointAssigner(scale=4, pos_num=3).assign(pts.cuda(), gtb.cuda(), None, gl.cuda())
    assert np.array_equal(res.gt_inds.cpu().numpy(), g['pa%d_gt_inds' % seed])
    assert np.array_equal(res.labels.cpu().numpy(), g['pa%d_labels' % seed])


def _ha(k):
    from pointtinybenchmark_amd import HungarianAssignerV2
    return HungarianAssignerV2(cls_costs=dict(type='FocalLossCost', weight=2.0),
                               reg_costs=dict(type='DisCostV2', weight=0.1, norm_with_img_wh=False), topk_k=k)


def _ulp_diff(a, b):
    """|a - b| in units of the last place for same-sign finite fp32 arrays (0 where bit-identical)."""
    ia, ib = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
    return np.abs(ia - ib)


def _check_hungarian(pred, logits, gt, labels, shp, k, record_property, tag, ref_cost=None, ref_inds=None):
    """The a13 gate.  The reference's cost bits depend on the HOST it runs on (its fp32 log is MKL VML, whose last bit
    differs between the Xeon build host and the EPYC GPU-box host: profiles/round2_log_probe.txt), so parity is stated as
    a chain of exact links plus a measured residual:
      1. device cost == the reference formula with a correctly rounded log, BIT FOR BIT (host independent);
      2. device indices == scipy.linear_sum_assignment run on the device's own cost matrix, BIT FOR BIT, always;
      3. device LSA on the reference's own cost bits == the reference's indices, BIT FOR BIT (fixtures only);
      4. residual vs what the reference computes on a given host: a 1-ulp difference of either log term moves a cost
         entry by <= 2^-23 * |2 * term| (many ulps of the entry when pos - neg cancels) plus one rounding of the sum, so the
         bound is |diff| <= 1e-6 + 2^-23 |entry|; the number of differing entries / indices is recorded."""
    ha = _ha(k)
    meta = dict(img_shape=shp)
    costT = ha.cost_t(pred.cuda(), logits.cuda(), gt.cuda(), labels.cuda(), meta)
    dev_cost = costT.t().contiguous().cpu()
    got = ha.assign(pred.cuda(), logits.cuda(), gt.cuda(), labels.cuda(), meta).gt_inds.cpu()
    # 1. cost bits against the correctly-rounded-log statement of the reference formula
    inds_cr, _, cost_cr = O.hungarian_assign_v2(pred, logits, gt, labels, shp, topk_k=k, log_mode='cr')
    n_cr = int((dev_cost != cost_cr).sum())
    assert n_cr == 0, '%s: %d of %d device cost entries differ from the correctly-rounded-log formula' % (
        tag, n_cr, dev_cost.numel())
    # 2. the device LSA is scipy on the device's own costs, tie for tie
    own = O.lsa_topk(dev_cost, k)
    n_own = int((got != own).sum())
    assert n_own == 0, '%s: %d indices differ from scipy on the device cost matrix' % (tag, n_own)
    assert torch.equal(got, inds_cr)
    # 4. residual against the host-dependent reference bits
    if ref_cost is None:
        ref_inds_t, _, ref_cost_t = O.hungarian_assign_v2(pred, logits, gt, labels, shp, topk_k=k)   # this host's log
        ref_cost, ref_inds = ref_cost_t.numpy(), ref_inds_t.numpy()
    ulp = _ulp_diff(dev_cost.numpy(), ref_cost)
    n_ent, n_idx = int((ulp > 0).sum()), int((got.numpy() != ref_inds).sum())
    record_property(tag + '_n_diff_cost_entries', n_ent)
    record_property(tag + '_n_diff_indices', n_idx)
    _RESIDUALS.append(dict(case=tag, entries=int(ulp.size), n_diff_cost_entries=n_ent, max_ulp=int(ulp.max()),
                           n_diff_indices=n_idx, positives=int((got > 0).sum())))
    aerr = np.abs(dev_cost.numpy().astype(np.float64) - ref_cost.astype(np.float64))
    worst = float((aerr - 2.0 ** -23 * np.abs(ref_cost.astype(np.float64))).max())
    assert worst <= 1e-6, '%s: a cost entry is %.3e (beyond one rounding) from the reference' % (tag, worst)
    assert n_ent <= 0.08 * ulp.size, '%s: %d of %d cost entries differ' % (tag, n_ent, ulp.size)
    return got, n_ent, n_idx


_RESIDUALS = []


@pytest.fixture(scope='module', autouse=True)
def _dump_residuals():
    """gpurun_out/assigner_residuals.json: the measured host-log residual per case (pulled back by the driver)."""
    yield
    if _RESIDUALS:
        import json
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, 'assigner_residuals.json'), 'w') as f:
            json.dump(_RESIDUALS, f, indent=1)


@pytest.mark.parametrize('case', range(5))
def test_hungarian_v2_vs_reference_fixture(golden_dir, case, record_property):
    """Fixtures = the reference's own HungarianAssignerV2 run on the build host (indices AND its fp32 cost matrix)."""
    from pointtinybenchmark_amd import ops
    g = np.load(os.path.join(golden_dir, 'assigners.npz'))
    gc = np.load(os.path.join(golden_dir, 'assigner_costs.npz'))
    n_side, G, C, k = [int(v) for v in g['ha%d_cfg' % case]]
    pred, logits, gt, labels, shp = assigner_inputs(200 + case, n_side, 4, G, C)
    ref_cost, ref_inds = gc['ha%d_cost' % case], g['ha%d_gt_inds' % case]
    # 3. the device LSA on the reference's own cost bits reproduces the reference's indices, unconditionally
    (on_ref,), status = ops.lsa_topk([torch.from_numpy(ref_cost).t().contiguous().cuda()], k)
    assert int(status[0]) == 0
    assert np.array_equal(on_ref.cpu().numpy(), ref_inds), '%d indices differ on identical cost bits' % int(
        (on_ref.cpu().numpy() != ref_inds).sum())
    got, n_ent, n_idx = _check_hungarian(pred, logits, gt, labels, shp, k, record_property, 'ha%d' % case, ref_cost, ref_inds)
    # the build host's MKL log is within 2 entries in 819 200 of the correctly rounded one on these inputs, and none of
    # those touches a tie: the device indices ARE the reference's
    assert n_idx == 0, 'ha%d: %d indices differ from the reference fixture (%d cost entries 1 ulp apart)' % (case, n_idx, n_ent)
    labels_ref = g['ha%d_labels' % case]
    res = _ha(k).assign(pred.cuda(), logits.cuda(), gt.cuda(), labels.cuda(), dict(img_shape=shp))
    assert np.array_equal(res.labels.cpu().numpy(), labels_ref)


@pytest.mark.parametrize('case', range(3))
def test_hungarian_v2_fewer_proposals_than_gts_topk1(golden_dir, case):
    """topk_k == 1 with fewer proposals than gts (hungarian_assigner.py:229-240): scipy solves the problem with the proposals
    as rows; fixtures = the reference's own HungarianAssignerV2 (oracle/gen_golden_r2.py ha_transposed)."""
    from oracle.gen_golden_r2 import HA_T_CASES
    g = np.load(os.path.join(golden_dir, 'assigner_transposed.npz'))
    n_side, G, C = HA_T_CASES[case]
    pred, logits, gt, labels, shp = assigner_inputs(300 + case, n_side, 4, G, C)
    res = _ha(1).assign(pred.cuda(), logits.cuda(), gt.cuda(), labels.cuda(), dict(img_shape=shp))
    inds = res.gt_inds.cpu().numpy()
    assert np.array_equal(inds, g['hat%d_gt_inds' % case]), (inds, g['hat%d_gt_inds' % case])
    assert np.array_equal(res.labels.cpu().numpy(), g['hat%d_labels' % case])
    assert len(set(inds.tolist())) == n_side * n_side and inds.min() >= 1          # every proposal a distinct gt
    # scipy on the device's own cost matrix (the unconditional link of the parity chain, as for the ordinary orientation)
    from scipy.optimize import linear_sum_assignment
    cost = _ha(1).cost_t(pred.cuda(), logits.cuda(), gt.cuda(), labels.cuda(), dict(img_shape=shp)).t().cpu().numpy()
    r, c = linear_sum_assignment(cost)
    want = np.zeros(n_side * n_side, dtype=np.int64)
    want[r] = c + 1
    assert np.array_equal(inds, want)


@pytest.mark.parametrize('seed', range(8))
def test_hungarian_v2_vs_oracle_seeds(seed, record_property):
    """Extra seeds against the oracle executed on THIS host (the GPU box's CPU: its MKL log differs from the correctly
    rounded one in ~2-5 % of the cost entries, so links 1-2 carry the proof and the residual is recorded)."""
    n_side, G, C, k = [(48, 12, 1, 5), (64, 30, 1, 5), (40, 5, 2, 3), (30, 64, 1, 5), (100, 40, 1, 5),
                       (20, 3, 1, 1), (160, 100, 1, 5), (56, 17, 5, 4)][seed]
    pred, logits, gt, labels, shp = assigner_inputs(900 + seed, n_side, 4, G, C)
    got, n_ent, n_idx = _check_hungarian(pred, logits, gt, labels, shp, k, record_property, 'seed%d' % seed)
    assert int((got > 0).sum()) == min(k, (n_side * n_side) // G) * G
    # north star: "assigner indices bit-identical".  Links 1-2 above are host independent; THIS comparison is against the
    # reference formula evaluated with this host's own fp32 log (MKL VML: 1 ulp off the correctly rounded value in up to a
    # few per cent of the arguments on EPYC hosts).  A flipped index here means one of those last-bit differences landed on
    # an exact tie of the assignment -- not a kernel defect, but the claim "bit-identical to the reference CPU path on this
    # box" would then be false for this input, so it fails loudly instead of being written to a JSON nobody reads.
    assert n_idx == 0, ('seed%d: %d assigner indices differ from the reference formula evaluated on THIS host (%d of %d cost '
                        'entries differ by one rounding of the host log); the device indices still equal scipy on the '
                        "device's own correctly-rounded-log costs" % (seed, n_idx, n_ent, n_side * n_side * G))


@pytest.mark.parametrize('case', range(5))
def test_device_lsa_reproduces_scipy_on_identical_costs(golden_dir, case):
    """Same fp32 cost matrix in -> bit-identical indices out, INCLUDING exactly tied optima (the L1 distance cost
    makes them common): the kernel follows scipy's scan order and tie rules."""
    from pointtinybenchmark_amd import ops
    g = np.load(os.path.join(golden_dir, 'assigners.npz'))
    n_side, G, C, k = [int(v) for v in g['ha%d_cfg' % case]]
    pred, logits, gt, labels, shp = assigner_inputs(200 + case, n_side, 4, G, C)
    inds, lab, cost = O.hungarian_assign_v2(pred, logits, gt, labels, shp, topk_k=k)
    (got,), status = ops.lsa_topk([cost.t().contiguous().cuda()], k)
    assert int(status[0]) == 0
    assert torch.equal(got.cpu(), inds), '%d indices differ from scipy run on the same cost matrix' % int(
        (got.cpu() != inds).sum())


def _lsa_both_kernels(costs, k):
    """ops.lsa_topk on the register-resident kernel (round 4) and on the memory-resident one: outputs of both."""
    from pointtinybenchmark_amd import ops
    outs = []
    for reg in (True, False):
        ops.LSA_REGISTER_KERNEL[0] = reg
        try:
            o, status = ops.lsa_topk(costs, k)
        finally:
            ops.LSA_REGISTER_KERNEL[0] = True
        assert int(status.abs().max()) == 0, status
        outs.append([t.cpu() for t in o])
    return outs


@pytest.mark.parametrize('case', ['25600x100_k5', '25600x32_k5_batch4', 'ties', 'tiny', '32768x7_k3', '9000x255_k2'])
def test_register_resident_lsa_equals_the_memory_resident_kernel_and_scipy(case):
    """csrc/assign.hip holds two statements of scipy's shortest-augmenting-path solver: ``lsa_topk_kernel`` (all per-column
    state in memory) and ``lsa_topk_reg_kernel`` (round 4: the search state of a row in registers / LDS).  Same arithmetic,
    same comparison order, same tie rules: indices must agree BIT FOR BIT with each other and with scipy (oracle.lsa_topk)
    on the same cost matrix -- full-size problems, a batch, integer costs full of exactly tied optima, and degenerate sizes."""
    g = torch.Generator().manual_seed(sum(map(ord, case)))
    if case == '25600x100_k5':
        specs, k = [(160, 100)], 5
    elif case == '25600x32_k5_batch4':
        specs, k = [(160, 32), (160, 17), (160, 32), (128, 40)], 5
    elif case == 'tiny':
        specs, k = [(2, 3), (3, 1), (1, 1), (4, 16)], 2            # one round only (4 // 3 == 1); one gt; 1 x 1; M == G
    elif case == '32768x7_k3':
        specs, k = [(None, 7)], 3                                   # the widest problem the register kernel takes
    elif case == '9000x255_k2':
        specs, k = [(None, 255)], 2
    else:
        specs, k = [(40, 12), (64, 30)], 4
    costs, refs = [], []
    for si, (n_side, G) in enumerate(specs):
        if case == 'ties':            # small integer costs: almost every optimum is tied, the tie rules decide every index
            M = n_side * n_side
            cost = torch.randint(0, 4, (M, G), generator=g).float()
        elif n_side is None:
            M = 32768 if G == 7 else 9000
            cost = torch.rand((M, G), generator=g) * 3 + (torch.randint(0, 3, (M, G), generator=g).float())
        else:
            pred, logits, gt, labels, shp = assigner_inputs(1300 + 10 * si + len(case), n_side, 4, G, 1)
            _, _, cost = O.hungarian_assign_v2(pred, logits, gt, labels, shp, topk_k=k, log_mode='cr')
            if cost is None:
                cost = torch.rand((n_side * n_side, G), generator=g)
        costs.append(cost.t().contiguous().cuda())
        if cost.shape[0] == cost.shape[1]:
            # ops.lsa_topk's contract: the gts are the solver's rows.  scipy keeps the PROPOSALS as rows of a square matrix, so
            # among exactly tied optima it may return another one -- HungarianAssignerV2.assign routes square problems through
            # transposed_inds for that reason (checked below); the kernel-level reference is scipy on the transpose
            from scipy.optimize import linear_sum_assignment
            r, c = linear_sum_assignment(cost.t().numpy())
            want = torch.zeros(cost.shape[0], dtype=torch.long)
            want[torch.from_numpy(c)] = torch.from_numpy(r) + 1
            refs.append(want)
        else:
            refs.append(O.lsa_topk(cost, k))
    reg, mem = _lsa_both_kernels(costs, k)
    for i, (a, b, r) in enumerate(zip(reg, mem, refs)):
        assert torch.equal(a, b), '%s[%d]: %d indices differ between the two kernels' % (case, i, int((a != b).sum()))
        assert torch.equal(a, r), '%s[%d]: %d indices differ from scipy' % (case, i, int((a != r).sum()))


def test_square_problem_with_tied_optima_follows_scipy_orientation():
    """As many proposals as gts and a cost matrix whose columns are identical (every assignment is optimal): the reference
    hands scipy the (M, G) matrix, which keeps the proposals as rows for a square problem -- the assigner must return THAT
    optimum (here the identity), through the single-image and the batched entry."""
    ha = _ha(5)
    pred, logits, gt, labels, shp = assigner_inputs(1334, 4, 4, 16, 1)
    gt = gt[:1].repeat(16, 1)                       # 16 identical gts: identical cost columns
    want, _, cost = O.hungarian_assign_v2(pred, logits, gt, labels, shp, topk_k=5, log_mode='cr')
    assert bool((cost == cost[:, :1]).all()) and sorted(want.tolist()) == list(range(1, 17))
    res = ha.assign(pred.cuda(), logits.cuda(), gt.cuda(), labels.cuda(), dict(img_shape=shp))
    assert torch.equal(res.gt_inds.cpu(), want), (res.gt_inds.cpu(), want)


def test_device_lsa_batched_problems():
    from pointtinybenchmark_amd import ops
    costs, refs = [], []
    for seed, (n_side, G, k) in enumerate([(20, 9, 3), (33, 40, 3), (16, 5, 3)]):
        pred, logits, gt, labels, shp = assigner_inputs(700 + seed, n_side, 4, G, 1)
        inds, _, cost = O.hungarian_assign_v2(pred, logits, gt, labels, shp, topk_k=3)
        costs.append(cost.t().contiguous().cuda())
        refs.append(inds)
    outs, status = ops.lsa_topk(costs, 3)
    for o, r in zip(outs, refs):
        assert torch.equal(o.cpu(), r)


def test_hungarian_edge_cases():
    ha = _ha(5)
    pred, logits, gt, labels, shp = assigner_inputs(1, 8, 4, 3, 1)
    res = ha.assign(pred.cuda(), logits.cuda(), torch.zeros((0, 2)).cuda(), torch.zeros((0,), dtype=torch.long).cuda(),
                    dict(img_shape=shp))
    assert bool((res.gt_inds == 0).all()) and bool((res.labels == -1).all())
    res = ha.assign(pred[:0].cuda(), logits[:0].cuda(), gt.cuda(), labels.cuda(), dict(img_shape=shp))
    assert len(res.gt_inds) == 0
