"""Mixed-precision training step (bf16 recorded forward, fp32 backward) against the fp32 step on the same weights and batch:
losses, cosine of the full gradient, relative L2 per large tensor (measurement tool; the bars live in
tests/test_gpu_train_step.py::test_mixed_precision_step_tracks_the_fp32_step; lives under tests/ because it shares the parity tests'
case table and model builder).
  python tests/report_mixed_precision_grads.py            the two small cases of the test
  DECOMPOSE=1 python tests/report_mixed_precision_grads.py   round 6: where the error comes from at the gate's shape (decompose())"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.gen_golden import CPR_CASES  # noqa: E402  (case definitions only)
from pointtinybenchmark_amd import synthetic  # noqa: E402
from pointtinybenchmark_amd.training import CprTrainer  # noqa: E402
from tests.test_gpu_cpr_parity import build_hip_locator, to_cuda  # noqa: E402

for name in (() if os.environ.get('DECOMPOSE', '0') == '1' else ('cpr_r18_c3_128', 'cpr_r50_c1_160_spread')):
    cfg = CPR_CASES[name]
    m, _ = build_hip_locator(cfg)
    batch = synthetic.synthetic_batch(cfg['batch'], cfg['height'], cfg['width'], cfg['num_gts'], cfg['num_classes'], cfg['seed'],
                                      cfg.get('ragged', False))
    cb = to_cuda(batch)
    tr = CprTrainer(m, lr=1e-3)
    l32 = {k: float(v) for k, v in tr.forward_backward(cb['img'], cb['img_metas'], cb['gt_bboxes'], cb['gt_labels']).items()}
    g32 = tr.flat_g.clone().double()
    m.set_compute_dtype('bf16')
    l16 = {k: float(v) for k, v in tr.forward_backward(cb['img'], cb['img_metas'], cb['gt_bboxes'], cb['gt_labels']).items()}
    g16 = tr.flat_g.clone().double()
    cos = float(torch.dot(g16, g32) / (g16.norm() * g32.norm()))
    gmax = max(float(p.grad.norm()) for p in tr.params)
    rels, off = [], 0
    for p in tr.params:
        n = p.numel()
        a, b = g16[off:off + n], g32[off:off + n]
        off += n
        if float(b.norm()) >= 1e-2 * gmax:
            rels.append(float((a - b).norm() / b.norm()))
    print('%s: losses fp32 %s | bf16 forward %s' % (name, {k: round(v, 5) for k, v in l32.items()}, {k: round(v, 5) for k, v in l16.items()}))
    print('   gradient: cosine %.5f, |g16| / |g32| %.4f, relative L2 over the %d large tensors: median %.4f max %.4f'
          % (cos, float(g16.norm() / g32.norm()), len(rels), sorted(rels)[len(rels) // 2], max(rels)))


def decompose(depth=50, size=640, batch=2, warm_steps=3):
    """Round 6: WHERE the mixed-precision gradient error comes from, at the shape and state of bench.py's gradient gate (R50 640^2,
    B = 2, weights a few optimizer steps away from the seeded initialisation).  Against the fp32 HIP step on the same weights:
      A  bf16 forward + bf16 weight / data gradients (the product's mixed-precision step)
      B  bf16 forward + fp32 backward             (training.MIXED_BF16: wgrad = dgrad = False)
      Bw bf16 forward + bf16 weight gradients only,  Bd bf16 forward + bf16 data gradients only
      C  fp32 forward + bf16 backward rules        (MIXED_BF16['force'])
    per block (all tensors of a block concatenated) and for the worst tensors; then layer3.0.conv2's stride-2 data gradient alone
    (fp32 phase kernel vs an fp64 transposed convolution on the same inputs, and the same with the gradient map rounded to bf16)."""
    import bench
    import pointtinybenchmark_amd as P
    from pointtinybenchmark_amd import ops, training
    import torch.nn.functional as F
    m = P.build_detector(bench.model_cfg(depth=depth)).cuda()
    m.load_state_dict(synthetic.locator_state_dict(depth, 1, 0, 'cpr', 0), strict=True)
    m.train()
    b = synthetic.synthetic_batch(batch, size, size, 32, 1, 123)
    img, metas = b['img'].cuda(), b['img_metas']
    gtb, gtl = [x.cuda() for x in b['gt_bboxes']], [x.cuda() for x in b['gt_labels']]
    tr = CprTrainer(m, lr=0.02)
    for _ in range(warm_steps):
        tr.forward_backward(img, metas, gtb, gtl)
        tr.step()
    names = {id(p): n for n, p in m.named_parameters()}

    def grads(dtype, wgrad=True, dgrad=True, force=False):
        m.set_compute_dtype(dtype)
        training.MIXED_BF16.update(wgrad=wgrad, dgrad=dgrad, force=force)
        try:
            losses = tr.forward_backward(img, metas, gtb, gtl)
            torch.cuda.synchronize()
        finally:
            training.MIXED_BF16.update(wgrad=True, dgrad=True, force=False)
        return {names[id(p)]: p.grad.detach().double().clone() for p in tr.params}, {k: float(v) for k, v in losses.items()}

    ref, lref = grads('fp32')
    runs = [('A bf16 fwd + bf16 bwd', grads('bf16')), ('B bf16 fwd + fp32 bwd', grads('bf16', False, False)),
            ('Bw bf16 fwd + bf16 wgrad', grads('bf16', True, False)), ('Bd bf16 fwd + bf16 dgrad', grads('bf16', False, True)),
            ('C fp32 fwd + bf16 bwd', grads('fp32', True, True, True))]
    # W: the fp32 step with every trainable conv weight rounded to bf16 (fp32 arithmetic, fp32 activations): the share of the bf16 mode's
    # WEIGHT rounding alone -- the gradient's own sensitivity to a 2^-9 relative perturbation, no bf16 kernel involved
    saved = tr.flat_p.clone()
    with torch.no_grad():
        for p in tr.params:
            if p.dim() == 4:
                p.data.copy_(p.data.to(torch.bfloat16).float())
    from pointtinybenchmark_amd.layers import bump_weight_epoch
    bump_weight_epoch()
    runs.append(('W fp32 step, weights rounded to bf16', grads('fp32')))
    with torch.no_grad():
        tr.flat_p.copy_(saved)
    bump_weight_epoch()
    rep, _ = grads('fp32')
    print('R%d %d^2 B=%d, %d optimizer steps from the seeded weights; fp32 losses %s' % (depth, size, batch, warm_steps, {k: round(v, 5) for k, v in lref.items()}))
    print('fp32 step repeated: max per-tensor relative L2 %.2e (run-to-run noise of the reference itself)' % max(
        float((rep[k] - ref[k]).norm() / max(float(ref[k].norm()), 1e-30)) for k in ref))

    def block_of(name):
        parts = name.split('.')
        if parts[0] == 'backbone' and parts[1].startswith('layer'):
            return '.'.join(parts[:3])
        if parts[0] == 'neck':
            return '.'.join(parts[:3])
        if parts[0] == 'bbox_head':
            return '.'.join(parts[:3]) if parts[1].endswith('convs') else '.'.join(parts[:2])
        return '.'.join(parts[:2])
    blocks = {}
    for k in ref:
        blocks.setdefault(block_of(k), []).append(k)
    gmax = max(float(v.norm()) for v in ref.values())
    print('\nper block: relative L2 of the concatenated gradients (|g| = the block\'s fp32 gradient norm)')
    print('%-34s %10s ' % ('block', '|g|') + ' '.join('%10s' % n.split()[0] for n, _ in runs))
    for blk, keys in blocks.items():
        r = torch.cat([ref[k].flatten() for k in keys])
        row = []
        for _, (g, _l) in runs:
            a = torch.cat([g[k].flatten() for k in keys])
            row.append(float((a - r).norm() / max(float(r.norm()), 1e-30)))
        print('%-34s %10.3e ' % (blk, float(r.norm())) + ' '.join('%10.4f' % v for v in row))
    print('\nworst tensors of run A (tensors with |g| >= 1e-3 of the largest), with the other runs beside them')
    rows = []
    for k in ref:
        n = float(ref[k].norm())
        if n >= 1e-3 * gmax:
            rows.append((float((runs[0][1][0][k] - ref[k]).norm() / n), k))
    rows.sort(reverse=True)
    for e, k in rows[:12]:
        others = ' '.join('%s %.4f' % (nme.split()[0], float((g[k] - ref[k]).norm() / float(ref[k].norm()))) for nme, (g, _l) in runs[1:])
        print('  %-44s A %.4f | %s | |g| %.3e' % (k, e, others, float(ref[k].norm())))
    print('\nlosses: ' + '; '.join('%s %s' % (n.split()[0], {k: round(v, 5) for k, v in l.items() if 'loss' in k}) for n, (_g, l) in runs))
    gA, gB = runs[0][1][0], runs[1][1][0]
    dAB = sorted(((float((gA[k] - gB[k]).norm() / float(gB[k].norm())), k) for _e, k in rows), reverse=True)
    print('\nbackward kernels alone: run A against run B (the SAME bf16 forward, bf16 vs fp32 weight / data gradients), per tensor: max %.4f (%s), median %.4f'
          % (dAB[0][0], dAB[0][1], dAB[len(dAB) // 2][0]))
    for n, (g, _l) in runs:
        fa, fr = torch.cat([g[k].flatten() for k in ref]), torch.cat([ref[k].flatten() for k in ref])
        print('%-28s cosine %.6f  |g| ratio %.4f  max per-tensor rel L2 %.4f' % (n, float(torch.dot(fa, fr) / (fa.norm() * fr.norm())), float(fa.norm() / fr.norm()),
                                                                               max(float((g[k] - ref[k]).norm() / float(ref[k].norm())) for _e, k in rows)))

    # (iv) the stride-2 3x3 data gradient of layer3.0.conv2 on its own
    conv = m.backbone.layer3[0].conv2
    w = conv.weight.detach()
    gen = torch.Generator().manual_seed(5)
    hw = size // 16
    dy = torch.randn((batch, hw, hw, w.shape[0]), generator=gen).cuda() * torch.rand((1, 1, 1, w.shape[0]), generator=gen).cuda()
    pt = ops.dgrad_pack(w, 2, 1)
    dx = ops.conv2d_dgrad(dy, pt, (2 * hw, 2 * hw), 2)
    dy16 = dy.to(torch.bfloat16).float()
    dx16 = ops.conv2d_dgrad(dy16, pt, (2 * hw, 2 * hw), 2)
    torch.cuda.synchronize()
    ref64 = F.conv_transpose2d(dy.permute(0, 3, 1, 2).double().cpu(), w.double().cpu(), stride=2, padding=1, output_padding=1).permute(0, 2, 3, 1)
    e32 = float((dx.double().cpu() - ref64).norm() / ref64.norm())
    e16 = float((dx16.double().cpu() - ref64).norm() / ref64.norm())
    print('\nlayer3.0.conv2 (3x3 stride 2, %d -> %d) data gradient alone, %dx%d gradient map: fp32 phase kernel vs fp64 transposed conv: rel L2 %.2e; '
          'with the gradient map rounded to bf16 first: %.2e (the rounding a bf16 data gradient of this layer WOULD add; the product keeps it fp32)'
          % (w.shape[1], w.shape[0], hw, hw, e32, e16))


if __name__ == '__main__' and os.environ.get('DECOMPOSE', '0') == '1':
    decompose()
