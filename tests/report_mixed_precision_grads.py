"""Mixed-precision training step (bf16 recorded forward, fp32 backward) against the fp32 step on the same weights and batch:
losses, cosine of the full gradient, relative L2 per large tensor (measurement tool; the bars live in
tests/test_gpu_train_step.py::test_mixed_precision_step_tracks_the_fp32_step; lives under tests/ because it shares the parity tests'
case table and model builder).
  python tests/report_mixed_precision_grads.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.gen_golden import CPR_CASES  # noqa: E402  (case definitions only)
from pointtinybenchmark_amd import synthetic  # noqa: E402
from pointtinybenchmark_amd.training import CprTrainer  # noqa: E402
from tests.test_gpu_cpr_parity import build_hip_locator, to_cuda  # noqa: E402

for name in ('cpr_r18_c3_128', 'cpr_r50_c1_160_spread'):
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
