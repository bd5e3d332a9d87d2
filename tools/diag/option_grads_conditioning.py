"""CPU: how well conditioned is one option-gradient fixture?  The options oracle's autograd gradient in fp32 against the same in
fp64 (same weights, same batch): per tensor, the worst strided-sample deviation relative to the tensor's max -- the quantity the
GPU test bounds.  A fixture whose fp32-vs-fp64 deviation is already near the bar cannot tell a wrong backward from rounding
(a ReLU boundary flip somewhere upstream): pick another seed or state a wider bar.
PERTURB=1e-5 replaces the fp32 pass by an fp64 pass whose conv / linear weights carry that much relative noise (the size of the
GPU path's fp32 Winograd rounding): the fixture's sensitivity at the scale the GPU comparison actually probes.
python tools/diag/option_grads_conditioning.py NAME [SEED ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import cpr_oracle as O, cpr_options_oracle as OO  # noqa: E402
from oracle.gen_golden import grad_sample_index  # noqa: E402
from oracle.gen_golden_r2 import case_inputs  # noqa: E402
from oracle.gen_golden_r5 import grad_option_cfg  # noqa: E402


PERTURB = float(os.environ.get('PERTURB', '0'))


def grads(cfg, dtype, perturb=0.0):
    sd, batch = case_inputs(cfg)
    sd = {k: (v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    if perturb:
        gen = torch.Generator().manual_seed(7)
        for k, v in sd.items():
            if v.is_floating_point() and v.dim() >= 2:
                v.mul_(1 + perturb * torch.randn(v.shape, generator=gen, dtype=torch.float64))
    keys = [k for k in sd if sd[k].is_floating_point() and not any(t in k for t in ('running_', 'num_batches'))
            and not k.startswith('backbone.conv1') and not k.startswith('backbone.bn1') and not k.startswith('backbone.layer1')]
    for k in keys:
        sd[k].requires_grad_(True)
    img = batch['img'].to(dtype)
    boxes = batch["gt_bboxes"]                       # geometry stays in fp32 (points, validity, masks identical in both passes)
    feats = O.fpn_forward(sd, O.resnet_forward(sd, img, cfg['depth']), cfg['start_level'], 1)
    cf, _ = O.cpr_head_forward(sd, feats)
    inf = OO.ins_tower_forward(sd, feats)[0] if cfg.get('ins_tower') else None
    l, _ = OO.cpr_loss(sd, cf[0], boxes, batch['gt_labels'], batch['img_metas'], cfg, ins_feat=inf)
    tot = sum(v for k, v in l.items() if 'loss' in k)
    tot.backward()
    return float(tot), {k: sd[k].grad.detach().double().flatten() for k in keys if sd[k].grad is not None}


name = sys.argv[1]
seeds = [int(s) for s in sys.argv[2:]] or [None]
torch.set_num_threads(16)
for seed in seeds:
    cfg = grad_option_cfg(name)
    if seed is not None:
        cfg = dict(cfg, seed=seed)
    t32, g32 = grads(cfg, torch.float64, PERTURB) if PERTURB else grads(cfg, torch.float32)
    t64, g64 = grads(cfg, torch.float64)
    gmax = max(float(g.norm()) for g in g64.values())
    rows = []
    for k, r in g64.items():
        idx = torch.from_numpy(grad_sample_index(r.numel()))
        d = float((g32[k][idx] - r[idx]).abs().max()) / max(float(r[idx].abs().max()), 1e-5 * gmax)
        rows.append((d, float((g32[k] - r).norm() / max(float(r.norm()), 1e-30)), k))
    rows.sort(reverse=True)
    print('%s seed %s: loss fp32 %.7f fp64 %.7f; worst sampled deviation / tensor max (and rel L2):' % (name, cfg['seed'], t32, t64))
    for d, e, k in rows[:4]:
        print('   %-48s %.3e  (%.3e)' % (k, d, e))
