"""GPU box: per-tensor gradient errors of one CPRHead option case (oracle.gen_golden_r2 option name) -- HIP trainer vs torch
autograd over the options oracle on the host, full tensors (relative L2, worst first).  python tools/diag/option_grads_report.py NAME"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import cpr_oracle as O, cpr_options_oracle as OO  # noqa: E402
from oracle.gen_golden_r2 import case_inputs  # noqa: E402
from oracle.gen_golden_r5 import grad_option_cfg as option_cfg  # noqa: E402
from tests.test_gpu_options import build_hip, cuda_batch  # noqa: E402
from pointtinybenchmark_amd.training import CprTrainer  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'ins_tower_fc'
cfg = option_cfg(name)
if os.environ.get('SEED'):
    cfg = dict(cfg, seed=int(os.environ['SEED']))
m, batch = build_hip(cfg)
cb = cuda_batch(batch)
tr = CprTrainer(m)
losses = tr.forward_backward(cb['img'], cb['img_metas'], cb['gt_bboxes'], cb['gt_labels'])
torch.cuda.synchronize()
got = {k: p.grad.detach().cpu().double() for k, p in m.named_parameters() if p.requires_grad}
sd, _ = case_inputs(cfg)
sd = {k: v.clone() for k, v in sd.items()}
for k in got:
    sd[k].requires_grad_(True)
torch.set_num_threads(16)
feats = O.fpn_forward(sd, O.resnet_forward(sd, batch['img'], cfg['depth']), cfg['start_level'], 1)
cf, _ = O.cpr_head_forward(sd, feats)
inf = OO.ins_tower_forward(sd, feats)[0] if cfg.get('ins_tower') else None
l, _ = OO.cpr_loss(sd, cf[0], batch['gt_bboxes'], batch['gt_labels'], batch['img_metas'], cfg, ins_feat=inf)
tot = sum(v for k, v in l.items() if 'loss' in k)
tot.backward()
print('loss hip %.8f oracle %.8f' % (float(sum(v for k, v in losses.items() if 'loss' in k)), float(tot)))
rows = []
for k, g in got.items():
    r = sd[k].grad.double()
    rows.append((float((g - r).norm() / max(float(r.norm()), 1e-30)), float((g - r).abs().max() / max(float(r.abs().max()), 1e-30)),
                 float(r.norm()), k))
rows.sort(reverse=True)
if os.environ.get('SUMMARY'):       # one line per (case, seed): is the fixture smooth (no ReLU-boundary flip between the CPU and GPU forwards)?
    gm = max(r[2] for r in rows)
    live = [r for r in rows if r[2] > 1e-5 * gm]
    print('%s seed %s: worst rel_l2 %.2e (%s)  worst max_abs/max %.2e (%s)' % (
        name, cfg['seed'], live[0][0], live[0][3], max(r[1] for r in live), max(live, key=lambda r: r[1])[3]))
    sys.exit(0)
show = rows if os.environ.get('REPORT_ALL') else rows[:16]
if os.environ.get('REPORT_ALL'):
    show = sorted([r for r in rows if r[3].startswith('bbox_head') or r[3].startswith('neck')], key=lambda r: r[3])
for e, emax, n, k in show:
    print('%-48s rel_l2 %.3e  max_abs/max %.3e  |g| %.3e' % (k, e, emax, n))
