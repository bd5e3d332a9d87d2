"""Build-container measurement (needs /root/reference; never run on the GPU box): the REFERENCE's own classes (ResNet + FPN +
CPRHead through oracle/ref_loader.py) timed beside the CPU oracle ("port", oracle/cpr_oracle.py) on the same host, the same
threads and the same B=2 640x640 batch -- answers whether bench.py's cpu_baseline (kind 'port', the only thing that travels to
the GPU box) is as fast as the original.  Output: one JSON object (profiles/round3_cpu_reference_vs_port.json).
  python tools/ref_vs_port_cpu.py [--threads 16] [--steps 3]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cpr_oracle as O, ref_loader  # noqa: E402
from oracle.gen_golden import GN  # noqa: E402
from pointtinybenchmark_amd import synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--threads', type=int, default=16)
ap.add_argument('--steps', type=int, default=3)
args = ap.parse_args()
assert ref_loader.available(), 'needs /root/reference'
torch.set_num_threads(args.threads)
R = ref_loader.load()
sd = synthetic.locator_state_dict(50, 1, 0, 'cpr', 0)
batch = synthetic.synthetic_batch(2, 640, 640, 32, 1, 0)
alpha = 0.25
backbone = R.ResNet(depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1, norm_cfg=dict(type='BN', requires_grad=True),
                    norm_eval=True, style='pytorch')
neck = R.FPN(in_channels=[256, 512, 1024, 2048], out_channels=256, start_level=0, add_extra_convs='on_input', num_outs=1, norm_cfg=GN)
head = R.CPRHead(norm_cfg=GN, num_classes=1, in_channels=256, feat_channels=256, stacked_convs=4, strides=[4],
                 loss_mil=dict(type='MILLoss', binary_ins=False, loss_weight=alpha), loss_type=0,
                 loss_cfg=dict(with_neg=True, neg_loss_weight=1 - alpha, refine_bag_policy='independent_with_gt_bag',
                               random_remove_rate=0.4, with_gt_loss=True, gt_loss_weight=alpha, with_mil_loss=True),
                 normal_cfg=dict(prob_cls_type='sigmoid', out_bg_cls=False),
                 train_pts_extractor=dict(pos_generator=dict(type='CirclePtFeatGenerator', radius=5),
                                          neg_generator=dict(type='OutCirclePtFeatGenerator', radius=5, class_wise=True)),
                 refine_pts_extractor=dict(pos_generator=dict(type='CirclePtFeatGenerator', radius=5),
                                           neg_generator=dict(type='OutCirclePtFeatGenerator', radius=5, keep_wh=True, class_wise=True)),
                 point_refiner=dict(merge_th=0.1, refine_th=0.1, classify_filter=True))
backbone.load_state_dict({k[9:]: v for k, v in sd.items() if k.startswith('backbone.')}, strict=True)
neck.load_state_dict({k[5:]: v for k, v in sd.items() if k.startswith('neck.')}, strict=True)
head.load_state_dict({k[10:]: v for k, v in sd.items() if k.startswith('bbox_head.')}, strict=True)
for m in (backbone, neck, head):
    m.train()


def ref_step():
    with torch.no_grad():
        cls_feat, ins_feat = head(neck(backbone(batch['img'])))
        return head.loss(cls_feat, ins_feat, batch['gt_bboxes'], batch['gt_labels'], batch['img_metas'])


def port_step():
    with torch.no_grad():
        return O.locator_forward_train(sd, batch, 50, 0, 4, 5, 1)[0]


def timed(fn):
    fn()
    ts = []
    for _ in range(args.steps):
        t0 = time.perf_counter()
        out = fn()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2], {k: float(v) for k, v in out.items()}


t_ref, l_ref = timed(ref_step)
t_port, l_port = timed(port_step)
model = [ln.split(':', 1)[1].strip() for ln in open('/proc/cpuinfo') if ln.startswith('model name')][:1]
print(json.dumps(dict(
    what='median of %d forward+loss steps, B=2 640x640, %d threads, torch %s, build container (no GPU)' % (args.steps, args.threads, torch.__version__),
    host_cpu=model[0] if model else None,
    reference_classes=dict(s_per_step=round(t_ref, 3), img_per_s=round(2 / t_ref, 3), losses=l_ref),
    port_oracle=dict(s_per_step=round(t_port, 3), img_per_s=round(2 / t_port, 3), losses=l_port),
    port_over_reference_speed=round(t_ref / t_port, 3)), indent=1))
