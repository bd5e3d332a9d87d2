"""Measurement build, same box, same process: the bf16 forward + loss step with the 256 x 256 tile on the ping-pong instance
(the dispatch rule, hook word 1) against the round-5 instances (word 1 + 8192: shape 0 stays on weights-direct / lock-step),
interleaved rounds.  configs[4] = --depth 101 --size 1024 --batch 8; R50 640^2 = --depth 50 --size 640 --batch 64."""
import os
os.environ.setdefault('CPR_BENCH_HOOKS', '1')
import argparse
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import pointtinybenchmark_amd as P  # noqa: E402
from pointtinybenchmark_amd import _lib, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--depth', type=int, default=101)
ap.add_argument('--size', type=int, default=1024)
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--steps', type=int, default=20)
ap.add_argument('--rounds', type=int, default=3)
ap.add_argument('--train', action='store_true', help='the mixed-precision training step (native trainer) instead of forward + loss')
args = ap.parse_args()
model = P.build_detector(bench.model_cfg(depth=args.depth)).cuda()
model.load_state_dict(synthetic.locator_state_dict(args.depth, 1, 0, 'cpr', 0), strict=True)
model.set_compute_dtype('bf16')
batch = synthetic.synthetic_batch(args.batch, args.size, args.size, 32, 1, 0)
img = batch['img'].cuda()
gtb = [b.cuda() for b in batch['gt_bboxes']]
gtl = [l.cuda() for l in batch['gt_labels']]
if args.train:
    from pointtinybenchmark_amd.training import CprTrainer
    model.train()
    tr = CprTrainer(model, lr=1e-4)

    def step():
        tr.forward_backward(img, batch['img_metas'], gtb, gtl)
        tr.step()
else:
    def step():
        with torch.no_grad():
            model.forward_train(img, batch['img_metas'], gtb, gtl)
res = {}
for rnd in range(args.rounds + 1):
    for word, name in ((1, 'ping-pong (dispatch rule)'), (1 + 8192, 'round-5 instances')):
        _lib.call('cpr_bf16_set_dma', word)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        if rnd:
            res.setdefault(name, []).append(ms)
for name, v in res.items():
    print('R%d %d^2 B=%d %s %-28s: ms per step %s -> best %.3f ms = %.1f img/s' % (
        args.depth, args.size, args.batch, 'train' if args.train else 'fwd+loss', name, ' '.join('%.3f' % x for x in v), min(v), args.batch / min(v) * 1e3))
_lib.call('cpr_bf16_set_dma', 1)
