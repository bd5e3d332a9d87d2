"""The bf16 weight gradient (csrc/conv_wgrad_bf16.hip) against the fp32 one on the training step's big layers (HIP events).
  python tools/wgrad_bf16_bench.py [--batch 64]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointtinybenchmark_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--iters', type=int, default=5)
args = ap.parse_args()
g = torch.Generator().manual_seed(0)


def timed(fn):
    for _ in range(2):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(args.iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / args.iters


for (H, W, Cin, Cout, k) in [(160, 160, 256, 256, 3), (160, 160, 256, 256, 1), (40, 40, 256, 256, 3), (40, 40, 1024, 256, 1),
                             (80, 80, 512, 128, 1), (20, 20, 512, 512, 3)]:
    B = args.batch
    x = torch.randn((B, H, W, Cin), generator=g).cuda()
    dy = (torch.randn((B, H, W, Cout), generator=g) * 0.1).cuda()
    xb = x.bfloat16()
    t32 = timed(lambda: ops.conv2d_wgrad(dy, x, (Cout, Cin, k, k), 1, k // 2))
    t16 = timed(lambda: ops.conv_wgrad_bf16(dy, xb, (Cout, Cin, k, k)))
    flops = 2.0 * B * H * W * Cin * Cout * k * k
    print('%dx%d %d->%d k%d B=%d: fp32 %.3f ms | bf16 %.3f ms (%.0f TF algorithmic)' % (H, W, Cin, Cout, k, B, t32, t16, flops / t16 / 1e9))
