"""Runs one weight-gradient shape repeatedly (for rocprofv3 --pmc passes / quick A-B timing).  Default: the CPR head's
3x3 256->256 layer on a (B,160,160,256) map, plain input."""
import os
os.environ.setdefault('CPR_BENCH_HOOKS', '1')   # measurement build (libcprhip_bench.so: python -m pointtinybenchmark_amd.build --bench-hooks)
import argparse
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointtinybenchmark_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=16)
ap.add_argument('--hw', type=int, default=160)
ap.add_argument('--cin', type=int, default=256)
ap.add_argument('--cout', type=int, default=256)
ap.add_argument('--k', type=int, default=3)
ap.add_argument('--iters', type=int, default=10)
ap.add_argument('--xf', action='store_true')
ap.add_argument('--no-wino', action='store_true', help='keep the 3x3 layer on the direct weight-gradient kernel')
ap.add_argument('--ablate', type=int, default=0)
args = ap.parse_args()
from pointtinybenchmark_amd import _lib  # noqa: E402
_lib.call('cpr_wgrad_set_ablation', args.ablate)
ops.WINOGRAD[0] = not args.no_wino
g = torch.Generator().manual_seed(0)
x = torch.randn((args.batch, args.hw, args.hw, args.cin), generator=g).cuda()
dy = torch.randn((args.batch, args.hw, args.hw, args.cout), generator=g).cuda()
ab = None
if args.xf:
    ab = ((torch.rand((args.batch, args.cin), generator=g) + 0.5).cuda(), torch.randn((args.batch, args.cin), generator=g).cuda())
shape = (args.cout, args.cin, args.k, args.k)
out = torch.empty(shape, device='cuda')
for _ in range(3):
    ops.conv2d_wgrad(dy, x, shape, 1, args.k // 2, in_ab=ab, in_relu=True, out=out)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(args.iters):
    ops.conv2d_wgrad(dy, x, shape, 1, args.k // 2, in_ab=ab, in_relu=True, out=out)
e.record()
torch.cuda.synchronize()
t = s.elapsed_time(e) / args.iters
fl = 2.0 * dy.numel() * args.cin * args.k * args.k
print('wgrad %s on %s ablate %d: %.3f ms, %.1f TFLOP/s' % (shape, tuple(x.shape), args.ablate, t, fl / t / 1e9))
