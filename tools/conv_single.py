"""Runs one conv shape repeatedly (for rocprofv3 --pmc passes).  Default: the CPR head's 3x3 256->256 conv on a
(B,160,160,256) map with the fused GN-apply input and GN-stats epilogue, exactly as the forward launches it."""
import os
os.environ.setdefault('CPR_BENCH_HOOKS', '1')   # measurement build (libcprhip_bench.so: python -m pointtinybenchmark_amd.build --bench-hooks)
import argparse
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointtinybenchmark_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--hw', type=int, default=160)
ap.add_argument('--cin', type=int, default=256)
ap.add_argument('--cout', type=int, default=256)
ap.add_argument('--k', type=int, default=3)
ap.add_argument('--iters', type=int, default=10)
ap.add_argument('--plain', action='store_true')
ap.add_argument('--gn-stats', action='store_true', help='plain input, GroupNorm statistics in the epilogue (the forward\'s dominant launch)')
ap.add_argument('--res', action='store_true', help='bottleneck conv3 form: folded-BN scale / shift + residual + ReLU (with --plain)')
ap.add_argument('--check-against', type=int, default=-1, help='bf16: also run with this --bf16-dma word and require bit-equal outputs')
ap.add_argument('--bf16', action='store_true', help='bf16 compute mode kernel (plain input, GN stats epilogue)')
ap.add_argument('--no-wino', action='store_true', help='keep the 3x3 layer on the direct implicit GEMM')
ap.add_argument('--b8', action='store_true', help='Winograd layer with channel-blocked input and output + fused input affine')
ap.add_argument('--wino-sched', type=int, default=0)
ap.add_argument('--wino-ablate', type=int, default=0)
ap.add_argument('--ablate', type=int, default=0)
ap.add_argument('--pipeline', type=int, default=1)
ap.add_argument('--w32', default='0,2,100', help='conv_wino32 debug: ablate,workgroups per CU,stagger percent')
ap.add_argument('--wfrag', type=int, default=1, help='bf16 mode: 0 = ignore the fragment-order weight image (the 256 x 256 tile then stages both operands through LDS: A/B of the round-5 instance)')
ap.add_argument('--bf16-dma', type=int, default=1, help='bf16 mode: 0 = keep the layer on the register-staged kernel (A/B); + 32 * f: f = 1..3 forces DMA tile 256x256 / 128x256 / 256x128, 4 = the round-3 rule')
args = ap.parse_args()
from pointtinybenchmark_amd import _lib  # noqa: E402
ops.WINOGRAD[0] = not args.no_wino
_lib.call('cpr_conv_set_ablation', args.ablate)
_lib.call('cpr_conv_set_pipeline', args.pipeline)
_lib.call('cpr_wino_set_variant', args.wino_sched, args.wino_ablate)
_lib.call('cpr_bf16_set_dma', args.bf16_dma)
_lib.call('cpr_bf16_set_wfrag', args.wfrag)
_lib.call('cpr_wino32_set_debug', *[int(v) for v in args.w32.split(',')])
g = torch.Generator().manual_seed(0)
x = torch.randn((args.batch, args.hw, args.hw, args.cin), generator=g).cuda()
w = (torch.randn((args.cout, args.cin, args.k, args.k), generator=g) * 0.02).cuda()
pc = ops.PackedConv(w, 1, args.k // 2, torch.bfloat16 if args.bf16 else torch.float32)
if args.bf16:
    x = x.bfloat16()
    args.gn_stats = not args.plain
a = (torch.rand((args.batch, args.cin), generator=g) + 0.5).cuda()
b = torch.randn((args.batch, args.cin), generator=g).cuda()
if args.b8:
    xb = x.view(args.batch, args.hw, args.hw, args.cin // 8, 8).permute(0, 3, 1, 2, 4).contiguous()
    run_b8 = lambda: ops.conv3x3_wino(xb, pc, gn_part=True, in_ab=(a, b), in_relu=True, out_b8=True)
if args.res:
    oh = args.hw
    res_t = torch.randn((args.batch, oh, oh, args.cout), generator=g).to(x.dtype).cuda()
    sc_t = (torch.rand(args.cout, generator=g) + 0.5).cuda()
    bi_t = torch.randn(args.cout, generator=g).cuda()
    plain = lambda: ops.conv2d(x, pc, scale=sc_t, bias=bi_t, residual=res_t, relu=True)
else:
    plain = lambda: ops.conv2d(x, pc)
for _ in range(args.iters):
    if args.b8:
        run_b8()
    elif args.plain:
        plain()
    elif args.gn_stats:
        ops.conv2d(x, pc, gn_part=True)
    else:
        ops.conv2d(x, pc, in_ab=(a, b), in_relu=True, gn_part=True)
torch.cuda.synchronize()
if args.check_against >= 0:
    f = plain if args.plain else (lambda: ops.conv2d(x, pc, gn_part=True))
    a1 = f()
    _lib.call('cpr_bf16_set_dma', args.check_against)
    _lib.call('cpr_bf16_set_wfrag', 0)
    a2 = f()
    _lib.call('cpr_bf16_set_dma', args.bf16_dma)
    _lib.call('cpr_bf16_set_wfrag', args.wfrag)
    torch.cuda.synchronize()
    a1, a2 = (a1 if isinstance(a1, tuple) else (a1,)), (a2 if isinstance(a2, tuple) else (a2,))
    print('bit-equal to word %d without the fragment image: %s' % (args.check_against, all(torch.equal(u, v) for u, v in zip(a1, a2))))
ops.TRACE_CONV_VARIANT[0] = True
(run_b8 if args.b8 else plain if args.plain else (lambda: ops.conv2d(x, pc, gn_part=True)) if args.gn_stats else (lambda: None))()
variant = ops.TRACE_CONV_VARIANT[1]
ops.TRACE_CONV_VARIANT[0] = False
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(args.iters):
    if args.b8:
        run_b8()
    elif args.plain:
        plain()
    elif args.gn_stats:
        ops.conv2d(x, pc, gn_part=True)
    else:
        ops.conv2d(x, pc, in_ab=(a, b), in_relu=True, gn_part=True)
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / args.iters
fl = 2.0 * args.batch * args.hw * args.hw * args.cout * args.cin * args.k * args.k
print('conv %dx%d %d->%d k%d B=%d: %.3f ms  %.1f TFLOP/s  variant %s' % (args.hw, args.hw, args.cin, args.cout, args.k, args.batch,
                                                              ms, fl / ms / 1e9, variant))
