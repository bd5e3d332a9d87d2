"""Winograd forward kernel: staging variants (conv_wino.hip VAR) x tile -> XCD mappings (tpx), measurement build only.
Every variant must reproduce the VAR = 0 / tpx = 0 output BIT for bit (same arithmetic, different staging), then the
variants are timed in interleaved rounds (one process, HIP events per launch batch).
  python tools/wino_var.py [--batch 64] [--rounds 3] [--iters 5]"""
import os
os.environ.setdefault('CPR_BENCH_HOOKS', '1')
import argparse
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointtinybenchmark_amd import ops, _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--rounds', type=int, default=3)
ap.add_argument('--iters', type=int, default=5)
ap.add_argument('--vars', default='0,1,2,3,4,5')
ap.add_argument('--tpx', default='0,1,2')
ap.add_argument('--sched', default='0')
args = ap.parse_args()
VARS = [int(v) for v in args.vars.split(',')]
TPX = [int(v) for v in args.tpx.split(',')]
SCHED = [int(v) for v in args.sched.split(',')]
g = torch.Generator().manual_seed(0)


def case(B, H, W, Cin, Cout, b8):
    x = torch.randn((B, H, W, Cin), generator=g).cuda()
    w = (torch.randn((Cout, Cin, 3, 3), generator=g) * 0.02).cuda()
    pc = ops.PackedConv(w, 1, 1, torch.float32)
    a = (torch.rand((B, Cin), generator=g) + 0.5).cuda()
    b = torch.randn((B, Cin), generator=g).cuda()
    if b8:
        xb = x.view(B, H, W, Cin // 8, 8).permute(0, 3, 1, 2, 4).contiguous()
        return lambda: ops.conv3x3_wino(xb, pc, gn_part=True, in_ab=(a, b), in_relu=True, out_b8=True)
    return lambda: ops.conv3x3_wino(x, pc, gn_part=True)


def setv(var, tpx, sched):
    _lib.call('cpr_wino_set_variant', sched, 0)
    _lib.call('cpr_wino_set_staging', var, tpx)


# ---- bit-equality of every variant on small / ragged / short-K shapes and the head shape at B = 2
bad = 0
for (B, H, W, Cin, Cout) in [(2, 160, 160, 256, 256), (3, 40, 40, 256, 256), (2, 150, 134, 64, 64), (1, 84, 100, 32, 128),
                             (5, 48, 48, 128, 512)]:
    for b8 in (True, False):
        run = case(B, H, W, Cin, Cout, b8)
        setv(0, 0, 0)
        ref, refp = run()
        ref, refp = ref.clone(), refp.clone()
        for var in VARS:
            for tpx in TPX:
                for sched in SCHED:
                    setv(var, tpx, sched)
                    for rep in range(3):   # a staging race would not show on every launch
                        out, part = run()
                        torch.cuda.synchronize()
                        if not (torch.equal(out, ref) and torch.equal(part, refp)):
                            bad += 1
                            print('MISMATCH B=%d %dx%d %d->%d b8=%d var=%d tpx=%d sched=%d rep=%d maxdiff %.3e'
                                  % (B, H, W, Cin, Cout, b8, var, tpx, sched, rep, float((out - ref).abs().max())), flush=True)
                            break
print('bit-equality: %s' % ('ALL VARIANTS EQUAL' if bad == 0 else '%d MISMATCHES' % bad), flush=True)

# ---- timing at the headline shape
B = args.batch
for b8 in (True, False):
    run = case(B, 160, 160, 256, 256, b8)
    cfgs = [(v, t, s) for v in VARS for t in TPX for s in SCHED]
    times = {c: [] for c in cfgs}
    for c in cfgs:
        setv(*c)
        run(); run()
    torch.cuda.synchronize()
    for r in range(args.rounds):
        for c in cfgs:
            setv(*c)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(args.iters):
                run()
            e.record()
            torch.cuda.synchronize()
            times[c].append(s.elapsed_time(e) / args.iters)
    print('--- 3x3 256->256 160x160 B=%d %s (ms per launch: min / median over %d rounds)'
          % (B, 'channel-blocked + fused affine + GN stats' if b8 else 'NHWC plain + GN stats', args.rounds))
    for c in cfgs:
        t = sorted(times[c])
        print('var %d tpx %d sched %d : %.3f / %.3f' % (c[0], c[1], c[2], t[0], t[len(t) // 2]), flush=True)
setv(-1, -1, 0)
