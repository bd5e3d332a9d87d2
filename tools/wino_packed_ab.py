"""Winograd forward kernel: the packed-fp32 input transform / fused affine (product) against the scalar instruction form
(measurement build, ablation bit 6): outputs must be BIT-equal, then both are timed in interleaved rounds on the same box.
  python tools/wino_packed_ab.py [--batch 64] [--rounds 5] [--iters 5]"""
import os
os.environ.setdefault('CPR_BENCH_HOOKS', '1')
import argparse
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointtinybenchmark_amd import ops, _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--rounds', type=int, default=5)
ap.add_argument('--iters', type=int, default=5)
args = ap.parse_args()
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


def form(scalar):
    _lib.call('cpr_wino_set_variant', 0, 64 if scalar else 0)


bad = 0
for (B, H, W, Cin, Cout) in [(2, 160, 160, 256, 256), (3, 40, 40, 256, 256), (2, 150, 134, 64, 64), (1, 84, 100, 32, 128)]:
    for b8 in (True, False):
        f = case(B, H, W, Cin, Cout, b8)
        form(False)
        ref = [t.clone() for t in f()]
        form(True)
        out = f()
        for r, o in zip(ref, out):
            if not torch.equal(r, o):
                bad += 1
                print('MISMATCH', (B, H, W, Cin, Cout), 'b8' if b8 else 'nhwc', float((r - o).abs().max()))
print('bit-equality packed vs scalar form:', 'EQUAL' if bad == 0 else '%d MISMATCHES' % bad)

for b8, name in ((True, 'channel-blocked + fused affine + GN stats'), (False, 'NHWC plain + GN stats')):
    f = case(args.batch, 160, 160, 256, 256, b8)
    times = {False: [], True: []}
    for sc in (False, True):
        form(sc)
        for _ in range(3):
            f()
    for _ in range(args.rounds):
        for sc in (False, True):
            form(sc)
            f()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(args.iters):
                f()
            e.record()
            torch.cuda.synchronize()
            times[sc].append(s.elapsed_time(e) / args.iters)
    print('--- 3x3 256->256 160x160 B=%d %s (ms per launch, min / median over %d rounds)' % (args.batch, name, args.rounds))
    for sc in (False, True):
        t = sorted(times[sc])
        print('%-7s: %.3f / %.3f' % ('scalar' if sc else 'packed', t[0], t[len(t) // 2]))
form(False)
