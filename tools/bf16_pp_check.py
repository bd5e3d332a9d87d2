"""Measurement build: the two-group ping-pong 256 x 256 bf16 instance (csrc/conv_bf16_pp.hip, hook word 1 + 32 * 7 = 225) against the
lock-step LDS-staged instance (word 1, fragment image off) -- bit-equality on big-tile shapes (several launches each: a staging race
shows as a difference that comes and goes), then timings of the three instances and of the ping-pong instance's compile-time
ablations on the layer shapes that matter, then its phase clocks.  `--words` adds hook words to the timing table."""
import os
os.environ.setdefault('CPR_BENCH_HOOKS', '1')
import argparse
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointtinybenchmark_amd import ops, _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--words', default='', help='comma-separated extra cpr_bf16_set_dma words for the timing table')
ap.add_argument('--iters', type=int, default=10)
ap.add_argument('--skip-check', action='store_true')
args = ap.parse_args()
PP = 225                  # hook word: the ping-pong instance wherever the 256 x 256 tile fits (1 + 32 * 7)
KEEP0 = 8192              # ablate bit 9: shape 0 stays on its own instances (weights-direct with the fragment image, lock-step without)
STAMPS = 16384

CHECK = [   # N, Cin, H, W, Cout, k, stride, pad, flags
    (8, 64, 128, 128, 256, 3, 1, 1, 'gn'), (8, 128, 128, 128, 256, 3, 1, 1, 'bn res relu'), (10, 128, 121, 119, 256, 3, 1, 1, 'bias relu'),
    (6, 128, 256, 192, 512, 3, 2, 1, 'bn'), (8, 1024, 128, 128, 256, 1, 1, 0, 'bias f32out'), (8, 128, 128, 128, 256, 1, 1, 0, 'bn res relu'),
    (8, 64, 128, 128, 256, 1, 1, 0, 'bn res relu'), (8, 64, 96, 128, 512, 3, 1, 1, 'bias relu gn'), (4, 192, 160, 160, 256, 3, 1, 1, 'gn'),
    (8, 192, 128, 128, 256, 3, 1, 1, 'bn relu'),
]


def make(case, seed):
    N, Cin, H, W, Cout, k, stride, pad, flags = case
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((N, H, W, Cin), generator=g).bfloat16().cuda()
    w = (torch.randn((Cout, Cin, k, k), generator=g) / (Cin * k * k) ** 0.5).cuda()
    pc = ops.PackedConv(w, stride, pad, torch.bfloat16)
    kw = {}
    if 'bn' in flags:
        kw['scale'] = (torch.rand(Cout, generator=g) + 0.5).cuda()
    if 'bn' in flags or 'bias' in flags:
        kw['bias'] = torch.randn(Cout, generator=g).cuda()
    OH, OW = pc.out_hw(H, W)
    if 'res' in flags:
        kw['residual'] = torch.randn((N, OH, OW, Cout), generator=g).bfloat16().cuda()
    kw['relu'] = 'relu' in flags
    kw['gn_part'] = 'gn' in flags
    if 'f32out' in flags:
        kw['out_dtype'] = torch.float32
    return x, pc, kw


def run(word, wfrag, x, pc, kw):
    _lib.call('cpr_bf16_set_dma', word)
    _lib.call('cpr_bf16_set_wfrag', wfrag)
    ops.TRACE_CONV_VARIANT[0] = True
    out = ops.conv2d(x, pc, **kw)
    ops.TRACE_CONV_VARIANT[0] = False
    return (out if isinstance(out, tuple) else (out,)), ops.TRACE_CONV_VARIANT[1][1]


if not args.skip_check:
    bad = 0
    for ci, case in enumerate(CHECK):
        x, pc, kw = make(case, ci)
        ref, v0 = run(1, 0, x, pc, kw)
        for word in (PP,):
            for rep in range(6):
                got, v1 = run(word, 0, x, pc, kw)
                torch.cuda.synchronize()
                eq = all(torch.equal(u, v) for u, v in zip(ref, got))
                if not eq:
                    bad += 1
                    d = max(float((u.float() - v.float()).abs().max()) for u, v in zip(ref, got))
                    print('DIFF case %d word %d rep %d: variants %d / %d, max abs %.3e' % (ci, word, rep, v0, v1, d))
        print('case %d %s: lock-step variant %d, ping-pong variant %d' % (ci, case, v0, v1), flush=True)
    print('bit-equality: %s' % ('ALL EQUAL' if bad == 0 else '%d DIFFERENCES' % bad), flush=True)

TIMING_ALL = [  # label, N, Cin, HW, Cout, k, flags
    ('head 3x3 256->256 gn B=64 160^2', 64, 256, 160, 256, 3, 'gn'),
    ('head 3x3 256->256 gn B=8 256^2 (cfg4)', 8, 256, 256, 256, 3, 'gn'),
    ('3x3 512->512 B=64 80^2', 64, 512, 80, 512, 3, 'bn relu'),
    ('1x1 256->1024 +res B=8 128^2', 8, 256, 128, 1024, 1, 'bn res relu'),
    ('1x1 64->256 +res B=8 256^2', 8, 64, 256, 256, 1, 'bn res relu'),
    ('1x1 512->256 B=8 256^2', 8, 512, 256, 256, 1, 'bn res relu'),
    ('1x1 1024->256 B=64 40^2', 64, 1024, 40, 256, 1, 'bn relu'),
]
words = [(1 + KEEP0, 1, 'weights-direct'), (1, 0, 'lock-step LDS'), (PP, 0, 'ping-pong'), (1, 1, 'dispatch rule'),
         (PP + 8, 0, 'pp no requests'), (PP + 16, 0, 'pp no reads'), (PP + 24, 0, 'pp neither'), (PP + 2, 0, 'pp no counted wait'),
         (PP + 4, 0, 'pp same-KB requests'), (PP + 2048, 0, 'pp no row arithmetic'), (PP + 4096, 0, 'pp requests first')] + \
    [(int(w), 0, 'word %s' % w) for w in args.words.split(',') if w]
TIMING = TIMING_ALL
for label, N, Cin, HW, Cout, k, flags in TIMING:
    x, pc, kw = make((N, Cin, HW, HW, Cout, k, 1, k // 2, flags), 7)
    fl = 2.0 * N * HW * HW * Cout * Cin * k * k
    line = []
    for rnd in range(2):            # two interleaved rounds, the second is reported
        line = []
        for word, wf, name in words:
            _, v = run(word, wf, x, pc, kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(args.iters):
                ops.conv2d(x, pc, **kw)
            e.record()
            torch.cuda.synchronize()
            ms = s.elapsed_time(e) / args.iters
            line.append('%s %.3f ms %.0f TF (v%d)' % (name, ms, fl / ms / 1e9, v))
    print('%-40s | %s' % (label, ' | '.join(line)), flush=True)

# phase clocks (s_memtime) of chunk 8 of workgroup 16 (tile 2: its own output rows carry the stamps), wave 0 of each group: PP_STAMP in csrc/conv_bf16_pp.hip
x, pc, kw = make((8, 256, 256, 256, 256, 3, 1, 1, 'gn'), 7)
for word, name in ((PP, 'ping-pong'), (PP + 8, 'pp no requests'), (PP + 16, 'pp no reads'), (PP + 24, 'pp neither')):
    for rep in range(2):
        _lib.call('cpr_bf16_set_dma', word + STAMPS)
        _lib.call('cpr_bf16_set_wfrag', 0)
        out = ops.conv2d(x, pc, gn_part=True)[0]
        torch.cuda.synchronize()
        t = out.view(-1)[512 * 256:512 * 256 + 4 * 32].view(torch.int64).cpu().tolist()
    for g in range(2):
        v = t[16 * g:16 * g + 11]
        d = [v[i + 1] - v[i] for i in range(10)]
        print('%-28s group %d: La issue %d wait %d barrier %d | Ma issue %d barrier %d | Lb issue %d wait %d barrier %d | Mb issue %d barrier %d | chunk %d'
              % (name, g, d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8], d[9], v[10] - v[0]), flush=True)
_lib.call('cpr_bf16_set_dma', 1)
_lib.call('cpr_bf16_set_wfrag', 1)
