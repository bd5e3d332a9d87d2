"""Per-layer timing of every distinct conv launch of the CPR R50-FPN forward (HIP events, 20 launches each).
Prints flops, minimum HBM bytes, time, TFLOP/s, GB/s and the fraction of min(MFMA roof, HBM roof)."""
import os
os.environ.setdefault('CPR_BENCH_HOOKS', '1')   # measurement build (libcprhip_bench.so: python -m pointtinybenchmark_amd.build --bench-hooks)
import argparse
import json
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import pointtinybenchmark_amd as P  # noqa: E402
from pointtinybenchmark_amd import ops, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--out', default=None)
    ap.add_argument('--pipeline', type=int, default=1)
    ap.add_argument('--tile', default='0,0')
    ap.add_argument('--extra-lds', type=int, default=0, help='occupancy probe: dynamic LDS bytes added to every direct-conv launch')
    ap.add_argument('--dtype', default='fp32', choices=['fp32', 'bf16'])
    ap.add_argument('--bf16-dma', type=int, default=1, help='bf16 mode: 0 = every layer on the register-staged kernels (A/B)')
    ap.add_argument('--stream', type=int, default=1, help='0 = the streamed 1x1 kernel is never chosen (A/B of conv1x1_stream.hip)')
    ap.add_argument('--depth', type=int, default=50)
    ap.add_argument('--size', type=int, default=640, help='square image side (configs[4]: --depth 101 --size 1024 --dtype bf16 --batch 8)')
    args = ap.parse_args()
    from pointtinybenchmark_amd import _lib
    _lib.call('cpr_bf16_set_dma', args.bf16_dma)
    _lib.call('cpr_conv_set_stream', args.stream)
    _lib.call('cpr_conv_set_pipeline', args.pipeline)
    _lib.call('cpr_conv_force_tile', *[int(v) for v in args.tile.split(',')])
    _lib.call('cpr_conv_set_extra_lds', args.extra_lds)
    model = P.build_detector(bench.model_cfg(depth=args.depth)).cuda()
    model.load_state_dict(synthetic.locator_state_dict(args.depth, 1, 0, 'cpr', 0), strict=True)
    model.set_compute_dtype(args.dtype)
    batch = synthetic.synthetic_batch(args.batch, args.size, args.size, 32, 1, 0)
    img = batch['img'].cuda()
    gtb = [b.cuda() for b in batch['gt_bboxes']]
    gtl = [l.cuda() for l in batch['gt_labels']]
    calls = []
    orig = ops.conv2d

    def rec(x, pc, *a, **k):
        calls.append((x, pc, a, dict(k)))
        return orig(x, pc, *a, **k)
    ops.conv2d = rec
    duals = []
    orig_dual = ops.conv2d_dual

    def rec_dual(x, pc, x2, pc2, *a, **k):
        duals.append((x, pc, x2, pc2, a, dict(k)))
        return orig_dual(x, pc, x2, pc2, *a, **k)
    ops.conv2d_dual = rec_dual
    with torch.no_grad():
        model.forward_train(img, batch['img_metas'], gtb, gtl)
    ops.conv2d = orig
    ops.conv2d_dual = orig_dual
    torch.cuda.synchronize()
    uniq = {}
    for x, pc, a, k in calls:
        if ops.is_b8(x):      # channel-blocked (N, C/8, H, W, 8)
            N, H, W, Cin = x.shape[0], x.shape[2], x.shape[3], x.shape[1] * 8
        else:
            N, H, W, Cin = x.shape
        key = (N, H, W, Cin, pc.Cout, pc.KH, pc.stride, k.get('residual') is not None, k.get('in_ab') is not None,
               bool(k.get('gn_part')))
        uniq.setdefault(key, [0, x, pc, a, k])[0] += 1
    rows = []
    tot_t = tot_f = 0.0
    for key, (cnt, x, pc, a, k) in uniq.items():
        N, H, W, Cin, Cout, KH, stride, res, xform, gnp = key
        OH, OW = pc.out_hw(H, W)
        creal = 3 if Cin == 4 else Cin
        flops = 2.0 * N * OH * OW * Cout * KH * KH * creal
        byts = 4.0 * (N * H * W * Cin + N * OH * OW * Cout * (2 if res else 1) + Cout * KH * KH * Cin)
        for _ in range(3):
            orig(x, pc, *a, **k)
        ops.TRACE_CONV_VARIANT[0] = True
        orig(x, pc, *a, **k)
        variant = ops.TRACE_CONV_VARIANT[1]
        ops.TRACE_CONV_VARIANT[0] = False
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(args.iters):
            orig(x, pc, *a, **k)
        e.record()
        torch.cuda.synchronize()
        t = s.elapsed_time(e) / args.iters * 1e-3
        if args.dtype == 'bf16':
            byts *= 0.5
        roof = max(flops / (2500e12 if (args.dtype == 'bf16' and Cin != 4) else 157.3e12), byts / 6.3e12)
        rows.append(dict(key=str(key), count=cnt, ms=t * 1e3, tflops=flops / t / 1e12, gbs=byts / t / 1e9,
                         roof_frac=roof / t, gflop=flops / 1e9, mb=byts / 1e6, variant=str(variant)))
        tot_t += cnt * t
        tot_f += cnt * flops
    for x, pc, x2, pc2, a, k in duals:      # conv3 + projection shortcut in one launch (ops.conv2d_dual)
        N, H, W, Cin = x.shape
        flops = 2.0 * N * H * W * pc.Cout * (Cin + x2.shape[3])
        byts = 4.0 * (x.numel() + N * H * W * x2.shape[3] + N * H * W * pc.Cout + pc.Cout * (Cin + x2.shape[3]))
        for _ in range(3):
            orig_dual(x, pc, x2, pc2, *a, **k)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(args.iters):
            orig_dual(x, pc, x2, pc2, *a, **k)
        e.record()
        torch.cuda.synchronize()
        t = s.elapsed_time(e) / args.iters * 1e-3
        roof = max(flops / 157.3e12, byts / 6.3e12)
        key = 'dual ' + str((N, H, W, Cin, '+', x2.shape[1], x2.shape[3], 's%d' % pc2.stride, pc.Cout))
        rows.append(dict(key=key, count=1, ms=t * 1e3, tflops=flops / t / 1e12, gbs=byts / t / 1e9, roof_frac=roof / t,
                         gflop=flops / 1e9, mb=byts / 1e6))
        tot_t += t
        tot_f += flops
    rows.sort(key=lambda r: -r['ms'] * r['count'])
    print('%-62s %3s %8s %8s %8s %6s %7s' % ('N,H,W,Cin,Cout,K,s,res,xf,gn', 'cnt', 'ms', 'TF/s', 'GB/s', 'roof', 'tot ms'))
    for r in rows:
        print('%-62s %3d %8.3f %8.1f %8.0f %6.2f %7.2f  %s' % (r['key'], r['count'], r['ms'], r['tflops'], r['gbs'],
                                                              r['roof_frac'], r['ms'] * r['count'], r.get('variant', '')))
    print('conv total per step: %.2f ms, %.1f TF/s aggregate (B=%d)' % (tot_t * 1e3, tot_f / tot_t / 1e12, args.batch))
    if args.out:
        json.dump(rows, open(args.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
