"""The stem kernels of round 4 against the path they replace, ms per launch:
  (default)  bf16 mode: csrc/stem_bf16.hip (conv; conv + fused max-pool) vs the fp32 implicit-GEMM stem emitting bf16 + maxpool
  --fp32     fp32 mode: csrc/stem_f32.hip (conv + BN + ReLU + max-pool in one exact-fp32 kernel) vs implicit-GEMM stem + maxpool"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointtinybenchmark_amd import ops  # noqa: E402


def timeit(f, iters=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    g = torch.Generator().manual_seed(0)
    w = (torch.randn((64, 3, 7, 7), generator=g) * 0.05).cuda()
    sc, bi = (torch.rand(64, generator=g) + 0.5).cuda(), torch.randn(64, generator=g).cuda()
    pc = ops.PackedConv(w, 2, 3)
    fp32 = '--fp32' in sys.argv
    for (N, H, W) in ((8, 1024, 1024), (64, 640, 640), (8, 800, 1344), (2, 640, 640)):
        x = torch.randn((N, H, W, 4), device='cuda')
        if fp32:
            wp = ops.stem_weight_f32(w)
            t_new = timeit(lambda: ops.stem7x7s2_pool_f32(x, wp, sc, bi))
            t_conv = timeit(lambda: ops.conv2d(x, pc, scale=sc, bias=bi, relu=True))
            y = ops.conv2d(x, pc, scale=sc, bias=bi, relu=True)
            t_pool = timeit(lambda: ops.maxpool3x3s2(y))
            fl = 2.0 * N * ((H - 1) // 2 + 1) * ((W - 1) // 2 + 1) * 64 * 147
            print('stem fp32 %dx%dx%d: fused %.3f ms (%.1f TFLOP/s of the 147-product count)  implicit-GEMM conv %.3f ms + maxpool '
                  '%.3f ms' % (N, H, W, t_new, fl / t_new / 1e9, t_conv, t_pool))
            continue
        wp = ops.stem_weight_bf16(w)
        y = ops.stem7x7s2_bf16(x, wp, sc, bi)
        t_new = timeit(lambda: ops.stem7x7s2_bf16(x, wp, sc, bi))
        t_old = timeit(lambda: ops.conv2d(x, pc, scale=sc, bias=bi, relu=True, out_dtype=torch.bfloat16))
        t_pool = timeit(lambda: ops.maxpool3x3s2(y))
        t_fused = timeit(lambda: ops.stem7x7s2_pool_bf16(x, wp, sc, bi))
        byt = x.numel() * 4 + y.numel() * 2
        print('stem %dx%dx%d: bf16 kernel %.3f ms (%.0f GB/s of in+out)  fp32 kernel %.3f ms  maxpool(bf16) %.3f ms (%.0f GB/s)  '
              'fused %.3f ms' % (N, H, W, t_new, byt / t_new / 1e6, t_old, t_pool, (y.numel() * 2 * 1.25) / t_pool / 1e6, t_fused))


if __name__ == '__main__':
    main()
