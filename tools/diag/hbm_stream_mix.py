"""HBM ceilings for the traffic mixes of the HBM-bound 1x1 conv layers (measurement tool): torch elementwise kernels of the same
read : write ratios on 1.68 GB maps (the 160x160x256 fp32 map at B=64), HIP events, GB/s.
  python tools/diag/hbm_stream_mix.py"""
import torch

n = 64 * 160 * 160 * 256
a = torch.randn(n, device='cuda')
b = torch.randn(n, device='cuda')
q = torch.randn(n // 4, device='cuda')
c = torch.empty_like(a)


def timed(fn, byts, name, iters=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    t = s.elapsed_time(e) / iters * 1e-3
    print('%-58s %7.3f ms  %6.0f GB/s' % (name, t * 1e3, byts / t / 1e9))


timed(lambda: torch.sum(a), 4 * n, 'read only (sum of 1.68 GB)')
timed(lambda: c.fill_(1.0), 4 * n, 'write only (fill 1.68 GB)')
timed(lambda: c.copy_(a), 8 * n, 'copy: read 1 : write 1')
timed(lambda: torch.add(a, b, out=c), 12 * n, 'add: read 2 : write 1')
timed(lambda: torch.relu_(c), 8 * n, 'in place relu: read 1 : write 1')
qq = q.view(-1, 1).expand(-1, 4).reshape(-1)  # materialised below: the 64->256 layer reads 0.25 + 1 maps and writes 1
timed(lambda: torch.add(a, 1.0, out=c), 8 * n, 'add scalar: read 1 : write 1')
