"""Time the streaming logit projection on its real shape (B x 160 x 160 x 256 -> 2) with rotating buffers."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointtinybenchmark_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
g = torch.Generator(device='cuda').manual_seed(0)
xs = [torch.randn((B, 160, 160, 256), device='cuda', generator=g) for _ in range(4)]
w = torch.randn((2, 256), device='cuda', generator=g) * 0.01
b = torch.zeros(2, device='cuda')
a = torch.rand((B, 256), device='cuda', generator=g) + 0.5
bb = torch.randn((B, 256), device='cuda', generator=g)
pc = ops.PackedConv(w[:, :, None, None], 1, 0)
for name, fn in (('stream', lambda i: ops.logit_project(xs[i], w, b, (a, bb), True)),
                 ('mfma', lambda i: ops.conv2d(xs[i], pc, bias=b, in_ab=(a, bb), in_relu=True))):
    for i in range(4): fn(i)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(40): fn(i % 4)
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 40
    print(name, '%.4f ms  %.2f TB/s' % (t, xs[0].numel() * 4 / t / 1e9))
