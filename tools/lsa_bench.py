"""LSA kernel alone on a P2PNet-sized batch (16 images x 32 gts x 25 600 proposals, topk_k = 5): time per launch of the
register-resident and the memory-resident kernel, and (measurement build) the shader-clock breakdown of workgroup 0's phases."""
import ctypes
import os
import sys

os.environ.setdefault('CPR_BENCH_HOOKS', '1')
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pointtinybenchmark_amd import _lib, ops  # noqa: E402


def main():
    B, G, side, k = 16, 32, 160, 5
    g = torch.Generator().manual_seed(0)
    M = side * side
    ys, xs = torch.meshgrid(torch.arange(side) * 4.0, torch.arange(side) * 4.0, indexing='ij')
    pts = torch.stack([xs.reshape(-1), ys.reshape(-1)], 1)
    costs = []
    for b in range(B):
        gt = torch.rand((G, 2), generator=g) * 624 + 8
        cost = 0.1 * (pts[None] - gt[:, None]).abs().sum(-1) + 9.2 + torch.randn((G, M), generator=g) * 0.01     # (G, M)
        costs.append(cost.contiguous().cuda())
    for reg in (True, False):
        ops.LSA_REGISTER_KERNEL[0] = reg
        for _ in range(2):
            out, st = ops.lsa_topk(costs, k)
        torch.cuda.synchronize()
        if reg:
            _lib.call('cpr_lsa_phase_clocks', None, 1)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        n = 5
        for _ in range(n):
            out, st = ops.lsa_topk(costs, k)
        e.record()
        torch.cuda.synchronize()
        print('%s kernel: %.3f ms per launch (incl. host-side table uploads)' % ('register' if reg else 'memory  ', s.elapsed_time(e) / n))
        if reg:
            buf = (ctypes.c_longlong * 8)()
            _lib.call('cpr_lsa_phase_clocks', buf, 0)
            names = ['round set-up', 'search set-up', 'scan', 'reduce + wait', 'thread-0 bookkeeping', 'duals/augment/restore', '-', 'between rounds']
            tot = sum(buf)
            for nm, v in zip(names, buf):
                print('   %-24s %12d shader clocks over %d launches  %5.1f %%' % (nm, v, n, 100.0 * v / max(tot, 1)))
    ops.LSA_REGISTER_KERNEL[0] = True


if __name__ == '__main__':
    main()
