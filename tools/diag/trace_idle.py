"""GPU idle analysis of a `rocprofv3 --kernel-trace --output-format csv` file: per step (a step ends with the kernel whose name
starts with MARK) wall time, union of kernel time, kernel time per hardware queue, the largest idle gaps and which kernel follows
them.  python tools/diag/trace_idle.py TRACE.csv MARK [STEPS]   (e.g. MARK = sgd_kernel for the native training step,
loss_finalize_kernel for forward + loss)"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
mark = sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], r['Kernel_Name']) for r in rows)
ends = [i for i, e in enumerate(ev) if e[3].startswith(mark)]
assert len(ends) > steps, 'only %d kernels named %s*' % (len(ends), mark)
a, b = ends[-steps - 1], ends[-1]
seg = ev[a + 1:b + 1]
wall = (ev[b][1] - ev[a][1]) / 1e6
busy, cs, ce, gaps = 0, None, None, []
for s, e, q, n in seg:
    if ce is None or s > ce:
        if ce is not None:
            busy += ce - cs
            gaps.append((s - ce, n))
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
perq = collections.Counter()
for s, e, q, n in seg:
    perq[q] += e - s
print('%d steps: %.3f ms/step, GPU busy (union) %.3f ms/step = %.1f %%, %d launches/step' % (
    steps, wall / steps, busy / 1e6 / steps, 100 * busy / 1e6 / wall, len(seg) // steps))
print('kernel ms/step by hardware queue:', {q: round(v / 1e6 / steps, 2) for q, v in perq.items()})
gaps.sort(reverse=True)
print('idle %.3f ms/step in %d gaps/step; largest (us, next kernel):' % (sum(g for g, _ in gaps) / 1e6 / steps, len(gaps) // steps),
      [(round(g / 1e3, 1), n[:40]) for g, n in gaps[:8]])
c = collections.Counter()
for g, n in gaps:
    c[n[:48]] += g
print('idle by the kernel that follows (us/step):', [(k, round(v / 1e3 / steps, 1)) for k, v in c.most_common(10)])
