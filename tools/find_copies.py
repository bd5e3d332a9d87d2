"""Which host-side ops of one forward_train step launch device copies (rocclr copyBuffer / memcpy)?  torch.profiler, B=2."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import pointtinybenchmark_amd as P
from pointtinybenchmark_amd import synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
model = P.build_detector(bench.model_cfg()).cuda()
model.load_state_dict(synthetic.locator_state_dict(50, 1, 0, 'cpr', 0), strict=True)
batch = synthetic.synthetic_batch(B, 640, 640, 32, 1, 0)
img = batch['img'].cuda()
counts = [len(l) for l in batch['gt_labels']]
gtb = list(torch.split(torch.cat(batch['gt_bboxes']).cuda(), counts))
gtl = list(torch.split(torch.cat(batch['gt_labels']).cuda(), counts))
with torch.no_grad():
    for _ in range(3):
        model.forward_train(img, batch['img_metas'], gtb, gtl)
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        model.forward_train(img, batch['img_metas'], gtb, gtl)
        torch.cuda.synchronize()
ev = prof.events()
cp = [e for e in ev if 'copy' in e.name.lower() or 'memcpy' in e.name.lower()]
from collections import Counter
print(Counter(e.name for e in cp).most_common(12))
seen = Counter()
for e in cp:
    if e.stack:
        fr = [s for s in e.stack if 'pointtinybenchmark_amd' in s or 'bench.py' in s]
        seen[(e.name, fr[0] if fr else e.stack[0])] += 1
for k, v in seen.most_common(25):
    print(v, k)
