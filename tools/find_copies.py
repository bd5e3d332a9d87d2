"""Which host-side ops of one forward_train step launch device copies (rocclr copyBuffer / memcpy)?  torch.profiler, B=2.
TRAIN=1: one native training step (CprTrainer.forward_backward + step) instead; DEPTH=101 SIZE=1024 DTYPE=bf16: configs[4]."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import pointtinybenchmark_amd as P
from pointtinybenchmark_amd import synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
DEPTH, SIZE = int(os.environ.get('DEPTH', '50')), int(os.environ.get('SIZE', '640'))
model = P.build_detector(bench.model_cfg(depth=DEPTH)).cuda()
model.load_state_dict(synthetic.locator_state_dict(DEPTH, 1, 0, 'cpr', 0), strict=True)
model.set_compute_dtype(os.environ.get('DTYPE', 'fp32'))
batch = synthetic.synthetic_batch(B, SIZE, SIZE, 32, 1, 0)
img = batch['img'].cuda()
counts = [len(l) for l in batch['gt_labels']]
gtb = list(torch.split(torch.cat(batch['gt_bboxes']).cuda(), counts))
gtl = list(torch.split(torch.cat(batch['gt_labels']).cuda(), counts))
from torch.profiler import profile, ProfilerActivity
if os.environ.get('TRAIN'):
    from pointtinybenchmark_amd.training import CprTrainer
    model.train()
    tr = CprTrainer(model, lr=1e-3, momentum=0.9, weight_decay=1e-4, max_norm=35.0)

    def one():
        tr.forward_backward(img, batch['img_metas'], gtb, gtl)
        tr.step()
else:
    def one():
        with torch.no_grad():
            model.forward_train(img, batch['img_metas'], gtb, gtl)
for _ in range(3):
    one()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    one()
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
