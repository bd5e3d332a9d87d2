"""The reference's driver sequence on the drop-in classes, headline config (R50 640^2, B=64): train_step -> loss.backward() (autograd
bridge) -> clip_grad_norm_ -> torch.optim.SGD.step(), N steps -- what bench.py's train_step.torch_autograd times; for rocprofv3
kernel traces (tools/diag/trace_idle.py ... stem_pool_f32_kernel).  NATIVE=1: the native CprTrainer instead."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import torch  # noqa: E402
import pointtinybenchmark_amd as P  # noqa: E402
from pointtinybenchmark_amd import synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
B = int(os.environ.get('BATCH', '64'))
model = P.build_detector(bench.model_cfg()).cuda()
model.load_state_dict(synthetic.locator_state_dict(50, 1, 0, 'cpr', 0), strict=True)
model.train()
batch = synthetic.synthetic_batch(B, 640, 640, 32, 1, 0)
img = batch['img'].cuda()
counts = [len(l) for l in batch['gt_labels']]
gtb = list(torch.split(torch.cat(batch['gt_bboxes']).cuda(), counts))
gtl = list(torch.split(torch.cat(batch['gt_labels']).cuda(), counts))
data = dict(img=img, img_metas=batch['img_metas'], gt_bboxes=gtb, gt_labels=gtl)
if os.environ.get('NATIVE'):
    from pointtinybenchmark_amd.training import CprTrainer
    tr = CprTrainer(model, lr=1e-3, momentum=0.9, weight_decay=1e-4, max_norm=35.0)

    def step():
        tr.forward_backward(img, batch['img_metas'], gtb, gtl)
        tr.step()
else:
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=1e-3, momentum=0.9, weight_decay=1e-4)

    def step():
        opt.zero_grad(set_to_none=True)
        o = model.train_step(dict(data), opt)
        o['loss'].backward()
        torch.nn.utils.clip_grad_norm_(params, 35.0)
        opt.step()
for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    step()
torch.cuda.synchronize()
print('%.1f img/s (%.2f ms/step)' % (B * n / (time.perf_counter() - t0), (time.perf_counter() - t0) / n * 1e3))
