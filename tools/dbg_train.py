"""Sensitivity of the three-step training trajectory of tests/test_gpu_train_step.py::test_train_steps_match_torch_sgd:
prints the gradient norms of the three steps with the direct / Winograd 3x3 kernels and under a 1e-7 / 1e-6 relative
perturbation of the first batch, at lr 0.05 and 0.01 (the basis of that test's learning rate)."""
import os, sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from pointtinybenchmark_amd import ops, synthetic, _lib
from pointtinybenchmark_amd.training import CprTrainer
import test_gpu_train_step as T
cfg = T.CPR_CASES['cpr_r18_c3_128']
def batch(step, eps):
    b = synthetic.synthetic_batch(cfg['batch'], cfg['height'], cfg['width'], cfg['num_gts'], cfg['num_classes'], cfg['seed'] + step, True)
    img = b['img'].cuda()
    if eps:
        img = img * (1 + eps * torch.randn(img.shape, generator=torch.Generator().manual_seed(1)).cuda())
    return dict(img=img, img_metas=b['img_metas'], gt_bboxes=[x.cuda() for x in b['gt_bboxes']], gt_labels=[x.cuda() for x in b['gt_labels']])
for wino, eps, lr in ((False, 0, 0.05), (False, 1e-7, 0.05), (False, 1e-6, 0.05), (True, 0, 0.05), (False, 0, 0.01), (False, 1e-6, 0.01), (True, 0, 0.01)):
    ops.WINOGRAD[0] = wino
    m, sd0 = T.build_hip_locator(cfg)
    tr = CprTrainer(m, lr=lr, momentum=0.9, weight_decay=1e-4, max_norm=35.0)
    gn = []
    for s in range(3):
        tr.train_step(batch(s, eps if s == 0 else 0))
        gn.append(tr.grad_norm())
    print('wino', wino, 'eps', eps, 'lr', lr, ['%.6f' % g for g in gn])
