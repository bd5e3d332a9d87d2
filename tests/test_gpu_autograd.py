"""-m gpu: ``loss.backward()`` on the drop-in classes (pointtinybenchmark_amd/autograd_bridge.py).

The reference trains through torch autograd: ``BaseDetector.train_step`` returns a loss with a graph
(T/mmdet/models/detectors/base.py:214-247), mmcv's OptimizerHook calls ``loss.backward()``, clips and steps a torch optimizer,
MMDistributedDataParallel reduces the gradients (T/mmdet/apis/train.py:75-83,116-123).  Here the same calls run the HIP
backward kernels.  Bars: gradients BIT-equal to ``CprTrainer.forward_backward`` (same kernels, deterministic since round 4),
2e-3 against ``loss.backward()`` through the reference's own modules (tests/golden/cpr_grads_*.npz), torch.optim.SGD +
clip_grad_norm_ steps equal to the native optimizer's to rounding, torch DDP on a 1-rank RCCL group."""
import os
import warnings

import numpy as np
import pytest

from tests.conftest import free_port
import torch

from oracle.gen_golden import CPR_CASES
from pointtinybenchmark_amd import synthetic
from tests.test_gpu_cpr_parity import build_hip_locator, to_cuda

pytestmark = pytest.mark.gpu


def _data(cfg, seed=None):
    batch = synthetic.synthetic_batch(cfg['batch'], cfg['height'], cfg['width'], cfg['num_gts'], cfg['num_classes'],
                                      cfg['seed'] if seed is None else seed, cfg.get('ragged', False))
    cb = to_cuda(batch)
    return dict(img=cb['img'], img_metas=cb['img_metas'], gt_bboxes=cb['gt_bboxes'], gt_labels=cb['gt_labels'])


def _trainer_grads(cfg, data):
    from pointtinybenchmark_amd.training import CprTrainer
    m, _ = build_hip_locator(cfg)
    tr = CprTrainer(m)
    losses = tr.forward_backward(**data)
    torch.cuda.synchronize()
    return {k: p.grad.clone() for k, p in m.named_parameters() if p.requires_grad}, {k: float(v) for k, v in losses.items()}


@pytest.mark.parametrize('name', ['cpr_r18_c3_128', 'cpr_r50_c1_160_spread', 'cpr_r50_c80_s8_r8'])
def test_loss_backward_is_bit_equal_to_the_trainer(name):
    """BasicLocator.train_step (autograd on, exactly how the reference's runner calls it) -> loss.backward(): every
    trainable parameter's .grad equals CprTrainer.forward_backward's bit for bit, and the logged losses are the same floats."""
    cfg = CPR_CASES[name]
    data = _data(cfg)
    want, want_losses = _trainer_grads(cfg, data)
    m, _ = build_hip_locator(cfg)
    assert all(p.grad is None for p in m.parameters())
    out = m.train_step(dict(data), optimizer=None)
    assert out['loss'].requires_grad and out['loss'].grad_fn is not None, 'train_step must return a differentiable loss'
    for k, v in want_losses.items():
        assert out['log_vars'][k] == v, (k, out['log_vars'][k], v)
    out['loss'].backward()
    torch.cuda.synchronize()
    got = {k: p.grad for k, p in m.named_parameters() if p.requires_grad}
    assert sorted(got) == sorted(want)
    for k in want:
        assert got[k] is not None, 'no gradient reached %s' % k
        assert torch.equal(got[k], want[k]), '%s: max abs diff %.3e (|g| max %.3e)' % (
            k, float((got[k] - want[k]).abs().max()), float(want[k].abs().max()))
    for k, p in m.named_parameters():
        if not p.requires_grad:
            assert p.grad is None, 'frozen parameter %s received a gradient' % k
    # the consumed graph refuses a second backward instead of silently re-reading freed tapes
    with pytest.raises(Exception):
        out['loss'].backward()


@pytest.mark.parametrize('name', ['cpr_r18_c3_128', 'cpr_r50_c1_160_spread'])
def test_mixed_precision_loss_backward_is_bit_equal_to_the_trainer(name):
    """bf16 compute mode through the bridge (round 5; the reference analogue is Fp16OptimizerHook around an unmodified
    loss.backward(), T/mmdet/apis/train.py:116-119): bf16 recorded maps cross the Function boundaries behind fp32 carriers, the
    boundary gradients stay fp32 -- so every .grad must equal the mixed-precision CprTrainer's bit for bit, and the logged losses
    must be the same floats."""
    from pointtinybenchmark_amd.training import CprTrainer
    cfg = CPR_CASES[name]
    data = _data(cfg)
    m, _ = build_hip_locator(cfg)
    m.set_compute_dtype('bf16')
    tr = CprTrainer(m)
    losses = tr.forward_backward(**data)
    torch.cuda.synchronize()
    want = {k: p.grad.clone() for k, p in m.named_parameters() if p.requires_grad}
    want_losses = {k: float(v) for k, v in losses.items()}
    m2, _ = build_hip_locator(cfg)
    m2.set_compute_dtype('bf16')
    out = m2.train_step(dict(data), optimizer=None)
    assert out['loss'].requires_grad and out['loss'].grad_fn is not None, 'the bf16 mode must return a differentiable loss too'
    for k, v in want_losses.items():
        assert out['log_vars'][k] == v, (k, out['log_vars'][k], v)
    out['loss'].backward()
    torch.cuda.synchronize()
    for k, p in m2.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and p.grad.dtype == torch.float32, k
            assert torch.equal(p.grad, want[k]), '%s: max abs diff %.3e (|g| max %.3e)' % (
                k, float((p.grad - want[k]).abs().max()), float(want[k].abs().max()))
        else:
            assert p.grad is None, k


@pytest.mark.parametrize('name', ['cpr_r18_c3_128', 'cpr_r50_c1_160_spread', 'cpr_r50_c80_s8_r8'])
def test_loss_backward_matches_reference_autograd_golden(name):
    """The same bar tests/test_gpu_train_step.py holds the trainer to, on loss.backward(): total loss 1e-4, per-tensor norm
    2e-3 and strided samples 2e-3 against loss.backward() through the REFERENCE's own modules (tests/golden/cpr_grads_*)."""
    from oracle.gen_golden import grad_sample_index
    cfg = CPR_CASES[name]
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'cpr_grads_%s.npz' % name))
    m, _ = build_hip_locator(cfg)
    losses = m.forward_train(**_data(cfg))
    total = sum(v for k, v in losses.items() if 'loss' in k)
    total.backward()
    torch.cuda.synchronize()
    assert abs(float(total) - float(gold['total_loss'])) <= 1e-4 * max(1.0, abs(float(gold['total_loss'])))
    params = dict(m.named_parameters())
    keys = [k[len('norm:'):] for k in gold.files if k.startswith('norm:')]
    assert sorted(keys) == sorted(k for k, p in params.items() if p.requires_grad)
    gmax = max(float(gold['norm:' + k]) for k in keys)
    for k in keys:
        g = params[k].grad.detach().double().flatten().cpu()
        ref_n = float(gold['norm:' + k])
        assert abs(float(g.norm()) - ref_n) <= 2e-3 * ref_n + 1e-6 * gmax, (k, float(g.norm()), ref_n)
        smp = g[torch.from_numpy(grad_sample_index(g.numel()))].numpy()
        ref = gold['sample:' + k].astype(np.float64)
        assert np.abs(smp - ref).max() <= 2e-3 * max(np.abs(ref).max(), 1e-5 * gmax), k


def test_upstream_scale_and_accumulation():
    """What torch hands the loss Function is applied on the device: (2 * loss).backward() doubles every gradient exactly,
    a second step ACCUMULATES into .grad as autograd does, and a total that leaves a term out (here: neg_loss only) equals
    the trainer run with the other loss weights at zero."""
    cfg = CPR_CASES['cpr_r18_c3_128']
    data = _data(cfg)
    want, _ = _trainer_grads(cfg, data)
    m, _ = build_hip_locator(cfg)
    losses = m.forward_train(**data)
    (2.0 * sum(v for k, v in losses.items() if 'loss' in k)).backward()
    torch.cuda.synchronize()
    for k, p in m.named_parameters():
        if p.requires_grad:
            assert torch.equal(p.grad, 2.0 * want[k]), k
    losses = m.forward_train(**data)
    sum(v for k, v in losses.items() if 'loss' in k).backward()       # accumulates: 2g + g
    torch.cuda.synchronize()
    for k, p in m.named_parameters():
        if p.requires_grad:
            assert torch.allclose(p.grad, 3.0 * want[k], rtol=1e-6, atol=0), k
    # one term alone: the gradient of neg_loss is what is left of the total when the bag terms are removed
    m.zero_grad(set_to_none=True)
    losses = m.forward_train(**data)
    losses['neg_loss'].backward()
    neg = {k: p.grad.clone() for k, p in m.named_parameters() if p.requires_grad}
    m.zero_grad(set_to_none=True)
    losses = m.forward_train(**data)
    (losses['gt_loss'] + losses['pos_loss']).backward()
    torch.cuda.synchronize()
    for k, p in m.named_parameters():
        if p.requires_grad:
            s = neg[k] + p.grad
            scale = float(want[k].abs().max())
            assert float((s - want[k]).abs().max()) <= 2e-5 * scale + 1e-12, k


def test_torch_optimizer_and_clip_drive_the_drop_in_model():
    """mmcv's OptimizerHook sequence on the drop-in model -- zero_grad, loss.backward(), clip_grad_norm_(35), SGD(momentum
    0.9, weight decay 1e-4).step() -- for three steps, against the native trainer (same gradients; its clip + SGD kernel
    restates torch's update).  Also covers the pack caches: torch.optim updates parameters in place and every packed /
    folded / Winograd-transformed weight must be rebuilt from the new values."""
    from pointtinybenchmark_amd.training import CprTrainer
    cfg = CPR_CASES['cpr_r18_c3_128']
    data = _data(cfg)
    lr = 0.05
    ma, _ = build_hip_locator(cfg)
    tr = CprTrainer(ma, lr=lr, momentum=0.9, weight_decay=1e-4, max_norm=35.0)
    mb, _ = build_hip_locator(cfg)
    opt = torch.optim.SGD([p for p in mb.parameters() if p.requires_grad], lr=lr, momentum=0.9, weight_decay=1e-4)
    la, lb = [], []
    for _ in range(3):
        la.append(tr.train_step(dict(data))['log_vars']['loss'])
        out = mb.train_step(dict(data), opt)
        opt.zero_grad()
        out['loss'].backward()
        torch.nn.utils.clip_grad_norm_([p for p in mb.parameters() if p.requires_grad], 35.0)
        opt.step()
        lb.append(out['log_vars']['loss'])
    torch.cuda.synchronize()
    assert la[0] == lb[0]
    for a, b in zip(la, lb):
        assert abs(a - b) <= 2e-5 * max(1.0, abs(a)), (la, lb)
    assert la[2] != la[0], 'the optimizer must have moved the weights'
    pa, pb = dict(ma.named_parameters()), dict(mb.named_parameters())
    for k in pa:
        if pa[k].requires_grad:
            d = float((pa[k] - pb[k]).abs().max())
            assert d <= 2e-5 * max(float(pa[k].abs().max()), 1e-3), (k, d)


def test_ddp_wraps_the_drop_in_model_single_rank_rccl():
    """torch DistributedDataParallel (MMDistributedDataParallel's base class) around BasicLocator on a 1-rank RCCL group: the
    reducer's hooks must fire for the Functions' parameters while the backward runs, and the gradients must equal the
    unwrapped model's (an average over one rank).  find_unused_parameters=True as in the reference's train.py."""
    import torch.distributed as dist
    from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
    cfg = CPR_CASES['cpr_r18_c3_128']
    data = _data(cfg)
    want, _ = _trainer_grads(cfg, data)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(free_port()))
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        m, _ = build_hip_locator(cfg)
        ddp = torch.nn.parallel.DistributedDataParallel(m, device_ids=[torch.cuda.current_device()],
                                                        find_unused_parameters=True, bucket_cap_mb=8)
        fired = []

        def hook(state, bucket):
            fired.append(bucket.buffer().numel())
            return default_hooks.allreduce_hook(state, bucket)
        ddp.register_comm_hook(None, hook)
        losses = ddp(data['img'], data['img_metas'], return_loss=True, gt_bboxes=data['gt_bboxes'], gt_labels=data['gt_labels'])
        loss, log_vars = m._parse_losses(losses)
        loss.backward()
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    assert len(fired) >= 2 and sum(fired) == sum(v.numel() for v in want.values()), fired
    for k, p in m.named_parameters():
        if p.requires_grad:
            assert torch.equal(p.grad, want[k]), k


_DDP2_WORKER = r'''
import os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, %(root)r)
rank, port, out = int(sys.argv[1]), sys.argv[2], sys.argv[3]
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=port)
dist.init_process_group('gloo', rank=rank, world_size=2)      # two ranks on the box's one GPU: RCCL refuses that, gloo does not
torch.cuda.set_device(0)
from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
from oracle.gen_golden import CPR_CASES
from pointtinybenchmark_amd import synthetic
from tests.test_gpu_cpr_parity import build_hip_locator, to_cuda
cfg = CPR_CASES['cpr_r18_c3_128']
cb = to_cuda(synthetic.synthetic_batch(4, cfg['height'], cfg['width'], cfg['num_gts'], cfg['num_classes'], 5, True))
sl = slice(2 * rank, 2 * rank + 2)
m, _ = build_hip_locator(cfg)
ddp = torch.nn.parallel.DistributedDataParallel(m, device_ids=[0], find_unused_parameters=True, bucket_cap_mb=2)
fired = []
def hook(state, bucket):
    fired.append(bucket.buffer().numel())
    return default_hooks.allreduce_hook(state, bucket)
ddp.register_comm_hook(None, hook)
opt = torch.optim.SGD([p for p in m.parameters() if p.requires_grad], lr=0.01, momentum=0.9, weight_decay=1e-4)
logs = []
for it in range(2):                                            # the reference's driver: forward -> _parse_losses -> backward -> clip -> step
    opt.zero_grad()
    losses = ddp(cb['img'][sl].contiguous(), cb['img_metas'][sl], return_loss=True, gt_bboxes=cb['gt_bboxes'][sl],
                 gt_labels=cb['gt_labels'][sl])
    loss, log_vars = m._parse_losses(losses)
    loss.backward()
    if it == 0:
        g0 = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters() if p.requires_grad}
    torch.nn.utils.clip_grad_norm_([p for p in m.parameters() if p.requires_grad], 35.0)
    opt.step()
    logs.append(log_vars['loss'])
torch.cuda.synchronize()
torch.save(dict(g0=g0, p={k: p.detach().cpu() for k, p in m.named_parameters() if p.requires_grad}, logs=logs, fired=fired),
           out + '.%%d' %% rank)
dist.barrier(); dist.destroy_process_group()
'''


def test_ddp_two_ranks_on_the_gpu_average_the_gradients(tmp_path):
    """The reference's multi-GPU driver on the real kernels: torch DistributedDataParallel (MMDistributedDataParallel's base,
    T/mmdet/apis/train.py:75-83) around BasicLocator in TWO processes -- a gloo group, since both ranks share the box's one GPU
    -- each with its shard of a 4-image batch: ``loss.backward()`` runs the HIP backward behind the bridge's Functions, DDP's
    hooks reduce the buckets, clip_grad_norm_ + torch.optim.SGD step.  After backward every rank must hold the MEAN of the two
    shards' gradients (computed here by the native trainer on each shard), both ranks the same parameters after two steps."""
    import subprocess
    import sys
    from pointtinybenchmark_amd.training import CprTrainer
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'ddp2.py'
    script.write_text(_DDP2_WORKER % dict(root=root))
    port, out = str(free_port()), str(tmp_path / 'res')
    procs = [subprocess.Popen([sys.executable, str(script), str(r), port, out], cwd=root, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    logs = [p.communicate(timeout=600)[0].decode(errors='replace') for p in procs]
    assert all(p.returncode == 0 for p in procs), '\n'.join(l[-1500:] for l in logs)
    res = [torch.load(out + '.%d' % r) for r in range(2)]
    cfg = CPR_CASES['cpr_r18_c3_128']
    cb = to_cuda(synthetic.synthetic_batch(4, cfg['height'], cfg['width'], cfg['num_gts'], cfg['num_classes'], 5, True))
    m, _ = build_hip_locator(cfg)
    tr = CprTrainer(m)
    want = {}
    for r in range(2):
        sl = slice(2 * r, 2 * r + 2)
        tr.forward_backward(cb['img'][sl].contiguous(), cb['img_metas'][sl], cb['gt_bboxes'][sl], cb['gt_labels'][sl])
        torch.cuda.synchronize()
        for k, p in m.named_parameters():
            if p.requires_grad:
                want[k] = want.get(k, 0) + p.grad.detach().cpu() * 0.5
    assert sum(res[0]['fired']) >= sum(v.numel() for v in want.values()) and len(res[0]['fired']) >= 4, res[0]['fired'][:8]
    for k, w in want.items():
        for r in range(2):
            d = float((res[r]['g0'][k] - w).abs().max())
            assert d <= 1e-6 * max(float(w.abs().max()), 1e-6) + 1e-9, (r, k, d)
    for k in want:
        assert torch.equal(res[0]['p'][k], res[1]['p'][k]), k
    assert res[0]['logs'] == res[1]['logs']                      # _parse_losses all-reduces the logged means


def test_p2p_loss_backward_is_bit_equal_to_the_trainer():
    """The same bridge for BasicLocator(P2PHead) (configs[3]): per-image loss lists, (B, 2) upstream gradients."""
    import pointtinybenchmark_amd as P
    from bench import p2p_model_cfg
    from pointtinybenchmark_amd.training import P2PTrainer

    def build():
        m = P.build_detector(p2p_model_cfg(18)).cuda()
        m.load_state_dict(synthetic.locator_state_dict(18, 1, 0, 'p2p', 3, head_std=0.05), strict=True)
        m.train()
        return m
    batch = synthetic.synthetic_batch(2, 128, 160, 6, 1, seed=8)
    cb = to_cuda(batch)
    data = dict(img=cb['img'], img_metas=cb['img_metas'], gt_bboxes=cb['gt_bboxes'], gt_labels=cb['gt_labels'])
    ma = build()
    tr = P2PTrainer(ma)
    la = tr.forward_backward(**data)
    torch.cuda.synchronize()
    want = {k: p.grad.clone() for k, p in ma.named_parameters() if p.requires_grad}
    mb = build()
    out = mb.train_step(dict(data))
    assert out['loss'].requires_grad
    out['loss'].backward()
    torch.cuda.synchronize()
    la_total = float(sum(sum(v) for k, v in la.items() if 'loss' in k))
    assert abs(out['log_vars']['loss'] - la_total) <= 1e-6 * max(1.0, abs(la_total))
    for k, p in mb.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and torch.equal(p.grad, want[k]), k


def test_unsupported_options_keep_the_forward_only_path_and_say_so():
    """A model the bridge has no rule for (a stage with only some of its blocks frozen): with autograd on, forward_train still
    returns the losses, without a graph, and warns once; under no_grad nothing warns.  (The bf16 compute mode and the head
    options this test used until round 5 now all train through the bridge.)"""
    from pointtinybenchmark_amd import autograd_bridge
    cfg = CPR_CASES['cpr_r18_c3_128']
    data = _data(cfg)
    m, _ = build_hip_locator(cfg)
    for p in m.backbone.layer3[0].parameters():
        p.requires_grad_(False)
    assert 'partly frozen' in autograd_bridge.unsupported_reason(m)
    autograd_bridge._WARNED.clear()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        losses = m.forward_train(**data)
        m.forward_train(**data)
        with torch.no_grad():
            m.forward_train(**data)
    assert not any(v.requires_grad for v in losses.values())
    msgs = [str(x.message) for x in w if 'WITHOUT a graph' in str(x.message)]
    assert len(msgs) == 1 and 'partly frozen' in msgs[0], msgs


def test_backward_releases_the_recorded_maps():
    """The tapes hang off the Functions' ctx; after loss.backward() (and with the loss dropped) the recorded maps must be
    gone without waiting for the cycle collector: memory returns to the parameters + gradients level."""
    import gc
    cfg = CPR_CASES['cpr_r50_c1_160_spread']
    data = _data(cfg)
    m, _ = build_hip_locator(cfg)
    gc.disable()
    try:
        for _ in range(2):                                    # first step: allocator warm-up, pack caches, gradients
            out = m.train_step(dict(data))
            out['loss'].backward()
            del out
        torch.cuda.synchronize()
        base = torch.cuda.memory_allocated()
        out = m.train_step(dict(data))
        torch.cuda.synchronize()
        held = torch.cuda.memory_allocated() - base
        out['loss'].backward()
        del out
        torch.cuda.synchronize()
        after = torch.cuda.memory_allocated() - base
    finally:
        gc.enable()
    assert held > 5e6, 'the recorded forward should hold tens of MB of maps here (%d bytes)' % held
    assert after <= 0.1 * held, 'recorded maps survived the backward: %d of %d bytes still allocated' % (after, held)


@pytest.mark.parametrize('frozen', [2, 4])
def test_more_frozen_stages(frozen):
    """frozen_stages = 2 (layer2 frozen too: two stage Functions) and 4 (the whole backbone frozen: no stage Function, the lateral
    Function's inputs carry no gradient): loss.backward() fills exactly the trainable parameters, bit-equal to the native trainer."""
    import pointtinybenchmark_amd as P
    from bench import model_cfg
    from pointtinybenchmark_amd.training import CprTrainer

    def build():
        cfg = model_cfg(18, 2)
        cfg['backbone']['frozen_stages'] = frozen
        m = P.build_detector(cfg).cuda()
        m.load_state_dict(synthetic.locator_state_dict(18, 2, 0, 'cpr', 9, head_std=0.3), strict=True)
        m.train()
        return m
    batch = synthetic.synthetic_batch(2, 96, 128, 5, 2, seed=4)
    cb = to_cuda(batch)
    data = dict(img=cb['img'], img_metas=cb['img_metas'], gt_bboxes=cb['gt_bboxes'], gt_labels=cb['gt_labels'])
    ma = build()
    tr = CprTrainer(ma)
    tr.forward_backward(**data)
    torch.cuda.synchronize()
    want = {k: p.grad.clone() for k, p in ma.named_parameters() if p.requires_grad}
    assert not any(k.startswith('backbone.layer%d' % s) for k in want for s in range(1, frozen + 1))
    mb = build()
    out = mb.train_step(dict(data))
    out['loss'].backward()
    torch.cuda.synchronize()
    for k, p in mb.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and torch.equal(p.grad, want[k]), k
        else:
            assert p.grad is None, k
