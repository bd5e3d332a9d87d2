"""CPU-only checks of the host side: C-ABI surface, registry / config drop-in surface, state-dict layout,
and the N>1 path (log-var all-reduce) under gloo with world_size 2."""
import os
import re
import subprocess
import sys

import pytest

from tests.conftest import free_port
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference/TOV_mmdetection'
needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree only exists in the build container')


def test_library_exports_every_declared_symbol():
    """include/cpr_hip.h <-> libcprhip.so <-> the ctypes table: same set of entry points (no compute here)."""
    from pointtinybenchmark_amd import _lib, build
    build.build(verbose=False)
    hdr = open(os.path.join(ROOT, 'include', 'cpr_hip.h')).read()
    product, hooks = hdr.split('#ifdef CPR_BENCH_HOOKS')
    declared = set(re.findall(r'^\s*int\s+(cpr_\w+)\s*\(', product, flags=re.M))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.cpr_version() >= 1
    # the measurement switches (process-global state, results wrong by design) are NOT in the product library
    hook_names = set(re.findall(r'^\s*int\s+(cpr_\w+)\s*\(', hooks, flags=re.M))
    assert hook_names == set(_lib.BENCH_SIGNATURES)
    import ctypes
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in hook_names:
        assert not hasattr(raw, name), '%s must only exist in libcprhip_bench.so' % name
    out = subprocess.run(['nm', '-D', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r'\b(cpr_\w+)$', out, flags=re.M))
    assert exported == declared, exported ^ declared


def test_product_package_never_imports_the_oracle():
    for dp, _, files in os.walk(os.path.join(ROOT, 'pointtinybenchmark_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', src, flags=re.M), os.path.join(dp, f)


def test_ops_refuse_cpu_tensors():
    from pointtinybenchmark_amd import _lib, ops
    with pytest.raises(_lib.CprHipError):
        ops.gn_stats(torch.zeros((1, 4, 4, 256)))


def test_registry_names_and_state_dict_layout():
    import pointtinybenchmark_amd as P
    from pointtinybenchmark_amd import synthetic
    for name in ('BasicLocator', 'ResNet', 'FPN', 'CPRHead', 'P2PHead', 'MILLoss'):
        assert P.registry.MODELS.get(name) is not None, name
    for name in ('HungarianAssignerV2', 'PointAssigner'):
        assert P.BBOX_ASSIGNERS.get(name) is not None
    assert P.BBOX_SAMPLERS.get('PseudoSampler') and P.MATCH_COST.get('FocalLossCost') and P.MATCH_COST.get('DisCostV2')
    import bench
    for depth in (18, 50, 101):
        cfg = bench.model_cfg(depth)
        m = P.build_detector(cfg)
        sd = synthetic.locator_state_dict(depth)
        assert m.load_state_dict(sd, strict=True)
    m = P.build_detector(bench.model_cfg(50))
    n_train = sum(p.numel() for p in m.parameters() if p.requires_grad)
    assert n_train == 27219970     # 27.22 M trainable fp32 params (stem + layer1 frozen)
    with pytest.raises(KeyError):
        P.build_head(dict(type='CascadeCPRHead'))     # referenced by a DOTA config, defined nowhere in the reference


@needs_ref
@pytest.mark.parametrize('rel', [
    'configs2/TinyPersonV2/coarsepointv2/coarse_point_refine_r50_fpns4_1x_TinyPersonV2_640.py',
    'configs2/TinyPersonV2/coarsepointv2/coarse_point_refine_r50_fpns4_0.5x_TinyPersonV2_640.py',
    'configs2/COCO/coarsepointv2/coarse_point_refine_r50_fpn_1x_coco400.py',
    'configs2/COCO/coarsepointv2/coarse_point_refine_r101_fpn_1x_coco400.py',
    'configs2/TinyPersonV2/p2p/p2p_r50_fpns4_1x_fl_sl1_TinyPersonV2_640.py',
    'configs2/TinyPersonV2/p2p/p2p_r50_fpns4_0.5x_fl_sl1_TinyPersonV2_640.py',
    'configs2/COCO/p2p/p2p_r101_fpn_1x_fl_sl1_coco400_coarse.py',
    'configs2/COCO/p2p/p2p_r50_fpn_1x_fl_sl1_coco400_coarse.py',
    'configs2/COCO/p2p/p2p_r50_fpns4_1x_fl_sl1_coco.py',
    'configs2/DOTA/coarsepointv2/coarse_point_refine_r50_fpns4_1x_DOTA_1024.py',
    'configs2/DOTA/p2p/p2p_r50_fpn_1x_fl_sl1_DOTA_coarse.py',
    'configs2/_base_/models/cpr/coarse_point_refine_r50_fpns4_1x.py',    # refine_bag_policy='only_refine_bag'
])
def test_reference_configs_build_unmodified(rel):
    import pointtinybenchmark_amd as P
    from pointtinybenchmark_amd.config import Config
    cfg = Config.fromfile(os.path.join(REF, rel))
    m = P.build_detector(cfg.model, train_cfg=cfg.get('train_cfg'), test_cfg=cfg.get('test_cfg'))
    keys = set(m.state_dict())
    assert 'backbone.layer4.2.conv3.weight' in keys and 'neck.lateral_convs.0.gn.weight' in keys
    last = cfg.model.bbox_head.get('stacked_convs', 4) - 1
    assert 'bbox_head.cls_convs.%d.conv.weight' % last in keys and 'bbox_head.cls_out.weight' in keys
    if '_base_/models' not in rel:                                       # a bare model file has no schedule / runtime
        assert 'optimizer_config' in cfg and 'checkpoint_config' in cfg   # _base_ files merged
    if 'TinyPersonV2' in rel:
        assert cfg.optimizer_config.grad_clip.max_norm == 35            # _delete_ override handled
    cfg.merge_from_dict({'model.backbone.depth': 18, 'data.samples_per_gpu': 4})
    assert cfg.model.backbone.depth == 18 and cfg.data.samples_per_gpu == 4


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
rank = int(sys.argv[1]); os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = sys.argv[2]
dist.init_process_group('gloo', rank=rank, world_size=2)
from pointtinybenchmark_amd.detectors import BasicLocator
losses = {'gt_loss': torch.tensor(1.0 + rank), 'pos_loss': torch.tensor(2.0 * (rank + 1)), 'bag_acc': torch.tensor(50.0 * rank),
          'neg_loss': torch.tensor(0.5)}
loss, log_vars = BasicLocator._parse_losses(losses)
# loss is the LOCAL sum of keys containing "loss"; log vars are averaged over ranks (base.py:205-210)
assert abs(float(loss) - (1.0 + rank + 2.0 * (rank + 1) + 0.5)) < 1e-6, float(loss)
assert abs(log_vars['gt_loss'] - 1.5) < 1e-6 and abs(log_vars['pos_loss'] - 3.0) < 1e-6
assert abs(log_vars['bag_acc'] - 25.0) < 1e-6 and abs(log_vars['loss'] - 5.0) < 1e-6, log_vars
t = torch.tensor([float(rank + 1)], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)                 # bench.py: max-over-ranks timing
assert float(t) == 2.0
dist.barrier(); dist.destroy_process_group(); print('rank', rank, 'ok')
'''


def test_two_rank_gloo_log_var_reduction(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER % ROOT)
    port = str(free_port())
    procs = [subprocess.Popen([sys.executable, str(script), str(r), port], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


_BUCKET_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
rank = int(sys.argv[1]); os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = sys.argv[2]
dist.init_process_group('gloo', rank=rank, world_size=2)
from pointtinybenchmark_amd.training import GradBuckets
n = 950
flat = torch.arange(n, dtype=torch.float32) * (rank + 1)          # rank-dependent "gradients"
b = GradBuckets(flat, bucket_elems=300)                            # 300,300,350 (a tail under a quarter bucket is folded)
assert b.bounds == [0, 300, 600, 950], b.bounds
b.ready(250); assert b.next == 0 and not b.pending                 # bucket 0 not complete yet
b.ready(650); assert b.next == 2 and len(b.pending) == 2           # buckets 0 and 1 in flight, 2 still open
scale = b.finish()                                                 # launches the tail, waits for all
assert scale == 0.5 and b.next == 0 and not b.pending
ref = torch.arange(n, dtype=torch.float32) * 3.0                   # SUM over ranks; the optimizer applies 1/world
assert torch.equal(flat, ref), (flat[:4], ref[:4])
# a second step re-uses the object
flat.fill_(float(rank)); b.ready(n); b.finish(); assert torch.equal(flat, torch.ones(n))
dist.barrier(); dist.destroy_process_group(); print('rank', rank, 'ok')
'''


def test_two_rank_gloo_gradient_buckets(tmp_path):
    """The bucketed reducer of the training step (training.GradBuckets) under gloo, world_size 2."""
    script = tmp_path / 'bworker.py'
    script.write_text(_BUCKET_WORKER % ROOT)
    port = str(free_port())
    procs = [subprocess.Popen([sys.executable, str(script), str(r), port], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


_RS_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
rank = int(sys.argv[1]); os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = sys.argv[2]
dist.init_process_group('gloo', rank=rank, world_size=2)
from pointtinybenchmark_amd.training import GradBuckets
n = 951                                                            # odd: every bucket keeps a one-element tail
g = torch.Generator().manual_seed(7 + rank)
grads = torch.randn(n, generator=g)
a, b = grads.clone(), grads.clone()
ba = GradBuckets(a, bucket_elems=301)                              # the all-reduce reducer
bb = GradBuckets(b, bucket_elems=301, reducer='reduce_scatter')    # reduce_scatter into this rank's shard + all_gather back
assert ba.bounds == bb.bounds == [0, 301, 602, 951]
for bk in (ba, bb):
    bk.ready(400); assert bk.next == 1
    assert bk.finish() == 0.5
assert torch.equal(a, b), float((a - b).abs().max())               # two ranks: one summation order, bit-equal
other = torch.randn(n, generator=torch.Generator().manual_seed(7 + 1 - rank))
assert torch.equal(a, grads + other)
b.copy_(grads); bb.ready(n); bb.finish(); assert torch.equal(a, b)  # the shard buffers are reused
dist.barrier(); dist.destroy_process_group(); print('rank', rank, 'ok')
'''


def test_two_rank_gloo_reduce_scatter_reducer_equals_all_reduce(tmp_path):
    """reducer='reduce_scatter' (reduce_scatter_tensor + all_gather_into_tensor per bucket, tails all-reduced) gives the
    all-reduce reducer's sums bit for bit on two ranks."""
    script = tmp_path / 'rsworker.py'
    script.write_text(_RS_WORKER % ROOT)
    port = str(free_port())
    procs = [subprocess.Popen([sys.executable, str(script), str(r), port], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


_DDP_WORKER = r'''
import os, sys, torch, torch.nn as nn, torch.distributed as dist
sys.path.insert(0, %r)
rank = int(sys.argv[1]); os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = sys.argv[2]
dist.init_process_group('gloo', rank=rank, world_size=2)
from pointtinybenchmark_amd import autograd_bridge as AB

C = 6
class Vec(nn.Module):
    def __init__(self, seed, train=True):
        super().__init__()
        self.w = nn.Parameter(torch.randn(C, generator=torch.Generator().manual_seed(seed)), requires_grad=train)
class Backbone(nn.Module):
    res_layers = ['layer1', 'layer2', 'layer3']
    def __init__(self):
        super().__init__()
        self.layer1, self.layer2, self.layer3 = Vec(1, train=False), Vec(2), Vec(3)      # layer1 frozen (frozen_stages)
    def stem(self, img): return img * 0.5
    def run_stage(self, i, x, tape=None): return x * getattr(self, self.res_layers[i]).w
class Neck(nn.Module):
    in_channels, start_level = [C, C, C], 1
    def __init__(self):
        super().__init__()
        self.lateral_convs = nn.ModuleList([Vec(4), Vec(5)])
        self.fpn_convs = nn.ModuleList([Vec(6)])
class Head(nn.Module):
    def __init__(self):
        super().__init__()
        self.cls = Vec(7)
        self.never_used = Vec(8)          # find_unused_parameters: a trainable parameter no Function touches
class Model(nn.Module):
    def __init__(self):
        super().__init__()
        self.backbone, self.neck, self.bbox_head = Backbone(), Neck(), Head()
    def forward(self, img, scale):
        return AB.forward_train(self, img, [dict(scale=scale)], None, None)

class Engine:
    """CPU stand-in for training.BackwardEngine: the same segment API on closed-form elementwise math, so that the bridge's
    graph (parameters as Function inputs, gradients returned per segment) can be driven by torch DDP under gloo."""
    loss_vector_key = 'out5'
    def __init__(self, model): self.model, self._sink = model, {}
    def begin_step(self): pass
    def collect(self, params): return tuple(self._sink.pop(id(p), None) for p in params)
    def forward_stage(self, i, x):
        bb = self.model.backbone
        return bb.run_stage(i, x), [dict(x=x, w=getattr(bb, bb.res_layers[i]).w)]
    def backward_stage(self, tape, dout, need_in):
        rec = tape[0]
        self._sink[id(rec['w'])] = (dout * rec['x']).sum(0)
        return dout * rec['w'].detach() if need_in else None
    def forward_laterals(self, xs):
        ws = [m.w for m in self.model.neck.lateral_convs]
        return sum(x * w for x, w in zip(xs, ws)), dict(xs=list(xs), ws=ws)
    def backward_laterals(self, recs, dlat, need):
        for x, w in zip(recs['xs'], recs['ws']):
            self._sink[id(w)] = (dlat * x).sum(0)
        return [dlat * w.detach() if n else None for w, n in zip(recs['ws'], need)]
    def forward_head_loss(self, lat0, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None, gt_true_bboxes=None):
        o, c = self.model.neck.fpn_convs[0].w, self.model.bbox_head.cls.w
        t = lat0 * o * c * img_metas[0]['scale']
        out = torch.stack([t.sum(), (t * t).sum(), t.mean(), t.abs().sum(), t.new_tensor(1.0)])
        return out, dict(lat0=lat0, o=o, c=c, t=t, scale=img_metas[0]['scale'])
    def backward_head_loss(self, st, up):
        t = st['t']
        dt = up[0] + up[1] * 2 * t + up[2] / t.numel() + up[3] * torch.sign(t)
        o, c, lat0, sc = st['o'].detach(), st['c'].detach(), st['lat0'], st['scale']
        self._sink[id(st['o'])] = (dt * lat0 * c * sc).sum(0)
        self._sink[id(st['c'])] = (dt * lat0 * o * sc).sum(0)
        return dt * o * c * sc
    def loss_dict(self, out):
        return {'gt_loss': out[0], 'pos_loss': out[1], 'bag_acc': out[2], 'neg_loss': out[3]}

def reference(model, img, scale):          # the same math in plain torch autograd
    bb, nk, hd = model.backbone, model.neck, model.bbox_head
    x = img * 0.5
    feats = []
    for name in bb.res_layers:
        x = x * getattr(bb, name).w
        feats.append(x)
    lat0 = sum(f * m.w for f, m in zip(feats[1:], nk.lateral_convs))
    t = lat0 * nk.fpn_convs[0].w * hd.cls.w * scale
    return t.sum() + (t * t).sum() + t.abs().sum()

torch.manual_seed(0)
model = Model()
model._autograd_bridge_state = AB.Bridge(model, engine=Engine(model))
imgs = [torch.randn(4, C, generator=torch.Generator().manual_seed(10 + r)) for r in range(2)]
scales = [1.5, -0.75]
# expected: DDP averages the per-rank gradients
want = {}
for r in range(2):
    model.zero_grad()
    reference(model, imgs[r], scales[r]).backward()
    for k, p in model.named_parameters():
        if p.grad is not None:
            want[k] = want.get(k, 0) + p.grad.clone() / 2
model.zero_grad()
ddp = torch.nn.parallel.DistributedDataParallel(model, find_unused_parameters=True, bucket_cap_mb=1e-5)   # one bucket per tensor
fired = []
from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
def hook(state, bucket):
    fired.append(bucket.buffer().numel())
    return default_hooks.allreduce_hook(state, bucket)
ddp.register_comm_hook(None, hook)
losses = ddp(imgs[rank], scales[rank])
assert set(losses) == {'gt_loss', 'pos_loss', 'bag_acc', 'neg_loss'} and all(v.requires_grad for v in losses.values())
total = sum(v for k, v in losses.items() if 'loss' in k)         # BaseDetector._parse_losses: bag_acc is logged, not summed
assert torch.allclose(total, reference(model, imgs[rank], scales[rank]).detach())
total.backward()
assert len(fired) >= 5, fired                                     # the reducer's hooks ran for the Functions' parameters
for k, p in model.named_parameters():
    if not p.requires_grad:
        assert p.grad is None, k
    elif k in want:
        assert p.grad is not None and torch.allclose(p.grad, want[k], rtol=1e-5, atol=1e-6), (k, p.grad, want[k])
    else:
        assert k == 'bbox_head.never_used.w' and (p.grad is None or not p.grad.any()), k
# a second step re-uses the bridge (fresh graph, the consumed one is gone)
ddp.zero_grad()
l2 = ddp(imgs[rank], scales[rank]); sum(v for k, v in l2.items() if 'loss' in k).backward()
assert torch.allclose(model.backbone.layer2.w.grad, want['backbone.layer2.w'], rtol=1e-5, atol=1e-6)
dist.barrier(); dist.destroy_process_group(); print('rank', rank, 'ok')
'''


def test_two_rank_gloo_ddp_drives_the_autograd_bridge(tmp_path):
    """torch DistributedDataParallel (what mmcv's MMDistributedDataParallel subclasses, T/mmdet/apis/train.py:75-83) around a
    model whose forward is ``autograd_bridge.forward_train``: the three Functions take the parameters as inputs, so the DDP
    reducer's hooks fire on them, buckets are all-reduced and every rank ends with the AVERAGE of the per-rank gradients --
    including a frozen stage (no gradient) and a trainable parameter nothing uses (find_unused_parameters).  The HIP engine
    needs a GPU, so the segment API is served by a closed-form CPU stand-in; the Functions, the graph and DDP are real."""
    script = tmp_path / 'ddpworker.py'
    script.write_text(_DDP_WORKER % ROOT)
    port = str(free_port())
    procs = [subprocess.Popen([sys.executable, str(script), str(r), port], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


def test_gradient_buckets_single_process_is_a_no_op():
    from pointtinybenchmark_amd.training import GradBuckets
    flat = torch.ones(10)
    b = GradBuckets(flat, 4)
    b.ready(10)
    assert b.finish() == 1.0 and torch.equal(flat, torch.ones(10))


def test_header_is_plain_c(tmp_path):
    """include/cpr_hip.h is the C ABI: it must compile as C99 and as C++ without any HIP / torch header."""
    import shutil
    src = tmp_path / 'hdr.c'
    src.write_text('#include "cpr_hip.h"\nint main(void) { return cpr_version() < 0; }\n')
    inc = os.path.join(ROOT, 'include')
    for cc, flags in (('gcc', ['-std=c99']), ('g++', ['-std=c++17', '-x', 'c++'])):
        if shutil.which(cc) is None:
            pytest.skip(cc + ' not installed')
        subprocess.check_call([cc, '-Wall', '-Werror', '-I', inc, '-c', str(src), '-o', str(tmp_path / (cc + '.o'))] + flags)


_SYNC_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
rank = int(sys.argv[1]); os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = sys.argv[2]
dist.init_process_group('gloo', rank=rank, world_size=2)
import pointtinybenchmark_amd as P
from pointtinybenchmark_amd.training import CprTrainer
from bench import model_cfg
torch.manual_seed(100 + rank)                       # different per-rank initialisation, as unseeded processes would have
cfg = model_cfg(18)
model = P.build_detector(cfg)                       # CPU: construction only, no kernel runs
for b in model.buffers():
    if b.dtype.is_floating_point:
        b.add_(0.01 * rank)                         # buffers (BN running statistics) differ too
before = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone()
tr = CprTrainer(model, two_streams=False)           # constructor broadcasts rank 0's state
flat = torch.cat([t.detach().reshape(-1).float() for t in list(model.parameters()) + list(model.buffers())])
ref = flat.clone(); dist.broadcast(ref, src=0)
assert torch.equal(flat, ref), 'rank %%d differs from rank 0 after CprTrainer()' %% rank
if rank == 1:
    after = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    assert not torch.equal(before, after), 'the ranks were supposed to start from different weights'
tr.check_bindings()
sd = tr.state_dict()
assert all(v.untyped_storage().nbytes() == v.numel() * v.element_size() for v in sd.values())   # clones, not flat views
p0 = tr.params[0]; p0.data = p0.data.clone()        # re-binding must be caught
try:
    tr.check_bindings(); raise SystemExit('re-bound parameter not detected')
except AssertionError:
    pass
dist.barrier(); dist.destroy_process_group(); print('rank', rank, 'ok')
'''


def test_two_rank_gloo_trainer_broadcasts_initial_state(tmp_path):
    """CprTrainer replaces MMDistributedDataParallel: like DDP it must start every rank from rank 0's parameters and
    buffers (only gradients are averaged afterwards).  Also: the re-binding guard and the cloning state_dict()."""
    script = tmp_path / 'sworker.py'
    script.write_text(_SYNC_WORKER % ROOT)
    port = str(free_port())
    procs = [subprocess.Popen([sys.executable, str(script), str(r), port], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


def test_step_lr_schedule_matches_the_published_policy():
    """lr_config of the CPR configs (policy='step', warmup='linear', warmup_iters=500, warmup_ratio=0.001, step=[8, 11]) with
    optimizer lr 0.01 (coarse_point_refine_base_TinyPersonV2_640.py:99-106): values of mmcv's StepLrUpdaterHook formulas."""
    from pointtinybenchmark_amd.training import StepLrSchedule
    s = StepLrSchedule.from_config(dict(type='SGD', lr=0.01, momentum=0.9, weight_decay=0.0001),
                                   dict(policy='step', warmup='linear', warmup_iters=500, warmup_ratio=0.001, step=[8, 11]),
                                   iters_per_epoch=1000)
    assert abs(s.lr(0) - 0.01 * 0.001) < 1e-12                       # first iteration: base * warmup_ratio
    assert abs(s.lr(250) - 0.01 * (1 - 0.5 * 0.999)) < 1e-12         # half way through the warm-up
    assert s.lr(500) == 0.01 and s.lr(7999) == 0.01                  # regular lr until epoch 8
    assert abs(s.lr(8000) - 0.001) < 1e-15 and abs(s.lr(10999) - 0.001) < 1e-15
    assert abs(s.lr(11000) - 0.0001) < 1e-15
    short = StepLrSchedule(0.02, iters_per_epoch=10, step=(8, 11))   # warm-up longer than 8 epochs: both factors apply
    assert abs(short.lr(85) - 0.002 * (1 - (1 - 85 / 500) * 0.999)) < 1e-15


def test_bench_gpus_n_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no RANK in the environment must start its two ranks itself (torch.distributed.run on
    127.0.0.1) and print exactly ONE JSON line; --dry keeps the control flow (process group, barriers, max over ranks) and
    skips the device work, so the entry point is covered where no GPU exists."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '0',
                          '--dry', '--batch', '4'], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith('{')]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 3
    assert d['ms_per_step'] >= 19.0, d          # rank 1 sleeps 20 ms per step: the MAX over ranks is reported


def test_winograd_dispatch_rules_host_side():
    """Which layers run the fused Winograd kernels is host logic (ops.wino_eligible): 3x3 / stride 1 / pad 1, fp32,
    Cin % 16 == 0 (>= 32), Cout % 64 == 0 and at least 60 % of the 16x16 output regions filled."""
    import torch
    from pointtinybenchmark_amd import ops
    pc = ops.PackedConv(torch.zeros(64, 64, 3, 3), 1, 1)
    assert ops.wino_eligible(pc, 160, 160) and ops.wino_eligible(pc, 40, 40) and ops.wino_eligible(pc, 200, 336)
    assert not ops.wino_eligible(pc, 20, 20)                                             # 39 % of its regions
    assert not ops.wino_eligible(ops.PackedConv(torch.zeros(64, 64, 3, 3), 2, 1), 160, 160)      # stride 2
    assert not ops.wino_eligible(ops.PackedConv(torch.zeros(64, 64, 1, 1), 1, 0), 160, 160)      # 1x1
    assert not ops.wino_eligible(ops.PackedConv(torch.zeros(32, 64, 3, 3), 1, 1), 160, 160)      # Cout % 64
    assert not ops.wino_eligible(pc, 160, 160, torch.bfloat16)                                   # bf16 mode stays direct
    saved = ops.WINOGRAD[0]
    try:
        ops.WINOGRAD[0] = False
        assert not ops.wino_eligible(pc, 160, 160)                                               # CPR_WINOGRAD=0
    finally:
        ops.WINOGRAD[0] = saved
    x5 = torch.zeros(2, 8, 16, 16, 8)
    assert ops.is_b8(x5) and not ops.is_b8(torch.zeros(2, 16, 16, 64))


def test_bucket_layout_on_parameter_boundaries():
    """training.layout_buckets: buckets end on parameter boundaries of the flat gradient buffer, close at >= min, never pass max
    unless one parameter alone is larger, and the LAST bucket (the only reduction nothing can overlap) holds at most `tail`."""
    import random
    from pointtinybenchmark_amd.training import layout_buckets
    # the R50 CPR trainer's order: head (projection, 4 x [gn, gn, 3x3 conv]), FPN, layer4 ... layer2 (element counts)
    sizes = [256, 1, 256, 1] + [256, 256, 589824] * 5 + [256, 256, 65536, 256, 256, 131072, 256, 256, 262144, 256, 256, 524288]
    sizes += [1048576, 2048, 2048, 2359296, 512, 512, 1048576, 512, 512] * 3 + [2097152, 2048, 2048]
    sizes += [262144, 1024, 1024, 589824, 256, 256, 262144, 256, 256] * 6 + [524288, 1024, 1024]
    sizes += [65536, 512, 512, 147456, 128, 128, 65536, 128, 128] * 4 + [131072, 512, 512]
    mb = 262144
    b = layout_buckets(sizes, 25 * mb, 4 * mb, 1 * mb)
    ends = set()
    acc = 0
    for k in sizes:
        acc += k
        ends.add(acc)
    assert b[0] == 0 and b[-1] == sum(sizes) and all(x < y for x, y in zip(b, b[1:]))
    assert all(x in ends for x in b[1:]), 'bucket boundaries must be parameter boundaries'
    lens = [y - x for x, y in zip(b, b[1:])]
    assert lens[-1] <= 1 * mb, 'the tail bucket bounds the exposed part of the reducer'
    assert all(n <= 25 * mb for n in lens) and all(n >= 4 * mb for n in lens[:-2]), lens
    assert b[1] <= 3 * 589824 + 2048, 'the first bucket closes after at most two head tower layers'
    for _ in range(300):
        sz = [random.randint(1, 3000) for _ in range(random.randint(1, 40))]
        mx, mn, tl = random.randint(1, 5000), random.randint(1, 3000), random.randint(1, 2000)
        bb = layout_buckets(sz, mx, mn, tl)
        assert bb[0] == 0 and bb[-1] == sum(sz) and all(x < y for x, y in zip(bb, bb[1:])), (sz, bb)
        e, a = set(), 0
        for k in sz:
            a += k
            e.add(a)
        assert all(x in e for x in bb[1:])
        assert bb[-1] - bb[-2] <= max(tl, sz[-1])


def test_bench_config_presets_name_the_baseline_shapes():
    """`bench.py --config cfgN` = BASELINE.json configs[N] at its own shape; explicit flags win; --dry needs no GPU."""
    for cfg, want in (('cfg0', 'dry run'), ('cfg2', 'dry run')):
        out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--dry', '--config', cfg, '--steps', '1'],
                             capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr[-400:]
        assert want in out.stdout
    import importlib
    bench = importlib.import_module('bench')
    import argparse
    ns = argparse.Namespace(model='cpr', depth=50, height=800, width=1344, classes=80, stride=8, radius=8, dtype='fp32')
    assert bench.baseline_config_name(ns).startswith('configs[2]')
    ns = argparse.Namespace(model='cpr', depth=18, height=640, width=640, classes=1, stride=4, radius=5, dtype='fp32')
    assert bench.baseline_config_name(ns).startswith('configs[0]')
    cfgm = bench.model_cfg(50, 80, 1, 8, 8)
    assert cfgm['neck']['start_level'] == 1 and cfgm['bbox_head']['strides'] == [8] and cfgm['bbox_head']['num_classes'] == 80
    assert cfgm['bbox_head']['train_pts_extractor']['pos_generator']['radius'] == 8


def test_grad_report_of_the_bench_gates():
    """bench.grad_report (the numbers behind train_step.parity_gate and tests/test_gpu_fullsize_grads.py): per-tensor relative L2
    worst first, global-norm error, cosine, strided-sample error, and the nil flag for tensors whose true gradient is numerically 0."""
    from bench import grad_report
    g = torch.Generator().manual_seed(0)
    ref = {'a': torch.randn(1000, generator=g), 'b': torch.randn(64, 3, generator=g) * 10, 'nil': torch.zeros(17)}
    got = {k: v.clone() for k, v in ref.items()}
    got['a'] = got['a'] * (1 + 1e-3)
    got['b'][5, 1] += 0.5
    rep = grad_report(got, ref)
    rows = {d['key']: d for d in rep['rows']}
    assert abs(rows['a']['rel_l2'] - 1e-3) < 1e-6 and rows['nil']['nil'] and not rows['a']['nil']
    assert abs(rows['b']['rel_l2'] - 0.5 / float(ref['b'].double().norm())) < 1e-9
    assert rep['rows'][0]['key'] == 'b' or rep['rows'][0]['rel_l2'] >= rep['rows'][1]['rel_l2']
    n_ref = (float(ref['a'].double().norm()) ** 2 + float(ref['b'].double().norm()) ** 2) ** 0.5
    assert abs(rep['ref_norm'] - n_ref) < 1e-9 and 0 < rep['norm_rel'] < 1e-3 and 0.9999 < rep['cosine'] <= 1.0
    same = grad_report(ref, ref)
    assert same['norm_rel'] == 0 and abs(same['cosine'] - 1) < 1e-12 and all(d['rel_l2'] == 0 for d in same['rows'])


def test_loss_backward_dispatch_per_option_set():
    """Host logic of the training path (no GPU): which loss-backward kernels a CPRHead option set takes.  The shipped option set
    and the options that only change the towers / bag count (num_refine > 1 with independent bags, instance tower, FC layers, grid
    bags, align_corners) keep the SPECIALISED kernels -- their outputs are the bit-for-bit record of rounds 1-4; everything that
    changes the loss formula or the bag / annotated-point geometry goes to the general kernels.  Every option set trains."""
    from types import SimpleNamespace
    from oracle.gen_golden_r2 import cpr_head_kwargs
    from oracle.gen_golden_r5 import OPTION_GRAD_CASES, grad_option_cfg
    from pointtinybenchmark_amd.registry import build_head
    general = {'r2_independent': False, 'ins_tower': False, 'ins_tower_fc': False, 'ins_tower_fc_boundary': False,
               'fc2_shared': False, 'grid_circles': False, 'grid_circles_r2': False, 'align_corners': False,
               'align_corners_grid': False, 'grid_circles_fc': False,
               'softmax': True, 'normed_sigmoid_p1': True, 'normed_sigmoid_p2': True, 'binary_ins': True, 'allpos': True,
               'r2_merge_gt': True, 'r3_only_refine': True, 'bg_cls': True, 'no_mil_loss': True, 'no_neg': True,
               'combo_fc_softmax_merge': True, 'combo_tower_binary_normed': True}
    assert set(general) == set(OPTION_GRAD_CASES)
    for name, want in general.items():
        cfg = grad_option_cfg(name)
        head = build_head(dict(type='CPRHead', **cpr_head_kwargs(cfg)))
        assert head.train_step_supported(cfg.get('num_refine', 1)), name
        ex, R, G = head.train_pts_extractor, cfg.get('num_refine', 1), 7
        if ex.pos_is_grid:
            view = (1, ex.max_pos_num + 2 * R)
        else:
            from pointtinybenchmark_amd.dense_heads.cpr_head import circle_offsets
            view = (R, int(circle_offsets(ex.pos_radius, cfg['stride'], **ex.pos_kw).shape[0]) + 1)
        gts = SimpleNamespace(G=G, R=R, labels=torch.zeros(G, dtype=torch.int32), pt_labels=torch.zeros(G * R, dtype=torch.int32))
        bags, centres, labels, _ = head._loss_geometry(gts, view, None, torch.device('cpu'))
        assert labels.numel() == bags[0] and bags[0] * bags[1] == G * view[0] * view[1], (name, bags, view)
        assert head._loss_backward_general(R, bags, centres) == want, (name, bags, centres)


def test_package_asks_for_enough_hardware_queues():
    """RCCL's queues exhaust the HIP runtime's default of 4 hardware queues and the two-stream backward then serialises on one
    (pointtinybenchmark_amd/__init__.py: measured 119 vs 138 img/s on the configs[4] training line under torchrun): importing
    the package -- and bench.py, before it imports torch -- must leave GPU_MAX_HW_QUEUES set, without overriding a caller's choice."""
    code = ("import os, sys; sys.path.insert(0, %r); os.environ.pop('GPU_MAX_HW_QUEUES', None); import pointtinybenchmark_amd; "
            "print(os.environ['GPU_MAX_HW_QUEUES']); os.environ['GPU_MAX_HW_QUEUES'] = '12'; import importlib; "
            "importlib.reload(pointtinybenchmark_amd); print(os.environ['GPU_MAX_HW_QUEUES'])" % ROOT)
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-500:]
    assert out.stdout.split() == ['8', '12'], out.stdout
    head = open(os.path.join(ROOT, 'bench.py')).read().split('import torch')[0]
    assert "setdefault('GPU_MAX_HW_QUEUES'" in head
