"""The CPR training step beyond forward+loss: explicit backward over the recorded forward, a
DDP-style bucketed gradient all-reduce and the SGD update -- all on HIP kernels, no torch autograd.

What it replaces in the reference: ``loss.backward()`` (torch autograd over mmcv ConvModule / nn.GroupNorm /
F.grid_sample / the frozen-statistics BatchNorm of ResNet), MMDistributedDataParallel's gradient reducer
(T/mmdet/apis/train.py:75-86) and mmcv's OptimizerHook (clip_grad_norm_ max_norm=35 + torch.optim.SGD momentum 0.9,
weight decay 1e-4: T/configs/_base_/schedules/schedule_1x.py:2, configs2/_base_/.../base_TinyPersonV2_640.py:99-100).

Layout: all trainable parameters live in ONE flat fp32 buffer (``nn.Parameter.data`` are views into it), ordered by the
moment their gradient becomes final during the backward (head first, backbone stage 2 last); gradients and momentum are
flat buffers of the same layout.  Buckets are contiguous ranges of the gradient buffer: each is all-reduced (RCCL) as
soon as its last gradient has been enqueued, overlapping with the rest of the backward; the optimizer is one kernel over
the flat buffers (the clip coefficient is read from the device, no host sync)."""
import os
import warnings

import torch
import torch.distributed as dist

from . import ops
from .layers import bump_weight_epoch, folded_bn


WINO_DGRAD = [os.environ.get('CPR_WINO_DGRAD', '1') == '1']     # A/B switch (tools): 3x3 data gradients as Winograd + a mask pass


class GradBuckets:
    """Contiguous buckets over a flat gradient buffer.  ``ready(end)`` says gradients [0, end) are final; every bucket
    fully below ``end`` is reduced asynchronously (on the process group's own stream).  Works on CPU tensors with
    gloo (tests) and on device tensors with nccl (= RCCL).

    reducer='all_reduce' (default): one ``dist.all_reduce`` per bucket -- what MMDistributedDataParallel's reducer does
    (T/mmdet/apis/train.py:75-86).  reducer='reduce_scatter': the two halves of a ring all-reduce issued explicitly per
    bucket, ``reduce_scatter_tensor`` into this rank's shard then ``all_gather_into_tensor`` back into the bucket (xGMI is
    point to point: both halves move (N-1)/N of the bucket over each link once, and the split leaves room for a sharded
    optimizer between them).  Same sums as all_reduce up to the ring's summation order -- bit-equal on 2 ranks, where a
    sum has one order (tests/test_host_cpu.py).  timing=True (device tensors): per-bucket launch / completion events,
    ``timeline()`` after ``finish()`` -- what an N-GPU run needs to report overlap, not just img/s."""

    def __init__(self, flat, bucket_elems, group=None, force=False, reducer='all_reduce', timing=False, bounds=None):
        """force: issue the collectives even in a 1-rank group (a sum over one rank is the identity) -- lets a single GPU
        exercise the real RCCL path (tests).  bounds: explicit bucket boundaries (element offsets, 0 .. numel, ascending) --
        ``layout_buckets`` places them on parameter boundaries; default: uniform ``bucket_elems`` buckets."""
        assert reducer in ('all_reduce', 'reduce_scatter'), reducer
        self.flat, self.group, self.force, self.reducer = flat, group, force, reducer
        self.timing = bool(timing) and flat.is_cuda
        n = flat.numel()
        if bounds is not None:
            self.bounds = [int(b) for b in bounds]
            assert self.bounds[0] == 0 and self.bounds[-1] == n and all(a < b for a, b in zip(self.bounds, self.bounds[1:])), \
                'bucket bounds must ascend from 0 to numel'
        else:
            self.bounds = list(range(0, n, bucket_elems)) + [n]
            if len(self.bounds) > 2 and self.bounds[-1] - self.bounds[-2] < bucket_elems // 4:
                del self.bounds[-2]            # fold a small tail into the previous bucket
        self._shards = {}
        self._events, self._t0, self._t_end, self._last = [], None, None, None
        self.reset()

    @property
    def world_size(self):
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def reset(self):
        self.next, self.pending = 0, []

    def start_step(self):
        """timing: marks the start of the backward pass on the current stream (bucket times are relative to it)."""
        if self.timing:
            self._events = []
            self._t0 = torch.cuda.Event(enable_timing=True)
            self._t0.record()

    def _reduce(self, lo, hi):
        """Issue the reduction of gradients [lo, hi); returns the work handles in issue order."""
        if self.reducer == 'all_reduce':
            return [dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)]
        world = self.world_size
        m = (hi - lo) // world * world          # the part that splits evenly; a tail of < world elements is all-reduced
        works = []
        if m > 0:
            shard = self._shards.get((lo, m))
            if shard is None:
                shard = self._shards[(lo, m)] = torch.empty((m // world,), device=self.flat.device, dtype=self.flat.dtype)
            w = dist.reduce_scatter_tensor(shard, self.flat[lo:lo + m], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            if dist.get_backend(self.group) != 'nccl':
                w.wait()                        # host-side backends run async work on threads: order the two halves here
            else:
                works.append(w)                 # RCCL: both halves sit on the group's stream, in issue order
            works.append(dist.all_gather_into_tensor(self.flat[lo:lo + m], shard, group=self.group, async_op=True))
        if m < hi - lo:
            works.append(dist.all_reduce(self.flat[lo + m:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        return works

    @property
    def active(self):
        if not (dist.is_available() and dist.is_initialized()):
            return False
        return self.world_size > 1 or self.force

    def would_launch(self, end):
        return self.active and self.next + 1 < len(self.bounds) and self.bounds[self.next + 1] <= end

    def ready(self, end):
        if not self.active:
            return
        while self.next + 1 < len(self.bounds) and self.bounds[self.next + 1] <= end:
            lo, hi = self.bounds[self.next], self.bounds[self.next + 1]
            ev = None
            if self.timing:
                ev = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), lo, hi]
                ev[0].record()                  # the gradients of this bucket are final at this point of the backward
                self._events.append(ev)
            self.pending.append((self._reduce(lo, hi), ev))
            self.next += 1

    def finish(self):
        """All buckets reduced and visible to the current stream.  The sum is NOT divided here: the optimizer kernel
        applies 1/world_size (``grad_scale``) while it reads the gradient."""
        self.ready(self.flat.numel())
        if self.timing:
            self._t_end = torch.cuda.Event(enable_timing=True)
            self._t_end.record()                # end of the backward pass on the current stream
        for works, ev in self.pending:
            for w in works:
                w.wait()
            if ev is not None:
                ev[1].record()                  # the current stream has this bucket's sum
        if self.timing:
            self._last = (self._t0, self._t_end, self._events)
        self.reset()
        return 1.0 / self.world_size

    def timeline(self):
        """timing=True, after finish(): per bucket the time (ms since start_step) its gradients were final (= the reduction
        was issued) and the time the main stream held its sum; ``exposed_ms`` = what the step waited for the reducer after
        the backward had ended (0 = fully overlapped).  Synchronises the device."""
        if not self._last or self._last[0] is None:
            return None
        t0, t_end, events = self._last
        torch.cuda.synchronize()
        bwd = t0.elapsed_time(t_end)
        rows = [dict(bucket=i, mbytes=(hi - lo) * 4 / 1e6, issued_ms=t0.elapsed_time(a), done_ms=t0.elapsed_time(b))
                for i, (a, b, lo, hi) in enumerate(events)]
        done = max([r['done_ms'] for r in rows], default=bwd)
        return dict(reducer=self.reducer, world_size=self.world_size, backward_ms=bwd, exposed_ms=max(0.0, done - bwd),
                    buckets=rows)


def layout_buckets(sizes, max_elems, min_elems, tail_elems):
    """Bucket boundaries on PARAMETER boundaries of a flat gradient buffer whose parameters (``sizes``, elements each) lie in
    the order their gradients become final.  A bucket is closed as soon as it holds ``min_elems`` (so the head's first tower
    layer -- 2.4 MB, final a few milliseconds into the backward -- is on the wire while the rest of the head, 63 % of the
    backward's time, still computes) and never grows past ``max_elems`` unless one parameter alone is larger; the LAST bucket
    -- the only one whose reduction cannot overlap with any backward work, because its gradients are the last to be computed
    -- is cut to at most ``tail_elems``: the exposed part of the reducer is then bounded by design (a <= 1 MB all-reduce)
    instead of being whatever the uniform split left over (27 MB in round 3)."""
    ends, acc = [], 0
    for k in sizes:
        ends.append(acc + k)
        acc += k
    n = acc
    # tail: the longest suffix of whole parameters that fits tail_elems (at least the last parameter)
    tail_start = len(sizes) - 1
    t = sizes[-1]
    while tail_start > 0 and t + sizes[tail_start - 1] <= tail_elems:
        tail_start -= 1
        t += sizes[tail_start]
    bounds, lo = [0], 0
    for i, e in enumerate(ends[:tail_start]):
        nxt = ends[i + 1] if i + 1 < tail_start else None
        if e - lo >= min_elems or (nxt is not None and nxt - lo > max_elems):
            bounds.append(e)
            lo = e
    head_end = ends[tail_start - 1] if tail_start > 0 else 0
    if head_end > bounds[-1]:
        bounds.append(head_end)
    if n > bounds[-1]:
        bounds.append(n)
    return bounds


class StepLrSchedule:
    """The reference's learning-rate policy (``lr_config = dict(policy='step', warmup='linear', warmup_iters=500,
    warmup_ratio=0.001, step=[8, 11])``, T/configs2/TinyPersonV2/coarsepointv2/coarse_point_refine_base_TinyPersonV2_640.py:
    101-106, T/configs/_base_/schedules/schedule_1x.py:5-11) as a pure function of the iteration.  The hooks that implement
    it live in mmcv (StepLrUpdaterHook / LrUpdaterHook, un-vendored, mmcv-full 1.3.x): restated from the published
    algorithm -- parity unpinned --
        regular lr of epoch e      base_lr * gamma ** #{s in step : e >= s}
        iteration i < warmup_iters regular_lr * (1 - (1 - i / warmup_iters) * (1 - warmup_ratio))     ('linear')
    by_epoch=True: the epoch of iteration i is i // iters_per_epoch."""

    def __init__(self, base_lr, iters_per_epoch, step=(8, 11), gamma=0.1, warmup='linear', warmup_iters=500,
                 warmup_ratio=0.001):
        assert warmup in (None, 'linear', 'constant', 'exp') and iters_per_epoch > 0
        self.base_lr, self.iters_per_epoch, self.step, self.gamma = base_lr, int(iters_per_epoch), tuple(step), gamma
        self.warmup, self.warmup_iters, self.warmup_ratio = warmup, warmup_iters, warmup_ratio

    @classmethod
    def from_config(cls, optimizer, lr_config, iters_per_epoch):
        cfg = dict(lr_config)
        assert cfg.pop('policy') == 'step', 'only the step policy of the CPR / P2P configs is built'
        return cls(optimizer['lr'], iters_per_epoch, **cfg)

    def regular_lr(self, epoch):
        return self.base_lr * self.gamma ** sum(1 for s in self.step if epoch >= s)

    def lr(self, it):
        reg = self.regular_lr(it // self.iters_per_epoch)
        if self.warmup is None or it >= self.warmup_iters:
            return reg
        if self.warmup == 'constant':
            return reg * self.warmup_ratio
        if self.warmup == 'exp':
            return reg * self.warmup_ratio ** (1 - it / self.warmup_iters)
        return reg * (1 - (1 - it / self.warmup_iters) * (1 - self.warmup_ratio))


# mixed precision: dtype hand-offs fused into the producing kernels (CPR_MIXED_FUSED_CAST=0: the separate torch passes of rounds 3-4, A/B)
FUSED_CAST = os.environ.get('CPR_MIXED_FUSED_CAST', '1') != '0'
# measurement switches (tests/report_mixed_precision_grads.py: where the mixed-precision gradient error comes from): the weight /
# data gradients of the mixed-precision step on the fp32 kernels; force = bf16 backward rules behind an fp32 recorded forward
MIXED_BF16 = dict(wgrad=os.environ.get('CPR_MIXED_WGRAD', 'bf16') != 'fp32', dgrad=os.environ.get('CPR_MIXED_DGRAD', 'bf16') != 'fp32',
                  dgrad1x1=os.environ.get('CPR_MIXED_DGRAD_1X1', 'bf16') != 'fp32',
                  dz16=os.environ.get('CPR_MIXED_DZ16', '1') != '0',
                  dgrad_s2=os.environ.get('CPR_MIXED_DGRAD_S2', 'bf16') != 'fp32',
                  mask_mode=os.environ.get('CPR_MIXED_MASK_MODE', '1') != '0', force=False)


class BackwardEngine:
    """The backward rules of the recorded forward -- shared by ``CprTrainer`` (gradients written straight into the views of
    its flat buffer, bucketed reducer, native optimizer) and by the autograd bridge (``autograd_bridge.py``: the same rules
    behind ``torch.autograd.Function``s, gradients handed to torch, so that ``loss.backward()`` / DDP / torch.optim drive the
    drop-in classes as they drive the reference's).  ``_g(p)`` is where the gradient of parameter ``p`` is written."""
    _sink, side, _mixed = None, None, False

    def __init__(self, model, two_streams=True):
        self.model = model
        dev = next(model.parameters()).device
        self._mixed, self._wide = False, {}     # set per step (bf16 compute mode = mixed precision)
        self._sink = None                       # None: p.grad (CprTrainer); dict: id(p) -> fresh tensor (autograd bridge)
        # weight gradients run on a second stream: they are off the critical path (nothing downstream of the backward
        # chain reads them) and the deep layers' launches are too small to fill 256 CUs on their own
        # The two streams only overlap when they sit on different HARDWARE queues (__init__.py: GPU_MAX_HW_QUEUES): the side stream
        # is chosen by a probe (ops.concurrent_stream), not by luck of the stream pool -- a shared queue costs 2 .. 15 % of a step.
        self.side = None
        if two_streams and dev.type == 'cuda':
            self.side, concurrent = ops.concurrent_stream(dev)
            if not concurrent:
                warnings.warn('no hardware queue left for the weight-gradient stream: the backward runs on one queue '
                              '(GPU_MAX_HW_QUEUES=%s; export GPU_MAX_HW_QUEUES=8 -- or import pointtinybenchmark_amd -- BEFORE the process '
                              'touches the device: the HIP runtime reads it once, and RCCL takes several queues)'
                              % os.environ.get('GPU_MAX_HW_QUEUES', 'unset'))

    def _g(self, p):
        """Where the gradient of ``p`` is written."""
        if self._sink is None:
            return p.grad
        t = self._sink.get(id(p))
        if t is None:
            t = self._sink[id(p)] = torch.empty(p.shape, device=p.device, dtype=torch.float32)
        return t

    def collect(self, params):
        """autograd bridge: the gradients written since the last call, in the order of ``params`` (None where no rule wrote
        one), visible to the current stream."""
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)
        out = []
        cur = torch.cuda.current_stream()
        for p in params:
            t = self._sink.pop(id(p), None)
            if t is not None:
                t.record_stream(cur)            # may have been allocated while the side stream was current
            out.append(t)
        return tuple(out)

    def begin_step(self):
        self._mixed = self.model.backbone.compute_dtype == torch.bfloat16 or MIXED_BF16['force']
        self._wide = {}

    def _done(self, p):
        """The gradient of ``p`` (and of everything before it in the trainer's flat order) has been enqueued."""

    # ------------------------------------------------------------------ segment API of the autograd bridge
    # (autograd_bridge.py: one torch.autograd.Function per segment; each forward returns (output, state) and the matching
    # backward consumes the state.  Every boundary carries a true gradient tensor.)
    def forward_stage(self, i, x):
        bb = self.model.backbone
        tape = []
        out = bb.run_stage(i, x, tape)
        assert tape, 'a stage segment is only built for trainable stages'
        return out, tape

    def backward_stage(self, tape, dout, need_in):
        cache = self.model.backbone._cache
        dx = dout
        for idx in range(len(tape) - 1, -1, -1):
            rec = tape[idx]
            dx = self._block_backward(cache, rec['block'], rec, dx, need_dx=(idx > 0 or need_in), keep=idx > 0,
                                      mask_in=idx > 0 and rec['block'].downsample is None)
        return dx if need_in else None

    def forward_laterals(self, xs):
        tape = []
        lat = self.model.neck.run_laterals(list(xs), tape)
        return lat[0], {r['level']: r for r in tape}

    def backward_laterals(self, recs, dlat, need):
        """need[i]: whether the gradient wrt the i-th input (stage start_level + i) is wanted -> list of gradients / None."""
        neck = self.model.neck
        d_stage = self._backward_laterals(neck, recs, dlat, need_dx_of=lambda stage: need[stage - neck.start_level])
        return [d_stage.get(i + neck.start_level) for i in range(len(need))]

    def forward_head_loss(self, lat0, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None, gt_true_bboxes=None):
        """FPN output conv (lazy) -> head -> losses: returns the loss vector of the loss kernels and the backward state."""
        from .layers import conv_gn
        neck, head = self.model.neck, self.model.bbox_head
        rec = dict(kind='out', level=0)
        lazy = [conv_gn(neck._cache, neck.fpn_convs[0], lat0, materialize=False, save=rec)]
        _, saved = self._forward_head(head, lazy, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore, gt_true_bboxes)
        return saved[self.loss_vector_key], (rec, saved)

    def backward_head_loss(self, state, upstream):
        rec, saved = state
        dz = self._backward_head(self.model.bbox_head, saved, upstream=upstream)
        return self._backward_out_conv(rec, dz)

    loss_vector_key = 'out5'       # CPRHead: (gt_loss, pos_loss, bag_acc, neg_loss, num_sample)

    def loss_dict(self, out):
        return self.model.bbox_head._loss_dict(out)


    # ------------------------------------------------------------------ helpers
    def _param_side(self, fn, *tensors):
        """Run the parameter-gradient work ``fn`` on the side stream, ordered after everything enqueued so far on the
        main stream.  ``tensors``: its inputs that may be released by the main stream before the side stream ran."""
        if self.side is None:
            return fn()
        self.side.wait_stream(torch.cuda.current_stream())
        for t in tensors:
            if t is not None:
                t.record_stream(self.side)
        with torch.cuda.stream(self.side):
            fn()

    def _f32(self, t, keep=False):
        """A recorded map as the fp32 backward kernels read it (a bf16 map of the mixed-precision forward is widened, exactly).
        keep: the same recorded tensor is read again by the NEXT backward rule (a block's input is the output of the block
        before it) -- the widened copy is held until then instead of being made twice."""
        if t is None or t.dtype == torch.float32:
            return t
        hit = self._wide.pop(id(t), None)
        w = hit if hit is not None else t.float()
        if keep:
            self._wide[id(t)] = w
        return w

    def _gn_conv_backward(self, rec, dz, relu, need_dx, dx_bf16=False):
        """Backward of conv -> GN (-> ReLU) given dz wrt the module output.  Writes the three parameter gradients;
        returns the gradient wrt the conv INPUT as the consumer saw it (after the producer's pending affine, if any)."""
        cm = rec['module']
        w, gn = cm.conv.weight, cm.gn
        assert cm.conv.bias is None
        # mixed precision: ONE bf16 copy of the gradient map feeds the bf16 weight and data gradients
        dgrad16 = need_dx and MIXED_BF16['dgrad'] and rec['raw'].dtype == torch.bfloat16 and cm.conv.stride[0] == 1 and \
            w.shape[0] % 64 == 0
        wgrad16 = MIXED_BF16['wgrad'] and rec['x'].dtype == torch.bfloat16 and rec['in_ab'] is None and ops.conv_wgrad_bf16_supported(
            rec['x'].shape, w.shape, cm.conv.stride[0], cm.conv.padding[0], maps_bf16=True)
        if rec['raw'].dtype == torch.bfloat16 and FUSED_CAST:
            # the GroupNorm backward reads the bf16 recorded map as it is and writes the bf16 rounding of its result itself (and the
            # fp32 map only when a fp32 kernel still reads it): no widening / narrowing passes around it
            need32 = (not wgrad16) or (need_dx and not dgrad16)
            draw, _, _, d16 = ops.gn_bwd(rec['raw'], dz, rec['a'], rec['b'], rec['mean'], rec['rstd'], gn.weight, relu,
                                         out_dgamma=self._g(gn.weight), out_dbeta=self._g(gn.bias),
                                         want16=wgrad16 or dgrad16, want32=need32)
        else:
            draw, _, _ = ops.gn_bwd(self._f32(rec['raw']), dz, rec['a'], rec['b'], rec['mean'], rec['rstd'], gn.weight, relu,
                                    out_dgamma=self._g(gn.weight), out_dbeta=self._g(gn.bias))
            d16 = draw.to(torch.bfloat16) if wgrad16 else None
        if wgrad16:
            x = rec['x']       # the weight gradient on the bf16 matrix pipe, straight from the recorded map
            gw = self._g(w)
            self._param_side(lambda: ops.conv_wgrad_bf16(d16, x, w.shape, out=gw), d16, x)
        else:
            x = self._f32(rec['x'])
            gw = self._g(w)
            self._param_side(lambda: ops.conv2d_wgrad(draw, x, w.shape, cm.conv.stride[0], cm.conv.padding[0],
                                                      in_ab=rec['in_ab'], in_relu=rec['in_relu'], out=gw), draw, x)
        if not need_dx:
            return None
        if dgrad16:
            # dx_bf16 (round 6): the caller hands the result to another bf16 GroupNorm backward, which reads a bf16 gradient map as it is
            return self._dgrad_bf16(draw if d16 is None else d16, w, cm.conv.padding[0],
                                    torch.bfloat16 if dx_bf16 else torch.float32)
        pt = ops.dgrad_pack(w, cm.conv.stride[0], cm.conv.padding[0])
        return ops.conv2d_dgrad(draw, pt, (rec['x'].shape[1], rec['x'].shape[2]), cm.conv.stride[0])

    @staticmethod
    def _dgrad_bf16(dy, w, padding, out_dtype=torch.float32):
        """Mixed precision: the data gradient of a stride-1 conv on the bf16 matrix pipe -- a forward conv of the bf16-rounded
        gradient map with the rotated weights (channels swapped, taps flipped, padding K-1-p), fp32 out.  The weight gradient
        next to it keeps reading the fp32 map."""
        if ops.PACK_BF16_KERNEL[0]:
            pc = ops.PackedConv.for_dgrad_bf16(w, padding)
        else:
            k = w.shape[2]
            wt = w.detach().flip(2, 3).permute(1, 0, 2, 3)
            pc = ops.PackedConv(wt, 1, k - 1 - padding, torch.bfloat16)
        return ops.conv2d(dy if dy.dtype == torch.bfloat16 else dy.to(torch.bfloat16), pc, out_dtype=out_dtype)

    # ------------------------------------------------------------------ CPR head
    @staticmethod
    def _forward_head(head, lazy, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore, gt_true_bboxes):
        tape, save = [], {}
        losses = head.forward_train_lazy(lazy, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore, gt_true_bboxes,
                                         tape=tape, save=save)
        save['tape'] = tape
        return losses, save

    def _backward_head(self, head, s, upstream=None):
        """upstream (5,): gradient of the caller's total wrt the forward's loss vector (autograd bridge); None = unit weights."""
        head_tape = s['tape']
        C = head.num_cls_out                # classifier outputs (num_classes, + 1 with out_bg_cls)
        cfg = head.loss_cfg
        J = s['lmap'].shape[-1]
        Jd = 4 if J <= 4 else (J + 31) // 32 * 32
        w_mil = head.loss_mil.loss_weight if cfg.get('with_mil_loss', True) else 0.0
        w_gt = cfg.get('gt_loss_weight', 1.0) if cfg.get('with_gt_loss', False) else 0.0
        w_neg = cfg.get('neg_loss_weight', 1.0) if cfg.get('with_neg', True) else 0.0
        fc = head.num_cls_fcs > 0
        plist = s.get('plist')              # grid generator / align_corners=True: the gather walks the forward's point list
        if s.get('general'):
            # any other loss option (probability type, binary_ins, AllPosLoss, bag policy, gt_loss_type, out_bg_cls, no MIL term):
            # the general kernels leave the bag-entry gradients un-gathered; the gather onto the logit map is a second call
            dmap, dbag = ops.cpr_loss_bwd_general(s['lmap'], s['neg_mask'], s['out5'], s['bag_logits'], s['valid'], s['labels'],
                                                  s['bag_ws'], s['bags'], s['centres'], s['ins_off'], C, Jd, w_mil, w_gt, w_neg,
                                                  gt_weight=s['gt_weight'], eps=head.loss_mil.eps, upstream=upstream,
                                                  prob_type=head.prob_type, norm_p=head.norm_p, binary_ins=head.binary_ins,
                                                  allpos=head.loss_mil.allpos, neg_from_gt=s['neg_from_gt'])
            if not fc and plist is None:
                ops.bag_gather_bwd(dbag, s['centers'], s['gt_img'], s['offsets'], dmap, s['stride'], s['radius_cells'])
        else:
            dmap, dbag = ops.cpr_loss_bwd(s['lmap'], s['neg_mask'], s['out5'], s['bag_logits'], s['valid'], s['labels'],
                                          s['bag_ws'], s['centers'], s['gt_img'], s['offsets'], s['ins_off'], C, s['stride'],
                                          w_mil, w_gt, w_neg, Jd, gt_weight=s['gt_weight'], eps=head.loss_mil.eps,
                                          upstream=upstream, radius_cells=s['radius_cells'], gather=not fc and plist is None)
        if fc:
            return self._backward_head_fc(head, s, dmap, dbag, J, Jd)
        dbias = None
        if plist is not None:
            # padding slots / dropped taps sampled the projection's BIAS (zero features through the Linear, cpr_head.py:323-324):
            # their share of the bag gradient goes to the classifier biases, not onto the map
            dbias = ops.bag_points_gather_bwd(dbag, plist['pts'], plist['code'], s['gt_img'], dmap, s['stride'], plist['align'],
                                              want_bias=plist.get('pad', False))
        if not head.ins_share_head_feat:
            return self._backward_head_two_towers(head, s, dmap, J, Jd, dbias)
        # ---- logit projection (cls_out ++ ins_out as one 1x1 conv over the un-normalised last tower layer)
        shared = head.ins_share_head_classifier
        wcat = head.cls_out.weight if shared else torch.cat([head.cls_out.weight, head.ins_out.weight], 0)
        wpad = torch.zeros((Jd, wcat.shape[1], 1, 1), device=wcat.device, dtype=torch.float32)
        wpad[:J, :, 0, 0] = wcat.detach()
        gw = ops.conv2d_wgrad(dmap, self._f32(s['feat']), wpad.shape, 1, 0, in_ab=s['ab'], in_relu=True)
        _, gb = ops.relu_bwd_colsum(dmap, None, want_g=False)
        if dbias is not None:
            gb[:J] += dbias
        self._g(head.cls_out.weight).copy_(gw[:C, :, 0, 0])
        self._g(head.cls_out.bias).copy_(gb[:C])
        if not shared:                      # (binary_ins: the instance classifier has 2 C rows)
            self._g(head.ins_out.weight).copy_(gw[C:J, :, 0, 0])
            self._g(head.ins_out.bias).copy_(gb[C:J])
        self._done(head.ins_out.bias if not shared else head.cls_out.bias)
        dz = ops.conv2d_dgrad(dmap, ops.dgrad_pack(wpad, 1, 0), (dmap.shape[1], dmap.shape[2]))
        # ---- tower, last layer first; layer 0 consumes the (un-activated) FPN output
        order = list(reversed(head_tape))
        for i, rec in enumerate(order):
            # mixed precision: between two layers of the tower the gradient map travels in bf16 (the data gradient writes it, the
            # GroupNorm backward below reads it: 6 bytes per element less over the three passes); the tower's input gradient stays fp32
            nxt = order[i + 1] if i + 1 < len(order) else None
            to16 = nxt is not None and self._mixed and MIXED_BF16['dz16'] and FUSED_CAST and nxt['raw'].dtype == torch.bfloat16
            dz = self._gn_conv_backward(rec, dz, relu=True, need_dx=True, dx_bf16=to16)
            self._done(rec['module'].conv.weight)
        return dz

    def _backward_head_two_towers(self, head, s, dmap, J, Jd, dbias=None):
        """ins_share_head_feat=False (cpr_head.py:992-1008,1037-1040,1061-1070): the class logits (channels [0, C) of the map the
        loss reads) are cls_out over the class tower's last layer, the instance logits (channels [C, 2C)) ins_out over the instance
        tower's.  Each projection is a 1x1 conv whose padded weight holds its classifier's rows and zeros elsewhere, so the shared
        gradient map feeds both weight / data gradients unchanged; the towers' gradients meet at the FPN output."""
        C = head.num_cls_out
        assert J == C + head.ins_out.weight.shape[0], 'two towers: [cls ++ ins] logits'
        _, gb = ops.relu_bwd_colsum(dmap, None, want_g=False)
        if dbias is not None:
            gb[:J] += dbias
        same = head.ins_out is head.cls_out                      # ins_share_head_classifier on two towers: one Linear, two inputs
        dz, first = None, True
        for name, lo, mod, feat, ab, tape in (('cls', 0, head.cls_out, s['feat'], s['ab'], s['tape']),
                                              ('ins', C, head.ins_out, s['ifeat'], s['iab'], s['ins_tape'])):
            n = mod.weight.shape[0]
            wpad = torch.zeros((Jd, mod.weight.shape[1], 1, 1), device=dmap.device, dtype=torch.float32)
            wpad[lo:lo + n, :, 0, 0] = mod.weight.detach()
            gw = ops.conv2d_wgrad(dmap, self._f32(feat), wpad.shape, 1, 0, in_ab=ab, in_relu=True)
            if same and not first:
                self._g(mod.weight).add_(gw[lo:lo + n, :, 0, 0])
                self._g(mod.bias).add_(gb[lo:lo + n])
            else:
                self._g(mod.weight).copy_(gw[lo:lo + n, :, 0, 0])
                self._g(mod.bias).copy_(gb[lo:lo + n])
            first = False
            if name == 'ins':
                self._done(head.ins_out.bias)
            d = ops.conv2d_dgrad(dmap, ops.dgrad_pack(wpad, 1, 0), (dmap.shape[1], dmap.shape[2]))
            for rec in reversed(tape):
                d = self._gn_conv_backward(rec, d, relu=True, need_dx=True)
                if name == 'ins':
                    self._done(rec['module'].conv.weight)
            dz = d if dz is None else ops.axpby(dz, d, 1.0, 1.0)
        return dz

    # ------------------------------------------------------------------ CPRHead with FC layers (num_cls_fcs > 0)
    def _fc_chain_backward(self, dlogit, acts, fcs, rows, touched):
        """Backward of [relu(fc_i(.))]* -> classifier rows over an NHWC block.  dlogit (N,H,W,Jd): gradient wrt the Jd-padded
        logits; acts = [x0, x1 .. xn] (input and every FC output); fcs: the nn.Linear layers; rows: [(classifier module, first
        logit channel)] -- the classifiers that read x_n, placed at their channels of the padded projection (other channels of
        dlogit meet zero rows).  Parameter gradients are written on first touch and accumulated after (``touched``: ids).
        Returns the gradient wrt x0."""
        def put(p, g):
            if id(p) in touched:
                self._g(p).add_(g)
            else:
                self._g(p).copy_(g)
                touched.add(id(p))
        Jd = dlogit.shape[-1]
        xn = acts[-1]
        hw = (xn.shape[1], xn.shape[2])
        wpad = torch.zeros((Jd, xn.shape[-1], 1, 1), device=dlogit.device, dtype=torch.float32)
        for mod, lo in rows:
            wpad[lo:lo + mod.weight.shape[0], :, 0, 0] = mod.weight.detach()
        gw = ops.conv2d_wgrad(dlogit, xn, wpad.shape, 1, 0)
        _, gb = ops.relu_bwd_colsum(dlogit, None, want_g=False)
        for mod, lo in rows:
            n = mod.weight.shape[0]
            put(mod.weight, gw[lo:lo + n, :, 0, 0])
            put(mod.bias, gb[lo:lo + n])
        # x_n is a ReLU output (n >= 1 FC layers): the data gradient comes back through that ReLU in the conv epilogue
        d = ops.conv2d_dgrad(dlogit, ops.dgrad_pack(wpad, 1, 0), hw, mask=xn)
        for i in range(len(fcs) - 1, -1, -1):
            fc, x = fcs[i], acts[i]
            w4 = fc.weight.detach()[:, :, None, None]
            put(fc.weight, ops.conv2d_wgrad(d, x, tuple(w4.shape), 1, 0)[:, :, 0, 0])
            put(fc.bias, ops.relu_bwd_colsum(d, None, want_g=False)[1])
            d = ops.conv2d_dgrad(d, ops.dgrad_pack(w4, 1, 0), hw, mask=x if i > 0 else None)
        return d

    def _backward_head_fc(self, head, s, dmap, dbag, J, Jd):
        """num_cls_fcs > 0 (cpr_head.py:999-1005,1055-1072): the classifiers read relu(fc(.)) of the normalised, activated tower
        output -- over the whole map for the negative-grid term (class path only: the instance logits of the map enter no loss),
        and over the bilinear SAMPLES of the features for the bags (the ReLUs do not commute with the interpolation, so the bag
        gradient reaches the feature map through the samples' taps, not through the logit map).  One chain per (path, tower);
        the FC / classifier gradients of the map and bag paths add up."""
        assert s['feat'].dtype == torch.float32, 'num_cls_fcs > 0 trains in the fp32 compute mode'
        C = head.num_cls_out
        acts, touched = s['fc'], set()
        G, K, _ = s['bag_logits'].shape
        dl_bag = torch.zeros((1, G * K, 1, Jd), device=dmap.device, dtype=torch.float32)
        dl_bag[0, :, 0, :J] = dbag.reshape(G * K, J)
        two = not head.ins_share_head_feat
        if two:
            towers = [('cls', s['feat'], s['tape'], list(head.cls_fcs), [(head.cls_out, 0)], acts['map'], acts['bag_cls']),
                      ('ins', s['ifeat'], s['ins_tape'], list(head.ins_fcs), [(head.ins_out, C)], None, acts['bag_ins'])]
        else:
            rows = [(head.cls_out, 0)] + ([] if head.ins_same_logits else [(head.ins_out, C)])
            towers = [('cls', s['feat'], s['tape'], list(head.cls_fcs), rows, acts['map'], acts['bag'])]
        dfeats = []
        for name, feat, tape, fcs, rows, a_map, a_bag in towers:
            if a_map is not None:
                dfeat = self._fc_chain_backward(dmap, a_map, fcs, rows, touched)
            else:
                dfeat = torch.zeros(tuple(feat.shape), device=dmap.device, dtype=torch.float32)
            ds = self._fc_chain_backward(dl_bag, a_bag, fcs, rows, touched)              # (1, G*K, 1, Cf)
            if s.get('plist') is not None:       # (features were sampled: padding slots / dropped taps hold zeros, no bias share)
                pl = s['plist']
                ops.bag_points_gather_bwd(ds.view(G, K, -1), pl['pts'], pl['code'], s['gt_img'], dfeat, s['stride'], pl['align'])
            else:
                ops.bag_gather_bwd(ds.view(G, K, -1), s['centers'], s['gt_img'], s['offsets'], dfeat, s['stride'], s['radius_cells'])
            dfeats.append((dfeat, tape))
        self._done((head.ins_fcs[0] if two else head.cls_fcs[0]).bias)       # classifiers and FC layers: complete (flat order)
        dz = None
        for dfeat, tape in dfeats:
            d = dfeat
            for rec in reversed(tape):
                d = self._gn_conv_backward(rec, d, relu=True, need_dx=True)
                self._done(rec['module'].conv.weight)
            dz = d if dz is None else ops.axpby(dz, d, 1.0, 1.0)
        return dz

    def _out_conv_backward(self, rec, dout_pad, n_out):
        """Backward of a biased output conv (nn.Conv2d) reading the un-normalised last tower layer with its GroupNorm
        affine (+ReLU) applied on load.  dout_pad (N,H,W,Cp): gradient wrt the conv output, channels padded to what the
        conv gradient kernels take (first n_out live).  Returns the gradient wrt the normalised, activated input."""
        conv = rec['conv']
        w = conv.weight
        Cp = dout_pad.shape[-1]
        wpad = torch.zeros((Cp,) + tuple(w.shape[1:]), device=w.device, dtype=torch.float32)
        wpad[:n_out] = w.detach()
        x, ab = self._f32(rec['x']), rec['in_ab']

        def param_grads():
            gw = ops.conv2d_wgrad(dout_pad, x, wpad.shape, conv.stride[0], conv.padding[0], in_ab=ab, in_relu=True)
            _, gb = ops.relu_bwd_colsum(dout_pad, None, want_g=False)
            self._g(w).copy_(gw[:n_out])
            self._g(conv.bias).copy_(gb[:n_out])
        self._param_side(param_grads, dout_pad, x)
        return ops.conv2d_dgrad(dout_pad, ops.dgrad_pack(wpad, conv.stride[0], conv.padding[0]),
                                (x.shape[1], x.shape[2]), conv.stride[0])

    # ------------------------------------------------------------------ FPN, backbone
    def _backward_neck(self, neck, neck_tape, dz):
        """Output conv, then the top-down chain from the finest lateral to the coarsest -> {stage: d(stage output)}."""
        lat_recs = {r['level']: r for r in neck_tape if r['kind'] == 'lateral'}
        out_recs = {r['level']: r for r in neck_tape if r['kind'] == 'out'}
        assert list(out_recs) == [0], 'num_outs == 1 (every shipped CPR config)'
        dlat = self._backward_out_conv(out_recs[0], dz)
        return self._backward_laterals(neck, lat_recs, dlat)

    def _backward_out_conv(self, rec, dz):
        """FPN output conv (3x3 + GN, no activation): dz wrt its normalised output -> gradient wrt the finest lateral sum."""
        dlat = self._gn_conv_backward(rec, dz, relu=False, need_dx=True)
        self._done(rec['module'].conv.weight)
        return dlat

    def _backward_laterals(self, neck, lat_recs, dlat, need_dx_of=None):
        """The top-down chain from the finest lateral to the coarsest -> {stage: d(stage output) or None}.
        need_dx_of(stage): whether the gradient wrt that backbone stage's output is wanted (default: the stage trains)."""
        d_stage = {}
        L = len(lat_recs)
        for i in range(L):
            rec = lat_recs[i]
            stage = i + neck.start_level
            need_dx = bool((need_dx_of or self._stage_trainable)(stage))
            d_stage[stage] = self._gn_conv_backward(rec, dlat, relu=False, need_dx=need_dx)
            self._done(rec['module'].conv.weight)
            if i + 1 < L:
                nxt = lat_recs[i + 1]['raw'].shape
                dlat = ops.upsample_add_bwd(dlat, tuple(nxt))
        return d_stage

    def _stage_trainable(self, stage):
        bb = self.model.backbone
        return any(p.requires_grad for p in getattr(bb, bb.res_layers[stage]).parameters())

    def _backward_backbone(self, bb, tape, d_stage):
        c = bb._cache
        dx = None            # gradient flowing down from the block above
        cur_stage = None
        for idx in range(len(tape) - 1, -1, -1):
            rec = tape[idx]
            blk, stage = rec['block'], rec['stage']
            if stage != cur_stage:       # entering a stage from above: its output also feeds an FPN lateral
                cur_stage = stage
                lat = d_stage.get(stage)
                if lat is not None:
                    dx = lat if dx is None else ops.axpby(dx, lat, 1.0, 1.0)
            assert dx is not None, 'no gradient reaches backbone stage %d' % stage
            need_dx = idx > 0
            # a block without a projection shortcut reads the output of the block below it and nothing else adds to that gradient
            dx = self._block_backward(c, blk, rec, dx, need_dx,
                                      mask_in=need_dx and blk.downsample is None and tape[idx - 1]['stage'] == stage)
        self._wide = {}      # nothing below the lowest trainable block reads a widened copy

    def _reads_bf16_only(self, conv, x, has_add, need_dx):
        """True when the backward rule of ``conv`` (input map x) reads nothing but the bf16 rounding of its output gradient: both of
        its gradients run on the bf16 matrix pipe (the conditions of _conv_bn_backward), so the producer need not write the fp32 map."""
        if not self._mixed:
            return False
        w = conv.weight
        k = conv.kernel_size[0]
        w16 = not w.requires_grad or (MIXED_BF16['wgrad'] and
                                      ops.conv_wgrad_bf16_supported(x.shape, w.shape, conv.stride[0], conv.padding[0],
                                                                    maps_bf16=x.dtype == torch.bfloat16))
        d16 = not need_dx or (MIXED_BF16['dgrad'] and conv.stride[0] == 1 and w.shape[0] % 64 == 0 and
                              (k == 3 and not has_add or k == 1 and MIXED_BF16['dgrad1x1'] and w.shape[1] % 64 == 0))
        return bool(w16 and d16)

    def _conv_bn_backward(self, cache, conv, bn, g, colsum, x, need_dx, mask=None, add=None, want_colsum=False, g16=None, want16=False,
                          need32=True):
        """conv -> folded eval-BN given g = d(pre-activation output) (un-scaled) and its column sums.  Parameter
        gradients go to the side stream; returns the data gradient wrt x -- with ``mask`` (x itself, when x is the
        output of a fused ReLU) already taken through that ReLU, with ``add`` summed in, with ``want_colsum`` as
        (gradient, column sums): all three ride in the conv epilogue (fp32 kernels) or in one streaming pass over the result
        (bf16 data gradients).  ``x`` / ``mask`` are the maps AS RECORDED (bf16 in the mixed-precision step): they are widened only
        where an fp32 kernel reads them.  g16: the bf16 rounding of g when the producer already wrote it; want16 (with
        want_colsum): also return the bf16 rounding of the result -> (gradient, column sums, gradient16 | None).  need32=False (the
        consumer reads only the bf16 rounding, _reads_bf16_only): a bf16 data gradient with a mask then runs in the kernel's mask mode
        -- ReLU backward, bf16 rounding and column sums in its epilogue, no fp32 map, no streaming pass -- and returns
        (None, column-sum partials, gradient16).  g may be None when g16 is given and both gradients of this conv are bf16."""
        scale, _ = folded_bn(cache, bn)
        inv_sigma = cache.get(('bn_is', id(bn)), [bn.running_var],
                              lambda: ops.bn_fold(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, True)[2])
        w = conv.weight
        k = conv.kernel_size[0]
        # mixed precision: ONE bf16 copy of the gradient map feeds the bf16 weight gradient and the bf16 data gradient (round 6: the
        # stride-1 1x1 layers too -- their fp32 form was 13 % of the step's kernel time, profiles/round5_train_cfg4_kernel_stats.csv)
        w16 = w.requires_grad and self._mixed and MIXED_BF16['wgrad'] and \
            ops.conv_wgrad_bf16_supported(x.shape, w.shape, conv.stride[0], conv.padding[0], maps_bf16=x.dtype == torch.bfloat16)
        d16 = need_dx and self._mixed and MIXED_BF16['dgrad'] and conv.stride[0] == 1 and w.shape[0] % 64 == 0 and \
            (k == 3 and add is None or k == 1 and MIXED_BF16['dgrad1x1'] and w.shape[1] % 64 == 0)
        if g16 is None and (w16 or d16):
            g16 = g.to(torch.bfloat16)
        if w.requires_grad:
            aff = bn.weight.requires_grad

            def param_grads():
                cs = colsum            # (TilePartials of a conv epilogue are summed inside bn_fold_bwd)
                gw = self._g(w)
                if w16:
                    ops.conv_wgrad_bf16(g16, x, w.shape, out=gw, stride=conv.stride[0])    # (x: the bf16 recorded map, or a widened copy that rounds back exactly)
                else:
                    ops.conv2d_wgrad(g, self._f32(x), w.shape, conv.stride[0], conv.padding[0], out=gw)
                ops.bn_fold_bwd(gw, w, scale, bn.running_mean, inv_sigma, cs,
                                out_dgamma=self._g(bn.weight) if aff else None, out_dbeta=self._g(bn.bias) if aff else None)
            # x may be a widened fp32 temporary of the mixed-precision step that the main stream frees right after this call
            self._param_side(param_grads, g, colsum, g16, x)
        if not need_dx:
            return None

        def finish(dx, add=None):
            """sum / mask / column sums / bf16 rounding as one streaming pass over the data gradient."""
            if mask is None and not want_colsum:
                return dx if add is None else ops.axpby(dx, add, 1.0, 1.0)
            keep = mask is not None or add is not None
            r = ops.relu_bwd_colsum(dx, mask, want_g=keep, want16=want16, add=add)
            dxm = r[0] if keep else dx
            if not want_colsum:
                return dxm
            if want16:
                return dxm, r[1], r[2]
            return dxm, r[1]
        if d16:
            # mixed precision: the data gradient on the bf16 matrix pipe (a forward conv of the bf16-rounded gradient map with the
            # rotated, BN-scaled weights, fp32 out), the residual add, the ReLU mask (the bf16 recorded map as it is) and the column
            # sums as streaming passes over the (small) result
            def pack16():
                if ops.PACK_BF16_KERNEL[0]:
                    return ops.PackedConv.for_dgrad_bf16(w, conv.padding[0], scale=scale)
                wt = (w.detach() * scale[:, None, None, None]).flip(2, 3).permute(1, 0, 2, 3)
                return ops.PackedConv(wt, 1, k - 1 - conv.padding[0], torch.bfloat16)
            pc16 = cache.get(('dgrad16', id(conv)), [w, bn.weight, bn.running_var], pack16,
                             refresh=(lambda pc: ('pack', w, ('bn', id(bn)), pc, 1)) if ops.PACK_BF16_KERNEL[0] else None)
            if MIXED_BF16['mask_mode'] and not need32 and mask is not None and add is None and want_colsum and want16 and \
                    mask.dtype == torch.bfloat16 and ops.conv2d_bf16_mask_slots(g16.shape, pc16) > 0:
                dx16, part = ops.conv2d(g16, pc16, residual=mask, res_mask=True, colsum=True)
                return None, part, dx16
            if MIXED_BF16['mask_mode'] and mask is not None and add is not None and want_colsum and want16 and \
                    mask.dtype == torch.bfloat16 and ops.conv2d_bf16_mask_slots(g16.shape, pc16, fused_add=True) > 0:
                # the block-boundary gradient: shortcut sum, mask, both roundings and the column sums in the data gradient's epilogue
                dx32, dx16, part = ops.conv2d_dgrad_bf16_fused(g16, pc16, mask, add)
                return dx32, part, dx16
            return finish(ops.conv2d(g16, pc16, out_dtype=torch.float32), add)
        if need_dx and self._mixed and MIXED_BF16['dgrad'] and MIXED_BF16['dgrad_s2'] and conv.stride[0] == 2 and k in (1, 3) and \
                conv.padding[0] == k // 2 and w.shape[0] % 64 == 0 and w.shape[1] % 64 == 0 and ops._PHASED[0]:
            # mixed precision, the strided layers of a stage's first block (round 6): the four parity sub-convolutions of the phase-
            # decomposed data gradient on the bf16 pipe (fp32 out, scattered / summed in fp32 as before)
            pt16 = cache.get(('dgrad16s2', id(conv)), [w, bn.weight, bn.running_var],
                             lambda: ops.dgrad_pack(w, 2, conv.padding[0], scale=scale, dtype=torch.bfloat16))
            if g16 is None:
                g16 = g.to(torch.bfloat16)
            return finish(pt16(g16, (x.shape[1], x.shape[2]), add=add))
        pt = cache.get(('dgrad', id(conv)), [w, bn.weight, bn.running_var],
                       lambda: ops.dgrad_pack(w, conv.stride[0], conv.padding[0], scale=scale),
                       refresh=lambda q: ('pack32', w, ('bn', id(bn)), q, 1) if isinstance(q, ops.PackedConv) and
                       q.dtype == torch.float32 and w.dtype == torch.float32 and w.is_contiguous() else None)
        if WINO_DGRAD[0] and k == 3 and conv.stride[0] == 1 and add is None and \
                (mask is not None or want_colsum) and ops.wino_eligible(pt, x.shape[1], x.shape[2], torch.float32):
            # a 3x3 stride-1 data gradient with a mask / column-sum epilogue would run the direct kernel (2.25x the multiplies of
            # the Winograd launch the plain form gets); the epilogue as one streaming pass over the result is cheaper
            return finish(ops.conv2d_dgrad(g, pt, (x.shape[1], x.shape[2]), 1))
        if mask is not None and add is not None:
            # the fp32 epilogue has one extra operand: the sum rides in it, mask / column sums / bf16 rounding are the streaming pass
            return finish(ops.conv2d_dgrad(g, pt, (x.shape[1], x.shape[2]), conv.stride[0], add=add))
        r = ops.conv2d_dgrad(g, pt, (x.shape[1], x.shape[2]), conv.stride[0], mask=self._f32(mask), add=add, colsum=want_colsum)
        if want_colsum and want16:
            return r[0], r[1], None
        return r

    def _block_backward(self, cache, blk, rec, dout, need_dx, keep=None, mask_in=False):
        """dout: the gradient of the block's output -- or the triple (g, column sums, bf16 g | None) when the block above has already
        taken it through this block's output ReLU.  mask_in (the caller guarantees that x is the output of the block whose backward
        comes next and that nothing else adds to its gradient): a block without a projection shortcut returns that triple for the
        block below -- shortcut sum, ReLU mask, column sums and bf16 rounding leave with the data gradient (round 6: they were an
        axpby launch here plus a pass of their own at the top of the next call)."""
        # the recorded maps are handed on as recorded (bf16 in the mixed-precision step): the ReLU masks are read as they are, the bf16
        # gradient kernels read x as it is, and only an fp32 fallback kernel widens what it reads (round 6: rounds 3-5 widened every
        # recorded map up front -- 5.6 % of the step's kernel time in torch copy kernels)
        mixed = self._mixed
        x = rec['x']
        if isinstance(dout, tuple):
            g3, cs3, g3h = dout
        else:
            r3 = ops.relu_bwd_colsum(dout, rec['out'], want16=mixed)          # also the shortcut gradient
            g3, cs3, g3h = r3 if mixed else (r3[0], r3[1], None)
        o1 = rec['o1']
        if blk.kind == 'bottleneck':
            o2 = rec['o2']
            g2, cs2, g2h = self._conv_bn_backward(cache, blk.conv3, blk.bn3, g3, cs3, o2, True, mask=o2, want_colsum=True, g16=g3h, want16=True,
                                                  need32=not self._reads_bf16_only(blk.conv2, o1, False, True))
            del o2
            g1, cs1, g1h = self._conv_bn_backward(cache, blk.conv2, blk.bn2, g2, cs2, o1, True, mask=o1, want_colsum=True, g16=g2h, want16=True,
                                                  need32=not self._reads_bf16_only(blk.conv1, x, blk.downsample is None, need_dx))
            last = blk.bn3
        else:
            g1, cs1, g1h = self._conv_bn_backward(cache, blk.conv2, blk.bn2, g3, cs3, o1, True, mask=o1, want_colsum=True, g16=g3h, want16=True,
                                                  need32=not self._reads_bf16_only(blk.conv1, x, blk.downsample is None, need_dx))
            last = blk.bn2
        del o1
        self._done(last.bias if last.bias.requires_grad else blk.conv2.weight)
        if blk.downsample is not None:     # the shortcut conv sees the same g3 (no activation on that branch)
            dx = self._conv_bn_backward(cache, blk.conv1, blk.bn1, g1, cs1, x, need_dx, g16=g1h)
            dx = self._conv_bn_backward(cache, blk.downsample[0], blk.downsample[1], g3, cs3, x, need_dx, add=dx, g16=g3h)
            tail = blk.downsample[1].bias if blk.downsample[1].bias.requires_grad else blk.downsample[0].weight
        elif mask_in and need_dx:
            dx = self._conv_bn_backward(cache, blk.conv1, blk.bn1, g1, cs1, x, True, mask=x, add=g3, want_colsum=True, g16=g1h, want16=mixed)
            dx = dx if mixed else (dx[0], dx[1], None)
            tail = blk.bn1.bias if blk.bn1.bias.requires_grad else blk.conv1.weight
        else:
            dx = self._conv_bn_backward(cache, blk.conv1, blk.bn1, g1, cs1, x, need_dx, add=g3, g16=g1h)
            tail = blk.bn1.bias if blk.bn1.bias.requires_grad else blk.conv1.weight
        self._done(tail)
        return dx


class CprTrainer(BackwardEngine):
    def __init__(self, model, lr=0.02, momentum=0.9, weight_decay=1e-4, max_norm=35.0, bucket_mb=25.0, group=None,
                 two_streams=True, force_collectives=False, schedule=None, reducer='all_reduce', reducer_timing=False,
                 min_bucket_mb=4.0, tail_bucket_mb=1.0):
        """schedule: a StepLrSchedule (or any object with ``lr(iteration)``); ``lr`` is then only the fallback of
        ``step(lr=...)``.  Constructing the trainer re-homes every trainable parameter: ``p.data`` becomes a view of ONE
        flat buffer (``flat_p``) and ``p.grad`` a view of ``flat_g``; do not re-bind them afterwards (``model.to()``,
        ``.float()``, ``p.data = ...``) -- ``step`` checks and refuses.  ``state_dict()`` below returns detached clones."""
        BackwardEngine.__init__(self, model, two_streams)
        self.schedule = schedule
        self.lr, self.momentum, self.weight_decay, self.max_norm = lr, momentum, weight_decay, max_norm
        order = self._backward_order()
        seen = {id(p) for p in order}
        missing = [n for n, p in model.named_parameters() if p.requires_grad and id(p) not in seen]
        assert not missing, 'trainable parameters without a backward rule: %s' % missing[:8]
        self.params = order
        dev = order[0].device
        n = sum(p.numel() for p in order)
        self.flat_p = torch.empty((n,), device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros((n,), device=dev, dtype=torch.float32)
        self.flat_m = torch.zeros((n,), device=dev, dtype=torch.float32)
        self._launch = None      # stream the gradient collectives are issued from (two-stream backward on a device)
        self.offset, off = {}, 0
        for p in order:
            k = p.numel()
            self.flat_p[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat_p[off:off + k].view(p.shape)
            p.grad = self.flat_g[off:off + k].view(p.shape)
            self.offset[id(p)] = (off, off + k)
            off += k
        mb = (1 << 20) / 4
        self.buckets = GradBuckets(self.flat_g, max(1, int(bucket_mb * mb)), group, force=force_collectives,
                                   reducer=reducer, timing=reducer_timing,
                                   bounds=layout_buckets([p.numel() for p in order], max(1, int(bucket_mb * mb)),
                                                         max(1, int(min_bucket_mb * mb)), max(1, int(tail_bucket_mb * mb))))
        self.norm2 = torch.zeros((1,), device=dev, dtype=torch.float64)
        self._ws = torch.empty((1024,), device=dev, dtype=torch.float64)
        self._pack_caches = None
        self.steps = 0
        self.group = group
        self.sync_initial_state()
        bump_weight_epoch()

    def sync_initial_state(self, src=0):
        """What MMDistributedDataParallel does at construction (T/mmdet/apis/train.py:75-86): every rank starts from rank
        ``src``'s module state -- parameters (trainable: the flat buffer in one collective; frozen ones singly) and buffers
        (BatchNorm running statistics).  Without it, unseeded per-process initialisation leaves the replicas on different
        weights for ever, because only gradients are averaged."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(self.group) == 1:
            return
        dist.broadcast(self.flat_p, src=src, group=self.group)
        dist.broadcast(self.flat_m, src=src, group=self.group)
        flat_ids = set(self.offset)
        for t in list(self.model.parameters()) + list(self.model.buffers()):
            if id(t) not in flat_ids and t.numel() > 0:
                dist.broadcast(t.data, src=src, group=self.group)

    def check_bindings(self):
        """A parameter that no longer aliases the flat buffers would be updated into stale memory: refuse."""
        lo, hi = self.flat_p.data_ptr(), self.flat_p.data_ptr() + self.flat_p.numel() * 4
        for p in (self.params[0], self.params[len(self.params) // 2], self.params[-1]):
            assert lo <= p.data_ptr() < hi and p.grad is not None and \
                self.flat_g.data_ptr() <= p.grad.data_ptr() < self.flat_g.data_ptr() + self.flat_g.numel() * 4, \
                'a parameter was re-bound after CprTrainer took it over (model.to() / .float() / p.data = ...): ' \
                'build the trainer AFTER moving the model and load weights with load_state_dict (in place)'

    def state_dict(self):
        """Detached clones of the model's tensors (saving the views would serialise the whole flat storage per key)."""
        return {k: v.detach().clone() for k, v in self.model.state_dict().items()}

    # ------------------------------------------------------------------ parameter order = gradient completion order
    def _backward_order(self):
        m = self.model
        head, neck, bb = m.bbox_head, m.neck, m.backbone
        out = []

        def add(*ps):
            for p in ps:
                if p is not None and p.requires_grad and all(p is not q for q in out):
                    out.append(p)
        self._head_param_order(head, add)
        for cm in neck.fpn_convs:
            add(cm.gn.weight, cm.gn.bias, cm.conv.weight)
        for cm in neck.lateral_convs:
            add(cm.gn.weight, cm.gn.bias, cm.conv.weight)
        for name in reversed(bb.res_layers):
            for blk in reversed(list(getattr(bb, name))):
                if blk.kind == 'bottleneck':
                    add(blk.conv3.weight, blk.bn3.weight, blk.bn3.bias)
                add(blk.conv2.weight, blk.bn2.weight, blk.bn2.bias, blk.conv1.weight, blk.bn1.weight, blk.bn1.bias)
                if blk.downsample is not None:
                    add(blk.downsample[0].weight, blk.downsample[1].weight, blk.downsample[1].bias)
        return out      # a trainable stem (frozen_stages < 0) has no backward rule and trips the constructor's check

    @staticmethod
    def _head_param_order(head, add):
        add(head.cls_out.weight, head.cls_out.bias, head.ins_out.weight, head.ins_out.bias)
        for fc in list(reversed(list(getattr(head, 'cls_fcs', [])))) + list(reversed(list(getattr(head, 'ins_fcs', [])))):
            add(fc.weight, fc.bias)                                    # num_cls_fcs > 0: weight before bias, ins_fcs[0] last
        for cm in reversed(list(head.cls_convs)):
            add(cm.gn.weight, cm.gn.bias, cm.conv.weight)
        for cm in reversed(list(getattr(head, 'ins_convs', []))):      # ins_share_head_feat=False: the second tower, last
            add(cm.gn.weight, cm.gn.bias, cm.conv.weight)

    def _done(self, p):
        """The gradient of ``p`` (and of everything before it in the flat order) has been enqueued."""
        end = self.offset[id(p)][1]
        if self.side is not None and self.buckets.would_launch(end):
            # The collective must see the gradients both streams wrote.  The process group orders its own stream behind the
            # stream that is current when the collective is issued: issue it from a third stream that waits for the two, so
            # NEITHER compute stream stalls (until round 5 the main stream waited for the side stream here, 20-37 times per
            # backward, i.e. the data-gradient chain could idle behind the weight gradients at every bucket boundary).
            if self._launch is None:
                self._launch = torch.cuda.Stream(device=self.flat_g.device)
            self._launch.wait_stream(torch.cuda.current_stream())
            self._launch.wait_stream(self.side)
            with torch.cuda.stream(self._launch):
                self.buckets.ready(end)
            return
        self.buckets.ready(end)

    # ------------------------------------------------------------------ forward (recorded) + backward
    def forward_backward(self, img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None, gt_true_bboxes=None):
        """-> dict of losses (device scalars, identical to BasicLocator.forward_train); parameter .grad filled (summed
        over ranks once ``step`` has waited for the buckets)."""
        m = self.model
        head, neck, bb = m.bbox_head, m.neck, m.backbone
        # bf16 compute mode = mixed precision (the reference analogue is mmcv's Fp16OptimizerHook, mmdet/apis/train.py:116-119): the
        # recorded forward runs on the bf16 kernels, the backward kernels are fp32 and read the recorded (bf16-rounded) maps
        # through ``_f32`` just in time; weights, gradients and the optimizer state stay fp32.
        batch_input_shape = tuple(img[0].size()[-2:])
        for meta in img_metas:
            meta['batch_input_shape'] = batch_input_shape
        self.begin_step()    # mixed-precision flag; _wide: id(recorded bf16 map) -> its fp32 copy while a second reader is to come
        bb_tape, neck_tape = [], []
        feats = bb(img, tape=bb_tape)
        lazy = neck.forward_lazy(feats, tape=neck_tape)
        losses, saved = self._forward_head(head, lazy, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore, gt_true_bboxes)
        self.buckets.reset()
        self.buckets.start_step()
        dz = self._backward_head(head, saved)              # gradient wrt the (normalised) FPN output
        d_stage = self._backward_neck(neck, neck_tape, dz)
        self._backward_backbone(bb, bb_tape, d_stage)
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)
        return losses

    # ------------------------------------------------------------------ optimizer
    def step(self, lr=None):
        """Wait for the gradient buckets, clip by the global norm, SGD-momentum update of every trainable parameter.
        lr: explicit override; else the schedule's value for this iteration; else the constructor's constant."""
        self.check_bindings()
        if lr is None and self.schedule is not None:
            lr = self.schedule.lr(self.steps)
        self.last_lr = self.lr if lr is None else lr
        grad_scale = self.buckets.finish()
        if self.max_norm and self.max_norm > 0:
            ops.grad_sumsq(self.flat_g, self.norm2, self._ws, accumulate=False)
        ops.sgd_step(self.flat_p, self.flat_g, self.flat_m, self.norm2, self.lr if lr is None else lr, self.momentum,
                     self.weight_decay, self.max_norm or 0.0, grad_scale, first=self.steps == 0)
        self.steps += 1
        bump_weight_epoch()
        # the folds / bf16 packs the last step registered are recomputed in place now, in two launches, instead of lapsing (round 6)
        if self._pack_caches is None:
            from .layers import _PackCache
            self._pack_caches = [m._cache for m in self.model.modules() if isinstance(getattr(m, '_cache', None), _PackCache)]
        for c in self._pack_caches:
            c.refresh_all()

    def grad_norm(self):
        """Global L2 norm of the (rank-averaged) gradient the last ``step`` clipped with (host sync: logging only)."""
        return float(self.norm2.sqrt()) / self.buckets.world_size

    def train_step(self, data, lr=None):
        """One optimisation step; returns BasicLocator.train_step's dict (base.py:214-247)."""
        losses = self.forward_backward(**data)
        self.step(lr)
        loss, log_vars = self.model._parse_losses(losses)
        return dict(loss=loss, log_vars=log_vars, num_samples=len(data['img_metas']))


class P2PHeadRules:
    """Head rules of BasicLocator(P2PHead) (BASELINE.json configs[3]; T/configs2/**/p2p/*.py): two towers, 3x3 output convs
    with bias, sigmoid-focal + SmoothL1 loss on the device Hungarian assignment (the assignment itself carries no gradient,
    as in the reference: it works on detached costs, hungarian_assigner.py:214-236).  Mixed into a BackwardEngine."""
    loss_vector_key = 'out'        # P2PHead: (B, 2) = (loss_cls, loss_pts) per image

    def loss_dict(self, out):
        B = out.shape[0]
        return {'loss_cls': [out[b, 0] for b in range(B)], 'loss_pts': [out[b, 1] for b in range(B)]}

    @staticmethod
    def _head_param_order(head, add):
        add(head.reg_out.weight, head.reg_out.bias)
        for cm in reversed(list(head.reg_convs)):
            add(cm.gn.weight, cm.gn.bias, cm.conv.weight)
        add(head.cls_out.weight, head.cls_out.bias)
        for cm in reversed(list(head.cls_convs)):
            add(cm.gn.weight, cm.gn.bias, cm.conv.weight)

    @staticmethod
    def _forward_head(head, lazy, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore, gt_true_bboxes):
        assert len(lazy) == 1 and head.num_points == 1, 'single level, one point per cell (the shipped P2P configs)'
        assert lazy[0][0].dtype == torch.float32, 'the mixed-precision step covers the CPR locator; P2PNet trains in fp32'
        raw, (a, b) = lazy[0]
        x = ops.gn_apply(raw, a, b, relu=False)                   # FPN output, materialised once for the two towers
        cls_tape, reg_tape, save = [], [], {}
        cls = ops.as_nchw(head._tower(head.cls_convs, head.cls_out, x, tape=cls_tape))
        reg = ops.as_nchw(head._tower(head.reg_convs, head.reg_out, x, tape=reg_tape))
        losses = head.loss([cls], [reg], gt_bboxes, gt_labels, img_metas, gt_bboxes_ignore=gt_bboxes_ignore, save=save)
        save.update(cls_tape=cls_tape, reg_tape=reg_tape, hw=tuple(x.shape[1:3]))
        return losses, save

    def _backward_head(self, head, s, upstream=None):
        from .dense_heads.p2p_head import _get
        lc, lr = head.loss_cls_cfg, head.loss_reg_cfg
        B, M, C = s['cls'].shape
        H, W = s['hw']
        Cp = 4 if C <= 4 else (C + 31) // 32 * 32
        dcls, dreg = ops.p2p_loss_bwd(s['cls'], s['pred'], s['gt_inds'], s['gt_pts'], s['gt_labels'], s['gt_start'],
                                      lc.get('alpha', 0.25), lc.get('gamma', 2.0), lr.get('beta', 1.0),
                                      _get(head.train_cfg, 'pos_weight', 1.0), _get(head.train_cfg, 'neg_weight', 1.0),
                                      head.reg_norm, lc.get('loss_weight', 1.0), lr.get('loss_weight', 1.0),
                                      head.pts_gamma, Cp, 4, upstream=upstream, cls_mode=head.cls_mode, reg_mode=head.reg_mode)
        dz = None
        for tape, dout, n_out, last in ((s['reg_tape'], dreg.view(B, H, W, 4), 2, head.reg_convs[0]),
                                        (s['cls_tape'], dcls.view(B, H, W, Cp), C, head.cls_convs[0])):
            d = self._out_conv_backward(tape[-1], dout, n_out)
            self._done(tape[-1]['conv'].bias)
            for rec in reversed(tape[:-1]):
                d = self._gn_conv_backward(rec, d, relu=True, need_dx=True)
                self._done(rec['module'].conv.weight)
            dz = d if dz is None else ops.axpby(dz, d, 1.0, 1.0)
        return dz


class P2PTrainer(P2PHeadRules, CprTrainer):
    """The same training step for BasicLocator(P2PHead)."""


class P2PBackwardEngine(P2PHeadRules, BackwardEngine):
    """The P2PNet backward rules without the trainer's flat buffers (autograd bridge)."""
