"""``loss.backward()`` for the drop-in classes: the recorded forward / HIP backward of ``training.BackwardEngine`` behind
``torch.autograd.Function``s, so that the reference's own training driver runs them unmodified --
``BaseDetector.train_step`` returns a loss WITH a graph (T/mmdet/models/detectors/base.py:214-247), mmcv's ``OptimizerHook``
calls ``loss.backward()`` + ``clip_grad_norm_`` + ``optimizer.step()``, and ``MMDistributedDataParallel`` / torch DDP
(T/mmdet/apis/train.py:75-83) hooks the parameters' gradient accumulators and all-reduces buckets while the backward runs.

One Function per module boundary at which a TRUE gradient tensor exists (the lazy GroupNorm hand-offs inside the neck / head
chain have none: a raw map travels with a pending affine):

    image -> stem + frozen stages (no graph)
          -> _StageFn (layer2) -> _StageFn (layer3) -> _StageFn (layer4)          d(stage output) between them
          -> _LateralsFn: lateral 1x1 convs + GN + top-down adds -> finest lateral sum        d(lateral sum)
          -> _HeadLossFn: FPN 3x3 output conv + head towers + projection + point stage + losses -> the loss vector

Every Function takes its trainable parameters as explicit inputs, so they sit in the autograd graph as leaves: gradient
accumulators fire per Function -- the head's 9.4 MB of gradients (63 % of the backward's time) are final and on the wire while
the neck and backbone still compute, as with the trainer's own buckets.  The arithmetic is the trainer's kernel for kernel:
``p.grad`` after ``loss.backward()`` is BIT-equal to ``CprTrainer.forward_backward``'s (tests/test_gpu_autograd.py), which is
pinned to ``loss.backward()`` through the reference's own modules by tests/golden/cpr_grads_*.npz.

Both compute modes (round 5: the bf16 mode = mixed precision, the reference analogue being mmcv's ``Fp16OptimizerHook`` around an
unmodified ``loss.backward()``, T/mmdet/apis/train.py:116-119 -- see ``Bridge.carrier`` for how bf16 maps cross the Function
boundaries), every CPRHead option set that runs forward (``CPRHead.train_step_supported``), one FPN output level, a frozen stem.  Anything else keeps the forward-only path and warns once."""
import os
import warnings

import torch
from torch.autograd.function import once_differentiable

from .training import BackwardEngine, P2PBackwardEngine


class _GtPack:
    """The non-tensor arguments of the head's loss (lists of per-image tensors, metas) carried through Function.apply."""

    def __init__(self, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore, gt_true_bboxes):
        self.img_metas, self.gt_bboxes, self.gt_labels = img_metas, gt_bboxes, gt_labels
        self.gt_bboxes_ignore, self.gt_true_bboxes = gt_bboxes_ignore, gt_true_bboxes


class Bridge:
    """Per-model state of the bridge: the backward engine (gradients into fresh tensors instead of ``p.grad``) and the
    parameter lists each Function takes."""

    def __init__(self, model, engine=None):
        """engine: an object with the segment API of training.BackwardEngine (tests drive the Functions with a CPU stand-in
        under gloo: tests/test_host_cpu.py); default: the HIP backward rules for the model's head."""
        self.model = model
        head = model.bbox_head
        if engine is None:
            two = os.environ.get('CPR_TRAIN_STREAMS', '2') != '1'
            engine = (P2PBackwardEngine if type(head).__name__ == 'P2PHead' else BackwardEngine)(model, two_streams=two)
        if getattr(engine, '_sink', None) is None:
            engine._sink = {}           # gradients go to fresh tensors handed to torch (any engine, also a caller's own)
        self.engine = engine
        bb, neck = model.backbone, model.neck
        self._maps = {}         # bf16 compute mode: data_ptr of an fp32 carrier -> the bf16 map it stands for (see carrier())
        self.stage_params = [[p for p in getattr(bb, name).parameters() if p.requires_grad] for name in bb.res_layers]
        self.lateral_params = [p for cm in neck.lateral_convs for p in cm.parameters() if p.requires_grad]
        self.head_params = [p for p in list(neck.fpn_convs[0].parameters()) + list(head.parameters()) if p.requires_grad]
        self.signature = signature(model)


    # ---- mixed precision (bf16 compute mode) behind the same Functions.  The recorded maps are bf16, the gradients the backward
    # kernels hand from segment to segment are fp32 -- and autograd casts a gradient to the dtype of the output it belongs to, so
    # a bf16 map cannot be a Function output without rounding every boundary gradient to 8 bits.  The boundary therefore carries
    # an fp32 STAND-IN of the map's shape (all strides 0: one element of storage, never read or written); the Functions look
    # the real map up here.  fp32 maps pass through unchanged.
    def carrier(self, real):
        if real.dtype == torch.float32:
            return real
        ph = torch.empty_strided(tuple(real.shape), (0,) * real.dim(), dtype=torch.float32, device=real.device)
        self._maps[ph.data_ptr()] = real
        return ph

    def real(self, t):
        if t.dtype == torch.float32 and self._maps and t.dim() > 0 and all(s == 0 for s in t.stride()):
            return self._maps.get(t.data_ptr(), t)
        return t


def signature(model):
    return tuple(p.requires_grad for p in model.parameters())


def unsupported_reason(model, gt_bboxes=None, gt_labels=None):
    """None when ``forward_train`` can be differentiable, else why not."""
    bb, neck, head = model.backbone, model.neck, model.bbox_head
    if neck is None or type(neck).__name__ != 'FPN' or len(neck.fpn_convs) != 1:
        return 'needs an FPN neck with num_outs == 1 (every shipped CPR / P2P config)'
    if bb.compute_dtype != torch.float32 and type(head).__name__ != 'CPRHead':
        return 'the mixed-precision step (bf16 compute mode) covers the CPR locator; P2PNet trains in fp32'
    if any(p.requires_grad for m in (bb.conv1, bb.bn1) for p in m.parameters()):
        return 'a trainable stem (frozen_stages < 0) has no backward rule'
    if tuple(bb.out_indices) != tuple(range(len(bb.res_layers))):
        return 'out_indices must name every stage'
    for name in bb.res_layers:
        if len({blk.conv1.weight.requires_grad for blk in getattr(bb, name)}) > 1:
            return 'stage %s is partly frozen' % name
    kind = type(head).__name__
    if kind == 'CPRHead':
        R = 1
        if gt_bboxes is not None and gt_labels is not None and len(gt_labels) and len(gt_labels[0]):
            R = max(1, int(gt_bboxes[0].shape[0]) // int(len(gt_labels[0])))     # (every image brings num_gts * R boxes: the head asserts)
        if not head.train_step_supported(R):
            return 'this CPRHead option set has no hand-written backward (CPRHead.train_step_supported)'
        if head.num_cls_fcs > 0 and bb.compute_dtype != torch.float32:
            return 'num_cls_fcs > 0 trains in the fp32 compute mode (the FC backward reads fp32 activations)'
    elif kind == 'P2PHead':
        if head.num_points != 1 or not getattr(head, 'train_cfg', None):
            return 'P2PHead trains with one point per cell and a train_cfg (the shipped P2P configs)'
    else:
        return 'no backward rules for head %s' % kind
    return None


_WARNED = set()


def warn_once(reason):
    if reason not in _WARNED:
        _WARNED.add(reason)
        warnings.warn('BasicLocator.forward_train was called with autograd enabled but returns losses WITHOUT a graph: %s.  '
                      'Wrap forward-only calls in torch.no_grad() to silence this.' % reason, stacklevel=3)


def get_bridge(model):
    br = getattr(model, '_autograd_bridge_state', None)
    if br is None or br.signature != signature(model):
        br = Bridge(model)
        model._autograd_bridge_state = br
    return br


# ------------------------------------------------------------------------------------------------ Functions
# Thin: every Function calls one forward_* / backward_* pair of the engine (training.BackwardEngine).  An autograd OUTPUT must
# not be reachable from its own ctx (ctx -> state -> output -> grad_fn -> ctx would be a reference cycle that keeps a batch of
# maps alive until the garbage collector runs): the engine's state holds the output's storage under another tensor object
# (``.detach()`` aliases), never the returned object itself.
_CONSUMED = 'the recorded forward of this step was already consumed (a second backward / retain_graph is not supported)'


class _StageFn(torch.autograd.Function):
    """One trainable ResNet stage (T/mmdet/models/backbones/resnet.py:262-302 per block) on an NHWC map."""

    @staticmethod
    def forward(ctx, bridge, stage, x, *params):
        out, tape = bridge.engine.forward_stage(stage, bridge.real(x))
        out = bridge.carrier(out)
        if tape and isinstance(tape[-1], dict) and tape[-1].get('out') is out:
            tape[-1]['out'] = out.detach()
        ctx.bridge, ctx.tape, ctx.params = bridge, tape, params
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        eng, tape = ctx.bridge.engine, ctx.tape
        assert tape is not None, _CONSUMED
        dx = eng.backward_stage(tape, dout.contiguous(), ctx.needs_input_grad[2])
        grads = eng.collect(ctx.params)
        ctx.tape = None
        return (None, None, dx) + grads


class _LateralsFn(torch.autograd.Function):
    """FPN lateral convs + GroupNorm + the top-down nearest-upsample adds (T/mmdet/models/necks/fpn.py:166-188) -> the finest
    lateral sum (the only one an output conv reads when num_outs == 1)."""

    @staticmethod
    def forward(ctx, bridge, n_in, *args):
        xs, params = args[:n_in], args[n_in:]
        lat0, recs = bridge.engine.forward_laterals([bridge.real(x) for x in xs])
        ctx.bridge, ctx.recs, ctx.params, ctx.n_in = bridge, recs, params, n_in
        return bridge.carrier(lat0)

    @staticmethod
    @once_differentiable
    def backward(ctx, dlat):
        eng = ctx.bridge.engine
        assert ctx.recs is not None, _CONSUMED
        need = [ctx.needs_input_grad[2 + i] for i in range(ctx.n_in)]
        dxs = eng.backward_laterals(ctx.recs, dlat.contiguous(), need)
        grads = eng.collect(ctx.params)
        ctx.recs = None
        return (None, None) + tuple(dxs) + grads


class _HeadLossFn(torch.autograd.Function):
    """FPN output conv -> head towers -> projection / output convs -> point stage -> the loss vector: (5,) for CPRHead
    (gt_loss, pos_loss, bag_acc, neg_loss, num_sample; cpr_head.py:1101-1229), (B, 2) for P2PHead (loss_cls, loss_pts per
    image; p2p_head.py:172-248).  Everything between the finest lateral sum and the losses hands raw maps + pending GroupNorm
    affines from layer to layer, so this is the smallest unit with a true gradient on both sides."""

    @staticmethod
    def forward(ctx, bridge, lat0, pack, *params):
        eng = bridge.engine
        out, state = eng.forward_head_loss(bridge.real(lat0), pack.img_metas, pack.gt_bboxes, pack.gt_labels,
                                           pack.gt_bboxes_ignore, pack.gt_true_bboxes)
        saved = state[1] if isinstance(state, tuple) and len(state) == 2 and isinstance(state[1], dict) else None
        if saved is not None and saved.get(eng.loss_vector_key) is out:
            saved[eng.loss_vector_key] = out.detach()
        ctx.bridge, ctx.state, ctx.params = bridge, state, params
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        eng = ctx.bridge.engine
        assert ctx.state is not None, _CONSUMED
        # what torch hands over: d(total) / d(loss vector) -- ones where _parse_losses summed a term, zero on bag_acc; a scaled
        # total (loss scaling, gradient accumulation) arrives as that scale.  The loss-backward kernels multiply by it on the device
        dlat = eng.backward_head_loss(ctx.state, gout.contiguous().float())
        grads = eng.collect(ctx.params)
        ctx.state = None
        return (None, dlat, None) + grads


# ------------------------------------------------------------------------------------------------ entry point
def forward_train(model, img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None, gt_true_bboxes=None):
    """BasicLocator.forward_train with a graph: the same dict of losses, differentiable wrt every trainable parameter.  The
    CPR losses are the forward-only values up to fp32 summation order (<= 2e-6 relative, asserted by tests/test_gpu_train_step.py:
    the recorded forward fuses the producer GroupNorm into the 3x3 consumers' loads where the forward-only path materialises it
    and runs Winograd); P2PNet's differ in the last bits for one more reason -- the recorded path keeps the 3x3 output convs on
    the conv kernel (its backward walks them) where the forward-only path uses the tap projection.  The recorded maps live until
    ``loss.backward()`` has run or the returned losses die: evaluate losses you do not differentiate under ``torch.no_grad()``."""
    bridge = get_bridge(model)
    eng = bridge.engine
    eng.begin_step()
    eng._sink.clear()
    bridge._maps.clear()
    bb, neck = model.backbone, model.neck
    with torch.no_grad():
        x = bb.stem(img)
    feats = []
    for i in range(len(bb.res_layers)):
        params = bridge.stage_params[i]
        if params:
            x = _StageFn.apply(bridge, i, x, *params)
        else:
            with torch.no_grad():
                x = bb.run_stage(i, bridge.real(x))
        feats.append(x)
    assert len(feats) == len(neck.in_channels)
    xs = feats[neck.start_level:neck.start_level + len(neck.lateral_convs)]
    lat0 = _LateralsFn.apply(bridge, len(xs), *xs, *bridge.lateral_params)
    pack = _GtPack(img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore, gt_true_bboxes)
    out = _HeadLossFn.apply(bridge, lat0, pack, *bridge.head_params)
    bridge._maps.clear()          # every forward consumer has run; the backward reads the tapes
    return eng.loss_dict(out)
