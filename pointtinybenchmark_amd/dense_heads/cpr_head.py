"""CPRHead -- Coarse Point Refinement head (T/mmdet/models/point/dense_heads/cpr_head.py:898-1309).

Same registry name, constructor keys and method signatures as the reference; the work is done by HIP kernels:
  forward        4 x [conv3x3 -> GN32 -> ReLU] on the MFMA conv kernel; GN statistics ride in the conv epilogue
                 and GN-apply+ReLU is folded into the NEXT conv's input load (only the last layer materialises)
  loss           ONE 1x1 conv projects the head features to a (N,H,W,J) logit map ([cls ++ ins]); the negative
                 grid mask + probabilities + gfocal run on that map, positive bags are sampled from it
                 (valid because num_cls_fcs == 0: Linear and bilinear sampling commute), MIL + gt loss in one
                 wave per bag; no host sync, the four scalars stay on the device
  get_bboxes     same extraction, then PointRefiner as one wave per gt.
Single FPN level like every shipped config (the reference asserts the single level itself: cpr_head.py:799,1152).
Options beyond the shipped configs, all on the same kernels: num_refine > 1 inputs with the three
``refine_bag_policy`` values and both ``gt_loss_type`` values, ``GridCirclesPtFeatGenerator`` bags, ``softmax`` /
``normed_sigmoid`` class probabilities, ``binary_ins``, ``AllPosLoss``, ``num_cls_fcs > 0``; round 3:
``ins_share_head_feat=False`` (a second tower ``ins_convs`` / ``ins_fcs`` for the instance classifier, cpr_head.py:992-1008,
1037,1068), ``out_bg_cls=True`` for one class (:953), ``PointRefiner(return_score_type='max')`` (:840-842).
generator ``align_corners=True`` (:73-93,126).
``AnchorPtFeatGenerator(scale_factor != 1)`` and ``GridEllipsePtFeatGenerator`` raise in the reference itself and are refused.
Training (``training.BackwardEngine`` / ``autograd_bridge``): every option set listed above has a hand-written backward since round 5
(``train_step_supported``): the shipped configs' on the specialised loss-backward kernels, the other loss options on the general
ones, grid bags / align_corners through the point-list gather (csrc/backward.hip)."""
import math

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from ..layers import ConvModule, _PackCache, conv_gn
from ..registry import HEADS, build_loss

_NEG_GENERATORS = ('OutCirclePtFeatGenerator', 'OutGridCirclesPtFeatGenerator', 'AnchorPtFeatGenerator')


def circle_offsets(radius, stride, base_num_point=8, start_angle=0, same_num_all_radius=False):
    """Ring offsets of CirclePtFeatGenerator.get_point_neighbours (cpr_head.py:474-497), computed on the HOST
    with torch's CPU cos/sin so the bag coordinates are the reference's bit for bit."""
    out = []
    for i in range(radius):
        r = (i + 1) * stride
        n = base_num_point if same_num_all_radius else base_num_point * (i + 1)
        ang = torch.arange(n).float() / n * 360 + start_angle
        ang = ang / 360 * np.pi * 2
        out.append(torch.stack([r * torch.cos(ang), r * torch.sin(ang)], dim=-1))
    return torch.cat(out).float().contiguous()


def sqrt_threshold(thr):
    """Smallest fp32 t with torch.sqrt(t) >= thr on the host CPU.  The reference compares
    ``cdist(...).min() >= stride*radius`` (cpr_head.py:278); cdist = sqrt(d2) and torch's CPU sqrt is monotone
    but not correctly rounded, so the exact d2-space threshold is read off the host implementation instead of
    being assumed to be thr*thr."""
    thr = np.float32(thr)
    c = np.float32(thr * thr)
    cand = [c]
    lo = hi = c
    for _ in range(64):
        lo = np.nextafter(lo, np.float32(-np.inf), dtype=np.float32)
        hi = np.nextafter(hi, np.float32(np.inf), dtype=np.float32)
        cand += [lo, hi]
    cand = np.sort(np.array(cand, dtype=np.float32))
    ok = (torch.sqrt(torch.from_numpy(cand)) >= float(thr)).numpy()
    first = int(np.argmax(ok))
    assert ok[first:].all() and not ok[:first].any(), 'host sqrt is not monotone around the threshold'
    return float(cand[first])


class _Extractor:
    """Parsed ``train_pts_extractor`` / ``refine_pts_extractor`` config (PointExtractor, cpr_head.py:602-662).
    pos generator: CirclePtFeatGenerator (rings around every annotated / refine point, bilinear samples) or
    GridCirclesPtFeatGenerator (the grid points inside the circles, cpr_head.py:405-438)."""

    def __init__(self, pos_generator, neg_generator, strides, num_classes):
        pg, ng = dict(pos_generator), dict(neg_generator)
        ptype, ntype = pg.pop('type'), ng.pop('type')
        if ptype == 'GridEllipsePtFeatGenerator':
            # the reference class cannot run: `c` is a (num_gts, 2) unit vector where a focal distance was meant, so
            # `2 * a.reshape(-1, 1, 1)` has 2*num_gts rows against (num_gts, H, W) distances and get_max_pos_num returns a
            # tensor that torch.zeros() rejects (cpr_head.py:368-402) -- there is no behaviour to reproduce
            raise NotImplementedError('GridEllipsePtFeatGenerator raises inside the reference itself (cpr_head.py:382-402)')
        assert ptype in ('CirclePtFeatGenerator', 'GridCirclesPtFeatGenerator'), ptype
        assert ntype in _NEG_GENERATORS, ntype
        # align_corners=True (PtFeatGenerator, cpr_head.py:126-127,73-93): grid 2x/(w-1)-1 and zeros padding for the bilinear
        # samples of the positive generator; the grid (negative) generators take cell values and never sample
        self.align_corners = bool(pg.pop('align_corners', False))
        self.pos_is_grid = ptype == 'GridCirclesPtFeatGenerator'
        self.pos_radius = pg.pop('radius')
        if self.pos_is_grid:
            mp = pg.pop('max_pos_num', -1)
            self.max_pos_num = mp if mp > 0 else 2 * (2 * self.pos_radius) ** 2          # cpr_head.py:434-438
            assert pg.pop('scale_factor', None) in (None, 1.0) and not pg, pg
        else:
            self.pos_kw = dict(start_angle=pg.pop('start_angle', 0), base_num_point=pg.pop('base_num_point', 8),
                               same_num_all_radius=pg.pop('same_num_all_radius', False))
            assert pg.pop('append_center', True) and not pg, pg
        self.neg_is_anchor = ntype == 'AnchorPtFeatGenerator'
        if self.neg_is_anchor and ng.get('scale_factor', None) not in (None, 1, 1.0):
            # generate() hands scale_factor to F.interpolate as its second POSITIONAL argument, i.e. as `size`
            # (cpr_head.py:229-231): the reference raises a TypeError on the first call (tests/golden/cpr_options_r3.npz)
            raise NotImplementedError('AnchorPtFeatGenerator(scale_factor=%r) raises inside the reference itself '
                                      '(cpr_head.py:229-231)' % (ng['scale_factor'],))
        self.neg_radius = ng.get('radius', 0)
        self.neg_class_wise = ng.get('class_wise', False)
        self.strides, self.num_classes = strides, num_classes
        self._off = {}

    def offsets(self, stride, device):
        key = (stride, str(device))
        if key not in self._off:
            off = circle_offsets(self.pos_radius, stride, **self.pos_kw)
            # how far (in grid cells, rounded up) a bag point can lie from its centre: read off the offsets themselves, so a
            # non-integer radius or any start_angle sizes the backward's gather window correctly (ops.cpr_loss_bwd)
            self._off[('cells', stride)] = int(math.ceil(float(off.abs().max()) / stride)) if off.numel() else 0
            self._off[key] = off.to(device)
        return self._off[key]

    def window_radius_cells(self, stride, device):
        self.offsets(stride, device)
        return self._off[('cells', stride)]


def cat_rows(tensors):
    """torch.cat(tensors) along dim 0 -- without any copy when the list is what ``torch.split`` of ONE contiguous tensor
    produced (GpuImagePipeline hands the per-image gt lists over like that): consecutive views of one storage are re-joined
    as a view.  Otherwise a plain cat (one small copy per image)."""
    t0 = tensors[0]
    if len(tensors) > 1 and t0.is_contiguous():
        end = t0.data_ptr() + t0.numel() * t0.element_size()
        for t in tensors[1:]:
            if not (t.is_contiguous() and t.dtype == t0.dtype and t.shape[1:] == t0.shape[1:] and t.data_ptr() == end and
                    t.untyped_storage().data_ptr() == t0.untyped_storage().data_ptr()):
                return torch.cat(list(tensors))
            end += t.numel() * t.element_size()
        rows = sum(t.shape[0] for t in tensors)
        return torch.as_strided(t0, (rows,) + tuple(t0.shape[1:]), t0.stride())
    return torch.cat(list(tensors)) if len(tensors) > 1 else t0


class _Gts:
    """The annotated / refine points of a batch in the kernels' CSR form.  With num_refine = R > 1 an image brings
    (num_gts*R, 4) pseudo boxes (cpr_head.py:1240-1246): ``points`` holds all of them (gt-major), ``pt_*`` describe points,
    the un-prefixed fields describe gts."""
    pass


@HEADS.register_module()
class CPRHead(nn.Module):
    accepts_b8 = True     # forward_train_lazy / _tower read a channel-blocked neck output (ops.is_b8) directly
    def __init__(self, num_classes, in_channels, num_cls_fcs=0, fc_out_channels=1024,
                 train_pts_extractor=dict(pos_generator=dict(type='CirclePtFeatGenerator', radius=5),
                                          neg_generator=dict(type='OutCirclePtFeatGenerator', radius=3)),
                 refine_pts_extractor=dict(pos_generator=dict(type='CirclePtFeatGenerator', radius=5),
                                           neg_generator=dict(type='AnchorPtFeatGenerator', scale_factor=1.0)),
                 point_refiner=dict(), ins_share_head_feat=True, ins_share_head_classifier=False,
                 loss_mil=dict(type='MILLoss', binary_ins=False, loss_weight=1.0), loss_type=0,
                 loss_cfg=dict(with_neg=True, neg_loss_weight=1.0, refine_bag_policy='independent_with_gt_bag',
                               random_remove_rate=0.4, with_gt_loss=False, gt_loss_weight=1.0, with_mil_loss=True),
                 normal_cfg=dict(prob_cls_type='sigmoid', out_bg_cls=False), init_cfg=None, debug=False,
                 debug_info=dict(), other_info=dict(),
                 # AnchorFreeHead kwargs (T/mmdet/models/dense_heads/anchor_free_head.py:40-87)
                 feat_channels=256, stacked_convs=4, strides=(4, 8, 16, 32, 64), dcn_on_last_conv=False,
                 conv_bias='auto', loss_cls=None, loss_bbox=None, conv_cfg=None, norm_cfg=None, train_cfg=None,
                 test_cfg=None):
        super().__init__()
        # loss_type != 0 has no loss{n} method in the reference either (cpr_head.py:1116: getattr(self, 'loss%d'))
        assert num_cls_fcs >= 0 and loss_type == 0, 'loss_type != 0 does not exist'
        assert num_cls_fcs == 0 or fc_out_channels % 32 == 0, 'fc_out_channels must be a multiple of 32'
        self.num_cls_fcs, self.fc_out_channels = num_cls_fcs, fc_out_channels
        self.prob_type = normal_cfg.get('prob_cls_type', 'sigmoid')
        self.norm_p = float(normal_cfg.get('normed_sigmoid_p', 1))
        if self.prob_type not in ('sigmoid', 'softmax', 'normed_sigmoid'):
            raise ValueError(self.prob_type)                     # as get_cls_prob does (cpr_head.py:1097)
        # out_bg_cls=True (one more classifier output, never a label): runs in the reference for ONE class only -- for C > 1
        # the (.., C) validity mask meets (.., C+1) probabilities in gfocal_loss (cpr_head.py:1226) and broadcasting fails
        self.out_bg_cls = bool(normal_cfg.get('out_bg_cls', False))
        assert not self.out_bg_cls or num_classes == 1, 'out_bg_cls=True cannot run in the reference for num_classes > 1'
        assert norm_cfg is not None and norm_cfg['type'] == 'GN' and not dcn_on_last_conv and not debug
        self.num_classes = self.cls_out_channels = num_classes
        self.num_cls_out = num_classes + 1 if self.out_bg_cls else num_classes          # cpr_head.py:953
        self.in_channels, self.feat_channels, self.stacked_convs = in_channels, feat_channels, stacked_convs
        self.strides = list(strides)
        self.ins_share_head_feat, self.ins_share_head_classifier = ins_share_head_feat, ins_share_head_classifier
        self.binary_ins = bool(loss_mil.get('binary_ins', False))
        assert not (self.binary_ins and ins_share_head_classifier)            # cpr_head.py:1012
        self.loss_cfg, self.loss_type, self.normal_cfg = dict(loss_cfg), loss_type, dict(normal_cfg)
        assert self.loss_cfg.get('refine_bag_policy', 'independent_with_gt_bag') in (
            'independent_with_gt_bag', 'merge_to_gt_bag', 'only_refine_bag'), self.loss_cfg['refine_bag_policy']
        if self.loss_cfg.get('gt_loss_type', 'gt_refine') not in ('gt_refine', 'gt'):
            raise NotImplementedError('gt_loss_type=%r (the reference raises too: cpr_head.py:1167)' % self.loss_cfg['gt_loss_type'])
        self.train_cfg, self.test_cfg, self.other_info = train_cfg, test_cfg, dict(other_info)
        self.cls_convs = nn.ModuleList()
        self.ins_convs = nn.ModuleList()
        chn = in_channels
        for _ in range(stacked_convs):
            self.cls_convs.append(ConvModule(chn, feat_channels, 3, padding=1, conv_cfg=conv_cfg, norm_cfg=norm_cfg))
            if not ins_share_head_feat:                                               # cpr_head.py:992-996
                self.ins_convs.append(ConvModule(chn, feat_channels, 3, padding=1, conv_cfg=conv_cfg, norm_cfg=norm_cfg))
            chn = feat_channels
        self.cls_fcs = nn.ModuleList()     # ins_share_head_feat: shared by the cls and ins classifiers (cpr_head.py:999-1005)
        self.ins_fcs = nn.ModuleList()
        for _ in range(num_cls_fcs):
            self.cls_fcs.append(nn.Linear(chn, fc_out_channels))
            if not ins_share_head_feat:
                self.ins_fcs.append(nn.Linear(chn, fc_out_channels))
            chn = fc_out_channels
        self.cls_out = nn.Linear(chn, self.num_cls_out)
        self.ins_out = self.cls_out if ins_share_head_classifier else \
            nn.Linear(chn, self.num_cls_out * 2 if self.binary_ins else self.num_cls_out)      # cpr_head.py:1009-1011
        # one projection serves both classifiers only when features AND classifier are shared (cpr_head.py:1068-1070)
        self.ins_same_logits = ins_share_head_feat and ins_share_head_classifier
        self.loss_mil = build_loss(loss_mil)
        self.loss_cls = self.loss_mil
        self.train_pts_extractor = _Extractor(**train_pts_extractor, strides=self.strides, num_classes=num_classes)
        self.refine_pts_extractor = _Extractor(**refine_pts_extractor, strides=self.strides, num_classes=num_classes)
        pr = dict(gt_alpha=0.5, merge_th=0.05, refine_th=0.05, classify_filter=False, return_score_type='mean',
                  nearest_filter=True)
        pr.update(point_refiner)
        if pr['return_score_type'] not in ('mean', 'max'):
            raise ValueError(pr['return_score_type'])                              # cpr_head.py:845
        self.point_refiner = pr
        self._cache = _PackCache()
        self._thr = {}
        self.init_weights()

    def train_step_supported(self, num_refine=1):
        """Which option sets have a hand-written backward (training.BackwardEngine).  The shipped configs' (sigmoid, MILLoss,
        circle bags: the specialised kernels); since round 5 also a separate instance tower (``ins_share_head_feat=False``,
        cpr_head.py:992-1008,1037), FC layers between the sampled features and the classifiers (``num_cls_fcs > 0``,
        :999-1005,1055-1059), num_refine > 1 inputs with every ``refine_bag_policy`` / ``gt_loss_type`` (:1159-1211), softmax /
        normed_sigmoid probabilities (:1080-1099), ``binary_ins``, ``AllPosLoss``, ``out_bg_cls`` and ``with_mil_loss=False`` (the
        general loss-backward kernels, csrc/backward.hip), the grid generator and ``align_corners=True`` sampling (the point-list
        gather, ``cpr_bag_points_gather_bwd``) and ``with_neg=False``: every option set the head can run forward."""
        return True

    def _loss_backward_general(self, num_refine, bags, centres):
        """True when the loss gradient needs the general kernels (anything the specialised sigmoid / MIL / one-bag-per-row /
        centre-is-the-last-entry kernels do not cover)."""
        cfg = self.loss_cfg
        default_geometry = bags[1] == bags[3] and bags[2] == 0 and (centres[2] == 0 or centres == (bags[3] - 1, bags[3], 1, 1))
        return (self.prob_type != 'sigmoid' or self.binary_ins or self.loss_mil.allpos or self.out_bg_cls or
                not default_geometry or not cfg.get('with_mil_loss', True) or not cfg.get('with_neg', True))

    # ------------------------------------------------------------------ init (cpr_head.py:939-948)
    def init_weights(self):
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.normal_(m.weight, 0, 0.01)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        nn.init.constant_(self.cls_out.bias, float(-math.log((1 - 0.01) / 0.01)))

    # ------------------------------------------------------------------ forward (cpr_head.py:1030-1043)
    def forward(self, feats):
        cls, ins = [], []
        for x in feats:
            f = self.forward_single(x)
            cls.append(f[0])
            ins.append(f[1])
        return cls, ins

    def _tower(self, x, ab=None, in_relu=True, tape=None, own_input=False, convs=None):
        """4 x [conv3x3 -> GN -> ReLU]; returns the LAST layer un-normalised: (raw, (a, b)).
        own_input: nobody else reads ``x`` (its pending GroupNorm may be applied in place).
        convs: the tower's layers (cls_convs; ins_convs for the instance tower of ins_share_head_feat=False)."""
        convs = self.cls_convs if convs is None else convs
        last = len(convs) - 1
        for i, m in enumerate(convs):
            rec = None
            if tape is not None:
                rec = dict(kind='tower', level=i)
                tape.append(rec)
            # forward only: layers hand their raw output to the next one channel-blocked (the Winograd kernel's fast input
            # form); the last layer stays NHWC for the logit projection.  conv_gn ignores the request when it cannot serve it
            x, ab = conv_gn(self._cache, m, x, in_ab=ab, in_relu=(in_relu if i == 0 else True), materialize=False,
                            save=rec, consume_input=(own_input or i > 0), out_b8=(tape is None and i < last))
        if ops.is_b8(x):       # (a one-layer tower cannot get here; kept for stacked_convs the configs do not use)
            x = ops.gn_apply_b8(x)
        return x, ab

    def forward_single(self, x):
        xin = ops.from_nchw(x)
        raw, ab = self._tower(xin)
        out = ops.as_nchw(ops.gn_apply(raw, ab[0], ab[1], relu=True, out=raw))
        if self.ins_share_head_feat:
            return out, out
        raw2, ab2 = self._tower(xin, convs=self.ins_convs)                           # cpr_head.py:1037-1040
        return out, ops.as_nchw(ops.gn_apply(raw2, ab2[0], ab2[1], relu=True, out=raw2))

    def forward_train(self, x, img_metas, gt_bboxes, gt_labels=None, gt_bboxes_ignore=None, gt_true_bboxes=None,
                      proposal_cfg=None, **kwargs):
        outs = self(x)
        losses = self.loss(*outs, gt_bboxes, gt_labels, img_metas, gt_bboxes_ignore=gt_bboxes_ignore,
                           gt_true_bboxes=gt_true_bboxes)
        if proposal_cfg is None:
            return losses
        return losses, self.get_bboxes(*outs, img_metas, cfg=proposal_cfg)

    def forward_train_lazy(self, lazy_feats, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None,
                           gt_true_bboxes=None, tape=None, save=None):
        """Fused training path used by BasicLocator: the neck hands over (raw, (a, b)) and neither the neck output nor
        the last tower layer is ever materialised in normalised form -- the consumer convs (tower layer 0, the logit
        projection) apply the GroupNorm affine (+ReLU) on load.  Same arithmetic as forward() + loss()."""
        assert len(lazy_feats) == 1
        raw0, ab0 = lazy_feats[0]
        ins = None
        if not self.ins_share_head_feat:        # the instance tower reads the neck output first (the cls tower may consume it)
            ins_tape = [] if tape is not None else None          # training: the second tower is recorded like the first
            ins = [self._tower(raw0, ab0, in_relu=False, convs=self.ins_convs, tape=ins_tape)]
            if save is not None:
                save['ins_tape'] = ins_tape
        raw, ab = self._tower(raw0, ab0, in_relu=False, tape=tape, own_input=True)
        return self.loss([(raw, ab)], ins, gt_bboxes, gt_labels, img_metas, gt_bboxes_ignore=gt_bboxes_ignore,
                         gt_true_bboxes=gt_true_bboxes, save=save)

    # ------------------------------------------------------------------ shared extraction
    def _fc_stack(self, rows_nhwc, ins=False, acts=None):
        """get_pts_outs.forward_with_fc (cpr_head.py:1055-1059): relu(fc_i(.)) as 1x1 convs over an NHWC block of rows
        (ins: the instance tower's ``ins_fcs``).  acts (list, training): receives the input and every layer's output."""
        x = rows_nhwc
        if acts is not None:
            acts.append(x)
        for i, fc in enumerate(self.ins_fcs if ins else self.cls_fcs):
            pc, bias = self._cache.get(('fc', ins, i, x.dtype), [fc.weight, fc.bias], lambda fc=fc, x=x: (
                ops.PackedConv(fc.weight.detach()[:, :, None, None], 1, 0, x.dtype), fc.bias.detach().float().contiguous()))
            x = ops.conv2d(x, pc, bias=bias, relu=True)
            if acts is not None:
                acts.append(x)
        return x

    def _proj(self, dt, part=None):
        """The classifiers as ONE packed 1x1 conv: rows [cls_out ++ ins_out] (cls_out alone when features and classifier are
        shared); part='cls' / 'ins': one classifier alone (ins_share_head_feat=False: each reads its own tower)."""
        def make():
            mods = {None: [self.cls_out] + ([] if self.ins_same_logits else [self.ins_out]),
                    'cls': [self.cls_out], 'ins': [self.ins_out]}[part]
            wt = torch.cat([m.weight for m in mods], 0).detach()[:, :, None, None]
            return (ops.PackedConv(wt, 1, 0, dt), torch.cat([m.bias for m in mods], 0).detach().float().contiguous(),
                    wt[:, :, 0, 0].float().contiguous())
        srcs = [self.cls_out.weight, self.cls_out.bias, self.ins_out.weight, self.ins_out.bias]
        return self._cache.get(('proj', dt, part), srcs, make)

    def _logit_map(self, feat_nhwc, in_ab=None, part=None, acts=None):
        """(N,H,W,256) -> (N,H,W,J) with J = [cls(C) ++ ins(C, or 2C with binary_ins)] (or C when features and classifier are
        shared; part='cls' / 'ins': that classifier's logits alone, through its own FC stack).
        in_ab: the input is the raw last-layer conv output and (a, b) its GroupNorm affine (+ReLU), applied on load.
        With num_cls_fcs > 0 the (materialised) input first runs through the FC stack."""
        assert part is not None or self.ins_share_head_feat, 'two towers: project each with its own part'
        dt = feat_nhwc.dtype
        if self.num_cls_fcs > 0:
            assert in_ab is None
            feat_nhwc = self._fc_stack(feat_nhwc, ins=(part == 'ins' and not self.ins_share_head_feat), acts=acts)

        pc, bias, w_rows = self._proj(dt, part)
        if in_ab is not None and ((feat_nhwc.shape[1] * feat_nhwc.shape[2]) % 128 != 0 or dt != torch.float32):
            feat_nhwc, in_ab = ops.gn_apply(feat_nhwc, in_ab[0], in_ab[1], relu=True), None
        # the logit map is always fp32 (the loss / sampling kernels are shared by both compute modes)
        if dt == torch.float32 and feat_nhwc.shape[0] * feat_nhwc.shape[1] * feat_nhwc.shape[2] >= 4096:
            # few output channels: a pure HBM stream (csrc/project.hip) instead of a mostly-empty MFMA tile
            out = ops.logit_project(feat_nhwc, w_rows, bias, in_ab, in_relu=True)
            if out is not None:
                return out
        return ops.conv2d(feat_nhwc, pc, bias=bias, in_ab=in_ab, in_relu=True, out_dtype=torch.float32)

    def _lmap_all(self, feat, ab=None, ifeat=None, iab=None, acts=None):
        """The logit map the loss reads: [cls ++ ins] channels.  Two towers (ins_share_head_feat=False): the class logits
        come from the class tower's map, the instance logits from the instance tower's (cpr_head.py:1061-1070).
        acts (dict, training with num_cls_fcs > 0): acts['map'] receives the FC activations of the class path over the map."""
        rec = None if acts is None else acts.setdefault('map', [])
        if self.ins_share_head_feat:
            return self._logit_map(feat, ab, acts=rec)
        return torch.cat([self._logit_map(feat, ab, 'cls', acts=rec), self._logit_map(ifeat, iab, 'ins')], dim=-1).contiguous()

    def _sample(self, ex, src, gts, stride, pad, rec=None):
        """The positive generator on one map: bag points, validity, samples and the bag view.  pad: what a slot / tap without
        a feature contributes (the projection's bias on a logit map, nothing on a feature map).  rec (dict, training): receives
        the grid generator's entry codes (what the point-list gather of the backward walks)."""
        if ex.pos_is_grid:
            # the reference pads to max_pos_num + num_refine grid slots and THEN appends the num_refine points (cpr_head.py:325-349)
            kmax = ex.max_pos_num + gts.R
            pts, valid, out, count, cell = ops.grid_bag(src, gts.points, gts.gt_img, gts.R, kmax, ex.pos_radius * stride, stride,
                                                        pad_value=pad, align_corners=ex.align_corners, want_cell=True)
            if rec is not None:
                rec['code'] = cell
            # the reference fails inside generate() when a bag overflows (cpr_head.py:331-333: shape mismatch on assignment)
            worst = int(count.max().item()) if count.numel() else 0
            if worst > kmax:
                raise RuntimeError('GridCirclesPtFeatGenerator: %d grid points in one bag > max_pos_num + num_refine = %d'
                                   % (worst, kmax))
            return pts, valid, out, (1, kmax + gts.R)
        pts, valid, out = ops.bag_sample(src, gts.points, gts.pt_img, gts.pad_hw, ex.offsets(stride, src.device), stride,
                                         align_corners=ex.align_corners, pad_value=pad if ex.align_corners else None)
        return pts, valid, out, (gts.R, pts.shape[1])

    def _bags(self, ex, feat, lmap, gts, stride, ifeat=None, part=None, acts=None, rec=None):
        """Bag points (E,2), validity (E) and bag logits (E,J) of the positive generator, E = G * entries-per-gt, plus
        the bag view (sub_bags per gt, entries per sub-bag).  num_cls_fcs == 0: Linear commutes with bilinear sampling, so
        the logits are sampled from the projected map ``lmap``.  Otherwise the 256-channel features are sampled and run
        through the FC stack + classifiers (the ReLUs in between do not commute with the interpolation): ``feat`` for the
        class logits and, with two towers, ``ifeat`` for the instance logits (part='cls': class logits only)."""
        if self.num_cls_fcs == 0:
            # padding slots hold zero features in the reference (:323-324): on the projected map that is the projection's bias
            pad = None
            if ex.pos_is_grid or ex.align_corners:
                pad = self._proj(lmap.dtype, part)[1] if (self.ins_share_head_feat or part is not None) else \
                    torch.cat([self._proj(lmap.dtype, 'cls')[1], self._proj(lmap.dtype, 'ins')[1]])
            pts, valid, out, view = self._sample(ex, lmap, gts, stride, pad, rec)
            if rec is not None:
                rec['pad'] = pad is not None
        else:
            # acts (dict, training): the sampled features and the FC activations of each path ('bag': shared features;
            # 'bag_cls' / 'bag_ins': the two towers)
            pts, valid, out, view = self._sample(ex, feat, gts, stride, None, rec)
            E, K, Cf = out.shape
            if self.ins_share_head_feat:
                out = self._logit_map(out.view(1, E * K, 1, Cf), part=part,
                                      acts=None if acts is None else acts.setdefault('bag', [])).view(E, K, -1)
            else:
                parts = [self._logit_map(out.view(1, E * K, 1, Cf), part='cls',
                                         acts=None if acts is None else acts.setdefault('bag_cls', [])).view(E, K, -1)]
                if part is None:
                    _, _, iout, _ = self._sample(ex, ifeat, gts, stride, None)
                    parts.append(self._logit_map(iout.view(1, E * K, 1, Cf), part='ins',
                                                 acts=None if acts is None else acts.setdefault('bag_ins', [])).view(E, K, -1))
                out = torch.cat(parts, dim=-1).contiguous()
        G = gts.G
        if rec is not None:
            rec['pts'] = pts
        return pts.view(G, -1, 2), valid.view(G, -1), out.view(G, -1, out.shape[-1]), view

    def _gt_tensors(self, gt_bboxes, gt_labels, img_metas, device, shape_key='pad_shape'):
        counts = [int(len(l)) for l in gt_labels]
        assert len(counts) > 0 and all(c > 0 for c in counts), 'CPRHead does not support empty-gt images ' \
            '(the reference asserts too: cpr_head.py:1102,1242)'
        R = gt_bboxes[0].shape[0] // counts[0]
        for b, c in zip(gt_bboxes, counts):
            assert b.shape[0] == c * R and R >= 1, 'every image must bring num_gts * num_refine pseudo boxes'
        g = _Gts()
        g.R, g.counts, g.G = R, counts, sum(counts)
        boxes = cat_rows([b if b.dtype == torch.float32 else b.float() for b in gt_bboxes]).contiguous()
        labels = cat_rows(list(gt_labels)).to(torch.int32).contiguous()
        start = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        gt_img = np.repeat(np.arange(len(counts), dtype=np.int32), counts)
        hw = np.array([[m[shape_key][0], m[shape_key][1]] for m in img_metas], dtype=np.int32).reshape(-1)
        ihw = np.array([[m['img_shape'][0], m['img_shape'][1]] for m in img_metas], dtype=np.int32).reshape(-1)
        parts = [start, gt_img, hw, ihw] + ([start * R, np.repeat(gt_img, R)] if R > 1 else [])
        meta = torch.from_numpy(np.concatenate(parts)).to(device)
        cuts = np.cumsum([0] + [len(p) for p in parts])
        g.gt_start, g.gt_img, g.pad_hw, g.img_hw = [meta[cuts[i]:cuts[i + 1]] for i in range(4)]
        g.pt_start, g.pt_img = (meta[cuts[4]:cuts[5]], meta[cuts[5]:cuts[6]]) if R > 1 else (g.gt_start, g.gt_img)
        g.points = ops.box_centers(boxes.to(device))              # (G*R, 2), gt-major
        g.labels = labels.to(device)
        g.pt_labels = g.labels.repeat_interleave(R).contiguous() if R > 1 else g.labels
        return g

    def _d2_threshold(self, stride, radius):
        key = (stride, radius)
        if key not in self._thr:
            self._thr[key] = sqrt_threshold(stride * radius)
        return self._thr[key]

    def _loss_geometry(self, gts, view, gt_weights, dev):
        """Bag / annotated-point geometry of loss0 (cpr_head.py:1159-1211) for ops.mil_loss, plus per-bag labels / weights."""
        Rv, Kv = view
        G, cfg = gts.G, self.loss_cfg
        w = None if gt_weights is None else torch.cat(list(gt_weights)).float().to(dev).contiguous()
        policy = cfg.get('refine_bag_policy', 'independent_with_gt_bag')
        only_first = cfg.get('gt_loss_type', 'gt_refine') == 'gt'
        if Rv == 1 or policy == 'independent_with_gt_bag':
            bags = (G * Rv, Kv, 0, Kv)
            centres = (Kv - 1, Kv, 1, Rv if only_first else 1)
            labels = gts.pt_labels if Rv > 1 else gts.labels
            if w is not None and Rv > 1:
                w = w.repeat_interleave(Rv).contiguous()
        else:
            si = 1 if policy == 'only_refine_bag' else 0
            bags = (G, Rv * Kv, si * Kv, (Rv - si) * Kv)
            centres = (Kv - 1, Kv, 1 if only_first else Rv, 1)
            labels = gts.labels
        if not cfg.get('with_gt_loss', False):
            centres = (0, 1, 0, 1)
        return bags, centres, labels, w

    # ------------------------------------------------------------------ loss (cpr_head.py:1101-1229)
    def loss(self, cls_feat, ins_feat, gt_bboxes, gt_labels, img_metas, gt_bboxes_ignore=None, gt_true_bboxes=None,
             gt_weights=None, save=None, lmap=None):
        assert len(gt_labels) > 0
        assert len(cls_feat) == 1, 'single FPN level (the reference asserts the same: cpr_head.py:1152)'
        # C = classifier outputs (num_classes, + 1 with out_bg_cls: the labels never name the extra output)
        ex, C, stride = self.train_pts_extractor, self.num_cls_out, self.strides[0]

        def unpack(f):
            return f if isinstance(f, tuple) else (ops.from_nchw(f), None)      # (raw, (a, b)) from forward_train_lazy
        feat, ab = unpack(cls_feat[0])
        ifeat = iab = None
        if not self.ins_share_head_feat:
            ifeat, iab = unpack(ins_feat[0])
        dev = feat.device
        fc_acts = None
        if self.num_cls_fcs > 0:                          # the FC path samples the normalised, activated features
            if ab is not None:                            # (out of place: in training the raw map is on the tape)
                feat, ab = ops.gn_apply(feat, ab[0], ab[1], relu=True), None
            if iab is not None:
                ifeat, iab = ops.gn_apply(ifeat, iab[0], iab[1], relu=True), None
            fc_acts = {} if save is not None else None    # training: the FC activations of the map and bag paths (round 5)
        if lmap is None:                            # (a caller that replays a hipGraph hands the projected map over)
            lmap = self._lmap_all(feat, ab, ifeat, iab, acts=fc_acts)
        gts = self._gt_tensors(gt_bboxes, gt_labels, img_metas, dev)
        # grid generators / align_corners=True sampling: the backward gathers through the forward's own point list
        plist = {} if (save is not None and (ex.pos_is_grid or ex.align_corners)) else None
        _, valid, bag_logits, view = self._bags(ex, feat, lmap, gts, stride, ifeat=ifeat, acts=fc_acts, rec=plist)
        cfg = self.loss_cfg
        with_mil, with_gt, with_neg = cfg.get('with_mil_loss', True), cfg.get('with_gt_loss', False), cfg.get('with_neg', True)
        assert with_mil or with_gt, 'loss0 needs num_pos from the MIL or the gt loss (cpr_head.py:1180,1213,1227)'
        partial = neg_mask = None
        if with_neg:
            # the reference cannot train with it either: loss0 unpacks a (H, W, C) tensor into two dims (cpr_head.py:1221)
            assert not ex.neg_is_anchor, 'AnchorPtFeatGenerator is a refine-time generator only'
            neg_mask, partial = ops.neg_mask_loss(lmap, gts.points, gts.pt_labels, gts.pt_start, gts.pad_hw, C, stride,
                                                  self._d2_threshold(stride, ex.neg_radius), self.loss_mil.eps,
                                                  ex.neg_class_wise, self.prob_type, self.norm_p,
                                                  mask_classes=self.num_classes)
        bags, centres, labels, w = self._loss_geometry(gts, view, gt_weights, dev)
        ins_off = 0 if self.ins_same_logits else C
        out = self.loss_mil.forward_logits(bag_logits, ins_off, valid, labels, C, w, partial,
                                           cfg.get('gt_loss_weight', 1.0), cfg.get('neg_loss_weight', 1.0),
                                           want_bag_ws=save is not None, bags=bags, centres=centres,
                                           prob_type=self.prob_type, norm_p=self.norm_p, neg_from_gt=not with_mil)
        if save is not None:
            out, bag_ws = out
            # one row per BAG: with num_refine = R > 1 (independent bags) the (G, R * Kv) layout is (G * R, Kv) in memory, every
            # refine point a bag around its own centre (gts.points is gt-major, pt_img ascends)
            # rows = sampled points (gt-major, one per annotated / refine point), Kv entries each; with the independent policy a
            # row is a bag, the merged policies span several rows per bag (geometry in ``bags`` / ``centres``)
            rows, Kv = gts.G * view[0], view[1]
            save.update(feat=feat, ab=ab, ifeat=ifeat, iab=iab, lmap=lmap, neg_mask=neg_mask, out5=out,
                        bag_logits=bag_logits.view(rows, Kv, -1), valid=valid.view(rows, Kv),
                        labels=labels, gt_weight=w, bag_ws=bag_ws, centers=gts.points, ins_off=ins_off, stride=stride,
                        fc=fc_acts, bags=bags, centres=centres, general=self._loss_backward_general(gts.R, bags, centres),
                        neg_from_gt=not with_mil, plist=plist)
            if plist is None:      # ring offsets around every (gt, refine) point: the window gather
                save.update(gt_img=gts.pt_img, offsets=ex.offsets(stride, dev), radius_cells=ex.window_radius_cells(stride, dev))
            else:                  # one row per gt (grid) or per point (circle + align_corners); codes only for grid bags
                plist.setdefault('code', None)
                plist.update(pts=plist['pts'].reshape(rows, Kv, 2), align=ex.align_corners)
                save.update(gt_img=gts.gt_img if ex.pos_is_grid else gts.pt_img, offsets=None, radius_cells=0)
        return self._loss_dict(out)

    def _loss_dict(self, out):
        """The (gt_loss, pos_loss, bag_acc, neg_loss, num_sample) vector of the loss kernels -> the reference's dict
        (cpr_head.py:1131-1229: only the enabled terms appear)."""
        cfg = self.loss_cfg
        losses = {}
        if cfg.get('with_gt_loss', False):
            losses['gt_loss'] = out[0]
        if cfg.get('with_mil_loss', True):
            losses['pos_loss'], losses['bag_acc'] = out[1], out[2]
        if cfg.get('with_neg', True):
            losses['neg_loss'] = out[3]
        return losses

    # ------------------------------------------------------------------ refine (cpr_head.py:1231-1283)
    def refine_points(self, cls_feat, img_metas, gt_bboxes, gt_labels, not_refine=None):
        """Extraction + PointRefiner of get_bboxes: returns the kernel outputs for all gts of the batch
        (gts, bag pts (G,Kt,2), refine_pts (G,2), scores (G), not_refine (G) u8, chosen (G,Kt) u8)."""
        assert len(cls_feat) == 1
        ex, C, stride, pr = self.refine_pts_extractor, self.num_cls_out, self.strides[0], self.point_refiner
        feat = ops.from_nchw(cls_feat[0])
        dev = feat.device
        # the refiner reads class probabilities only (cpr_head.py:780-850): with two towers the class tower's logits suffice
        part = None if self.ins_share_head_feat else 'cls'
        lmap = self._logit_map(feat, part=part)
        gts = self._gt_tensors(gt_bboxes, gt_labels, img_metas, dev)
        # PointRefiner.refine_single asserts gt_r_pts == gt_r_points[:, :1] (cpr_head.py:809): with CirclePtFeatGenerator
        # bags it only accepts one point per gt; the grid generators keep one annotated point per bag and take R > 1
        assert ex.pos_is_grid or gts.R == 1, 'num_refine > 1 inputs cannot be refined (the reference asserts: cpr_head.py:809)'
        pts, valid, bag_logits, (Rv, Kv) = self._bags(ex, feat, lmap, gts, stride, part=part)
        # the grid (negative) branch is computed but unused by the reference at refine time (cpr_head.py:794-804)
        nr_in = None if not_refine is None else torch.cat(list(not_refine)).to(torch.uint8).to(dev).contiguous()
        rp, sc, nr, chosen = ops.refine(bag_logits, pts, valid, gts.points, gts.labels, gts.gt_img, gts.gt_start,
                                        gts.img_hw, C, pr['gt_alpha'], pr['merge_th'], pr['refine_th'],
                                        pr['nearest_filter'], pr['classify_filter'], nr_in, sub_bags=Rv,
                                        ctr_stride=gts.R, prob_type=self.prob_type, norm_p=self.norm_p,
                                        score_max=pr['return_score_type'] == 'max')
        return gts, pts, rp, sc, nr, chosen

    def get_bboxes(self, cls_feat, ins_feat, img_metas, cfg=None, rescale=False, with_nms=True, gt_bboxes=None,
                   gt_labels=None, gt_bboxes_ignore=None, gt_true_bboxes=None, gt_anns_id=None, not_refine=None,
                   cascade_out_fmt=False):
        assert gt_labels is not None and len(gt_labels) > 0 and gt_anns_id is not None
        gts, pts, rp, sc, nr, chosen = self.refine_points(cls_feat, img_metas, gt_bboxes, gt_labels, not_refine)
        dev = rp.device
        boxes = torch.cat([rp - 8.0, rp + 8.0], dim=-1)          # center_to_pseudo_bbox, 16x16 (:1303-1309)
        out, nr_list, s = [], [], 0
        for b, n in enumerate(gts.counts):
            bx = boxes[s:s + n]
            if rescale:
                bx = bx / bx.new_tensor(img_metas[b]['scale_factor'])
            ann = gt_anns_id[b].to(dev).unsqueeze(-1).type_as(sc)
            cols = [bx, sc[s:s + n, None], ann]
            if self.other_info.get('out_geo', False):
                cols.append(self._geo(pts[s:s + n], chosen[s:s + n], rp[s:s + n], img_metas[b], rescale))
            out.append((torch.cat(cols, dim=-1), gt_labels[b]))
            nr_list.append(nr[s:s + n].bool())
            s += n
        if cascade_out_fmt:
            return out, nr_list
        if not with_nms:
            raise NotImplementedError
        return out

    @staticmethod
    def _geo(pts, chosen, rp, img_meta, rescale):
        """get_geo_output + scale_geos + fill_list_to_tensor (cpr_head.py:852-864,1285-1288,54-60): per gt
        [refined pt, chosen pts...], rescaled, then padded with -1 to the longest row of the IMAGE."""
        G, K, _ = pts.shape
        m = int(chosen.sum(dim=1).max().item()) if G else 0
        order = torch.argsort(chosen.to(torch.int16), dim=1, descending=True, stable=True)[:, :m]
        sel = torch.cat([rp[:, None], torch.gather(pts, 1, order[..., None].expand(-1, -1, 2))], dim=1)
        keep = torch.cat([chosen.new_ones((G, 1)), torch.gather(chosen, 1, order)], dim=1).bool()
        if rescale:
            sel = sel / sel.new_tensor(img_meta['scale_factor'][:2])
        geo = torch.where(keep[..., None], sel, sel.new_full((), -1.0))
        return geo.reshape(G, -1)

    def simple_test(self, feats, img_metas, rescale=False, **gt_kwargs):
        """dense_test_mixins.simple_test_bboxes (T/mmdet/models/dense_heads/dense_test_mixins.py:15-36)."""
        outs = self(feats)
        return self.get_bboxes(*outs, img_metas, rescale=rescale, **gt_kwargs)

    @staticmethod
    def pseudo_bbox_to_center(gt_bboxes):
        return [ops.box_centers(b.float().contiguous()) for b in gt_bboxes]
