"""P2PHead -- P2PNet head with multi-class support (T/mmdet/models/point/dense_heads/p2p_head.py:18-572).

  forward        two towers of 4 x [conv3x3 -> GN32 -> ReLU] + 3x3 cls_out / reg_out; GroupNorm statistics ride in the
                 conv epilogues and every GN-apply+ReLU is folded into the consumer conv's load (nothing materialised)
  loss           decode kernel -> cost kernel (the shipped FocalLossCost + DisCostV2 pair fused; any other list of the registered costs
                 through the general kernel, round 6) -> ONE batched device LSA launch for all images (topk_k rounds) -> fused
                 classification (sigmoid focal | BCE | softmax CE) + regression (SmoothL1 | MSE | L1) loss straight from the assignment
  get_bboxes     sigmoid row-max -> radix-select top-k -> 16x16 pseudo boxes -> bitmask NMS
NHWC makes the reference's permute/reshape of the output maps (p2p_head.py:143-148) a free view."""
import math

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from ..core.assigners import HungarianAssignerV2
from ..core.point_generator import PointGenerator
from ..core.post_processing import multiclass_nms, multiclass_nms_batched
from ..layers import ConvModule, _PackCache, conv_gn, packed_conv
from ..registry import HEADS, build_assigner, build_sampler


TAP_PROJECTION = [True]          # test hook: False keeps the output convs on the 3x3 matrix-core kernel (A/B, tests)
BATCHED_POSTPROCESS = [True]     # test hook: False runs _get_bboxes_single image by image (the two must agree exactly)


def _get(cfg, key, default=None):
    if cfg is None:
        return default
    return cfg.get(key, default) if isinstance(cfg, dict) else getattr(cfg, key, default)


@HEADS.register_module()
class P2PHead(nn.Module):
    def __init__(self, num_classes, in_channels,
                 point_anchor=[(-0.25, -0.25), (0.25, -0.25), (0.25, 0.25), (-0.25, 0.25)],
                 assign_before_pred=False, pts_gamma=100. / 8, reg_norm=1. / 8,
                 loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                 loss_reg=dict(type='MSELoss', loss_weight=2e-4), init_cfg=None,
                 feat_channels=256, stacked_convs=4, strides=(4, 8, 16, 32, 64), dcn_on_last_conv=False,
                 conv_bias='auto', loss_bbox=None, conv_cfg=None, norm_cfg=None, train_cfg=None, test_cfg=None):
        super().__init__()
        # round 6: the head's own defaults (CrossEntropyLoss(use_sigmoid) + MSELoss, p2p_head.py:38-46) and the other registered losses
        # the fused loss kernels take (ops.P2P_CLS_MODES / P2P_REG_MODES), next to the shipped FocalLoss + SmoothL1Loss
        assert loss_cls['type'] in ('FocalLoss', 'CrossEntropyLoss') and loss_reg['type'] in ops.P2P_REG_MODES, (loss_cls, loss_reg)
        assert loss_cls['type'] != 'FocalLoss' or loss_cls.get('use_sigmoid', False), 'FocalLoss: only the sigmoid version exists (focal_loss.py:170)'
        assert norm_cfg is not None and norm_cfg['type'] == 'GN' and not dcn_on_last_conv
        self.point_anchor = torch.FloatTensor(point_anchor)
        self.num_points = len(point_anchor)
        self.use_sigmoid_cls = loss_cls.get('use_sigmoid', False)          # p2p_head.py:61-65
        self.num_classes = num_classes
        self.num_cls_out = self.cls_out_channels = num_classes if self.use_sigmoid_cls else num_classes + 1
        self.loss_cls_type = loss_cls['type']
        self.cls_mode = ops.P2P_CLS_MODES['FocalLoss' if loss_cls['type'] == 'FocalLoss' else
                                          'CrossEntropyLoss_sigmoid' if self.use_sigmoid_cls else 'CrossEntropyLoss']
        self.reg_mode = ops.P2P_REG_MODES[loss_reg['type']]
        self.loss_cls_cfg, self.loss_reg_cfg = dict(loss_cls), dict(loss_reg)
        self.in_channels, self.feat_channels, self.stacked_convs = in_channels, feat_channels, stacked_convs
        self.strides, self.conv_bias = list(strides), conv_bias
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.assign_before_pred, self.pts_gamma, self.reg_norm = assign_before_pred, pts_gamma, reg_norm
        self.point_generators = [PointGenerator() for _ in self.strides]
        if self.train_cfg:
            self.assigner = build_assigner(_get(self.train_cfg, 'assigner'))
            self.sampler = build_sampler(_get(self.train_cfg, 'sampler'))
            assert isinstance(self.assigner, HungarianAssignerV2)
        self.cls_convs, self.reg_convs = nn.ModuleList(), nn.ModuleList()
        for i in range(stacked_convs):
            chn = in_channels if i == 0 else feat_channels
            self.cls_convs.append(ConvModule(chn, feat_channels, 3, padding=1, conv_cfg=conv_cfg, norm_cfg=norm_cfg,
                                             bias=conv_bias))
            self.reg_convs.append(ConvModule(chn, feat_channels, 3, padding=1, conv_cfg=conv_cfg, norm_cfg=norm_cfg,
                                             bias=conv_bias))
        self.cls_out = nn.Conv2d(feat_channels, self.num_cls_out * self.num_points, 3, padding=1)
        self.reg_out = nn.Conv2d(feat_channels, self.num_points * 2, 3, padding=1)
        self._cache = _PackCache()
        self.init_weights()

    def init_weights(self):  # p2p_head.py:46-57
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, 0, 0.01)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        nn.init.constant_(self.cls_out.bias, float(-math.log((1 - 0.01) / 0.01)))

    # ------------------------------------------------------------------ forward (p2p_head.py:107-123)
    def _tower(self, convs, out_conv, x, tape=None):
        """tape (list): training mode -- one record per conv+GN layer and a final one for the output conv."""
        ab = None
        for m in convs:
            rec = None
            if tape is not None:
                rec = dict(kind='tower')
                tape.append(rec)
            x, ab = conv_gn(self._cache, m, x, in_ab=ab, in_relu=True, materialize=False, save=rec)
        H, W = x.shape[1:3]
        J = out_conv.out_channels
        if tape is None and TAP_PROJECTION[0] and x.dtype == torch.float32 and 9 * J <= 64 and out_conv.kernel_size == (3, 3) \
                and out_conv.padding == (1, 1) and out_conv.stride == (1, 1):
            # forward only: a 3x3 conv with 1-2 output channels wastes 97 % of a 64-cout matrix-core tile; as a 1x1 projection to
            # 9 J tap responses + a tap sum it is one (fused-GroupNorm) GEMM over the map and a 30 MB gather (csrc/postproc.hip)
            pc1 = self._cache.get(('tap1x1', id(out_conv)), [out_conv.weight], lambda: ops.PackedConv(
                out_conv.weight.detach().permute(2, 3, 0, 1).reshape(9 * J, -1)[:, :, None, None].contiguous(), 1, 0))
            if (H * W) % 128 == 0:
                R = ops.conv2d(x, pc1, in_ab=ab, in_relu=True)
            else:
                R = ops.conv2d(ops.gn_apply(x, ab[0], ab[1], relu=True), pc1)
            return ops.tap_sum3x3(R, out_conv.bias.detach(), J)
        pc = packed_conv(self._cache, out_conv)
        if tape is not None:
            tape.append(dict(kind='out', conv=out_conv, x=x, in_ab=ab))
        if (H * W) % 128 == 0:
            return ops.conv2d(x, pc, bias=out_conv.bias, in_ab=ab, in_relu=True)
        return ops.conv2d(ops.gn_apply(x, ab[0], ab[1], relu=True), pc, bias=out_conv.bias)

    def forward_single(self, feat):
        x = ops.from_nchw(feat)
        return (ops.as_nchw(self._tower(self.cls_convs, self.cls_out, x)),
                ops.as_nchw(self._tower(self.reg_convs, self.reg_out, x)))

    def forward(self, feats):
        outs = [self.forward_single(f) for f in feats]
        return [o[0] for o in outs], [o[1] for o in outs]

    def forward_train(self, x, img_metas, gt_bboxes, gt_labels=None, gt_bboxes_ignore=None, proposal_cfg=None, **kw):
        outs = self(x)
        return self.loss(*outs, gt_bboxes, gt_labels, img_metas, gt_bboxes_ignore=gt_bboxes_ignore)

    # ------------------------------------------------------------------ points (p2p_head.py:125-170, 425-465)
    def get_pred_points(self, cls_outs, pts_outs, img_metas):
        """p2p_head.py:125-170: every level decoded by the decode kernel, levels concatenated along the proposal axis in level order
        (the reference's torch.cat over levels)."""
        assert len(cls_outs) == len(pts_outs) == len(self.strides)
        pa = None
        anchors, preds, valids, clss = [], [], [], []
        for lvl, stride in enumerate(self.strides):
            cls = ops.from_nchw(cls_outs[lvl])
            reg = ops.from_nchw(pts_outs[lvl])
            B, H, W, _ = cls.shape
            if pa is None:
                pa = self._cache.get('pa', [], lambda: self.point_anchor.to(cls.device).contiguous())
            pred, anchor = ops.p2p_decode(reg, pa, stride, self.pts_gamma, want_anchor=True)
            clss.append(cls.reshape(B, H * W * self.num_points, self.num_cls_out))
            anchors.append(anchor), preds.append(pred)
            # valid_flags (p2p_head.py:451-463, PointGenerator.valid_flags): cells beyond ceil(pad_shape / stride) -- an image
            # padded less than the batch maximum -- are invalid: not assigned, label weight 0
            valid = None
            for b, m in enumerate(img_metas):
                ph, pw = m['pad_shape'][:2]
                vh, vw = min(int(np.ceil(ph / stride)), H), min(int(np.ceil(pw / stride)), W)
                if vh != H or vw != W:
                    if valid is None:
                        valid = torch.ones((B, H, W, self.num_points), dtype=torch.bool)
                    valid[b, vh:] = False
                    valid[b, :, vw:] = False
            valids.append(None if valid is None else valid.reshape(B, -1))
        if len(self.strides) == 1:
            anchor, pred, cls = anchors[0], preds[0], clss[0]
        else:
            anchor, pred, cls = torch.cat(anchors, 1), torch.cat(preds, 1), torch.cat(clss, 1)
        B = cls.shape[0]
        if all(v is None for v in valids):
            valid = torch.ones((B, cls.shape[1]), dtype=torch.bool, device=cls.device)
            valid.all_valid = True
        else:
            valid = torch.cat([torch.ones((B, c.shape[1]), dtype=torch.bool) if v is None else v for v, c in zip(valids, clss)], 1).to(cls.device)
            valid.all_valid = False
        return anchor, pred, valid, cls

    @staticmethod
    def pseudo_bbox_to_center(gt_bboxes):
        return [ops.box_centers(b.float().contiguous()) for b in gt_bboxes]

    # ------------------------------------------------------------------ loss (p2p_head.py:172-328)
    def assign_batch(self, proposals, cls, gt_points, gt_labels, img_metas, valid=None):
        """All images' Hungarian problems in one LSA launch.  Returns gt_inds (B, M) int64: j+1 = gt j, 0 = background,
        -1 = invalid cell (``valid`` false: not offered to the assigner, label weight 0 -- p2p_head.py:288-305).
        Degenerate images follow HungarianAssignerV2.assign (hungarian_assigner.py:207-219,251): no gts -> all background;
        fewer proposals than gts with topk_k > 1 -> the loop never runs, all background."""
        a = self.assigner
        B, M = proposals.shape[:2]
        all_valid = True if valid is None else getattr(valid, 'all_valid', None)
        if all_valid is None:
            all_valid = bool(valid.all())
        masked = not all_valid
        out = proposals.new_zeros((B, M), dtype=torch.long)
        costs, where, costs_t, where_t = [], [], [], []
        for b in range(B):
            idx = torch.nonzero(valid[b], as_tuple=False).squeeze(1) if masked else None
            if masked:
                out[b][~valid[b]] = -1
            props, c = (proposals[b][idx], cls[b][idx]) if masked else (proposals[b], cls[b])
            G, Mb = gt_points[b].shape[0], props.shape[0]
            if G == 0 or Mb == 0:
                continue
            if Mb <= G:
                if a.topk_k == 1 or Mb == G:      # (a square problem: one round assigns everything, in scipy's orientation)
                    # linear_sum_assignment on the (Mb, G) cost (hungarian_assigner.py:229-240): every proposal gets a
                    # distinct gt (HungarianAssignerV2.transposed_inds)
                    costs_t.append(a.cost_t(props.contiguous(), c.contiguous(), gt_points[b], gt_labels[b], img_metas[b]))
                    where_t.append((b, idx))
                # topk_k > 1: `cost_new.shape[0] // num_gts != 0` is false at once, nothing is assigned (:245-248)
                continue
            costs.append(a.cost_t(props.contiguous(), c.contiguous(), gt_points[b], gt_labels[b], img_metas[b]))
            where.append((b, idx))
        if costs:
            inds, _ = ops.lsa_topk(costs, a.topk_k)
            for (b, idx), gi in zip(where, inds):
                if idx is None:
                    out[b] = gi
                else:
                    out[b][idx] = gi
        if costs_t:
            for (b, idx), gi in zip(where_t, a.transposed_inds(costs_t)):
                if idx is None:
                    out[b] = gi
                else:
                    out[b][idx] = gi
        return out

    def loss(self, cls_outs, pts_outs, gt_bboxes, gt_labels, img_metas, gt_bboxes_ignore=None, save=None):
        for gb in gt_bboxes:
            assert len(gb) > 0, gt_bboxes
        anchor, pred, valid, cls = self.get_pred_points(cls_outs, pts_outs, img_metas)
        dev = cls.device
        gt_points = self.pseudo_bbox_to_center([b.to(dev) for b in gt_bboxes])
        gt_labels = [l.to(dev) for l in gt_labels]
        proposals = anchor if self.assign_before_pred else pred
        gt_inds = self.assign_batch(proposals, cls, gt_points, gt_labels, img_metas, valid)
        counts = [len(l) for l in gt_labels]
        start = torch.from_numpy(np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)).to(dev)
        lc, lr = self.loss_cls_cfg, self.loss_reg_cfg
        out = ops.p2p_loss(cls.contiguous(), pred, gt_inds.contiguous(), torch.cat(gt_points).contiguous(),
                           torch.cat(gt_labels).to(torch.int32).contiguous(), start, lc.get('alpha', 0.25),
                           lc.get('gamma', 2.0), lr.get('beta', 1.0), _get(self.train_cfg, 'pos_weight', 1.0),
                           _get(self.train_cfg, 'neg_weight', 1.0), self.reg_norm, lc.get('loss_weight', 1.0),
                           lr.get('loss_weight', 1.0), self.cls_mode, self.reg_mode)
        B = out.shape[0]
        if save is not None:      # what the loss backward re-reads (training.P2PTrainer)
            save.update(cls=cls.contiguous(), pred=pred, gt_inds=gt_inds.contiguous(), gt_pts=torch.cat(gt_points).contiguous(),
                        gt_labels=torch.cat(gt_labels).to(torch.int32).contiguous(), gt_start=start, out=out)
        return {'loss_cls': [out[b, 0] for b in range(B)], 'loss_pts': [out[b, 1] for b in range(B)]}

    def get_targets(self, pred_pts, valid_flag_list, cls_outs_list, gt_points, gt_labels, img_metas,
                    gt_points_ignore=None, unmap_outputs=True):
        """Reference-format targets (p2p_head.py:250-328) derived from the device assignment."""
        valid = valid_flag_list if torch.is_tensor(valid_flag_list) else torch.stack(list(valid_flag_list))
        gt_inds = self.assign_batch(pred_pts, cls_outs_list, gt_points, gt_labels, img_metas, valid)
        labels, lw, tgt, w = [], [], [], []
        pos_w = _get(self.train_cfg, 'pos_weight', 1.0)
        neg_w = _get(self.train_cfg, 'neg_weight', 1.0)
        for b in range(gt_inds.shape[0]):
            gi = gt_inds[b]
            pos = gi > 0
            lab = gi.new_full(gi.shape, self.num_classes)
            lab[pos] = gt_labels[b][gi[pos] - 1]
            t = pred_pts.new_zeros((gi.shape[0], 2))
            t[pos] = gt_points[b][gi[pos] - 1]
            ww = pred_pts.new_zeros((gi.shape[0], 2))
            ww[pos] = 1.0
            l_w = pred_pts.new_full(gi.shape, 1.0 if neg_w <= 0 else neg_w)
            l_w[pos] = pos_w
            inv = gi < 0                       # unmap(fill=0) of the reference: label 0, weight 0 on invalid cells
            lab[inv] = 0
            l_w[inv] = 0
            labels.append(lab), lw.append(l_w), tgt.append(t), w.append(ww)
        return labels, lw, tgt, w

    # ------------------------------------------------------------------ inference (p2p_head.py:330-423)
    def get_bboxes(self, cls_outs, pts_outs, img_metas, cfg=None, rescale=False, with_nms=True):
        """p2p_head.py:330-343.  ``with_nms`` is accepted and, as in the reference, NOT forwarded: its get_bboxes calls
        _get_bboxes_single without it (:339-340), so the NMS branch always runs; the no-NMS branch (:405-423) is reachable only by
        calling _get_bboxes_single directly (softmax heads: see there)."""
        anchor, pred, valid, cls = self.get_pred_points(cls_outs, pts_outs, img_metas)
        if BATCHED_POSTPROCESS[0] and self.use_sigmoid_cls and len(self.strides) == 1:
            return self._get_bboxes_batched(pred[..., :2], cls, img_metas, cfg, rescale, True)
        res = []
        for b in range(len(img_metas)):
            pts_scores, labels = self._get_bboxes_single(pred[b][..., :2], valid[b], cls[b], img_metas[b]['img_shape'],
                                                         img_metas[b]['scale_factor'], cfg, rescale)
            res.append((self.center_to_pseudo_bbox([pts_scores])[0], labels))
        return res

    def _scores(self, logits):
        """cls_score.sigmoid() / .softmax(-1) of p2p_head.py:362 -> (scores (n, C'), the per-proposal maximum over the FOREGROUND classes).
        sigmoid: torch's CPU bits (the scores order top-k and NMS); softmax (CrossEntropyLoss without use_sigmoid): torch's device softmax."""
        if self.use_sigmoid_cls:
            if self.num_cls_out == 1:
                m = ops.rowmax_sigmoid(logits)
                return m[:, None], m
            return ops.sigmoid_exact(logits), ops.rowmax_sigmoid(logits)
        sc = torch.softmax(logits, dim=-1)
        return sc, sc[:, :-1].max(dim=1)[0]

    def _get_bboxes_single(self, pred_pts, valid_flag, cls_outs, img_shape, scale_factor, cfg, rescale=False,
                           with_nms=True):
        cfg = self.test_cfg if cfg is None else cfg
        nms_pre = _get(cfg, 'nms_pre', -1)
        # p2p_head.py:357-376: the concatenated proposals are cut into len(strides) EQUAL chunks (a reshape, whatever the levels' true
        # sizes are) and the top nms_pre of every chunk survive -- restated as it stands
        L = len(self.strides)
        assert pred_pts.shape[0] % L == 0, 'the reference reshapes the proposals to (len(strides), -1, 2) (p2p_head.py:357)'
        pts_l, sc_l = [], []
        for logits, pp in zip(cls_outs.contiguous().reshape(L, -1, self.num_cls_out), pred_pts.reshape(L, -1, 2)):
            logits = logits.contiguous()
            scores, rowmax = self._scores(logits)
            if 0 < nms_pre < logits.shape[0]:
                _, topk_inds = ops.topk_desc(rowmax.contiguous(), nms_pre)
                scores, pp = scores[topk_inds], pp[topk_inds]
            x = pp[:, 0].clamp(min=0, max=img_shape[1])
            y = pp[:, 1].clamp(min=0, max=img_shape[0])
            pts_l.append(torch.stack([x, y], dim=-1)), sc_l.append(scores)
        pts, scores = (pts_l[0], sc_l[0]) if L == 1 else (torch.cat(pts_l), torch.cat(sc_l))
        if rescale:
            pts = pts / pts.new_tensor(scale_factor[:2])
        if self.use_sigmoid_cls:
            scores = torch.cat([scores, scores.new_zeros(scores.shape[0], 1)], dim=1)
        if not with_nms:
            # p2p_head.py:405-423: every (point, class) pair -- the background column included, as the reference expands num_cls_out + 1
            # ... it expands to num_cls_out columns of a (n, num_cls_out [+ 1]) score matrix: restated through its own reshape
            n, Cs = scores.shape
            cols = self.num_cls_out
            mp = pts[:, None].expand(n, cols, 2).reshape(-1, 2)
            labels = torch.arange(cols, dtype=torch.long, device=pts.device).view(1, -1).expand(n, cols)
            # mlvl_scores (n, Cs) is flattened whole while points / labels are expanded to num_cls_out columns: equal only when Cs == cols
            assert Cs == cols, 'with_nms=False indexes (n * %d) points with (n * %d) scores in the reference (p2p_head.py:408-418): ' \
                'only a softmax head (no padded background column) runs there' % (cols, Cs)
            ms, labels = scores.reshape(-1), labels.reshape(-1)
            inds = (ms > _get(cfg, 'score_thr')).nonzero(as_tuple=False).squeeze(1)
            points, sc, labels = mp[inds], ms[inds], labels[inds]
            dets = torch.cat([points, sc[:, None]], -1)
            max_per_img = _get(cfg, 'max_per_img')
            if 0 < max_per_img < len(sc):
                _, idx = ops.topk_desc(sc.contiguous(), max_per_img)
                dets, labels = dets[idx], labels[idx]
            return dets, labels
        wh = pts.new_tensor(_get(self.test_cfg, 'pseudo_wh', (16, 16)))
        boxes = torch.cat([pts - wh / 2, pts + wh / 2], dim=-1)
        dets, labels = multiclass_nms(boxes, scores, _get(cfg, 'score_thr'), _get(cfg, 'nms'), _get(cfg, 'max_per_img'))
        ctr = torch.stack([(dets[:, 0] + dets[:, 2]) * 0.5, (dets[:, 1] + dets[:, 3]) * 0.5, dets[:, 4]], dim=-1)
        return ctr, labels

    def _get_bboxes_batched(self, pred_pts, cls, img_metas, cfg, rescale=False, with_nms=True):
        """_get_bboxes_single (p2p_head.py:345-423) for the whole batch at once: one row-max-sigmoid, one top-k, one candidate
        compaction and one NMS launch over all images, and a single host read (candidate + keep counts) where the per-image
        loop pays two synchronisations per image.  pred_pts (B, M, 2), cls (B, M, C).  Same arithmetic per element as the
        per-image path (``BATCHED_POSTPROCESS[0] = False`` keeps that one: tests compare the two)."""
        cfg = self.test_cfg if cfg is None else cfg
        assert with_nms
        B, M, C = cls.shape
        dev = cls.device
        nms_pre = _get(cfg, 'nms_pre', -1)
        logits = cls.contiguous()
        if 0 < nms_pre < M:
            rs = ops.rowmax_sigmoid(logits.view(B * M, C)).view(B, M)
            _, topk_inds = ops.topk_desc_batched(rs, nms_pre)
            logits = torch.gather(logits, 1, topk_inds[..., None].expand(-1, -1, C))
            pred_pts = torch.gather(pred_pts, 1, topk_inds[..., None].expand(-1, -1, 2))
        n = logits.shape[1]
        flat = logits.reshape(B * n, C)
        scores = (ops.rowmax_sigmoid(flat)[:, None] if self.num_cls_out == 1 else ops.sigmoid_exact(flat)).view(B, n, C)
        shapes = [tuple(m['img_shape'][:2]) for m in img_metas]
        if len(set(shapes)) == 1:
            x = pred_pts[..., 0].clamp(min=0, max=shapes[0][1])
            y = pred_pts[..., 1].clamp(min=0, max=shapes[0][0])
        else:
            lim = torch.tensor([[s[1], s[0]] for s in shapes], dtype=torch.float32).to(dev)          # (B, 2): (w, h)
            x = torch.minimum(pred_pts[..., 0].clamp(min=0), lim[:, 0:1])
            y = torch.minimum(pred_pts[..., 1].clamp(min=0), lim[:, 1:2])
        pts = torch.stack([x, y], dim=-1)
        if rescale:
            sf = torch.tensor([list(m['scale_factor'][:2]) for m in img_metas], dtype=torch.float32).to(dev)
            pts = pts / sf[:, None, :]
        scores = torch.cat([scores, scores.new_zeros(B, n, 1)], dim=2)
        wh = pts.new_tensor(_get(self.test_cfg, 'pseudo_wh', (16, 16)))
        boxes = torch.cat([pts - wh / 2, pts + wh / 2], dim=-1)
        res = []
        for dets, labels in multiclass_nms_batched(boxes, scores, _get(cfg, 'score_thr'), _get(cfg, 'nms'),
                                                   _get(cfg, 'max_per_img')):
            ctr = torch.stack([(dets[:, 0] + dets[:, 2]) * 0.5, (dets[:, 1] + dets[:, 3]) * 0.5, dets[:, 4]], dim=-1)
            res.append((self.center_to_pseudo_bbox([ctr])[0], labels))
        return res

    def center_to_pseudo_bbox(self, center_scores):
        wh = center_scores[0].new_tensor(_get(self.test_cfg, 'pseudo_wh', (16, 16)))
        return [torch.cat([c[:, :2] - wh / 2, c[:, :2] + wh / 2, c[:, 2:]], dim=-1) for c in center_scores]

    def simple_test(self, feats, img_metas, rescale=False, **kwargs):
        return self.get_bboxes(*self(feats), img_metas, rescale=rescale)

    # ------------------------------------------------------------------ tile / flip TTA merge (p2p_head.py:487-572)
    @staticmethod
    def bbox_mapping_back(bboxes, img_shape, scale_factor, flip, flip_direction, tile_offset=None):
        """T/mmdet/core/bbox/transforms.py:5-31,62-80 (the fork adds ``tile_offset``): un-flip, divide by the test scale, move a
        tile's detections to their place in the original image."""
        new = bboxes
        if flip:
            assert flip_direction in ('horizontal', 'vertical', 'diagonal')
            new = bboxes.clone()
            if flip_direction in ('horizontal', 'diagonal'):
                new[..., 0::4] = img_shape[1] - bboxes[..., 2::4]
                new[..., 2::4] = img_shape[1] - bboxes[..., 0::4]
            if flip_direction in ('vertical', 'diagonal'):
                new[..., 1::4] = img_shape[0] - bboxes[..., 3::4]
                new[..., 3::4] = img_shape[0] - bboxes[..., 1::4]
        new = new.view(-1, 4) / new.new_tensor(scale_factor)
        assert tile_offset is None or (isinstance(tile_offset, (tuple, list)) and len(tile_offset) == 2), \
            'tile_offset must be None or (dx, dy) or [dx, dy]'
        if tile_offset is not None:
            dx, dy = tile_offset
            new[:, [0, 2]] += dx
            new[:, [1, 3]] += dy
        return new.view(bboxes.shape)

    def aug_test(self, feats, img_metas, rescale=False):
        return self.aug_test_bboxes(feats, img_metas, rescale=rescale)

    def aug_test_bboxes(self, feats, img_metas, rescale=False):
        """Test-time augmentation of the fork's tile inference (p2p_head.py:487-572): every augmentation (one image each) runs
        forward + get_bboxes (top-k, pseudo-box NMS) in ITS frame, the surviving pseudo boxes are mapped back to the original
        image (flip, scale, tile offset), and one more class-aware NMS runs over the union.  feats: list (augmentations) of
        feature tuples; img_metas: list of one-element lists.  -> [(dets (n, 5), labels (n,))]."""
        aug_bboxes, aug_scores = [], []
        for x, img_meta in zip(feats, img_metas):
            assert len(img_meta) == 1, 'one image per augmentation (dense_test_mixins.py:192)'
            dets, labels = self.get_bboxes(*self(x), img_meta, self.test_cfg, False, True)[0]
            scores = dets.new_zeros((dets.shape[0], self.num_classes))
            scores[torch.arange(dets.shape[0], device=dets.device), labels] = dets[:, 4]
            m = img_meta[0]
            aug_bboxes.append(self.bbox_mapping_back(dets[:, :4], m['img_shape'], m['scale_factor'], m['flip'],
                                                     m['flip_direction'], m.get('tile_offset', None)))
            aug_scores.append(scores)
        merged_bboxes, merged_scores = torch.cat(aug_bboxes, dim=0), torch.cat(aug_scores, dim=0)
        merged_scores = torch.cat([merged_scores, merged_scores.new_zeros(merged_scores.shape[0], 1)], dim=1)   # bg column
        det_bboxes, det_labels = multiclass_nms(merged_bboxes, merged_scores, _get(self.test_cfg, 'score_thr'),
                                                _get(self.test_cfg, 'nms'), _get(self.test_cfg, 'max_per_img'))
        if not rescale:
            det_bboxes = det_bboxes.clone()
            det_bboxes[:, :4] *= det_bboxes.new_tensor(img_metas[0][0]['scale_factor'])
        return [(det_bboxes, det_labels)]
