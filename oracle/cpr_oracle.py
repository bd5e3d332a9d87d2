"""CPU oracle for the CPR / P2P point-localization hot path.

TEST INFRASTRUCTURE ONLY: a functional torch-CPU (fp32) restatement of the reference's
algorithm, keyed on a state dict with the reference's parameter names.  It is the checker
for the HIP path (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg) and is never
imported by ``pointtinybenchmark_amd``.

Parity pinning: every function below is compared against fixtures in ``tests/golden/`` that were generated from the
reference's OWN classes (loaded read-only by ``oracle/ref_loader.py``) by ``oracle/gen_golden.py`` /
``oracle/gen_golden_r2.py`` (build container only; the comparison, ``tests/test_oracle_golden.py``, runs everywhere):
feature maps, bag points, validity, negative masks, assigner indices and the PointRefiner's chosen-point masks bit for bit,
logits / losses to 2e-6, gradients through torch autograd to 1e-3.  The reference's own known-answer vectors for this path
(PointAssigner, T/tests/test_utils/test_assigner.py:155-194) are asserted in ``tests/test_gpu_assigners.py``.
One op is PARITY UNPINNED: ``batched_nms`` (mmcv-full, un-vendored) -- see its docstring.

Citations: ``T/`` = /root/reference/TOV_mmdetection/.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# ------------------------------------------------------------------------------------------------
# backbone / neck / head towers  (floating point; torch fp32 reference of the conv kernels)
# ------------------------------------------------------------------------------------------------
ARCH = {18: ('basic', (2, 2, 2, 2)), 34: ('basic', (3, 4, 6, 3)), 50: ('bottleneck', (3, 4, 6, 3)),
        101: ('bottleneck', (3, 4, 23, 3)), 152: ('bottleneck', (3, 8, 36, 3))}


def _bn_eval(x, sd, p, eps=1e-5):
    # BN is always in eval mode on this path (norm_eval=True; T/mmdet/models/backbones/resnet.py:647-657)
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'], sd[p + '.weight'], sd[p + '.bias'],
                        False, 0.0, eps)


def resnet_forward(sd, x, depth=50, out_indices=(0, 1, 2, 3), prefix='backbone.'):
    """T/mmdet/models/backbones/resnet.py:630-645 (stem 564-610, Bottleneck 262-302 with stride on
    conv2 = style 'pytorch' 153-158, BasicBlock 66-93, downsample T/mmdet/models/utils/res_layer.py:40-60)."""
    kind, blocks = ARCH[depth]
    x = F.conv2d(x, sd[prefix + 'conv1.weight'], None, 2, 3)
    x = F.relu(_bn_eval(x, sd, prefix + 'bn1'))
    x = F.max_pool2d(x, 3, 2, 1)
    outs = []
    for li, nb in enumerate(blocks):
        for bi in range(nb):
            p = '%slayer%d.%d.' % (prefix, li + 1, bi)
            stride = 2 if (bi == 0 and li > 0) else 1
            identity = x
            if kind == 'bottleneck':
                o = F.relu(_bn_eval(F.conv2d(x, sd[p + 'conv1.weight']), sd, p + 'bn1'))
                o = F.relu(_bn_eval(F.conv2d(o, sd[p + 'conv2.weight'], None, stride, 1), sd, p + 'bn2'))
                o = _bn_eval(F.conv2d(o, sd[p + 'conv3.weight']), sd, p + 'bn3')
            else:
                o = F.relu(_bn_eval(F.conv2d(x, sd[p + 'conv1.weight'], None, stride, 1), sd, p + 'bn1'))
                o = _bn_eval(F.conv2d(o, sd[p + 'conv2.weight'], None, 1, 1), sd, p + 'bn2')
            if (p + 'downsample.0.weight') in sd:
                identity = _bn_eval(F.conv2d(x, sd[p + 'downsample.0.weight'], None, stride), sd, p + 'downsample.1')
            x = F.relu(o + identity)
        if li in out_indices:
            outs.append(x)
    return tuple(outs)


def _conv_gn(x, sd, p, padding, relu, groups=32, eps=1e-5):
    # mmcv ConvModule order conv -> norm -> act; bias='auto' => no conv bias when a norm follows
    x = F.conv2d(x, sd[p + '.conv.weight'], sd.get(p + '.conv.bias'), 1, padding)
    x = F.group_norm(x, groups, sd[p + '.gn.weight'], sd[p + '.gn.bias'], eps)
    return F.relu(x) if relu else x


def fpn_forward(sd, inputs, start_level=0, num_outs=1, prefix='neck.'):
    """T/mmdet/models/necks/fpn.py:166-194 with the fork's num_outs < #levels edit (96,134,193):
    all laterals (1x1 conv + GN, no act) and the top-down nearest-upsample adds run; only the first
    ``num_outs`` 3x3 fpn_convs (+GN, no act) are evaluated."""
    lat = [_conv_gn(inputs[i + start_level], sd, '%slateral_convs.%d' % (prefix, i), 0, False)
           for i in range(len(inputs) - start_level)]
    for i in range(len(lat) - 1, 0, -1):
        lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode='nearest')
    return tuple(_conv_gn(lat[i], sd, '%sfpn_convs.%d' % (prefix, i), 1, False) for i in range(min(len(lat), num_outs)))


def cpr_head_forward(sd, feats, stacked_convs=4, prefix='bbox_head.'):
    """T/mmdet/models/point/dense_heads/cpr_head.py:1030-1043 (ins_share_head_feat=True)."""
    cls_feats = []
    for x in feats:
        for i in range(stacked_convs):
            x = _conv_gn(x, sd, '%scls_convs.%d' % (prefix, i), 1, True)
        cls_feats.append(x)
    return cls_feats, cls_feats


# ------------------------------------------------------------------------------------------------
# CPR point generators  (cpr_head.py:125-290, 442-497)
# ------------------------------------------------------------------------------------------------
def circle_offsets(radius, stride, base_num_point=8, start_angle=0.0):
    """Ring offsets of CirclePtFeatGenerator.get_point_neighbours (cpr_head.py:474-497):
    ring i (1..R): radius i*stride, 8*i points at angles k/(8i)*360+start (deg), fp32 cos/sin."""
    out = []
    for i in range(radius):
        r = (i + 1) * stride
        n = base_num_point * (i + 1)
        ang = torch.arange(n).float() / n * 360 + start_angle
        ang = ang / 360 * np.pi * 2
        out.append(torch.stack([r * torch.cos(ang), r * torch.sin(ang)], dim=-1))
    return torch.cat(out)


def bag_points(centers, radius, stride):
    """centers (G,2) -> (G, K, 2): rings then the centre LAST (cpr_head.py:492-496)."""
    pts = circle_offsets(radius, stride).unsqueeze(0) + centers.reshape(-1, 1, 2)
    return torch.cat([pts, centers.unsqueeze(1)], dim=1)


def inside(pts, h, w):
    """get_point_valid (cpr_head.py:172-180)."""
    return (0 <= pts[..., 0]) & (pts[..., 0] < w) & (0 <= pts[..., 1]) & (pts[..., 1] < h)


def sample_bilinear(feat, pts_over_stride, align_corners=False):
    """grid_sample wrapper of cpr_head.py:73-93 (align_corners=False: border padding; True: zeros padding).
    feat (1,C,H,W); pts (G,K,2) already divided by the stride -> (G,K,C)."""
    h, w = feat.shape[2:]
    wh = feat.new_tensor([w, h])
    if align_corners:
        grid = 2 * pts_over_stride.unsqueeze(0) / (wh - 1) - 1
        return F.grid_sample(feat, grid, align_corners=True, padding_mode='zeros').permute(0, 2, 3, 1)[0]
    grid = (2 * pts_over_stride.unsqueeze(0) + 1) / wh - 1
    return F.grid_sample(feat, grid, align_corners=False, padding_mode='border').permute(0, 2, 3, 1)[0]


def grid_points(h, w, stride):
    """AnchorPtFeatGenerator.anchor_points (cpr_head.py:240-244): (x*s + s/2, y*s + s/2), row-major."""
    y, x = torch.meshgrid(torch.arange(h), torch.arange(w), indexing='ij')
    return torch.stack([x, y], dim=-1) * stride + stride / 2


def neg_valid_mask(h, w, stride, radius, centers, labels, num_classes, pad_h, pad_w, class_wise=True):
    """OutCirclePtFeatGenerator.generate (cpr_head.py:254-290): valid(N, C) bool.
    NOTE the distance is torch.cdist (matmul-based for > 25 rows on CPU): that exact op is kept here so
    the mask is the reference's, bit for bit."""
    pts = grid_points(h, w, stride).flatten(0, 1).float()
    v = inside(pts, pad_h, pad_w)[:, None].repeat(1, num_classes).float()
    if class_wise:
        for l in sorted(set(labels.tolist())):
            c = centers[labels == l]
            dist = torch.cdist(pts, c, p=2)
            v[:, l] = v[:, l] * (dist.min(dim=1)[0] >= stride * radius).float()
    else:
        dist = torch.cdist(pts, centers, 2)
        v = v * (dist.min(dim=1)[0] >= stride * radius)[:, None].float()
    return pts, v.bool()


# ------------------------------------------------------------------------------------------------
# CPR loss  (cpr_head.py:1101-1229, multi_instance_learning_loss.py:148-203)
# ------------------------------------------------------------------------------------------------
def gfocal(p, q, w, eps=1e-6):
    l1 = (p - q) ** 2
    l2 = q * (p + eps).log() + (1 - q) * (1 - p + eps).log()
    return -(l1 * l2 * w).sum(dim=-1)


def mil_loss(bag_cls_prob, bag_ins_outs, labels, valid, loss_weight=1.0, eps=1e-6):
    """MILLoss.forward, binary_ins=False, gfocal (multi_instance_learning_loss.py:153-203).
    bag_cls_prob/bag_ins_outs (B,N,C), valid (B,N,1) float, labels (B,)."""
    B, N, C = bag_cls_prob.shape
    prob_ins = bag_ins_outs.reshape(B, N, C, 1).softmax(dim=1) * valid.unsqueeze(-1)
    prob_ins = F.normalize(prob_ins, dim=1, p=1)
    prob = (bag_cls_prob.unsqueeze(-1) * prob_ins).sum(dim=1)[..., 0]      # (B, C)
    # accuracy(prob, labels): top-1 over classes, in percent (T/mmdet/models/losses/accuracy.py)
    acc = (prob.argmax(dim=1) == labels).float().sum() * (100.0 / B) if B > 0 else prob.new_tensor(0.)
    label_weights = (valid.sum(dim=1) > 0).float()                           # (B, 1)
    onehot = F.one_hot(labels, C).float()
    num_sample = max(float((label_weights.sum(dim=-1) > 0).float().sum().item()), 1.0)
    loss = gfocal(prob, onehot, label_weights, eps)
    return loss.sum() / num_sample * loss_weight, acc, num_sample


def cpr_points_and_logits(sd, cls_feat, gt_bboxes, gt_labels, img_metas, stride, radius, num_classes,
                          prefix='bbox_head.'):
    """Extraction + scoring shared by loss and refine (cpr_head.py:614-662, 1045-1078), single FPN level,
    num_refine=1, ins_share_head_feat=True; the optional FC stack (num_cls_fcs > 0, get_pts_outs.forward_with_fc
    :1055-1059) is read off the state dict: relu(fc_i(...)) shared by the cls and ins classifiers.  Returns per-image lists."""
    Wc, bc = sd[prefix + 'cls_out.weight'], sd[prefix + 'cls_out.bias']
    Wi, bi = sd[prefix + 'ins_out.weight'], sd[prefix + 'ins_out.bias']
    nfc = 0
    while prefix + 'cls_fcs.%d.weight' % nfc in sd:
        nfc += 1

    def fcs(x):
        shape = x.shape
        x = x.flatten(0, -2)
        for i in range(nfc):
            x = F.relu(F.linear(x, sd[prefix + 'cls_fcs.%d.weight' % i], sd[prefix + 'cls_fcs.%d.bias' % i]))
        return x.reshape(*shape[:-1], -1)
    out = []
    for b in range(len(gt_bboxes)):
        centers = (gt_bboxes[b][:, :2] + gt_bboxes[b][:, 2:]) / 2           # pseudo_bbox_to_center :1293-1301
        ph, pw = img_metas[b]['pad_shape'][:2]
        feat = cls_feat[b:b + 1]
        pts = bag_points(centers, radius, stride)                             # (G,K,2)
        valid = inside(pts, ph, pw)                                           # (G,K)
        bag_feat = sample_bilinear(feat, pts / stride)                        # (G,K,256)
        h, w = feat.shape[2:]
        npts, nvalid = neg_valid_mask(h, w, stride, radius, centers, gt_labels[b], num_classes, ph, pw)
        nfeat = fcs(feat.permute(0, 2, 3, 1)[0].flatten(0, 1))                # (N,256) -> FC stack
        bag_feat = fcs(bag_feat)
        out.append(dict(centers=centers, pts=pts, valid=valid,
                        cls_logit=F.linear(bag_feat, Wc, bc), ins_logit=F.linear(bag_feat, Wi, bi),
                        neg_pts=npts, neg_valid=nvalid, neg_logit=F.linear(nfeat, Wc, bc)))
    return out


def cpr_loss(sd, cls_feat, gt_bboxes, gt_labels, img_metas, stride=4, radius=5, num_classes=1,
             mil_weight=0.25, neg_weight=0.75, gt_weight=0.25, with_gt_loss=True, prefix='bbox_head.'):
    """CPRHead.loss -> loss0 (cpr_head.py:1101-1229) for refine_bag_policy='independent_with_gt_bag',
    num_refine=1, sigmoid probabilities, gt_loss_type='gt_refine'.  random_remove (1119-1129) only edits
    the stride column of pts and has no effect on any output, so it is omitted."""
    per = cpr_points_and_logits(sd, cls_feat, gt_bboxes, gt_labels, img_metas, stride, radius, num_classes, prefix)
    labels = torch.cat(gt_labels)
    cls_logit = torch.cat([p['cls_logit'] for p in per])                     # (G,K,C)
    ins_logit = torch.cat([p['ins_logit'] for p in per])
    valid = torch.cat([p['valid'] for p in per]).unsqueeze(-1).float()       # (G,K,1)
    losses = {}
    if with_gt_loss:
        gt_prob = cls_logit[:, -1, :].sigmoid()
        gt_w = valid[:, -1, :] * 1.0
        num_pos = max((gt_w > 0).sum(), 1)
        onehot = F.one_hot(labels, num_classes).float()
        losses['gt_loss'] = gt_weight * (gfocal(gt_prob, onehot, gt_w).sum() / num_pos)
    pos_loss, bag_acc, num_pos = mil_loss(cls_logit.sigmoid(), ins_logit, labels, valid, mil_weight)
    losses['pos_loss'], losses['bag_acc'] = pos_loss, bag_acc
    neg_prob = torch.cat([p['neg_logit'] for p in per]).sigmoid()
    neg_valid = torch.cat([p['neg_valid'] for p in per]).float()
    losses['neg_loss'] = neg_weight * (gfocal(neg_prob, torch.zeros_like(neg_prob), neg_valid).sum() / num_pos)
    return losses, per


# ------------------------------------------------------------------------------------------------
# CPR refine  (cpr_head.py:711-850, 1231-1283)
# ------------------------------------------------------------------------------------------------
def cpr_refine(sd, cls_feat, gt_bboxes, gt_labels, gt_anns_id, img_metas, stride=4, radius=5, num_classes=1,
               gt_alpha=0.5, merge_th=0.1, refine_th=0.1, classify_filter=True, nearest_filter=True,
               rescale=False, prefix='bbox_head.'):
    """CPRHead.get_bboxes + PointRefiner.refine_single, num_refine=1, return_score_type='mean'.
    Returns per image dict(dets (G,6), labels, refine_pts, scores, not_refine, merge_valid (G,K) bool)."""
    per = cpr_points_and_logits(sd, cls_feat, gt_bboxes, gt_labels, img_metas, stride, radius, num_classes, prefix)
    res = []
    for b, p in enumerate(per):
        labels, centers = gt_labels[b], p['centers']
        G, K = p['valid'].shape
        prob = p['cls_logit'].sigmoid()                                       # (G,K,C)
        mv = p['valid'].clone()
        if nearest_filter:                                                    # :711-743 class-wise
            nv = torch.ones(G, K, dtype=torch.bool)
            for l in sorted(set(labels.tolist())):
                idx = torch.nonzero(labels == l).squeeze(1)
                if len(idx) > 1:
                    d = torch.cdist(p['pts'][idx].flatten(0, 1), centers[idx], p=2)
                    closest = d.min(dim=1)[1].reshape(len(idx), K)
                    nv[idx] = closest == torch.arange(len(idx)).reshape(-1, 1)
            mv &= nv
        if classify_filter:                                                   # :745-756
            mv &= prob.max(dim=-1)[1] == labels.reshape(-1, 1)
        ar = torch.arange(G)
        pl = prob[ar, :, labels]                                              # (G,K)
        gp = pl[:, -1:]
        mv &= (pl > merge_th) & (pl > gp * gt_alpha)                          # :823
        ih, iw = img_metas[b]['img_shape'][:2]
        x, y = p['pts'][..., 0], p['pts'][..., 1]
        mv &= (x < iw) & (x >= 0) & (y < ih) & (y >= 0)                       # :773-778
        pm = pl * mv.float()
        wgt = pm / (pm.sum(dim=1, keepdim=True) + 1e-8)
        rp = (p['pts'] * wgt.unsqueeze(-1)).sum(dim=1)                        # :833
        sc = pm.sum(dim=-1) / ((pm > 0).float().sum(dim=-1) + 1e-8)           # :835
        nr = sc < refine_th
        rp[nr] = centers[nr]
        box = torch.cat([rp - 8, rp + 8], dim=-1)                             # center_to_pseudo_bbox :1303-1309
        if rescale:
            box = box / box.new_tensor(img_metas[b]['scale_factor'])
        dets = torch.cat([box, sc[:, None], gt_anns_id[b][:, None].type_as(sc)], dim=-1)
        res.append(dict(dets=dets, labels=labels, refine_pts=rp, scores=sc, not_refine=nr, merge_valid=wgt > 0))
    return res


# ------------------------------------------------------------------------------------------------
# assigners  (integer outputs: bit-exact bar)
# ------------------------------------------------------------------------------------------------
def point_assign(points, gt_bboxes, gt_labels=None, scale=4, pos_num=3):
    """PointAssigner.assign (T/mmdet/core/bbox/assigners/point_assigner.py:23-133).
    points (n,3)=(x,y,stride); returns gt_inds (n,) int64 (0 bg, j+1 = gt j), labels or None."""
    n, k = points.shape[0], gt_bboxes.shape[0]
    if k == 0 or n == 0:
        return (torch.zeros(n, dtype=torch.long),
                None if gt_labels is None else torch.full((n,), -1, dtype=torch.long))
    xy, lvl = points[:, :2], torch.log2(points[:, 2]).int()
    lmin, lmax = lvl.min(), lvl.max()
    gxy = (gt_bboxes[:, :2] + gt_bboxes[:, 2:]) / 2
    gwh = (gt_bboxes[:, 2:] - gt_bboxes[:, :2]).clamp(min=1e-6)
    glvl = ((torch.log2(gwh[:, 0] / scale) + torch.log2(gwh[:, 1] / scale)) / 2).int()
    glvl = torch.clamp(glvl, min=lmin, max=lmax)
    inds = torch.zeros(n, dtype=torch.long)
    best = torch.full((n,), float('inf'))
    rng = torch.arange(n)
    for j in range(k):
        sel = glvl[j] == lvl
        pidx = rng[sel]
        d = ((xy[sel] - gxy[[j]]) / gwh[[j]]).norm(dim=1)
        md, mi = torch.topk(d, pos_num, largest=False)
        mp = pidx[mi]
        closer = md < best[mp]
        mp = mp[closer]
        inds[mp] = j + 1
        best[mp] = md[closer]
    lab = None
    if gt_labels is not None:
        lab = torch.full((n,), -1, dtype=torch.long)
        pos = inds > 0
        lab[pos] = gt_labels[inds[pos] - 1]
    return inds, lab


def _log_cr(x):
    """Correctly rounded fp32 log.  torch's CPU ``log`` goes through MKL VML, whose last bit depends on the HOST (Xeon /
    AVX-512 kernel: 0.007 % of values 1 ulp off this; EPYC kernel: 1.7 %, measured with tools/diag/log_probe.py,
    profiles/round2_log_probe.txt), so the reference's cost bits are not machine independent.  This is the
    host-independent limit both approximate and what the HIP cost kernel computes ((float)log((double)x), checked against
    glibc on 2^22 inputs: tools/diag/logcr.hip).  Evaluated in x87 extended precision (glibc logl, 64-bit mantissa) and
    rounded ONCE to fp32 -- torch's own float64 log is MKL as well and not accurate enough on every host to be the judge."""
    v = np.log(x.detach().numpy().astype(np.longdouble)).astype(np.float32)
    return torch.from_numpy(v)


def _sigmoid_vector_path(x):
    """torch.sigmoid as ATen's VECTOR loop computes it (1 / (1 + Sleef_expf_u10(-x))) for EVERY element.  On a contiguous
    tensor ATen runs the last numel mod (2 x vector width) elements through the scalar lambda instead (glibc expf), whose
    last bit can differ: the same logit gives 0x1.cd0e34p-5 as element 9987 of 10000 and 0x1.cd0e30p-5 inside a longer
    tensor.  Padding to a multiple of 64 keeps every real element in the vector body.  (P2P score maps hold
    H/4 * W/4 * C elements with H, W multiples of 32, i.e. a multiple of 64: the reference has no tail there.)"""
    flat = x.reshape(-1)
    pad = (-flat.numel()) % 64
    return torch.cat([flat, flat.new_zeros(pad + 64)]).sigmoid()[:flat.numel()].reshape(x.shape)


def focal_loss_cost(cls_pred, gt_labels, weight=1.0, alpha=0.25, gamma=2, eps=1e-12, log_mode='host'):
    """FocalLossCost.__call__ (T/mmdet/core/bbox/match_costs/match_cost.py:84-100).
    log_mode='host': torch's CPU ops as the reference executes them on THIS machine;
    'cr': the host- and shape-independent statement of the same formula (correctly rounded log, vector-path sigmoid)."""
    log = _log_cr if log_mode == 'cr' else torch.log
    p = _sigmoid_vector_path(cls_pred) if log_mode == 'cr' else cls_pred.sigmoid()
    neg = -log(1 - p + eps) * (1 - alpha) * p.pow(gamma)
    pos = -log(p + eps) * alpha * (1 - p).pow(gamma)
    return (pos[:, gt_labels] - neg[:, gt_labels]) * weight


def dis_cost_v2(pred, gt, img_shape, weight=1.0, norm_with_img_wh=True, p=1):
    """DisCostV2.__call__ (match_cost.py:197-214)."""
    factor = 1.0
    if norm_with_img_wh:
        k = pred.shape[-1] // 2
        h, w = img_shape[:2]
        factor = gt.new_tensor([w, h] * k).unsqueeze(0)
    return torch.cdist(pred / factor, gt / factor, p=p) * weight


def lsa_topk(cost, topk_k):
    """The topk_k>1 branch of HungarianAssignerV2.assign (hungarian_assigner.py:244-268): up to topk_k rounds
    of scipy linear_sum_assignment, assigned rows removed each round.  topk_k == 1 is one plain LSA (229-243).
    cost (M,G) fp32 (scipy promotes to float64).  Returns gt_inds (M,) int64."""
    from scipy.optimize import linear_sum_assignment
    M, G = cost.shape
    inds = torch.zeros(M, dtype=torch.long)
    if topk_k == 1:
        r, c = linear_sum_assignment(cost)
        inds[torch.from_numpy(r)] = torch.from_numpy(c) + 1
        return inds
    index = torch.arange(M)
    cur = cost
    num = 0
    while cur.shape[0] // G != 0 and num + 1 <= topk_k:
        num += 1
        r, c = linear_sum_assignment(cur)
        rows = index[torch.from_numpy(r)]
        inds[rows] = torch.from_numpy(c) + 1
        index = torch.nonzero(inds == 0).squeeze(1)
        cur = cost[inds == 0]
    return inds


def hungarian_assign_v2(pred_pts, cls_pred, gt_pts, gt_labels, img_shape, topk_k=5,
                        cls_weight=2.0, dis_weight=0.1, dis_norm=False, log_mode='host', p=1):
    """HungarianAssignerV2.assign (hungarian_assigner.py:166-270) with the P2P config's costs
    (T/configs2/TinyPersonV2/p2p/p2p_r50_fpns4_1x_fl_sl1_TinyPersonV2_640.py:55-63):
    FocalLossCost(weight=2.0) + DisCostV2(weight=0.1, p=1)."""
    M, G = pred_pts.shape[0], gt_pts.shape[0]
    labels = torch.full((M,), -1, dtype=torch.long)
    if G == 0 or M == 0:
        return torch.full((M,), 0 if G == 0 else -1, dtype=torch.long), labels, None
    cost = focal_loss_cost(cls_pred, gt_labels, cls_weight, log_mode=log_mode) + dis_cost_v2(
        pred_pts, gt_pts, img_shape, dis_weight, dis_norm, p)
    inds = lsa_topk(cost, topk_k)
    pos = inds > 0
    labels[pos] = gt_labels[inds[pos] - 1]
    return inds, labels, cost


# ------------------------------------------------------------------------------------------------
# P2P head  (p2p_head.py) and pseudo-box NMS  (bbox_nms.py + mmcv batched_nms, third-party)
# ------------------------------------------------------------------------------------------------
def p2p_head_forward(sd, feats, stacked_convs=4, prefix='bbox_head.'):
    """P2PHead.forward_single (p2p_head.py:113-123)."""
    cls_outs, pts_outs = [], []
    for x in feats:
        c = r = x
        for i in range(stacked_convs):
            c = _conv_gn(c, sd, '%scls_convs.%d' % (prefix, i), 1, True)
            r = _conv_gn(r, sd, '%sreg_convs.%d' % (prefix, i), 1, True)
        cls_outs.append(F.conv2d(c, sd[prefix + 'cls_out.weight'], sd[prefix + 'cls_out.bias'], 1, 1))
        pts_outs.append(F.conv2d(r, sd[prefix + 'reg_out.weight'], sd[prefix + 'reg_out.bias'], 1, 1))
    return cls_outs, pts_outs


def p2p_loss(cls_outs, pts_outs, gt_bboxes, gt_labels, img_shape, stride=4, topk_k=5, alpha=0.25, gamma=2.0,
             beta=1.0 / 9.0, w_cls=1.0, w_reg=0.5, pts_gamma=1.0, reg_norm=1.0, point_anchor=((0., 0.),), pos_weight=1.0,
             neg_weight=1.0, log_mode='host'):
    """P2PHead.loss with the shipped config (p2p_head.py:172-248; T/configs2/TinyPersonV2/p2p/p2p_r50_fpns4_1x_fl_sl1_TinyPersonV2_640.py):
    get_pred_points (:125-170, pred = anchor + point_anchor * stride + reg * gamma * stride), per image HungarianAssignerV2 ->
    PseudoSampler -> sample_result_to_target (:308-328), then per image FocalLoss (py_sigmoid_focal_loss, losses/focal_loss.py:11-56,
    label weights, avg_factor = positives of the whole batch) and SmoothL1Loss (losses/smooth_l1_loss.py:11-28) on
    pred / stride / reg_norm.  cls_outs / pts_outs: (B, k*C, H, W) / (B, 2k, H, W) as the head returns them.
    -> {'loss_cls': [B scalars], 'loss_pts': [B scalars]}, gt_inds list."""
    import torch.nn.functional as F
    B, kc, H, W = cls_outs.shape
    k = len(point_anchor)
    C = kc // k
    anchor = p2p_grid_points(H, W, stride)[:, None, :2] + torch.tensor(point_anchor, dtype=torch.float32)[None] * stride    # (HW, k, 2)
    reg = pts_outs.permute(0, 2, 3, 1).reshape(B, H * W, k, 2)
    pred = (anchor[None] + reg * pts_gamma * stride).reshape(B, H * W * k, 2)
    cls = cls_outs.permute(0, 2, 3, 1).reshape(B, H * W * k, C)
    labels, lw, tgt, pw, inds_all = [], [], [], [], []
    for b in range(B):
        ctr = (gt_bboxes[b][:, :2] + gt_bboxes[b][:, 2:]) / 2
        # the assignment carries no gradient (the reference builds its cost from detached tensors: hungarian_assigner.py:214-236)
        inds, _, _ = hungarian_assign_v2(pred[b].detach(), cls[b].detach(), ctr, gt_labels[b], img_shape, topk_k=topk_k,
                                         log_mode=log_mode)
        pos = inds > 0
        lab = torch.full((pred.shape[1],), C, dtype=torch.long)
        lab[pos] = gt_labels[b][inds[pos] - 1]
        w = torch.full((pred.shape[1],), 1.0 if neg_weight <= 0 else neg_weight)
        w[pos] = pos_weight
        t = torch.zeros((pred.shape[1], 2))
        t[pos] = ctr[inds[pos] - 1]
        ww = torch.zeros((pred.shape[1], 2))
        ww[pos] = 1.0
        labels.append(lab), lw.append(w), tgt.append(t), pw.append(ww), inds_all.append(inds)
    num_pos = sum(int((w[:, 0] > 0).sum()) for w in pw)
    loss_cls, loss_pts = [], []
    for b in range(B):
        target = F.one_hot(labels[b], num_classes=C + 1)[:, :C].type_as(cls)
        ps = cls[b].sigmoid()
        pt = (1 - ps) * target + ps * (1 - target)
        fw = (alpha * target + (1 - alpha) * (1 - target)) * pt.pow(gamma)
        l = F.binary_cross_entropy_with_logits(cls[b], target, reduction='none') * fw * lw[b].view(-1, 1)
        loss_cls.append(w_cls * l.sum() / num_pos)
        diff = torch.abs(pred[b] / stride / reg_norm - tgt[b] / stride / reg_norm)
        sl = torch.where(diff < beta, 0.5 * diff * diff / beta, diff - 0.5 * beta) * pw[b]
        loss_pts.append(w_reg * sl.sum() / num_pos)
    return {'loss_cls': loss_cls, 'loss_pts': loss_pts}, inds_all


def p2p_grid_points(h, w, stride):
    """PointGenerator.grid_points (T/mmdet/core/anchor/point_generator.py:17-25): (x*s, y*s, s), no half-stride."""
    sx = torch.arange(0., w) * stride
    sy = torch.arange(0., h) * stride
    xx = sx.repeat(h)
    yy = sy.view(-1, 1).repeat(1, w).view(-1)
    return torch.stack([xx, yy, torch.full_like(xx, stride)], dim=-1)


def batched_nms(boxes, scores, idxs, iou_thr):
    """Restatement of mmcv-full 1.3.x ``mmcv.ops.nms.batched_nms`` (third-party, pin >=1.3.2,<=1.4.0,
    T/mmdet/__init__.py:18-26; call site T/mmdet/core/post_processing/bbox_nms.py:85).  PARITY UNPINNED:
    the reference tree holds no golden vector for NMS; semantics restated from the published algorithm:
    boxes_for_nms = boxes + idxs*(boxes.max()+1); sort by score desc; greedy suppress IoU > thr (offset 0).
    Returns keep indices (into the inputs) in descending-score order."""
    if boxes.numel() == 0:
        return torch.zeros(0, dtype=torch.long)
    b = boxes + (idxs.to(boxes) * (boxes.max() + 1))[:, None]
    order = torch.argsort(scores, descending=True, stable=True)
    b = b[order]
    n = len(b)
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    supp = np.zeros(n, dtype=bool)
    bn, an = b.numpy(), area.numpy()
    keep = []
    for i in range(n):
        if supp[i]:
            continue
        keep.append(i)
        xx1 = np.maximum(bn[i, 0], bn[i + 1:, 0])
        yy1 = np.maximum(bn[i, 1], bn[i + 1:, 1])
        xx2 = np.minimum(bn[i, 2], bn[i + 1:, 2])
        yy2 = np.minimum(bn[i, 3], bn[i + 1:, 3])
        inter = np.clip(xx2 - xx1, 0, None) * np.clip(yy2 - yy1, 0, None)
        iou = inter / (an[i] + an[i + 1:] - inter)
        supp[i + 1:] |= iou > np.float32(iou_thr)
    return order[torch.tensor(keep, dtype=torch.long)]


def multiclass_nms(multi_bboxes, multi_scores, score_thr, iou_thr, max_num=-1):
    """multiclass_nms (T/mmdet/core/post_processing/bbox_nms.py:7-94), boxes shared across classes (n,4),
    multi_scores (n, C+1) with the last column = background.  Returns dets (k,5), labels (k,), keep-inds."""
    C = multi_scores.size(1) - 1
    bboxes = multi_bboxes[:, None].expand(multi_scores.size(0), C, 4).reshape(-1, 4)
    scores = multi_scores[:, :-1].reshape(-1)
    labels = torch.arange(C, dtype=torch.long).view(1, -1).expand(multi_scores.size(0), C).reshape(-1)
    valid = scores > score_thr
    inds = valid.nonzero(as_tuple=False).squeeze(1)
    bboxes, scores, labels = bboxes[inds], scores[inds], labels[inds]
    if inds.numel() == 0:
        return torch.zeros((0, 5)), labels, inds
    keep = batched_nms(bboxes, scores, labels, iou_thr)
    if max_num > 0:
        keep = keep[:max_num]
    return torch.cat([bboxes[keep], scores[keep, None]], -1), labels[keep], inds[keep]


def p2p_get_points_single(cls_scores, pts_preds, points, img_shape, nms_pre=2000, score_thr=0.05, iou_thr=0.2,
                          max_per_img=1000, pseudo_wh=(16, 16)):
    """P2PHead._get_bboxes_single (p2p_head.py:345-405), use_sigmoid_cls=True, single level given as
    cls_scores (M,C) logits, pts_preds (M,2), points (M,3).  Returns dets (k,3)=(cx,cy,score), labels, topk idx."""
    scores = cls_scores.sigmoid()
    topk_inds = None
    if 0 < nms_pre < scores.shape[0]:
        max_scores, _ = scores.max(dim=1)
        _, topk_inds = max_scores.topk(nms_pre)
        pts_preds, scores = pts_preds[topk_inds], scores[topk_inds]
    x = pts_preds[:, 0].clamp(min=0, max=img_shape[1])
    y = pts_preds[:, 1].clamp(min=0, max=img_shape[0])
    ctr = torch.stack([x, y], dim=-1)
    wh = ctr.new_tensor(pseudo_wh)
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], dim=-1)
    scores = torch.cat([scores, scores.new_zeros(scores.shape[0], 1)], dim=1)
    dets, labels, keep = multiclass_nms(boxes, scores, score_thr, iou_thr, max_per_img)
    out = torch.stack([(dets[:, 0] + dets[:, 2]) / 2, (dets[:, 1] + dets[:, 3]) / 2, dets[:, 4]], dim=-1) \
        if len(dets) else torch.zeros((0, 3))
    return out, labels, topk_inds, keep


# ------------------------------------------------------------------------------------------------
# whole step (the benchmarked unit): backbone -> neck -> head -> loss
# ------------------------------------------------------------------------------------------------
def locator_forward_train(sd, batch, depth=50, start_level=0, stride=4, radius=5, num_classes=1, **loss_kw):
    """BasicLocator.forward_train (T/mmdet/models/point/detectors/locator.py:20-32)."""
    feats = fpn_forward(sd, resnet_forward(sd, batch['img'], depth), start_level, 1)
    cls_feat, _ = cpr_head_forward(sd, feats)
    losses, per = cpr_loss(sd, cls_feat[0], batch['gt_bboxes'], batch['gt_labels'], batch['img_metas'], stride,
                           radius, num_classes, **loss_kw)
    return losses, cls_feat[0], per
