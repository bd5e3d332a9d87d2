"""CPU oracle for the CPRHead options no shipped config uses.  TEST INFRASTRUCTURE ONLY.

A functional torch-CPU restatement of CPRHead.loss / loss0 / get_bboxes + PointRefiner for
  * num_refine > 1 inputs: (num_gts * R, 4) pseudo boxes per image, the three ``refine_bag_policy`` values and both
    ``gt_loss_type`` values                                   T/mmdet/models/point/dense_heads/cpr_head.py:1101-1217
  * GridCirclesPtFeatGenerator bags                            cpr_head.py:296-352,405-438
  * ``softmax`` / ``normed_sigmoid`` class probabilities        cpr_head.py:1080-1099
  * MILLoss(binary_ins=True), AllPosLoss                       T/mmdet/models/losses/multi_instance_learning_loss.py:153-243
  * (round 3) ins_share_head_feat=False: a second tower ``ins_convs`` / ``ins_fcs`` feeds the instance classifier
                                                               cpr_head.py:992-1008,1030-1043,1055-1072
  * (round 3) out_bg_cls=True for one class (a background output beside the class, never a label)   cpr_head.py:953
  * (round 3) PointRefiner(return_score_type='max')            cpr_head.py:840-842
  * (round 3) generator align_corners=True (grid 2x/(w-1)-1, zeros padding)   cpr_head.py:73-93,126
pinned by tests/golden/cpr_options.npz / cpr_options_r3.npz (the reference's own classes, oracle/gen_golden_r2.py) in
tests/test_oracle_golden.py.  ``cfg`` is an oracle.gen_golden_r2 case dict."""
import torch
import torch.nn.functional as F

from oracle import cpr_oracle as O


def cls_prob(logit, cfg):
    """get_cls_prob (cpr_head.py:1080-1099) on (..., C) class logits."""
    t = cfg.get('prob', 'sigmoid')
    if t == 'softmax':
        return logit.softmax(dim=-1)
    p = logit.sigmoid()
    if t == 'normed_sigmoid':
        p = F.normalize(p, p=cfg.get('norm_p', 1), dim=-1)
    return p


def grid_circle_bag(feat, centers, radius, stride, max_pos_num, align_corners=False):
    """GridPtFeatGenerator.generate + GridCirclesPtFeatGenerator.get_chosen_neighbours for one image.
    feat (1,C,H,W), centers (G,R,2) -> pts (G, Kmax+2R... , 2), valid, sampled feats (G, ., C)."""
    h, w = feat.shape[2:]
    G, R, _ = centers.shape
    pts = O.grid_points(h, w, stride).float()                                            # (H,W,2)
    dis = torch.norm(pts.reshape(1, h, w, 1, 2) - centers.reshape(G, 1, 1, R, 2), p=2, dim=-1)
    chosens = torch.any(dis <= radius * stride, dim=-1)                                  # (G,H,W)
    kmax = (max_pos_num if max_pos_num > 0 else 2 * (2 * radius) ** 2) + R
    fmap = feat.permute(0, 2, 3, 1)[0]
    out_pts = torch.zeros(G, kmax, 2)
    out_feat = torch.zeros(G, kmax, fmap.shape[-1])
    valid = torch.ones(G, kmax, dtype=torch.bool)
    for i in range(G):
        sel = pts[chosens[i]]
        out_pts[i, :len(sel)] = sel
        out_feat[i, :len(sel)] = fmap[chosens[i]]
        valid[i, len(sel):] = False
    cfeat = O.sample_bilinear(feat, centers / stride, align_corners)                     # (G,R,C)
    out_pts = torch.cat([out_pts, centers.flip(dims=(1,))], dim=1)
    out_feat = torch.cat([out_feat, cfeat.flip(dims=(1,))], dim=1)
    valid = torch.cat([valid, torch.ones(G, R, dtype=torch.bool)], dim=1)
    return out_pts, valid, out_feat


def ins_tower_forward(sd, feats, stacked_convs=4, prefix='bbox_head.'):
    """forward_single's second tower (ins_share_head_feat=False, cpr_head.py:1037-1040)."""
    out = []
    for x in feats:
        for i in range(stacked_convs):
            x = O._conv_gn(x, sd, '%sins_convs.%d' % (prefix, i), 1, True)
        out.append(x)
    return out


def _fc_stack(sd, x, name, prefix='bbox_head.'):
    """get_pts_outs.forward_with_fc (cpr_head.py:1055-1059) with the layers ``name``.{i} found in the state dict."""
    i = 0
    while prefix + '%s.%d.weight' % (name, i) in sd:
        x = F.relu(F.linear(x, sd[prefix + '%s.%d.weight' % (name, i)], sd[prefix + '%s.%d.bias' % (name, i)]))
        i += 1
    return x


def extract(sd, cls_feat, gt_bboxes, gt_labels, img_metas, cfg, prefix='bbox_head.', ins_feat=None):
    """PointExtractor + get_pts_outs for every image.  Returns per-image dicts with pts (G,Rv,Kv,2), valid (G,Rv,Kv),
    cls_logit / ins_logit (G,Rv,Kv,.), the negative mask and grid logits, centers (G,R,2).
    ins_feat: the instance tower's map when cfg['ins_tower'] (the same points are sampled from it)."""
    stride, radius, C = cfg['stride'], cfg['radius'], cfg['num_classes']
    Wc, bc = sd[prefix + 'cls_out.weight'], sd[prefix + 'cls_out.bias']
    Wi, bi = sd[prefix + 'ins_out.weight'], sd[prefix + 'ins_out.bias']
    assert (ins_feat is not None) == bool(cfg.get('ins_tower', False))
    out = []
    for b in range(len(gt_bboxes)):
        G = len(gt_labels[b])
        ctr = ((gt_bboxes[b][:, :2] + gt_bboxes[b][:, 2:]) / 2).reshape(G, -1, 2)        # (G,R,2)
        R = ctr.shape[1]
        ph, pw = img_metas[b]['pad_shape'][:2]
        feat = cls_feat[b:b + 1]
        if cfg.get('pos') == 'GridCirclesPtFeatGenerator':
            pts, valid, bag_feat = grid_circle_bag(feat, ctr, radius, stride, cfg.get('max_pos_num', -1),
                                                   cfg.get('align_corners', False))
            pts, valid, bag_feat = pts[:, None], valid[:, None], bag_feat[:, None]       # (G,1,K,.)
        else:
            pts = O.bag_points(ctr.reshape(-1, 2), radius, stride).reshape(G, R, -1, 2)
            valid = O.inside(pts, ph, pw)
            bag_feat = O.sample_bilinear(feat, pts.reshape(G * R, -1, 2) / stride,
                                         cfg.get('align_corners', False)).reshape(G, R, pts.shape[2], -1)
        ins_bag = bag_feat
        if ins_feat is not None:
            assert cfg.get('pos', 'CirclePtFeatGenerator') == 'CirclePtFeatGenerator'
            ins_bag = O.sample_bilinear(ins_feat[b:b + 1], pts.reshape(G * R, -1, 2) / stride,
                                        cfg.get('align_corners', False)).reshape(bag_feat.shape)
            ins_bag = _fc_stack(sd, ins_bag, 'ins_fcs', prefix)
        h, w = feat.shape[2:]
        # negative mask: every refine point counts, carrying its gt's label (cpr_head.py:271-275: centers.flatten(0, 1))
        npts, nvalid = O.neg_valid_mask(h, w, stride, radius, ctr.reshape(-1, 2), gt_labels[b].repeat_interleave(R), C, ph, pw)
        nfeat = _fc_stack(sd, feat.permute(0, 2, 3, 1)[0].flatten(0, 1), 'cls_fcs', prefix)
        bag_feat = _fc_stack(sd, bag_feat, 'cls_fcs', prefix)
        if ins_feat is None:
            ins_bag = bag_feat                      # ins_share_head_feat: the classifier inputs are shared (:1066)
        out.append(dict(centers=ctr, pts=pts, valid=valid, cls_logit=F.linear(bag_feat, Wc, bc),
                        ins_logit=F.linear(ins_bag, Wi, bi), neg_valid=nvalid, neg_logit=F.linear(nfeat, Wc, bc)))
    return out


def cpr_loss(sd, cls_feat, gt_bboxes, gt_labels, img_metas, cfg, mil_weight=0.25, neg_weight=0.75, gt_weight=0.25,
             ins_feat=None):
    """CPRHead.loss -> loss0 (cpr_head.py:1101-1229) with the options of ``cfg``."""
    # out_bg_cls: num_cls_out = C + 1 outputs; labels stay < C, the (.., C) validity masks broadcast over them (C = 1 only)
    C = cfg['num_classes'] + (1 if cfg.get('out_bg_cls', False) else 0)
    per = extract(sd, cls_feat, gt_bboxes, gt_labels, img_metas, cfg, ins_feat=ins_feat)
    labels = torch.cat(gt_labels)
    cls = torch.cat([p['cls_logit'] for p in per])                                       # (G,R,K,C)
    ins = torch.cat([p['ins_logit'] for p in per])
    valid = torch.cat([p['valid'] for p in per]).unsqueeze(-1).float()                   # (G,R,K,1)
    G, R, K, _ = cls.shape
    w = torch.ones(G)
    losses, num_pos = {}, None
    # gt loss (:1159-1184)
    gt_prob = cls_prob(cls[..., -1, :].reshape(G * R, -1), cfg)
    if cfg.get('gt_loss_type', 'gt_refine') == 'gt_refine':
        lab = labels.repeat_interleave(R)
        gw = valid[..., -1, :].reshape(G * R, -1) * w.repeat_interleave(R).reshape(-1, 1)
    else:
        lab = labels
        gw = valid[:, 0, -1, :] * w.reshape(-1, 1)
        gt_prob = gt_prob.reshape(G, R, -1)[:, 0]
    num_pos = max(float((gw > 0).sum()), 1.0)
    losses['gt_loss'] = gt_weight * (O.gfocal(gt_prob, F.one_hot(lab, C).float(), gw).sum() / num_pos)
    # MIL loss (:1186-1217)
    if cfg.get('with_mil_loss', True):
        policy = cfg.get('policy', 'independent_with_gt_bag')
        if policy == 'independent_with_gt_bag':
            bc_, bi_, bv_ = cls.reshape(G * R, K, -1), ins.reshape(G * R, K, -1), valid.reshape(G * R, K, 1)
            bl = labels.repeat_interleave(R)
        else:
            si = 1 if (policy == 'only_refine_bag' and R > 1) else 0
            bc_ = cls[:, si:].reshape(G, (R - si) * K, -1)
            bi_ = ins[:, si:].reshape(G, (R - si) * K, -1)
            bv_ = valid[:, si:].reshape(G, (R - si) * K, 1)
            bl = labels
        prob = cls_prob(bc_, cfg)
        if cfg.get('loss', 'MILLoss') == 'AllPosLoss':
            B, N, _ = prob.shape
            p2 = prob.reshape(B * N, C)
            l2 = bl.unsqueeze(-1).repeat(1, N).flatten()
            v2 = bv_.reshape(B * N, 1)
            num_pos = max(float((v2.sum(-1) > 0).sum()), 1.0)
            losses['pos_loss'] = O.gfocal(p2, F.one_hot(l2, C).float(), v2).sum() / num_pos * mil_weight
            losses['bag_acc'] = (p2.argmax(-1) == l2).float().mean() * 100
        else:
            B, N, _ = prob.shape
            pi = bi_.reshape(B, N, C, -1).softmax(dim=1) * bv_.unsqueeze(-1)
            pi = F.normalize(pi, dim=1, p=1)
            pb = (prob.unsqueeze(-1) * pi).sum(dim=1)                                    # (B,C,1|2)
            lw = (bv_.sum(dim=1) > 0).float()
            onehot = F.one_hot(bl, C).float()
            num_pos = max(float((lw.sum(-1) > 0).sum()), 1.0)
            loss = O.gfocal(pb[..., 0], onehot, lw).sum()
            if pb.shape[-1] == 2:                                                        # binary_ins (:179-184)
                loss = loss + O.gfocal(pb[..., 1], torch.zeros_like(onehot), lw).sum()
            losses['pos_loss'] = loss / num_pos * mil_weight
            losses['bag_acc'] = (pb[..., 0].argmax(-1) == bl).float().mean() * 100
    # negative loss (:1219-1228): averaged over the LAST num_pos computed above; loss_cfg with_neg=False drops the term (:1219)
    if cfg.get('with_neg', True):
        neg_prob = cls_prob(torch.cat([p['neg_logit'] for p in per]), cfg)
        neg_valid = torch.cat([p['neg_valid'] for p in per]).float()
        losses['neg_loss'] = neg_weight * (O.gfocal(neg_prob, torch.zeros_like(neg_prob), neg_valid).sum() / num_pos)
    return losses, per


def cpr_refine(sd, cls_feat, gt_bboxes, gt_labels, img_metas, cfg, gt_alpha=0.5, merge_th=0.1, refine_th=0.1,
               classify_filter=True, nearest_filter=True, not_refine=None, ins_feat=None):
    """PointRefiner.refine_single (cpr_head.py:780-850) on the bags of ``extract``.  Returns per image
    dict(refine_pts, scores, not_refine, chosen (G, Rv*Kv) bool, bag_pts)."""
    per = extract(sd, cls_feat, gt_bboxes, gt_labels, img_metas, cfg, ins_feat=ins_feat)
    res = []
    for b, p in enumerate(per):
        labels = gt_labels[b]
        G, Rv, Kv = p['valid'].shape
        assert Rv == 1 or cfg.get('pos') == 'GridCirclesPtFeatGenerator' or p['centers'].shape[1] == 1, \
            'the reference asserts on num_refine > 1 circle bags (cpr_head.py:809)'
        prob = cls_prob(p['cls_logit'], cfg)                                             # (G,Rv,Kv,C)
        pts = p['pts']
        mv = p['valid'].reshape(G, Rv * Kv).clone()
        gt_pts = pts[..., -1:, :]                                                        # (G,Rv,1,2)
        if nearest_filter:
            nv = torch.ones(G, Rv * Kv, dtype=torch.bool)
            for l in sorted(set(labels.tolist())):
                idx = torch.nonzero(labels == l).squeeze(1)
                if len(idx) > 1:
                    d = torch.cdist(pts[idx].flatten(0, -2), gt_pts[idx].flatten(0, -2), p=2)
                    closest = d.min(dim=1)[1].reshape(len(idx) * Rv, Kv)
                    cur = torch.arange(len(closest)).reshape(-1, 1)
                    nv[idx] = (closest == cur).reshape(len(idx), Rv * Kv)
            mv &= nv
        if classify_filter:
            mv &= (prob.max(dim=-1)[1] == labels.reshape(-1, 1, 1)).reshape(G, Rv * Kv)
        ar = torch.arange(G)
        pl = prob[ar, ..., labels].reshape(G, Rv * Kv)
        gp = prob[:, :, -1:, :][ar, 0, ..., labels].reshape(G, 1)
        mv &= (pl > merge_th) & (pl > gp * gt_alpha)
        ih, iw = img_metas[b]['img_shape'][:2]
        flat = pts.reshape(G, Rv * Kv, 2)
        x, y = flat[..., 0], flat[..., 1]
        mv &= (x < iw) & (x >= 0) & (y < ih) & (y >= 0)
        pm = pl * mv.float()
        wgt = pm / (pm.sum(dim=1, keepdim=True) + 1e-8)
        rp = (flat * wgt.unsqueeze(-1)).sum(dim=1)
        sc = pm.sum(dim=-1) / ((pm > 0).float().sum(dim=-1) + 1e-8)
        nr = sc < refine_th
        if not_refine is not None:
            nr = not_refine[b] | nr
        rp[nr] = p['centers'][:, 0][nr]
        if cfg.get('score_type', 'mean') == 'max':                                       # cpr_head.py:840-842
            sc = pm.max(dim=-1)[0]
            sc[sc == 0] = refine_th / 2
        res.append(dict(refine_pts=rp, scores=sc, not_refine=nr, chosen=wgt > 0, bag_pts=flat))
    return res
