"""Build container, CPU: would fp32 Winograd F(4x4,3x3) keep the parity bars on the five 256 -> 256 3x3 layers (FPN output conv + the four
CPRHead tower convs)?  VERDICT round 5, task 4, step 1: emulate the transform arithmetic in fp32 (torch CPU), run the oracle's forward with
those five layers replaced, and compare the head logits with an fp64 run of the same network -- next to the direct fp32 convolution and the
F(2x2,3x3) form the product runs today.  Stop rule: head-logit error above 3e-5 (a third of the 1e-4 bar), or any refine selection moving.
  python tools/wino_f4_emulation.py            -> table on stdout (profiles/round6_wino_f4_emulation.txt)"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cpr_oracle as O  # noqa: E402
from oracle.gen_golden import CPR_CASES  # noqa: E402
from pointtinybenchmark_amd import synthetic  # noqa: E402

# F(4x4,3x3) with the interpolation points 0, +-1, +-2 (Lavin & Gray), F(2x2,3x3) with 0, +-1
BT4 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)
AT4 = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)
BT2 = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G2 = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT2 = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def wino_conv(x, w, m):
    """3x3 / stride 1 / pad 1 convolution as Winograd F(m x m, 3x3); every step in x.dtype (fp32: one rounding per operation of the
    transforms, fp32 accumulation over the channels -- what a device kernel would do, up to summation order)."""
    BT, G, AT = (BT4, G4, AT4) if m == 4 else (BT2, G2, AT2)
    BT, G, AT = BT.to(x.dtype), G.to(x.dtype), AT.to(x.dtype)
    B, C, H, W = x.shape
    assert H % m == 0 and W % m == 0
    t = m + 2
    xp = F.pad(x, (1, 1, 1, 1))
    d = xp.unfold(2, t, m).unfold(3, t, m)                                   # (B, C, th, tw, t, t)
    th, tw = d.shape[2], d.shape[3]
    V = torch.einsum('ij,bcyxjk,lk->bcyxil', BT, d, BT)                      # B^T d B
    U = torch.einsum('ij,ocjk,lk->ocil', G, w, G)                            # G g G^T
    M = torch.einsum('ocil,bcyxil->boyxil', U, V)                            # 36 (16) channel GEMMs
    Y = torch.einsum('ij,boyxjk,lk->boyxil', AT, M, AT)                      # A^T M A  -> (B, O, th, tw, m, m)
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(B, w.shape[0], th * m, tw * m)


FIVE = ['neck.fpn_convs.0'] + ['bbox_head.cls_convs.%d' % i for i in range(4)]


def tower(sd, batch, cfg, mode, dtype):
    """Oracle backbone + FPN + head tower with the five 3x3 256 -> 256 layers computed by ``mode`` in ``dtype`` -> cls_feat (B, 256, h, w)."""
    sdt = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
    orig = O._conv_gn

    def conv_gn(x, sd_, p, padding, relu, groups=32, eps=1e-5):
        if mode != 'direct' and p in FIVE:
            y = wino_conv(x, sd_[p + '.conv.weight'], 4 if mode == 'f4' else 2)
            y = F.group_norm(y, groups, sd_[p + '.gn.weight'], sd_[p + '.gn.bias'], eps)
            return F.relu(y) if relu else y
        return orig(x, sd_, p, padding, relu, groups, eps)
    O._conv_gn = conv_gn
    try:
        feats = O.fpn_forward(sdt, O.resnet_forward(sdt, batch['img'].to(dtype), cfg['depth']), cfg['start_level'], 1)
        cls_feat, _ = O.cpr_head_forward(sdt, feats)
    finally:
        O._conv_gn = orig
    return cls_feat[0]


def logits(sd, feat, geo, stride):
    """cls / ins logits of the bag points and the negative-grid logits from a tower output, in feat.dtype (geometry from the fp32 oracle run)."""
    dt = feat.dtype
    Wc, bc, Wi, bi = [sd['bbox_head.' + k].to(dt) for k in ('cls_out.weight', 'cls_out.bias', 'ins_out.weight', 'ins_out.bias')]
    out = []
    for b, g in enumerate(geo):
        bag = O.sample_bilinear(feat[b:b + 1], g['pts'].to(dt) / stride)
        grid = feat[b].permute(1, 2, 0).flatten(0, 1)
        out.append(torch.cat([F.linear(bag, Wc, bc).flatten(), F.linear(bag, Wi, bi).flatten(), F.linear(grid, Wc, bc).flatten()]))
    return out


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    cases = [('cpr_r50_c1_160', CPR_CASES['cpr_r50_c1_160']), ('cpr_r50_c1_160_spread', CPR_CASES['cpr_r50_c1_160_spread']),
             ('gate sample R50 640^2 B=1', dict(depth=50, num_classes=1, start_level=0, stride=4, radius=5, head_std=0.01, seed=0, batch=1,
                                                height=640, width=640, num_gts=32))]
    print('head-logit error against an fp64 run of the same network (max over cls / ins / negative-grid logits), tower-feature error relative to')
    print('its maximum, and how many PointRefiner outputs (chosen points, not_refine flags) differ from the direct fp32 convolution\'s;')
    print('bar for F(4x4): logit error <= 3e-5 (a third of the 1e-4 parity bar) and no selection moved')
    for name, cfg in cases:
        sd = synthetic.locator_state_dict(cfg['depth'], cfg['num_classes'], cfg['start_level'], 'cpr', cfg['seed'], cfg['head_std'])
        batch = synthetic.synthetic_batch(cfg['batch'], cfg['height'], cfg['width'], cfg['num_gts'], cfg['num_classes'], cfg['seed'], cfg.get('ragged', False))
        args = (batch['gt_bboxes'], batch['gt_labels'], batch['img_metas'], cfg['stride'], cfg['radius'], cfg['num_classes'])
        with torch.no_grad():
            f64 = tower(sd, batch, cfg, 'direct', torch.float64)
            base = None
            for mode in ('direct', 'f2', 'f4'):
                f32 = tower(sd, batch, cfg, mode, torch.float32)
                geo = O.cpr_points_and_logits(sd, f32, *args)
                l32, l64 = logits(sd, f32, geo, cfg['stride']), logits(sd, f64, geo, cfg['stride'])
                lerr = max(float((a.double() - b).abs().max()) for a, b in zip(l32, l64))
                ferr = float((f32.double() - f64).abs().max() / f64.abs().max())
                ref = O.cpr_refine(sd, f32, batch['gt_bboxes'], batch['gt_labels'], batch['gt_anns_id'], batch['img_metas'], cfg['stride'], cfg['radius'],
                                   cfg['num_classes'])
                sel = [(r['merge_valid'], r['not_refine']) for r in ref]
                if base is None:
                    base = sel
                moved = sum(int((a[0] != b[0]).sum()) + int((a[1] != b[1]).sum()) for a, b in zip(sel, base))
                print('%-28s %-7s max logit error %.3e   tower feature error / max %.3e   refine selections that differ from direct fp32: %d'
                      % (name, {'direct': 'direct', 'f2': 'F(2x2)', 'f4': 'F(4x4)'}[mode], lerr, ferr, moved), flush=True)


if __name__ == '__main__':
    main()
