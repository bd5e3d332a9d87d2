"""Parameter containers + the fused conv/norm building blocks used by backbone, neck and heads.

nn.Conv2d / nn.BatchNorm2d / nn.GroupNorm objects are used ONLY as parameter holders so the state-dict
keys match the reference checkpoint layout; their ``forward`` is never called -- all
compute goes through ``ops`` (HIP kernels)."""
import os

import torch
import torch.nn as nn

from . import ops


_WEIGHT_EPOCH = [0]
_STALE_PACKS = __import__('os').environ.get('CPR_EXPERIMENT_STALE_PACKS', '0') == '1'


def bump_weight_epoch():
    """Called by the native optimizer: its kernels update parameters through raw pointers, which torch's version counter
    does not see."""
    if _STALE_PACKS:       # measurement only (CPR_EXPERIMENT_STALE_PACKS=1: WRONG results): what the per-step re-packs and re-folds cost
        return
    _WEIGHT_EPOCH[0] += 1


class _PackCache:
    """Repacked weights / folded norms, rebuilt when the source parameter is modified in place."""

    def __init__(self):
        self._d = {}
        self._jobs = {}          # key -> (tensors, job): entries the native trainer refreshes IN PLACE after its optimizer step
        self._tables = None      # device job tables of refresh_all (rebuilt when a job is registered or a pointer moved)

    @staticmethod
    def _ver(tensors):
        ver = tuple((t.data_ptr(), t._version) for t in tensors)
        if any(t.requires_grad for t in tensors):
            ver = ver + (_WEIGHT_EPOCH[0],)
        return ver

    def get(self, key, tensors, make, refresh=None):
        """refresh (optional): val -> job, how the native trainer re-computes ``val`` IN PLACE after an optimizer step instead of
        letting the entry lapse (refresh_all): ('fold', bn, scale | None, shift | None, inv | None) or
        ('pack' | 'pack32', weight, key of the fold entry whose scale is multiplied in | None, packed_conv, transpose) (bf16 / fp32 pack)."""
        ver = self._ver(tensors)
        hit = self._d.get(key)
        if hit is not None and hit[0] == ver:
            self._order_behind(hit)
            return hit[1]
        val = make()
        if refresh is not None and REFRESH_IN_PLACE[0]:
            job = refresh(val)
            if job is not None:
                self._jobs[key] = (list(tensors), job)
                self._tables = None
        elif key in self._jobs:
            del self._jobs[key]
            self._tables = None
        # whatever make() enqueued (pack kernels, torch ops building a bf16 pack / a folded norm / a bias vector) ran on the
        # CURRENT stream: a reader on another stream (sub-batches of CPR_STREAMS > 1, the trainer's side stream) must order
        # itself behind it -- the event lives with the entry until it has completed
        ev = None
        if torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
            ev = torch.cuda.Event()
            ev.record()
        self._d[key] = [ver, val, ev, torch.cuda.current_stream().cuda_stream if ev is not None else None]
        return val

    def refresh_all(self):
        """Called by the native trainer right after its optimizer step (the parameters changed through raw pointers, the weight epoch
        was bumped): every registered fold / bf16 pack is recomputed in place by ONE multi-tensor launch per kind (csrc/pack.hip,
        cpr_bn_fold_multi / cpr_pack_weights_bf16_multi: bit for bit the single-tensor kernels) and its entry re-stamped with the
        current version, instead of ~320 lazy rebuilds -- launches, allocations and Python -- spread over the next step (4.3 of the
        37 ms of a configs[4] step, profiles/round6_stale_packs_ab.txt).  Entries without a job lapse and rebuild as before."""
        import numpy as np
        from . import _lib
        live = []
        for k, (t, j) in self._jobs.items():
            if k not in self._d:
                continue
            if j[0] in ('pack', 'pack32') and j[2] is not None:
                # a data-gradient pack multiplies a folded-BN scale in: only while that fold is refreshed in place too (else this
                # entry lapses with the epoch like any other and is rebuilt from the new fold)
                if j[2] not in self._jobs or j[2] not in self._d:
                    continue
            live.append((k, t, j))
        if not live:
            return

        def scale_of(j):
            return None if j[2] is None else self._d[j[2]][1][0]
        ptrs = tuple(x.data_ptr() for _, t, _ in live for x in t) + \
            tuple((0 if scale_of(j) is None else scale_of(j).data_ptr(), j[3].w.data_ptr(),
                   0 if getattr(j[3], 'wfrag', None) is None else j[3].wfrag.data_ptr()) for _, _, j in live if j[0] in ('pack', 'pack32'))
        if self._tables is None or self._tables[0] != ptrs:
            folds, packs, packs32, blocks, blocks32, max_c = [], [], [], 0, 0, 1
            for k, t, j in live:
                if j[0] == 'fold':
                    _, bn, sc, sh, inv = j
                    C = bn.weight.numel()
                    max_c = max(max_c, C)
                    folds.append((bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                                  0 if sc is None else sc.data_ptr(), 0 if sh is None else sh.data_ptr(),
                                  0 if inv is None else inv.data_ptr(), C, float(bn.eps)))
            # folds first: the data-gradient packs multiply the refreshed scale in
            for k, t, j in live:
                if j[0] == 'pack':
                    _, w, _, pc, transpose = j
                    sc = scale_of(j)
                    O, I, KH, KW = w.shape
                    rows, cols = (I, O) if transpose else (O, I)
                    nb = max(1, min(64, (rows * (KH * KW * cols // 2) + 255) // 256))
                    packs.append((w.data_ptr(), 0 if sc is None else sc.data_ptr(), pc.w.data_ptr(),
                                  0 if pc.wfrag is None else pc.wfrag.data_ptr(), O, I, KH, KW, int(transpose), blocks, nb, 0))
                    blocks += nb
                elif j[0] == 'pack32':          # the fp32 implicit-GEMM pack [rows][Kpad] (forward: colsp = padded Cin, dgrad: padded Cout)
                    _, w, _, pc, transpose = j
                    sc = scale_of(j)
                    O, I, KH, KW = w.shape
                    rows = I if transpose else O
                    nb = max(1, min(64, (rows * pc.Kpad + 255) // 256))
                    packs32.append((w.data_ptr(), 0 if sc is None else sc.data_ptr(), pc.w.data_ptr(), nb, 0,
                                    O, I, KH, KW, pc.Cin, pc.Kpad, int(transpose), blocks32))
                    blocks32 += nb
            dev = live[0][1][0].device
            fdt = np.dtype([('p', '<u8', (7,)), ('C', '<i4'), ('eps', '<f4')])
            pdt = np.dtype([('p', '<u8', (4,)), ('i', '<i4', (8,))])
            p32dt = np.dtype([('p', '<u8', (3,)), ('i', '<i4', (10,))])
            ft = pt = p32t = None
            if packs32:
                qa = np.zeros((len(packs32),), dtype=p32dt)
                for i, q in enumerate(packs32):
                    qa[i] = (q[:3], q[3:])
                p32t = torch.from_numpy(qa.view(np.uint8).reshape(-1).copy()).to(dev)
            if folds:
                fa = np.zeros((len(folds),), dtype=fdt)
                for i, f in enumerate(folds):
                    fa[i] = (f[:7], f[7], f[8])
                ft = torch.from_numpy(fa.view(np.uint8).reshape(-1).copy()).to(dev)
            if packs:
                pa = np.zeros((len(packs),), dtype=pdt)
                for i, q in enumerate(packs):
                    pa[i] = (q[:4], q[4:])
                pt = torch.from_numpy(pa.view(np.uint8).reshape(-1).copy()).to(dev)
            self._tables = (ptrs, ft, len(folds), max_c, pt, len(packs), blocks, p32t, len(packs32), blocks32)
        _, ft, nf, max_c, pt, npk, blocks, p32t, np32, blocks32 = self._tables
        stream = torch.cuda.current_stream().cuda_stream
        if ft is not None:
            _lib.call('cpr_bn_fold_multi', ft.data_ptr(), nf, max_c, stream)
        if pt is not None:
            _lib.call('cpr_pack_weights_bf16_multi', pt.data_ptr(), npk, blocks, stream)
        if p32t is not None:
            _lib.call('cpr_pack_weights_multi', p32t.data_ptr(), np32, blocks32, stream)
            for k, t, j in live:        # the Winograd images of a refreshed fp32 pack: G g G^T again, in place (one launch each)
                if j[0] == 'pack32':
                    pc = j[3]
                    for attr, fn in (('wino', 'cpr_wino_pack_weights'), ('wino32', 'cpr_wino32_pack_weights')):
                        img = getattr(pc, attr, None)
                        if img is not None:
                            _lib.call(fn, pc.w.data_ptr(), img.data_ptr(), pc.Cin, pc.Cout, pc.Kpad, stream)
        ev = torch.cuda.Event()
        ev.record()
        for k, t, j in live:
            hit = self._d[k]
            hit[0], hit[2], hit[3] = self._ver(t), ev, stream
            if j[0] in ('pack', 'pack32'):
                j[3].ready = ev

    @staticmethod
    def _order_behind(entry):
        ev = entry[2]
        if ev is None or torch.cuda.is_current_stream_capturing():     # (a capture starts after a synchronised warm-up)
            return
        if ev.query():
            entry[2] = None
        elif torch.cuda.current_stream().cuda_stream != entry[3]:
            torch.cuda.current_stream().wait_event(ev)


# CPR_REFRESH_IN_PLACE=0: every fold / pack lapses with the weight epoch and is rebuilt lazily (rounds 3-5; A/B switch)
REFRESH_IN_PLACE = [__import__('os').environ.get('CPR_REFRESH_IN_PLACE', '1') != '0']


def packed_conv(cache, conv, dtype=torch.float32):
    def job(pc):        # the bf16 pack kernel's outputs can be refreshed in place; the fp32 / Winograd packs lapse as before
        if dtype == torch.bfloat16 and conv.weight.is_cuda and ops.PACK_BF16_KERNEL[0] and conv.weight.dtype == torch.float32 \
                and conv.weight.is_contiguous():
            return ('pack', conv.weight, None, pc, 0)
        if dtype == torch.float32 and conv.weight.is_cuda and conv.weight.dtype == torch.float32 and conv.weight.is_contiguous():
            return ('pack32', conv.weight, None, pc, 0)
        return None
    return cache.get(('pc', id(conv), dtype), [conv.weight],
                     lambda: ops.PackedConv(conv.weight, conv.stride[0], conv.padding[0], dtype), refresh=job)


def folded_bn(cache, bn):
    """Eval-mode BatchNorm as a per-channel affine for the conv epilogue:
    scale = gamma / sqrt(var + eps), shift = beta - mean * scale."""
    def make():
        if bn.weight.is_cuda:
            return ops.bn_fold(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)[:2]
        with torch.no_grad():
            scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).float().contiguous()
            shift = (bn.bias - bn.running_mean * scale).float().contiguous()
        return scale, shift
    def job(val):
        ok = bn.weight.is_cuda and all(t.dtype == torch.float32 and t.is_contiguous()
                                       for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var, val[0], val[1]))
        return ('fold', bn, val[0], val[1], None) if ok else None
    return cache.get(('bn', id(bn)), [bn.weight, bn.bias, bn.running_mean, bn.running_var], make, refresh=job)


class ConvModule(nn.Module):
    """conv -> norm -> act container with mmcv's attribute names (``conv``, ``gn``/``bn``), bias='auto'."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias='auto', conv_cfg=None,
                 norm_cfg=None, act_cfg=dict(type='ReLU'), inplace=True):
        super().__init__()
        assert conv_cfg is None or conv_cfg.get('type') in (None, 'Conv2d'), 'only plain Conv2d is on this path'
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        if bias == 'auto':
            bias = not self.with_norm
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=bias)
        self.norm_name = None
        if self.with_norm:
            cfg = dict(norm_cfg)
            t = cfg.pop('type')
            rg = cfg.pop('requires_grad', True)
            if t == 'GN':
                self.norm_name = 'gn'
                norm = nn.GroupNorm(num_channels=out_channels, **cfg)
            elif t == 'BN':
                self.norm_name = 'bn'
                norm = nn.BatchNorm2d(out_channels, **cfg)
            else:
                raise KeyError(t)
            for p in norm.parameters():
                p.requires_grad = rg
            self.add_module(self.norm_name, norm)
        if self.with_activation:
            assert act_cfg.get('type', 'ReLU') == 'ReLU'

    @property
    def norm(self):
        return getattr(self, self.norm_name) if self.norm_name else None


def conv_gn(cache, m, x, in_ab=None, in_relu=False, materialize=True, up=None, save=None, consume_input=False,
            out_b8=False):
    """Run a ConvModule(conv, GN[, ReLU]) on an NHWC tensor (or a channel-blocked one, ops.is_b8: Winograd layers only).

    The GroupNorm statistics come out of the conv epilogue (no extra pass: always for the Winograd layers, for the
    direct kernel when the map is tile-aligned); with ``materialize=False`` the raw conv output and the per-(image,
    channel) affine (a, b) are returned so the CONSUMER conv applies normalisation + ReLU while loading its input (no
    apply pass either).
    ``save`` (dict): training mode -- records what the backward needs (conv input and its pending affine, raw output,
    GroupNorm affine and statistics) and keeps the raw output intact (the apply pass goes out of place).
    ``consume_input``: the caller owns ``x`` and nobody else reads it -- a pending producer-GroupNorm may be applied in place.
    ``out_b8``: the consumer is a Winograd layer -- hand it the raw output channel-blocked (forward only, materialize=False).

    Fused-on-load vs one streaming apply pass: the Winograd layers (3x3, stride 1) transform every input element once per
    cout tile on its way into the 4x4 patch transform -- 2 packed FMA/max per 8 bytes, invisible behind the MFMAs -- so they
    always fuse.  The direct kernel's KxK consumers re-transform every element K*K x (cout tiles) times (3x3 256->256 at
    160x160, B=16: 3.52 ms fused vs 3.30 ms plain) while gn_apply touches it once at HBM speed (0.14 ms): those take the
    materialised input in forward-only mode, 1x1 consumers fuse (CPR_GN_FUSE_IN=1 forces the fused form).
    """
    pc = packed_conv(cache, m.conv, x.dtype)
    gn = m.norm
    blocked = ops.is_b8(x)
    if blocked:
        N, _, H, W, _ = x.shape
    else:
        N, H, W, _ = x.shape
    wino = x.dtype == torch.float32 and ops.wino_eligible(pc, H, W, x.dtype) and pc.Cin <= 512
    if blocked and not wino:       # a channel-blocked map reached a layer that cannot read it: back to NHWC (+ its pending affine)
        x = ops.gn_apply_b8(x, *(in_ab if in_ab is not None else (None, None)), relu=in_relu and in_ab is not None)
        in_ab, blocked = None, False
    OH, OW = pc.out_hw(H, W)
    fused_stats = wino or (OH * OW) % 128 == 0
    fuse_in = in_ab is not None and (wino or ((H * W) % 128 == 0 and pc.stride == 1 and (OH, OW) == (H, W)
                                              and x.dtype == torch.float32))   # the bf16 kernel does not fuse the producer GN
    if fuse_in and not wino and save is None and pc.KH * pc.KW > 1 and os.environ.get('CPR_GN_FUSE_IN', '0') != '1':
        fuse_in = False
    if in_ab is not None and not fuse_in:
        x = ops.gn_apply(x, in_ab[0], in_ab[1], relu=in_relu, out=x if (consume_input and save is None) else None)
        in_ab = None
    out_b8 = bool(out_b8) and wino and save is None and not materialize
    bias = m.conv.bias
    if fused_stats:
        raw, part = ops.conv2d(x, pc, bias=bias, in_ab=in_ab, in_relu=in_relu, gn_part=True, out_b8=out_b8)
    else:
        raw = ops.conv2d(x, pc, bias=bias, in_ab=in_ab, in_relu=in_relu)
        part = ops.gn_stats(raw)
    if save is not None:
        a, b, mean, rstd = ops.gn_finalize(part, gn.weight, gn.bias, N, OH * OW, gn.num_groups, gn.eps, want_stats=True)
        save.update(module=m, x=x, in_ab=in_ab, in_relu=in_relu, raw=raw, a=a, b=b, mean=mean, rstd=rstd)
    else:
        a, b = ops.gn_finalize(part, gn.weight, gn.bias, N, OH * OW, gn.num_groups, gn.eps)
    if not materialize:
        return raw, (a, b)
    return ops.gn_apply(raw, a, b, relu=m.with_activation, up=up, out=None if save is not None else raw)
