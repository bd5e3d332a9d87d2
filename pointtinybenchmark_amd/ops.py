"""Thin tensor -> C-ABI wrappers.  PyTorch here is plumbing only: it owns the device memory
(caller-allocated outputs and workspaces) and the stream; every op runs in a
hand-written HIP kernel from libcprhip.so on the current HIP stream.

Activations are NHWC fp32 (see csrc/conv_mfma.hip for why).  ``as_nchw`` / ``from_nchw`` expose them as
NCHW-shaped (channels_last-strided) tensors, which is what crosses the reference's module boundaries.
"""
import ctypes
import math
import os

import torch

from . import _lib


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _check(t, dtype=torch.float32):
    """dtype may be a tuple of accepted dtypes (fp32 / bf16 activations)."""
    if not t.is_cuda:
        raise _lib.CprHipError('HIP op called with a CPU tensor: the product path has no CPU fallback')
    ok = t.dtype in dtype if isinstance(dtype, tuple) else t.dtype == dtype
    assert ok and t.is_contiguous(), (t.dtype, t.stride())
    return t


ACT = (torch.float32, torch.bfloat16)   # activation dtypes: fp32 (default, parity mode) or bf16 (configs[4] mode)


def as_nchw(x_nhwc):
    """(N,H,W,C) buffer -> NCHW-shaped view (channels_last strides, no copy)."""
    return x_nhwc.permute(0, 3, 1, 2)


def from_nchw(x):
    """NCHW-shaped tensor -> contiguous (N,H,W,C) buffer; free when x is already channels_last."""
    y = x.permute(0, 2, 3, 1)
    if y.is_contiguous():
        return y
    return nchw_to_nhwc(x)


def nchw_to_nhwc(x):
    N, C, H, W = x.shape
    if C <= 4:  # network input: pad to 4 channels for the stem's float4 taps
        x = _check(x.contiguous())
        out = torch.empty((N, H, W, 4), device=x.device, dtype=torch.float32)
        _lib.call('cpr_nchw_to_nhwc4', _ptr(x), _ptr(out), N, C, H, W, _stream())
        return out
    return x.permute(0, 2, 3, 1).contiguous()  # layout copy only (not on the benchmarked path)


def nhwc_to_nchw_dense(x_nhwc):
    N, H, W, C = x_nhwc.shape
    out = torch.empty((N, C, H, W), device=x_nhwc.device, dtype=torch.float32)
    _lib.call('cpr_nhwc_to_nchw', _ptr(_check(x_nhwc)), _ptr(out), N, C, H, W, _stream())
    return out


class PackedConv:
    """Conv weight repacked for the implicit-GEMM kernel: [Cout][KH][KW][Cin'] fp32, rows padded to a
    multiple of 32 floats.  Cin' = 4 for the 3-channel stem (zero 4th channel)."""
    wfrag = None    # bf16: the weights in MFMA-fragment order (frag_image), built on first use for Cout % 256 == 0
    wino = None     # transformed weights of the Winograd path, built on first use (conv3x3_wino)
    wino32 = None   # the same for the two-workgroups-per-CU kernel (csrc/conv_wino32.hip: chunks of 4 input channels)
    ready = None    # event recorded behind the last pack kernel (weights / Winograd image); see pack_ready()

    def _packed(self):
        """A pack kernel was just enqueued on the current stream: consumers on OTHER streams (CPR_STREAMS > 1 sub-batches)
        must order themselves behind it."""
        if torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
            self.ready = torch.cuda.Event()
            self.ready.record()

    def __init__(self, weight, stride=1, padding=0, dtype=torch.float32):
        Cout, Cin, KH, KW = weight.shape
        w = weight.detach().to(torch.float32).permute(0, 2, 3, 1)  # OHWI
        self.dtype = dtype
        if dtype == torch.bfloat16:
            # bf16 kernel: K chunks of 64 elements, no stem mode (the 3-channel stem stays on the fp32 kernel)
            assert Cin % 64 == 0, 'bf16 conv needs Cin % 64 == 0'
            self.Cout, self.Cin, self.KH, self.KW, self.Kpad = Cout, Cin, KH, KW, KH * KW * Cin
            self.stride, self.padding = stride, padding
            if weight.is_cuda and PACK_BF16_KERNEL[0]:
                self._pack_bf16(weight, None, 0)      # one launch: the [Cout][K] image and, where it is used, the fragment image
                return
            self.w = w.reshape(Cout, KH * KW * Cin).to(torch.bfloat16).contiguous()
            return
        if Cin <= 4:
            cin_p = 4
        else:
            assert Cin % 32 == 0, 'input channels must be a multiple of 32 (or <= 4 for the stem)'
            cin_p = Cin
        K = KH * KW * cin_p
        Kpad = (K + 31) // 32 * 32
        self.Cout, self.Cin, self.KH, self.KW, self.Kpad = Cout, cin_p, KH, KW, Kpad
        self.stride, self.padding = stride, padding
        if weight.is_cuda:       # one HIP launch (the training step re-packs every conv after each optimizer update)
            src = weight.detach()
            if src.dtype != torch.float32 or not src.is_contiguous():
                src = src.float().contiguous()
            self.w = torch.empty((Cout, Kpad), device=weight.device, dtype=torch.float32)
            _lib.call('cpr_pack_weights', _ptr(src), None, _ptr(self.w), Cout, Cin, KH, KW, cin_p, Kpad, 0, _stream())
            self._packed()
            return
        wp = torch.zeros((Cout, KH, KW, cin_p), device=weight.device, dtype=torch.float32)
        wp[..., :Cin] = w
        packed = torch.zeros((Cout, Kpad), device=weight.device, dtype=torch.float32)
        packed[:, :K] = wp.reshape(Cout, K)
        self.w = packed.contiguous()

    def _pack_bf16(self, weight, scale, transpose):
        """csrc/pack.hip, cpr_pack_weights_bf16: OIHW fp32 master -> self.w (bf16 [rows][K]) and, for rows % 256 == 0, self.wfrag.
        transpose: the data-gradient pack (rows = input channels, taps flipped, ``scale`` of the forward conv folded in)."""
        src = weight.detach()
        if src.dtype != torch.float32 or not src.is_contiguous():
            src = src.float().contiguous()
        O, I, KH, KW = src.shape
        rows = self.Cout
        self.w = torch.empty((rows, self.Kpad), device=src.device, dtype=torch.bfloat16)
        want_frag = rows % 256 == 0 and self.Kpad % 64 == 0 and WFRAG[0]
        if want_frag:
            self.wfrag = torch.empty((rows // 64, self.Kpad // 16, 2, 2, 32, 8), device=src.device, dtype=torch.bfloat16)
        _lib.call('cpr_pack_weights_bf16', _ptr(src), _ptr(scale), _ptr(self.w), _ptr(self.wfrag if want_frag else None), O, I, KH, KW,
                  int(transpose), _stream())
        self._packed()

    @classmethod
    def for_dgrad_bf16(cls, weight, padding, scale=None, pad_override=None):
        """The bf16 pack of the stride-1 conv over dy that yields the data gradient of a stride-1 conv (mixed-precision step): in / out
        channels swapped, taps flipped, the forward conv's folded-BN ``scale`` multiplied in (fp32) before the rounding, padding
        K-1-p (``pad_override``: explicit padding, used by the per-parity sub-kernels of a strided conv, which need not be square).
        The same bits as PackedConv((w * scale).flip(2, 3).permute(1, 0, 2, 3), 1, K-1-p, bf16), one launch."""
        Cout, Cin, KH, KW = weight.shape
        assert (KH == KW or pad_override is not None) and weight.is_cuda and Cout % 64 == 0
        self = cls.__new__(cls)
        self.dtype = torch.bfloat16
        self.Cout, self.Cin, self.KH, self.KW, self.Kpad = Cin, Cout, KH, KW, KH * KW * Cout
        self.stride, self.padding = 1, (KH - 1 - padding if pad_override is None else pad_override)
        self._pack_bf16(weight, scale, 1)
        return self

    def frag_image(self):
        """bf16 weights in the fragment order of conv_bf16_dma_kernel<4, 2, 4, true> (include/cpr_hip.h, cpr_conv2d_fwd_bf16):
        [cout / 64][k / 16][j][lane = l + 32 h][8] = w[64 g + 2 l + j][16 ks + 8 h .. + 8] -- the 16 bytes lane (l, h) of cout block j
        feeds to one MFMA, so a wave's weight operand of a k-step is two coalesced 1 KB loads.  None when the layer cannot take
        the 256 x 256 tile (Cout % 256, K % 64)."""
        if self.dtype != torch.bfloat16 or self.Cout % 256 != 0 or self.Kpad % 64 != 0 or not WFRAG[0]:
            return None
        if self.wfrag is None:
            G, KS = self.Cout // 64, self.Kpad // 16
            self.wfrag = self.w.view(G, 32, 2, KS, 2, 8).permute(0, 3, 2, 4, 1, 5).contiguous()
            self._packed()      # built on the calling stream: the other sub-batch streams order themselves behind it (pack_ready)
        return self.wfrag

    @classmethod
    def for_dgrad(cls, weight, padding, scale=None, pad_override=None):
        """Weights of the stride-1 conv over dy that yields the data gradient: in/out channels swapped, taps flipped,
        optional per-output-channel scale of the FORWARD conv (folded BatchNorm) multiplied in, padding K-1-p
        (``pad_override``: explicit padding, used by the per-parity sub-kernels of a strided conv)."""
        Cout, Cin, KH, KW = weight.shape
        assert (KH == KW or pad_override is not None) and weight.is_cuda
        assert Cout % 32 == 0 or Cout <= 4, 'gradient channels must be a multiple of 32 (or <= 4)'
        self = cls.__new__(cls)
        self.dtype = torch.float32
        cols_p = 4 if Cout <= 4 else Cout
        K = KH * KW * cols_p
        Kpad = (K + 31) // 32 * 32
        self.Cout, self.Cin, self.KH, self.KW, self.Kpad = Cin, cols_p, KH, KW, Kpad
        self.stride, self.padding = 1, (KH - 1 - padding if pad_override is None else pad_override)
        src = weight.detach()
        if src.dtype != torch.float32 or not src.is_contiguous():
            src = src.float().contiguous()
        self.w = torch.empty((Cin, Kpad), device=weight.device, dtype=torch.float32)
        _lib.call('cpr_pack_weights', _ptr(src), _ptr(scale), _ptr(self.w), Cout, Cin, KH, KW, cols_p, Kpad, 1, _stream())
        self._packed()
        return self

    def out_hw(self, H, W):
        return ((H + 2 * self.padding - self.KH) // self.stride + 1,
                (W + 2 * self.padding - self.KW) // self.stride + 1)


CONV_RELU, CONV_OUT_BF16, CONV_RES_MASK, CONV_COLSUM = 1, 2, 4, 8      # include/cpr_hip.h CPR_CONV_*
# bf16 mode: hand the fragment-order weight image to the conv launcher (the 256 x 256 tile then loads its weight operand
# straight into registers, csrc/conv_bf16_dma.hip BD instance).  CPR_BF16_WFRAG=0 keeps both operands on the LDS-DMA path (A/B).
WFRAG = [os.environ.get('CPR_BF16_WFRAG', '1') != '0']
# bf16 weight packs by one HIP launch per layer (csrc/pack.hip); CPR_PACK_BF16_KERNEL=0: the torch expression of rounds 3-4 (A/B, tests)
PACK_BF16_KERNEL = [os.environ.get('CPR_PACK_BF16_KERNEL', '1') != '0']
# profilers (bench.py) set [0] = True; the template instance of the last conv launch is then left in [1] as
# (kind, code).  Host-side, single-threaded bookkeeping of a value the C ABI returns through an out-parameter.
TRACE_CONV_VARIANT = [False, None]

# 3x3 / stride 1 / pad 1 fp32 layers run as fused Winograd F(2x2,3x3) (csrc/conv_wino.hip, 2.25x fewer multiplies) when the
# map fills its 16x16 output regions well enough; CPR_WINOGRAD=0 keeps every layer on the direct implicit GEMM (A/B runs).
WINOGRAD = [os.environ.get('CPR_WINOGRAD', '1') != '0']
WINO_MIN_FILL = 0.6      # useful share of the 16x16 regions (40x40 -> 0.69 runs Winograd, 20x20 -> 0.39 stays direct)


def pack_ready(pc):
    """Order the current stream behind the kernels that packed ``pc`` (they may have run on another stream: the first
    sub-batch of a multi-stream forward packs, the others only read).  The event is dropped once it has completed."""
    ev = pc.ready
    if ev is None or torch.cuda.is_current_stream_capturing():      # (a capture starts after a synchronised warm-up)
        return
    if ev.query():
        pc.ready = None
    else:
        torch.cuda.current_stream().wait_event(ev)


def wino_eligible(pc, H, W, dtype=torch.float32):
    if not (WINOGRAD[0] and dtype == torch.float32 and pc.dtype == torch.float32 and pc.KH == 3 and pc.KW == 3 and
            pc.stride == 1 and pc.padding == 1 and pc.Cin % 16 == 0 and pc.Cin >= 32 and pc.Cout % 64 == 0):
        return False
    fill = (H * W) / float(((H + 15) // 16 * 16) * ((W + 15) // 16 * 16))
    return fill >= WINO_MIN_FILL


# Which fused Winograd kernel serves a 3x3 layer: '32' = the two-workgroups-per-CU form (csrc/conv_wino32.hip: 8 x 16 pixel
# regions, 4-wave workgroups), '64' = the one-workgroup-per-CU form (csrc/conv_wino.hip: 16 x 16 regions).  The choice is a function
# of the LAYER alone (never of the batch size: an image of a big batch must equal its single-image run bit for bit, and the two
# kernels differ in accumulation order).  CPR_WINO_TILE=64 / 32 forces one of them everywhere it can run (A/B runs, tests).
WINO_TILE = [os.environ.get('CPR_WINO_TILE', 'auto')]


def wino_instance(pc, H, W, fused_affine):
    """'32' or '64' for a Winograd-eligible layer."""
    can32 = pc.Cin % 8 == 0 and (not fused_affine or pc.Cin <= 256)
    if WINO_TILE[0] == '64' or not can32:
        return '64'
    if WINO_TILE[0] == '32':
        return '32'
    return WINO_AUTO(pc, H, W, fused_affine)


WINO32_MAX_PIXELS = 40 * 40


def WINO_AUTO(pc, H, W, fused_affine):
    """Keyed on the LAYER GEOMETRY alone (never on the batch size: an image of a big batch must equal its single-image run bit
    for bit, and the two kernels differ in accumulation order).  Measured per shape on MI355X (profiles/round4_wino32_ab.txt,
    DESIGN.md 4.1d): the two-workgroups-per-CU kernel loses on the large maps at every batch (160x160x256: 8.9 vs 7.0 ms at
    B = 64, 0.359 vs 0.319 at B = 2; 80x80x128: 0.63 vs 0.54; 160x160x64: 0.71 vs 0.59) and wins where one image brings fewer
    16 x 16 regions than a tenth of the chip: 40x40x256 at the reference's batch of 2 runs 0.061 vs 0.071 ms (B = 2: 18 regions of
    16 x 16 per image against 30 of 8 x 16), at B = 64 it costs 0.709 vs 0.672 ms.  Maps of at most 40 x 40 pixels (layer3 of
    a 640 x 640 input) therefore take the 8 x 16-region kernel at every batch: -14 % on those layers at B = 2, +5 % at B = 64
    (0.3 % of that step)."""
    return '32' if H * W <= WINO32_MAX_PIXELS else '64'


def wino_gn_slots(pc, H, W, fused_affine):
    """GroupNorm-statistics slots per image of the Winograd kernel that serves this layer."""
    if wino_instance(pc, H, W, fused_affine) == '32':
        return ((H + 7) // 8) * ((W + 15) // 16)
    return ((H + 15) // 16) * ((W + 15) // 16)


def is_b8(x):
    """Channel-blocked activation (N, C/8, H, W, 8): the layout Winograd layers hand to each other (csrc/conv_wino.hip)."""
    return x.dim() == 5 and x.shape[-1] == 8


def conv3x3_wino(x, pc, scale=None, bias=None, relu=False, gn_part=False, out=None, in_ab=None, in_relu=False,
                 out_b8=False):
    """Winograd F(2x2,3x3) path of conv2d for 3x3 / stride 1 / pad 1 fp32 layers.  x: NHWC (N,H,W,Cin) or channel-blocked
    (N,Cin/8,H,W,8); out_b8=True returns the blocked form (N,Cout/8,H,W,8).  in_ab=(a,b): input read as relu?(x*a+b).
    gn_part=True also returns one (sum, sumsq) slot per output region of the kernel that runs (16x16, or 8x16 for the
    two-workgroups-per-CU form): (N * wino_gn_slots(...), Cout, 2)."""
    _check(x, ACT)
    in_b8 = is_b8(x)
    if in_b8:
        N, _, H, W, _ = x.shape
        Cin = x.shape[1] * 8
    else:
        N, H, W, Cin = x.shape
    assert x.dtype == torch.float32 and Cin == pc.Cin and pc.KH == 3 and pc.stride == 1 and pc.padding == 1
    inst = wino_instance(pc, H, W, in_ab is not None)
    attr, pack_fn = ('wino32', 'cpr_wino32_pack_weights') if inst == '32' else ('wino', 'cpr_wino_pack_weights')
    if getattr(pc, attr) is None:     # G g G^T of the packed weights, once per PackedConv (= once per weight update)
        pack_ready(pc)
        wino = torch.empty((16 * pc.Cin * pc.Cout,), device=x.device, dtype=torch.float32)
        _lib.call(pack_fn, _ptr(pc.w), _ptr(wino), pc.Cin, pc.Cout, pc.Kpad, _stream())
        setattr(pc, attr, wino)
        pc._packed()
    pack_ready(pc)
    shape = (N, pc.Cout // 8, H, W, 8) if out_b8 else (N, H, W, pc.Cout)
    if out is None:
        out = torch.empty(shape, device=x.device, dtype=torch.float32)
    assert tuple(out.shape) == shape and out.is_contiguous()
    part = None
    if gn_part:
        part = torch.empty((N * wino_gn_slots(pc, H, W, in_ab is not None), pc.Cout, 2), device=x.device, dtype=torch.float32)
    a = b = None
    if in_ab is not None:
        a, b = in_ab
        assert Cin <= 512
    _lib.call('cpr_conv3x3_wino32_fwd' if inst == '32' else 'cpr_conv3x3_wino_fwd', _ptr(x), _ptr(getattr(pc, attr)), _ptr(out),
              _ptr(scale), _ptr(bias), _ptr(a), _ptr(b), _ptr(part), N, H, W, Cin, pc.Cout, CONV_RELU if relu else 0, int(in_relu),
              (1 if in_b8 else 0) | (2 if out_b8 else 0), _stream())
    if TRACE_CONV_VARIANT[0]:
        TRACE_CONV_VARIANT[1] = ('wino32' if inst == '32' else 'wino',
                                 (1 if in_b8 else 0) | (2 if out_b8 else 0) | (4 if in_ab is not None else 0))
    return (out, part) if gn_part else out


def gn_apply_b8(x, a=None, b=None, relu=False):
    """Channel-blocked (N,C/8,H,W,8) -> NHWC (N,H,W,C), optionally through y = relu?(x*a+b)."""
    assert is_b8(x)
    N, C8, H, W, _ = _check(x, ACT).shape
    y = torch.empty((N, H, W, C8 * 8), device=x.device, dtype=torch.float32)
    _lib.call('cpr_gn_apply_b8', _ptr(x), _ptr(a), _ptr(b), _ptr(y), N, H, W, C8 * 8, int(relu), _stream())
    return y


def conv2d(x, pc, scale=None, bias=None, residual=None, relu=False, in_ab=None, in_relu=False, gn_part=False,
           out=None, out_dtype=None, res_mask=False, colsum=False, out_b8=False):
    """x (N,H,W,Cin) -> (N,OH,OW,Cout).  Epilogue: *scale[c] + bias[c] (+residual) (ReLU).
    in_ab=(a,b) applies x*a[n,c]+b[n,c] (+ReLU) to the input on load (fused GroupNorm of the producer; fp32 only).
    gn_part=True also returns per-128-pixel-tile per-channel (sum, sumsq) partials of the output.
    bf16 inputs run the bf16 MFMA kernel (fp32 accumulate); out_dtype overrides the output type (the fp32 stem can
    emit bf16, the bf16 logit projection emits fp32).
    Backward helpers (fp32): res_mask=True turns ``residual`` into a ReLU mask source (out = residual > 0 ? v : 0);
    colsum=True also returns the per-channel sums of the output (C,), taken from the epilogue partials."""
    _check(x, ACT)
    pack_ready(pc)
    if is_b8(x):     # channel-blocked input: only the Winograd layers read it
        assert residual is None and not (res_mask or colsum) and out_dtype in (None, torch.float32) and \
            wino_eligible(pc, x.shape[2], x.shape[3], x.dtype), 'channel-blocked input needs a Winograd-eligible layer'
        return conv3x3_wino(x, pc, scale, bias, relu, gn_part, out, in_ab, in_relu, out_b8)
    N, H, W, Cin = x.shape
    assert Cin == pc.Cin, (Cin, pc.Cin)
    assert x.dtype == pc.dtype, 'weights were packed for %s, input is %s' % (pc.dtype, x.dtype)
    OH, OW = pc.out_hw(H, W)
    odt = out_dtype or x.dtype
    if residual is None and not (res_mask or colsum) and odt == torch.float32 and wino_eligible(pc, H, W, x.dtype) and \
            (in_ab is None or Cin <= 512):
        return conv3x3_wino(x, pc, scale, bias, relu, gn_part, out, in_ab, in_relu, out_b8)
    assert not out_b8, 'channel-blocked output is produced by the Winograd layers only (check wino_eligible first)'
    if out is None:
        out = torch.empty((N, OH, OW, pc.Cout), device=x.device, dtype=odt)
    part = None
    if gn_part:
        assert (OH * OW) % 128 == 0 and not colsum
        part = torch.empty((N * OH * OW // 128, pc.Cout, 2), device=x.device, dtype=torch.float32)
    if colsum and x.dtype != torch.bfloat16:
        part = torch.empty(((N * OH * OW + 63) // 64, pc.Cout, 2), device=x.device, dtype=torch.float32)
    variant = ctypes.c_int(0) if (colsum or TRACE_CONV_VARIANT[0]) else None   # [host] out-parameter of the launcher
    vref = ctypes.byref(variant) if variant is not None else None
    if x.dtype == torch.bfloat16:
        assert in_ab is None, 'the bf16 kernel does not fuse the producer GroupNorm (materialise with gn_apply)'
        if res_mask or colsum:
            # mask mode (round 6): data gradient + ReLU backward + column sums (+ the bf16 rounding: out_dtype) in one launch
            assert res_mask and colsum and residual is not None and residual.dtype == torch.bfloat16 and not relu and not gn_part
            slots = conv2d_bf16_mask_slots(x.shape, pc, odt)
            assert slots > 0, 'this shape has no mask-mode instance (ask conv2d_bf16_mask_slots first)'
            part = torch.empty((slots, pc.Cout, 2), device=x.device, dtype=torch.float32)
            _lib.call('cpr_conv2d_fwd_bf16', _ptr(x), _ptr(pc.w), _ptr(pc.frag_image()), _ptr(out), _ptr(scale), _ptr(bias), _ptr(residual),
                      _ptr(part), N, H, W, Cin, pc.Cout, pc.KH, pc.KW, pc.stride, pc.padding, pc.Kpad, 2,
                      int(odt == torch.float32), vref, _stream())
            if variant is not None:
                TRACE_CONV_VARIANT[1] = ('bf16', variant.value)
            return out, TilePartials(part, slots, pc.Cout)
        _lib.call('cpr_conv2d_fwd_bf16', _ptr(x), _ptr(pc.w), _ptr(pc.frag_image()), _ptr(out), _ptr(scale), _ptr(bias), _ptr(residual),
                  _ptr(part), N, H, W, Cin, pc.Cout, pc.KH, pc.KW, pc.stride, pc.padding, pc.Kpad, int(relu),
                  int(odt == torch.float32), vref, _stream())
        if variant is not None:
            TRACE_CONV_VARIANT[1] = ('bf16', variant.value)
        return (out, part) if gn_part else out
    a = b = None
    if in_ab is not None:
        a, b = in_ab
    flags = (CONV_RELU if relu else 0) | (CONV_OUT_BF16 if odt == torch.bfloat16 else 0) | \
        (CONV_RES_MASK if res_mask else 0) | (CONV_COLSUM if colsum else 0)
    _lib.call('cpr_conv2d_fwd', _ptr(x), _ptr(pc.w), _ptr(out), _ptr(scale), _ptr(bias),
              _ptr(residual), _ptr(a), _ptr(b), _ptr(part), N, H, W, Cin, pc.Cout, pc.KH, pc.KW, pc.stride,
              pc.padding, pc.Kpad, flags, int(in_relu), vref, _stream())
    if variant is not None:
        TRACE_CONV_VARIANT[1] = ('fp32', variant.value)
    if colsum:
        bm = variant.value // 1000000     # tile edge the launcher picked
        return out, TilePartials(part, (N * OH * OW + bm - 1) // bm, pc.Cout)
    return (out, part) if gn_part else out


def conv2d_bf16_mask_slots(x_shape, pc, out_dtype=torch.bfloat16, fused_add=False):
    """Column-sum slots a mask-mode launch (``conv2d(bf16 x, pc, residual=mask, res_mask=True, colsum=True)``) of this shape writes;
    0 = the shape's kernel does not know the mode (include/cpr_hip.h, cpr_conv2d_bf16_mask_slots).  fused_add: the form with the
    shortcut sum and two outputs (``conv2d_dgrad_bf16_fused``)."""
    N, H, W, Cin = x_shape
    if fused_add and (not WFRAG[0] or pc.Cout % 256 != 0):
        return 0
    return _lib.call('cpr_conv2d_bf16_mask_slots', N, H, W, Cin, pc.Cout, pc.KH, pc.KW, pc.stride, pc.padding,
                     2 if fused_add else int(out_dtype == torch.float32), positive=True)


def conv2d_dgrad_bf16_fused(g16, pc, mask, add):
    """The block-boundary data gradient of the mixed-precision backward in one launch (include/cpr_hip.h,
    cpr_conv2d_dgrad_bf16_fused): g = mask > 0 ? conv(g16, pc) + add : 0 -> (g fp32, g bf16, column-sum TilePartials).  g16 bf16
    (N,H,W,Cin), pc: the rotated, BN-scaled bf16 pack, mask: bf16 map of the output's shape, add: fp32 map of the output's shape."""
    _check(g16, ACT)
    pack_ready(pc)
    N, H, W, Cin = g16.shape
    OH, OW = pc.out_hw(H, W)
    shape = (N, OH, OW, pc.Cout)
    assert g16.dtype == torch.bfloat16 and pc.dtype == torch.bfloat16 and Cin == pc.Cin
    assert mask.dtype == torch.bfloat16 and add.dtype == torch.float32 and tuple(mask.shape) == shape and tuple(add.shape) == shape
    assert mask.is_contiguous() and add.is_contiguous()
    slots = conv2d_bf16_mask_slots(g16.shape, pc, fused_add=True)
    assert slots > 0, 'this shape has no fused instance (ask conv2d_bf16_mask_slots(..., fused_add=True) first)'
    out32 = torch.empty(shape, device=g16.device, dtype=torch.float32)
    out16 = torch.empty(shape, device=g16.device, dtype=torch.bfloat16)
    part = torch.empty((slots, pc.Cout, 2), device=g16.device, dtype=torch.float32)
    variant = ctypes.c_int(0) if TRACE_CONV_VARIANT[0] else None
    _lib.call('cpr_conv2d_dgrad_bf16_fused', _ptr(g16), _ptr(pc.w), _ptr(pc.frag_image()), _ptr(out32), _ptr(out16), _ptr(add), _ptr(mask),
              _ptr(part), N, H, W, Cin, pc.Cout, pc.KH, pc.KW, pc.stride, pc.padding, pc.Kpad,
              ctypes.byref(variant) if variant is not None else None, _stream())
    if variant is not None:
        TRACE_CONV_VARIANT[1] = ('bf16', variant.value)
    return out32, out16, TilePartials(part, slots, pc.Cout)


def conv1x1_stream(x, pc, scale=None, bias=None, residual=None, relu=False, res_mask=False):
    """The streamed 1x1 kernel (csrc/conv1x1_stream.hip) called explicitly -- ``conv2d`` takes it by itself for large launches;
    tests compare the two kernels bit for bit.  x (N, H, W, Cin) fp32, Cin in {64, 128, 256}."""
    N, H, W, Cin = x.shape
    assert pc.KH == 1 and pc.stride == 1 and pc.padding == 0 and pc.Kpad == Cin and x.dtype == torch.float32
    out = torch.empty((N, H, W, pc.Cout), device=x.device, dtype=torch.float32)
    flags = (CONV_RELU if relu else 0) | (CONV_RES_MASK if res_mask else 0)
    _lib.call('cpr_conv1x1_stream_fwd', _ptr(x), _ptr(pc.w), _ptr(out), _ptr(scale), _ptr(bias), _ptr(residual),
              N * H * W, Cin, pc.Cout, flags, _stream())
    return out


def conv2d_dual(x, pc, x2, pc2, scale=None, bias=None, scale2=None, bias2=None, relu=False, out=None):
    """relu?((conv(x, pc) * scale + bias) + (conv1x1(x2, pc2) * scale2 + bias2)) in one launch: conv3 + bn3 of a stage's first
    bottleneck together with its projection shortcut (resnet.py:262-302) -- the shortcut map never reaches HBM.  Bit-identical
    to ``conv2d(x, pc, scale, bias, residual=conv2d(x2, pc2, scale2, bias2), relu=relu)``.  fp32 NHWC; pc2 is 1x1, unpadded."""
    _check(x, ACT)
    _check(x2, ACT)
    pack_ready(pc)
    pack_ready(pc2)
    N, H, W, Cin = x.shape
    N2, H2, W2, Cin2 = x2.shape
    assert x.dtype == torch.float32 and x2.dtype == torch.float32 and pc.dtype == torch.float32 and pc2.dtype == torch.float32
    assert N == N2 and Cin == pc.Cin and Cin2 == pc2.Cin and pc.Cout == pc2.Cout and pc2.KH == 1 and pc2.KW == 1 and pc2.padding == 0
    OH, OW = pc.out_hw(H, W)
    assert (OH, OW) == pc2.out_hw(H2, W2), 'the two sources must produce the same output grid'
    if out is None:
        out = torch.empty((N, OH, OW, pc.Cout), device=x.device, dtype=torch.float32)
    variant = ctypes.c_int(0) if TRACE_CONV_VARIANT[0] else None
    _lib.call('cpr_conv2d_dual_fwd', _ptr(x), _ptr(pc.w), _ptr(x2), _ptr(pc2.w), _ptr(out), _ptr(scale), _ptr(bias),
              _ptr(scale2), _ptr(bias2), N, H, W, Cin, pc.Cout, pc.KH, pc.KW, pc.stride, pc.padding, pc.Kpad, H2, W2, Cin2,
              pc2.stride, pc2.Kpad, CONV_RELU if relu else 0, ctypes.byref(variant) if variant is not None else None, _stream())
    if variant is not None:
        TRACE_CONV_VARIANT[1] = ('fp32', variant.value)
    return out


class TilePartials:
    """Per-tile per-channel sums left by a conv epilogue; ``reduce()`` -> (C,) column sums on the CURRENT stream (the
    training step does it on its side stream, next to the only consumer, so the main chain never waits for it)."""

    def __init__(self, part, tiles, C):
        self.part, self.tiles, self.C = part, tiles, C

    def record_stream(self, stream):
        self.part.record_stream(stream)

    def reduce(self):
        cs = torch.empty((self.C,), device=self.part.device, dtype=torch.float32)
        ws = torch.empty((64 * self.C,), device=self.part.device, dtype=torch.float32)
        _lib.call('cpr_part_colsum', _ptr(self.part), _ptr(cs), _ptr(ws), self.tiles, self.C, _stream())
        return cs


def _sfx(x):
    return '_bf16' if x.dtype == torch.bfloat16 else ''


def bn_fold(gamma, beta, mean, var, eps, want_inv_sigma=False):
    """Eval-mode BatchNorm as a per-channel affine: (scale, shift[, inv_sigma])."""
    C = gamma.numel()
    dev = gamma.device
    scale = torch.empty((C,), device=dev, dtype=torch.float32)
    shift = torch.empty((C,), device=dev, dtype=torch.float32)
    inv = torch.empty((C,), device=dev, dtype=torch.float32) if want_inv_sigma else None
    _lib.call('cpr_bn_fold', _ptr(_check(gamma.detach())), _ptr(_check(beta.detach())), _ptr(_check(mean)), _ptr(_check(var)),
              float(eps), _ptr(scale), _ptr(shift), _ptr(inv), C, _stream())
    return scale, shift, inv


def stem_weight_f32(weight):
    """(64, 3, 7, 7) conv1 weight -> the (64, 154) fp32 image of csrc/stem_f32.hip: [cout][kh][kw * 3 + c], slot 21 of every
    kernel row zero."""
    assert tuple(weight.shape) == (64, 3, 7, 7), tuple(weight.shape)
    w = torch.zeros((64, 7, 22), device=weight.device, dtype=torch.float32)
    w[:, :, :21] = weight.detach().float().permute(0, 2, 3, 1).reshape(64, 7, 21)
    return w.reshape(64, 154).contiguous()


def stem7x7s2_pool_f32(x, wpack, scale=None, bias=None, planar=None):
    """The whole ResNet stem in one kernel, exact fp32: x (N,H,W,4) NHWC4 or the (N,3,H,W) network input -> (N,PH,PW,64) =
    maxpool3x3s2(ReLU(conv7x7/2(x) * scale + bias))."""
    N, H, W, layout = _stem_input(_check(x), planar)
    assert wpack.dtype == torch.float32 and tuple(wpack.shape) == (64, 154)
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.empty((N, (OH - 1) // 2 + 1, (OW - 1) // 2 + 1, 64), device=x.device, dtype=torch.float32)
    _lib.call('cpr_stem7x7s2_pool_f32', _ptr(x), _ptr(wpack), _ptr(scale), _ptr(bias), _ptr(out), N, H, W, layout, _stream())
    return out


def stem_weight_bf16(weight):
    """(64, 3, 7, 7) conv1 weight -> the (64, 224) bf16 image of csrc/stem_bf16.hip: k = (kh * 8 + kw) * 4 + c, zero at kw = 7
    and c = 3."""
    assert tuple(weight.shape) == (64, 3, 7, 7), tuple(weight.shape)
    w = torch.zeros((64, 7, 8, 4), device=weight.device, dtype=torch.float32)
    w[:, :, :7, :3] = weight.detach().float().permute(0, 2, 3, 1)
    return w.reshape(64, 224).to(torch.bfloat16).contiguous()


def _stem_input(x, planar=None):
    """(N, H, W, layout) of a stem input: NHWC4 (N,H,W,4) -> layout 0, the NCHW network input (N,3,H,W) -> layout 1.
    planar: say which one it is (a (N,3,W,4) tensor reads both ways); None = by shape, NHWC4 first."""
    assert x.dim() == 4 and x.dtype == torch.float32 and x.is_contiguous()
    if planar is None:
        planar = x.shape[-1] != 4
    if not planar:
        assert x.shape[-1] == 4, tuple(x.shape)
        return x.shape[0], x.shape[1], x.shape[2], 0
    assert x.shape[1] == 3, tuple(x.shape)
    return x.shape[0], x.shape[2], x.shape[3], 1


def stem_bf16_fits(x, planar=None):
    N, H, W, _ = _stem_input(x, planar)
    return N * ((H - 1) // 2 + 1) * ((W - 1) // 2 + 1) * 64 * 2 < (1 << 31)


def stem7x7s2_bf16(x, wpack, scale=None, bias=None, relu=True, planar=None):
    """ResNet stem of the bf16 compute mode: x (N,H,W,4) fp32 NHWC4 or (N,3,H,W) fp32 -> (N,OH,OW,64) bf16 =
    ReLU(conv7x7/2(x) * scale + bias)."""
    N, H, W, layout = _stem_input(_check(x), planar)
    assert wpack.dtype == torch.bfloat16 and tuple(wpack.shape) == (64, 224)
    out = torch.empty((N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, 64), device=x.device, dtype=torch.bfloat16)
    _lib.call('cpr_stem7x7s2_bf16', _ptr(x), _ptr(wpack), _ptr(scale), _ptr(bias), _ptr(out), N, H, W, int(relu), layout, _stream())
    return out


def stem7x7s2_pool_bf16(x, wpack, scale=None, bias=None, planar=None):
    """stem7x7s2_bf16 (with ReLU) + maxpool3x3s2 in one kernel: x (N,H,W,4) or (N,3,H,W) fp32 -> (N,PH,PW,64) bf16; same bits as
    the pair."""
    N, H, W, layout = _stem_input(_check(x), planar)
    assert wpack.dtype == torch.bfloat16 and tuple(wpack.shape) == (64, 224)
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.empty((N, (OH - 1) // 2 + 1, (OW - 1) // 2 + 1, 64), device=x.device, dtype=torch.bfloat16)
    _lib.call('cpr_stem7x7s2_pool_bf16', _ptr(x), _ptr(wpack), _ptr(scale), _ptr(bias), _ptr(out), N, H, W, layout, _stream())
    return out


def maxpool3x3s2(x):
    N, H, W, C = _check(x, ACT).shape
    out = torch.empty((N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C), device=x.device, dtype=x.dtype)
    _lib.call('cpr_maxpool3x3s2' + _sfx(x), _ptr(x), _ptr(out), N, H, W, C, _stream())
    return out


def gn_stats(x, slots=None):
    """Per (image, slot, channel) (sum, sumsq) partials of an NHWC tensor."""
    N, H, W, C = _check(x, ACT).shape
    HW = H * W
    if slots is None:
        slots = max(1, min(256, HW // 256))
    part = torch.empty((N * slots, C, 2), device=x.device, dtype=torch.float32)
    _lib.call('cpr_gn_stats' + _sfx(x), _ptr(x), _ptr(part), N, HW, C, slots, _stream())
    return part


def gn_finalize(part, gamma, beta, N, HW, groups=32, eps=1e-5, want_stats=False):
    """partials -> per (image, channel) affine (a, b) with y = x*a + b == GroupNorm(x)."""
    C = part.shape[1]
    P = part.shape[0] // N
    a = torch.empty((N, C), device=part.device, dtype=torch.float32)
    b = torch.empty((N, C), device=part.device, dtype=torch.float32)
    mean = rstd = None
    if want_stats:
        mean = torch.empty((N, groups), device=part.device, dtype=torch.float32)
        rstd = torch.empty((N, groups), device=part.device, dtype=torch.float32)
    _lib.call('cpr_gn_finalize', _ptr(part), _ptr(_check(gamma)), _ptr(_check(beta)), _ptr(a), _ptr(b), _ptr(mean),
              _ptr(rstd), N, P, C, groups, HW, float(eps), _stream())
    return (a, b, mean, rstd) if want_stats else (a, b)


def gn_apply(x, a, b, relu=False, up=None, out=None):
    """y = x*a[n,c] + b[n,c] (ReLU) (+ nearest-upsampled ``up``).  In place when out is x."""
    N, H, W, C = _check(x, ACT).shape
    if out is None:
        out = torch.empty_like(x)
    UH = UW = 0
    if up is not None:
        assert up.dtype == x.dtype
        UH, UW = up.shape[1], up.shape[2]
    _lib.call('cpr_gn_apply' + _sfx(x), _ptr(x), _ptr(a), _ptr(b), _ptr(up), _ptr(out), N, H, W, C, UH, UW, int(relu),
              _stream())
    return out


# ------------------------------------------------------------------------------------------------ CPR points
def box_centers(boxes):
    n = boxes.shape[0]
    out = torch.empty((n, 2), device=boxes.device, dtype=torch.float32)
    _lib.call('cpr_box_centers', _ptr(_check(boxes)), _ptr(out), n, _stream())
    return out


def logit_project(x, w, bias, in_ab=None, in_relu=True):
    """x (N,H,W,Cin) fp32, w (J,Cin), bias (J) -> (N,H,W,J): the streaming projection kernel (J <= 8, Cin in {64, 128,
    192, 256}); returns None when the shape is outside what it covers (the caller uses the MFMA conv)."""
    N, H, W, Cin = _check(x).shape
    J = w.shape[0]
    if J > 8 or Cin > 256 or Cin % 64:
        return None
    out = torch.empty((N, H, W, J), device=x.device, dtype=torch.float32)
    a, b = in_ab if in_ab is not None else (None, None)
    _lib.call('cpr_logit_project', _ptr(x), _ptr(_check(w)), _ptr(_check(bias)), _ptr(a), _ptr(b), _ptr(out), N, H * W, Cin, J,
              int(in_relu), _stream())
    return out


PROB_TYPES = {'sigmoid': 0, 'softmax': 1, 'normed_sigmoid': 2, 'identity': 3}


def neg_mask_loss(logit_map, centers, labels, gt_start, pad_hw, num_classes, stride, d2_thr, eps=1e-6,
                  class_wise=True, prob_type='sigmoid', norm_p=1.0, mask_classes=None):
    """logit_map (N,H,W,J) -> mask (N*H*W, C) uint8, partial sums (double).  centers/labels/gt_start: the annotated
    points in CSR form (with num_refine > 1: every refine point, carrying its gt's label).  mask_classes=1 with
    num_classes=2: out_bg_cls, the single class's validity covers the [class, background] outputs."""
    N, H, W, J = _check(logit_map).shape
    C = num_classes
    mask = torch.empty((N * H * W, C), device=logit_map.device, dtype=torch.uint8)
    nblk = (H * W * C + 255) // 256
    partial = torch.empty((N * nblk,), device=logit_map.device, dtype=torch.float64)
    _lib.call('cpr_neg_mask_loss', _ptr(logit_map), J, _ptr(centers), _ptr(labels), _ptr(gt_start), _ptr(pad_hw),
              _ptr(mask), _ptr(partial), N, H, W, C, float(stride), float(d2_thr), float(eps), int(class_wise),
              PROB_TYPES[prob_type], float(norm_p), C if mask_classes is None else int(mask_classes), None, _stream())
    return mask, partial


def bag_sample(fmap, centers, gt_img, pad_hw, offsets, stride, align_corners=False, pad_value=None):
    """fmap (N,H,W,J); one bag per row of centers; returns pts (G,K,2), valid (G,K) uint8, sampled (G,K,J).
    align_corners: the generators' align_corners=True sampling (zeros padding); pad_value (J,): contribution of a dropped tap per
    unit of weight (the projection's bias when fmap is the logit map)."""
    N, H, W, J = _check(fmap).shape
    G = centers.shape[0]
    K = (offsets.shape[0] if offsets is not None else 0) + 1
    pts = torch.empty((G, K, 2), device=fmap.device, dtype=torch.float32)
    valid = torch.empty((G, K), device=fmap.device, dtype=torch.uint8)
    out = torch.empty((G, K, J), device=fmap.device, dtype=torch.float32)
    _lib.call('cpr_bag_sample', _ptr(fmap), J, _ptr(centers), _ptr(gt_img), _ptr(pad_hw), _ptr(offsets), _ptr(pts),
              _ptr(valid), _ptr(out), G, K, H, W, float(stride), int(align_corners), _ptr(pad_value), _stream())
    return pts, valid, out


def grid_bag(fmap, points, gt_img, num_refine, max_pos_num, radius_px, stride, pad_value=None, align_corners=False,
             want_cell=False):
    """GridCirclesPtFeatGenerator bags: fmap (N,H,W,J), points (G*R,2), gt_img (G) -> pts (G,Kmax+R,2),
    valid (G,Kmax+R) uint8, sampled (G,Kmax+R,J), count (G) int32 (grid points found per gt).  pad_value (J,): what the
    padding slots hold (default zeros).  want_cell: also the entry codes (G,Kmax+R) int32 -- cell index | -1 padding slot |
    <= -2 refine point (bilinear sample) -- which ``bag_points_gather_bwd`` walks."""
    N, H, W, J = _check(fmap).shape
    R = int(num_refine)
    G = points.shape[0] // R
    Kt = int(max_pos_num) + R
    dev = fmap.device
    pts = torch.empty((G, Kt, 2), device=dev, dtype=torch.float32)
    valid = torch.empty((G, Kt), device=dev, dtype=torch.uint8)
    cell = torch.empty((G, Kt), device=dev, dtype=torch.int32)
    count = torch.empty((G,), device=dev, dtype=torch.int32)
    out = torch.empty((G, Kt, J), device=dev, dtype=torch.float32)
    assert pad_value is None or pad_value.numel() == J
    _lib.call('cpr_grid_bag', _ptr(fmap), J, _ptr(_check(points)), _ptr(gt_img), R, int(max_pos_num), float(radius_px),
              _ptr(pad_value), _ptr(pts), _ptr(valid), _ptr(cell), _ptr(count), _ptr(out), G, H, W, float(stride),
              int(align_corners), _stream())
    return (pts, valid, out, count, cell) if want_cell else (pts, valid, out, count)


def mil_loss(logits, ins_off, valid, labels, num_classes, neg_partial, w_mil, w_gt, w_neg, gt_weight=None, eps=1e-6,
             bags=None, centres=None, prob_type='sigmoid', norm_p=1.0, binary_ins=False, allpos=False,
             neg_from_gt=False):
    """logits (G,K,J) [entries = G*K rows], valid (G,K).  bags = (num_bags, stride, off, len) in entries (default: one bag
    per row of the (G,K) layout); centres = (off, stride, count, mod): the annotated-point loss entries of a bag (default:
    the last entry of every bag).  labels / gt_weight are per bag.
    -> (5,) tensor {gt_loss, pos_loss, bag_acc, neg_loss, num_sample} and the per-bag workspace (num_bags,5)."""
    G, K, J = _check(logits).shape
    nb, bstride, boff, blen = bags if bags is not None else (G, K, 0, K)
    coff, cstride, ccount, cmod = centres if centres is not None else (K - 1, K, 1, 1)
    assert labels.numel() == nb and (gt_weight is None or gt_weight.numel() == nb)
    assert (nb - 1) * bstride + boff + blen <= G * K
    bag = torch.empty((nb, 5), device=logits.device, dtype=torch.float32)
    out = torch.empty((5,), device=logits.device, dtype=torch.float32)
    npart = 0 if neg_partial is None else neg_partial.numel()
    _lib.call('cpr_mil_loss', _ptr(logits), J, ins_off, _ptr(valid), _ptr(labels), _ptr(gt_weight), _ptr(bag),
              _ptr(neg_partial), npart, nb, bstride, boff, blen, coff, cstride, ccount, cmod, num_classes, float(eps),
              PROB_TYPES[prob_type], float(norm_p), int(binary_ins), int(allpos), float(w_mil), float(w_gt),
              float(w_neg), int(neg_from_gt), _ptr(out), _stream())
    return out, bag


def refine(logits, pts, valid, centers, labels, gt_img, gt_start, img_hw, num_classes, gt_alpha, merge_th, refine_th,
           use_nearest=True, use_classify=False, not_refine_in=None, sub_bags=1, ctr_stride=None, prob_type='sigmoid',
           norm_p=1.0, score_max=False):
    """logits (G,Kt,J): a gt owns Kt = sub_bags*Kv entries; centers holds ctr_stride points per gt (default sub_bags)."""
    G, Kt, J = _check(logits).shape
    Rv = int(sub_bags)
    assert Kt % Rv == 0
    ctr_stride = Rv if ctr_stride is None else int(ctr_stride)
    assert centers.shape[0] == G * ctr_stride
    dev = logits.device
    rp = torch.empty((G, 2), device=dev, dtype=torch.float32)
    sc = torch.empty((G,), device=dev, dtype=torch.float32)
    nr = torch.empty((G,), device=dev, dtype=torch.uint8)
    chosen = torch.empty((G, Kt), device=dev, dtype=torch.uint8)
    _lib.call('cpr_refine', _ptr(logits), J, _ptr(pts), _ptr(valid), _ptr(centers), Rv, ctr_stride, _ptr(labels),
              _ptr(gt_img), _ptr(gt_start), _ptr(img_hw), _ptr(not_refine_in), _ptr(rp), _ptr(sc), _ptr(nr),
              _ptr(chosen), G, Kt, Kt // Rv, num_classes, PROB_TYPES[prob_type], float(norm_p), float(gt_alpha),
              float(merge_th), float(refine_th), int(use_nearest), int(use_classify), int(score_max), _stream())
    return rp, sc, nr, chosen


# ------------------------------------------------------------------------------------------------ assigners
def point_assign(points, gt_bboxes, scale=4, pos_num=3):
    n, k = points.shape[0], gt_bboxes.shape[0]
    dev = points.device
    inds = torch.zeros((n,), device=dev, dtype=torch.int64)
    if n == 0 or k == 0:
        return inds
    best = torch.empty((n,), device=dev, dtype=torch.float32)
    lvl = torch.empty((n,), device=dev, dtype=torch.int32)
    _lib.call('cpr_point_assign', _ptr(_check(points)), _ptr(_check(gt_bboxes)), n, k, float(scale), pos_num,
              _ptr(inds), _ptr(best), _ptr(lvl), _stream())
    return inds


def hungarian_cost(pred, logits, gt, labels, w_cls=2.0, alpha=0.25, gamma=2.0, eps=1e-12, w_dis=0.1, fx=1.0,
                   fy=1.0, p=1):
    """-> cost^T (G, M) fp32 (column-major view of the reference's (M, G) cost)."""
    M, G = pred.shape[0], gt.shape[0]
    costT = torch.empty((G, M), device=pred.device, dtype=torch.float32)
    _lib.call('cpr_hungarian_cost', _ptr(_check(pred)), pred.shape[1], _ptr(_check(logits)), logits.shape[1],
              _ptr(_check(gt)), _ptr(labels), _ptr(costT), M, G, float(w_cls), float(alpha), float(gamma),
              float(eps), float(w_dis), float(fx), float(fy), int(p), _stream())
    return costT


LSA_REGISTER_KERNEL = [True]     # test hook: False keeps the memory-resident kernel (the two must agree bit for bit)


def lsa_topk(costT_list, topk):
    """Solve a batch of independent assignment problems (one workgroup each).  costT_list: list of (G_b, M_b)
    fp32 tensors with M_b >= G_b.  Returns list of gt_inds (M_b,) int64 (0 = background, j+1 = gt j)."""
    dev = costT_list[0].device
    Ms = [c.shape[1] for c in costT_list]
    Gs = [c.shape[0] for c in costT_list]
    nb = len(costT_list)
    flat = torch.cat([c.reshape(-1) for c in costT_list]) if nb > 1 else costT_list[0].reshape(-1)
    cost_off, col_off, row_off = [0], [0], [0]
    for m, g in zip(Ms, Gs):
        cost_off.append(cost_off[-1] + m * g)
        col_off.append(col_off[-1] + m)
        row_off.append(row_off[-1] + g)
    # index tables: keep them alive in locals until after the launch (a temporary would be freed -- and its block
    # recycled by the caching allocator -- before the kernel reads it)
    t_m = torch.tensor(Ms, dtype=torch.int32, device=dev)
    t_g = torch.tensor(Gs, dtype=torch.int32, device=dev)
    t_off = torch.tensor([cost_off[:-1], col_off[:-1], row_off[:-1]], dtype=torch.int64, device=dev)
    tm, tg = col_off[-1], max(row_off[-1], 1)
    gt_inds = torch.zeros((tm,), device=dev, dtype=torch.int64)
    ws_v = torch.empty((tm,), device=dev, dtype=torch.float64)
    ws_spc = torch.empty((tm,), device=dev, dtype=torch.float64)
    ws_path = torch.empty((tm,), device=dev, dtype=torch.int32)
    ws_r4c = torch.empty((tm,), device=dev, dtype=torch.int32)
    ws_sc = torch.empty((tm,), device=dev, dtype=torch.uint8)
    ws_act = torch.empty((tm,), device=dev, dtype=torch.uint8)
    ws_cols = torch.empty((3, tm), device=dev, dtype=torch.int32)   # cols / remaining / pos
    ws_u = torch.empty((tg,), device=dev, dtype=torch.float64)
    ws_c4r = torch.empty((tg,), device=dev, dtype=torch.int32)
    ws_sr = torch.empty((tg,), device=dev, dtype=torch.uint8)
    status = torch.zeros((nb,), device=dev, dtype=torch.int32)
    _lib.call('cpr_lsa_topk', _ptr(flat), _ptr(t_m), _ptr(t_g), _ptr(t_off[0]), _ptr(t_off[1]), _ptr(t_off[2]), nb,
              int(topk), _ptr(gt_inds), _ptr(ws_v), _ptr(ws_spc),
              _ptr(ws_path), _ptr(ws_r4c), _ptr(ws_sc), _ptr(ws_act), _ptr(ws_cols[0]), _ptr(ws_cols[1]),
              _ptr(ws_cols[2]), _ptr(ws_u), _ptr(ws_c4r), _ptr(ws_sr), _ptr(status),
              max(Ms) if LSA_REGISTER_KERNEL[0] else 0, max(Gs), _stream())
    return [gt_inds[col_off[i]:col_off[i + 1]] for i in range(nb)], status


# ------------------------------------------------------------------------------------------------ P2P inference
def topk_desc(scores, k):
    """k largest of a 1-D fp32 tensor, sorted descending, ties by lower index.  -> (values, indices int64)"""
    n = _check(scores).numel()
    vals = torch.empty((k,), device=scores.device, dtype=torch.float32)
    idx = torch.empty((k,), device=scores.device, dtype=torch.int64)
    _lib.call('cpr_topk_desc', _ptr(scores), n, int(k), _ptr(vals), _ptr(idx), _stream())
    return vals, idx


def nms_candidates(boxes, scores, score_thr, factors=None):
    """(proposal, class) pairs above score_thr in row-major order.  boxes (n,4) or (n,4C), scores (n,C+1) (last column =
    background).  -> cand_boxes (k,4), cand_scores (k), cand_labels (k) int32, cand_inds (k) int64.
    One host read of k (the list is variable-length in the reference too)."""
    n, C = scores.shape[0], scores.shape[1] - 1
    dev = scores.device
    cap = max(n * C, 1)
    cb = torch.empty((cap, 4), device=dev, dtype=torch.float32)
    cs = torch.empty((cap,), device=dev, dtype=torch.float32)
    cl = torch.empty((cap,), device=dev, dtype=torch.int32)
    ci = torch.empty((cap,), device=dev, dtype=torch.int64)
    cnt = torch.zeros((1,), device=dev, dtype=torch.int32)
    _lib.call('cpr_nms_candidates', _ptr(_check(boxes)), boxes.shape[1], _ptr(_check(scores)), _ptr(factors), n, C,
              float(score_thr), _ptr(cb), _ptr(cs), _ptr(cl), _ptr(ci), _ptr(cnt), _stream())
    k = int(cnt.item())
    return cb[:k], cs[:k], cl[:k], ci[:k]


def nms(boxes, scores, labels, iou_thr):
    """Class-aware greedy NMS (batched_nms semantics).  -> keep indices (int64, descending score).
    One host read of the keep count (the reference's NMS output is variable-length too)."""
    n = boxes.shape[0]
    dev = boxes.device
    if n == 0:
        return torch.zeros((0,), device=dev, dtype=torch.int64)
    keep = torch.empty((n,), device=dev, dtype=torch.int64)
    num = torch.zeros((1,), device=dev, dtype=torch.int32)
    order = torch.empty((n,), device=dev, dtype=torch.int32)
    sboxes = torch.empty((n, 4), device=dev, dtype=torch.float32)
    # (sort workspace, then the pair mask -- whole for n <= 8192, else one band of 8192 rows + the carried removed bits)
    mask = torch.empty((_lib.call('cpr_nms_workspace', n, positive=True),), device=dev, dtype=torch.int64)
    _lib.call('cpr_nms', _ptr(_check(boxes)), _ptr(_check(scores)), _ptr(_check(labels, torch.int32)), n,
              float(iou_thr), _ptr(keep), _ptr(num), _ptr(order), _ptr(sboxes), _ptr(mask), _stream())
    return keep[:int(num.item())]


def topk_desc_batched(scores, k):
    """scores (B, n) fp32 -> (values (B, k), indices (B, k) int64): the k largest of every row, sorted descending, ties by
    lower index -- one launch for the batch."""
    B, n = _check(scores).shape
    vals = torch.empty((B, k), device=scores.device, dtype=torch.float32)
    idx = torch.empty((B, k), device=scores.device, dtype=torch.int64)
    _lib.call('cpr_topk_desc_batched', _ptr(scores), B, n, int(k), _ptr(vals), _ptr(idx), _stream())
    return vals, idx


def nms_candidates_batched(boxes, scores, score_thr, factors=None):
    """boxes (B, n, 4) or (B, n, 4C), scores (B, n, C+1) -> per-image slabs of capacity n * C: cand_boxes (B, cap, 4),
    cand_scores (B, cap), cand_labels (B, cap) int32, cand_inds (B, cap) int64 and count (B) int32 ON THE DEVICE (no host read)."""
    B, n, C = scores.shape[0], scores.shape[1], scores.shape[2] - 1
    dev = scores.device
    cap = max(n * C, 1)
    cb = torch.empty((B, cap, 4), device=dev, dtype=torch.float32)
    cs = torch.empty((B, cap), device=dev, dtype=torch.float32)
    cl = torch.empty((B, cap), device=dev, dtype=torch.int32)
    ci = torch.empty((B, cap), device=dev, dtype=torch.int64)
    cnt = torch.zeros((B,), device=dev, dtype=torch.int32)
    if n * C > 0:
        _lib.call('cpr_nms_candidates_batched', _ptr(_check(boxes)), boxes.shape[2], _ptr(_check(scores)), _ptr(factors), B, n, C,
                  float(score_thr), _ptr(cb), _ptr(cs), _ptr(cl), _ptr(ci), _ptr(cnt), _stream())
    return cb, cs, cl, ci, cnt


def nms_batched(boxes, scores, labels, counts, iou_thr):
    """Class-aware greedy NMS of B candidate slabs (nms_candidates_batched outputs): the candidate counts are read on the device.
    -> keep (B, cap) int64 slab-relative indices in descending score order, num_keep (B) int32 on the device."""
    B, cap = scores.shape
    dev = scores.device
    keep = torch.empty((B, cap), device=dev, dtype=torch.int64)
    num = torch.zeros((B,), device=dev, dtype=torch.int32)
    nblk = (cap + 63) // 64
    P = 1
    while P < cap:
        P <<= 1
    stride = max(cap * nblk, P)
    order = torch.empty((B, cap), device=dev, dtype=torch.int32)
    sboxes = torch.empty((B, cap, 4), device=dev, dtype=torch.float32)
    mask = torch.empty((B * stride,), device=dev, dtype=torch.int64)
    _lib.call('cpr_nms_batched', _ptr(_check(boxes)), _ptr(_check(scores)), _ptr(_check(labels, torch.int32)),
              _ptr(_check(counts, torch.int32)), B, cap, float(iou_thr), _ptr(keep), _ptr(num), _ptr(order), _ptr(sboxes),
              _ptr(mask), stride, _stream())
    return keep, num


def tap_sum3x3(R, bias, J):
    """R (N,H,W,9J) tap responses -> (N,H,W,J) = bias + the nine shifted responses (see csrc/postproc.hip tap_sum3x3_kernel)."""
    N, H, W, J9 = _check(R).shape
    assert J9 == 9 * J
    out = torch.empty((N, H, W, J), device=R.device, dtype=torch.float32)
    _lib.call('cpr_tap_sum3x3', _ptr(R), _ptr(bias), _ptr(out), N, H, W, J, _stream())
    return out


def p2p_decode(reg_nhwc, point_anchor, stride, gamma, want_anchor=False):
    """reg (N,H,W,2k) -> pred (N, H*W*k, 3) = (x, y, stride) [, anchor pts]."""
    N, H, W, C2 = _check(reg_nhwc).shape
    k = C2 // 2
    pred = torch.empty((N, H * W * k, 3), device=reg_nhwc.device, dtype=torch.float32)
    anchor = torch.empty_like(pred) if want_anchor else None
    _lib.call('cpr_p2p_decode', _ptr(reg_nhwc), _ptr(_check(point_anchor)), _ptr(pred), _ptr(anchor), N, H, W, k,
              float(stride), float(gamma), _stream())
    return (pred, anchor) if want_anchor else pred


def sigmoid_exact(x):
    """Elementwise sigmoid with torch's CPU bits (the scores that order top-k / NMS candidates)."""
    x = _check(x.contiguous())
    y = torch.empty_like(x)
    _lib.call('cpr_sigmoid', _ptr(x), _ptr(y), x.numel(), _stream())
    return y


def rowmax_sigmoid(logits):
    M, C = _check(logits).shape
    out = torch.empty((M,), device=logits.device, dtype=torch.float32)
    _lib.call('cpr_rowmax_sigmoid', _ptr(logits), _ptr(out), M, C, _stream())
    return out


P2P_CLS_MODES = {'FocalLoss': 0, 'CrossEntropyLoss_sigmoid': 1, 'CrossEntropyLoss': 2}
P2P_REG_MODES = {'SmoothL1Loss': 0, 'MSELoss': 1, 'L1Loss': 2}


def p2p_loss(logits, pred, gt_inds, gt_pts, gt_labels, gt_start, alpha, gamma, beta, pos_w, neg_w, reg_norm, w_cls,
             w_reg, cls_mode=0, reg_mode=0):
    """logits (B,M,C), pred (B,M,3), gt_inds (B,M) int64 -> (B,2) {loss_cls, loss_pts} per image.  cls_mode / reg_mode:
    P2P_CLS_MODES / P2P_REG_MODES (include/cpr_hip.h, cpr_p2p_loss)."""
    B, M, C = _check(logits).shape
    nblk = (M + 255) // 256
    ws = torch.empty((B * nblk * 3,), device=logits.device, dtype=torch.float64)
    out = torch.empty((B, 2), device=logits.device, dtype=torch.float32)
    _lib.call('cpr_p2p_loss', _ptr(logits), _ptr(_check(pred)), _ptr(_check(gt_inds, torch.int64)), _ptr(_check(gt_pts)),
              _ptr(_check(gt_labels, torch.int32)), _ptr(_check(gt_start, torch.int32)), _ptr(ws), _ptr(out), B, M, C,
              float(alpha), float(gamma), float(beta), float(pos_w), float(neg_w), float(reg_norm), float(w_cls),
              float(w_reg), int(cls_mode), int(reg_mode), _stream())
    return out


def match_cost(pred, logits, gt, labels, cls_terms, reg_terms):
    """The general cost matrix of the Hungarian assigners (csrc/assign.hip, match_cost_kernel) -> cost^T (G, M).  pred (M, 2) points or
    (M, 4) xyxy boxes, gt alike; cls_terms / reg_terms: lists of (type, weight, a, b, c, d) (include/cpr_hip.h, cpr_match_cost)."""
    import ctypes
    M, G = pred.shape[0], gt.shape[0]
    assert pred.shape[1] == gt.shape[1] and pred.shape[1] in (2, 4) and len(cls_terms) <= 4 and len(reg_terms) <= 4
    costT = torch.empty((G, M), device=pred.device, dtype=torch.float32)
    flat = [float(v) for t in list(cls_terms) + list(reg_terms) for v in (tuple(t) + (0.,) * 6)[:6]]
    terms = (ctypes.c_float * max(len(flat), 1))(*flat)
    _lib.call('cpr_match_cost', _ptr(_check(pred)), pred.shape[1], _ptr(_check(logits)), logits.shape[1], _ptr(_check(gt)),
              _ptr(_check(labels, torch.int32)), _ptr(costT), M, G, terms, len(cls_terms), len(reg_terms), _stream())
    return costT


# ------------------------------------------------------------------------------------------------ backward / optimizer
def dgrad_pack(weight, stride, padding, scale=None, dtype=torch.float32):
    """PackedConv that computes the data gradient of ``conv2d(x, weight, stride, padding)`` as a stride-1 forward conv
    over dy (zero-inserted first when stride > 1): channels swapped, taps flipped, padding K-1-p; ``scale`` (Cout,) is
    the forward conv's folded-BatchNorm scale, multiplied into the weights.  Stride-2 convs with k in {1, 3} and
    padding k//2 (every strided conv of the ResNet body) get the phase-decomposed form (PhasedDgrad)."""
    if stride == 2 and weight.shape[2] == weight.shape[3] and weight.shape[2] in (1, 3) and padding == weight.shape[2] // 2 \
            and _PHASED[0]:
        return PhasedDgrad(weight, stride, padding, scale, dtype)
    assert dtype == torch.float32, 'bf16: PackedConv.for_dgrad_bf16 (stride 1) or the phase-decomposed form (stride 2)'
    return PackedConv.for_dgrad(weight, padding, scale)


_PHASED = [True]     # test hook: False forces the zero-insertion form for strided convs


class PhasedDgrad:
    """Data gradient of a stride-2 conv (k in {1, 3}, padding k//2) without zero insertion: one stride-1 sub-convolution
    over dy per output-parity class (py, px) using only the taps that can reach that class (1+2+2+4 = 9 of the 9 taps
    instead of 4 x 9 over a dilated gradient), scattered to the strided positions."""

    def __init__(self, weight, stride, padding, scale=None, dtype=torch.float32):
        Cout, Cin, KH, KW = weight.shape
        assert stride == 2 and KH == KW and KH in (1, 3) and padding == KH // 2 and weight.is_cuda
        self.Cin, self.stride, self.classes, self.dtype = Cin, stride, [], dtype      # bf16 (round 6): the sub-convolutions on the bf16 pipe, fp32 out
        for py in range(2):
            khs = [kh for kh in range(KH) if (py + padding - kh) % 2 == 0]
            for px in range(2):
                kws = [kw for kw in range(KW) if (px + padding - kw) % 2 == 0]
                if not khs or not kws:
                    continue                       # no tap reaches this class (1x1: only (0, 0))
                # dy offsets (py+p-kh)/2 form a contiguous range [dmin, dmax]; a symmetric-pad conv has offsets t-P
                dh = [(py + padding - kh) // 2 for kh in khs]
                dw = [(px + padding - kw) // 2 for kw in kws]
                P = max(0, max(dh), max(dw), -min(dh), -min(dw))
                sub = weight.detach()[:, :, khs][:, :, :, kws].contiguous()    # ascending kh/kw; for_dgrad flips the taps
                pc = PackedConv.for_dgrad(sub, 0, scale, pad_override=P) if dtype == torch.float32 else \
                    PackedConv.for_dgrad_bf16(sub, 0, scale, pad_override=P)
                # flipped tap t <-> descending kh <-> offset dmin + t, conv offset t - P  =>  out row i' = i + dmin + P
                self.classes.append((py, px, pc, min(dh) + P, min(dw) + P))

    def __call__(self, dy, in_hw, add=None):
        N = dy.shape[0]
        H, W = in_hw
        dx = add.clone() if add is not None else torch.zeros((N, H, W, self.Cin), device=dy.device, dtype=torch.float32)
        assert dy.dtype == self.dtype
        for py, px, pc, sh, sw in self.classes:
            o = conv2d(dy, pc, out_dtype=torch.float32)
            _lib.call('cpr_phase_scatter_add', _ptr(o), _ptr(dx), N, o.shape[1], o.shape[2], self.Cin, H, W, py, px, sh, sw,
                      self.stride, _stream())
        return dx


def conv2d_dgrad(dy, pc_t, in_hw, stride=1, mask=None, add=None, colsum=False):
    """dx (N,H,W,Cin) of a conv whose transposed/flipped weights are ``pc_t`` (dgrad_pack).
    mask: the forward's post-ReLU input -- the result is the gradient BEFORE that ReLU (dx * (mask > 0)), fused into
    the epilogue; add: another gradient of the same tensor summed in the epilogue (before the mask when both are given); colsum: also return the
    per-channel sums of the result as TilePartials (-> (dx, partials); ``partials.reduce()`` gives the (C,) vector)."""
    N, OH, OW, Cout = _check(dy).shape
    H, W = in_hw
    if mask is not None and add is not None:
        # the conv epilogue has one extra operand (the sum OR the mask source): the sum rides in the epilogue, mask and column sums
        # are one streaming pass over the result
        g, cs = relu_bwd_colsum(conv2d_dgrad(dy, pc_t, in_hw, stride, add=add), mask)
        return (g, cs) if colsum else g
    if isinstance(pc_t, PhasedDgrad):
        dx = pc_t(dy, in_hw, add=add)
        if mask is not None or colsum:
            g, cs = relu_bwd_colsum(dx, mask, want_g=mask is not None)
            dx = g if mask is not None else dx
            return (dx, cs) if colsum else dx
        return dx
    if stride > 1:
        # dilated gradient of extent (H + 2p - K + 1): rows/cols past (O-1)*s are the zeros the forward never read
        He, We = H + 2 * (pc_t.KH - 1 - pc_t.padding) - pc_t.KH + 1, W + 2 * (pc_t.KW - 1 - pc_t.padding) - pc_t.KW + 1
        z = torch.empty((N, He, We, Cout), device=dy.device, dtype=torch.float32)
        _lib.call('cpr_zero_insert', _ptr(dy), _ptr(z), N, OH, OW, Cout, He, We, stride, _stream())
        dy = z
    res = mask if mask is not None else add
    out = conv2d(dy, pc_t, residual=res, res_mask=mask is not None, colsum=colsum)
    o = out[0] if colsum else out
    assert o.shape[1] == H and o.shape[2] == W, (o.shape, in_hw)
    return out


def wgrad_tn(on=None):
    """The kernel choice of the bf16 weight gradient for bf16 maps (include/cpr_hip.h, cpr_wgrad_bf16_set_tn): True = the pixel-major
    kernel (csrc/conv_wgrad_bf16_tn.hip, the default), False = channel-major rewrites + NT GEMM.  Returns the value before the
    call; on=None only queries."""
    return bool(_lib.call('cpr_wgrad_bf16_set_tn', -1 if on is None else int(bool(on)), positive=True))      # (the previous value)


def conv_wgrad_bf16_supported(x_shape, weight_shape, stride, padding, maps_bf16=False):
    """Shapes the bf16 weight gradient takes: stride 1, 1x1 or 3x3 / padding 1, Cout % 64 == 0 and -- maps_bf16 (both maps are bf16:
    the pixel-major kernel, csrc/conv_wgrad_bf16_tn.hip) Cin % 64 == 0, else (csrc/conv_wgrad_bf16.hip: channel-major rewrites + NT
    GEMM) Cin % 256 == 0 and, for 1x1 layers, Cout >= 256."""
    Cout, Cin, KH, KW = weight_shape
    tn = maps_bf16 and wgrad_tn()
    if KH == 1 and Cout < 256 and not tn:      # measured (tools/wgrad_bf16_bench.py): the two rewrites cost what the bf16 GEMM saves
        return False
    if not (stride == 1 or (tn and stride == 2)):      # the strided 3x3 / projection layers: the pixel-major kernel only
        return False
    return KH == KW and KH in (1, 3) and padding == KH // 2 and Cin % (64 if tn else 256) == 0 and Cout % 64 == 0 and \
        _lib.call('cpr_conv_wgrad_bf16_workspace_s', x_shape[0], x_shape[1], x_shape[2], Cin, Cout, KH, stride, positive=True) > 0


def conv_wgrad_bf16(dy, x, weight_shape, out=None, accumulate=False, stride=1):
    """Weight gradient of a conv (1x1 or 3x3 / padding 1) on the bf16 matrix cores (mixed-precision training step): dy (N,OH,OW,Cout)
    and x (N,H,W,Cin) bf16 or fp32 (rounded to bf16 on the way in) -> [Cout][Cin][k][k] fp32.  stride 2 needs both maps in bf16."""
    N, H, W, Cin = x.shape
    Cout, Cin_w, KH, KW = weight_shape
    OH, OW = (H + 2 * (KH // 2) - KH) // stride + 1, (W + 2 * (KH // 2) - KH) // stride + 1
    assert Cin_w == Cin and tuple(dy.shape) == (N, OH, OW, Cout) and KH == KW, (tuple(dy.shape), (N, OH, OW, Cout))
    assert x.dtype in (torch.float32, torch.bfloat16) and dy.dtype in (torch.float32, torch.bfloat16)
    assert x.is_contiguous() and dy.is_contiguous()
    units = _lib.call('cpr_conv_wgrad_bf16_workspace_s', N, H, W, Cin, Cout, KH, stride, positive=True)
    ws = torch.empty((units * 256,), device=x.device, dtype=torch.uint8)
    if out is None:
        assert not accumulate
        out = torch.empty(tuple(weight_shape), device=x.device, dtype=torch.float32)
    assert tuple(out.shape) == tuple(weight_shape) and out.is_contiguous() and out.dtype == torch.float32
    _lib.call('cpr_conv_wgrad_bf16_s', _ptr(dy), int(dy.dtype == torch.bfloat16), _ptr(x), int(x.dtype == torch.bfloat16), _ptr(out),
              _ptr(ws), N, H, W, Cin, Cout, KH, stride, int(accumulate), _stream())
    return out


def conv3x3_wino_wgrad(dy, x, weight_shape, in_ab=None, in_relu=False, grad=None, out=None):
    """Weight gradient of a 3x3 / stride 1 / pad 1 conv as fused Winograd F(2x2,3x3) (csrc/conv_wino_wgrad.hip: 16 GEMMs over
    the tiles, 2.25x fewer multiplies).  Same arguments as conv2d_wgrad; Cin % 64 == 0, Cout % 64 == 0."""
    N, H, W, Cin = _check(x).shape
    Cout = _check(dy).shape[-1]
    assert tuple(weight_shape) == (Cout, Cin, 3, 3) and tuple(dy.shape[:3]) == (N, H, W)
    n = _lib.call('cpr_conv3x3_wino_wgrad_workspace', N, H, W, Cin, Cout, positive=True)
    ws = torch.empty((n,), device=x.device, dtype=torch.float32)
    acc = grad is not None
    if grad is None:
        grad = out if out is not None else torch.empty(tuple(weight_shape), device=x.device, dtype=torch.float32)
    assert tuple(grad.shape) == tuple(weight_shape) and grad.is_contiguous()
    a, b = in_ab if in_ab is not None else (None, None)
    _lib.call('cpr_conv3x3_wino_wgrad', _ptr(dy), _ptr(x), _ptr(a), _ptr(b), _ptr(grad), _ptr(ws), N, H, W, Cin, Cout,
              int(in_relu), int(acc), _stream())
    return grad


def conv2d_wgrad(dy, x, weight_shape, stride, padding, in_ab=None, in_relu=False, grad=None, out=None):
    """grad_w [Cout][Cin][KH][KW]: accumulated into ``grad`` when given, written into ``out`` when given, else a new
    tensor.  in_ab: fused GroupNorm affine (+ReLU) of the input."""
    N, H, W, Cin = _check(x).shape
    _, OH, OW, Cout = _check(dy).shape
    KH, KW = weight_shape[2], weight_shape[3]
    assert weight_shape[0] == Cout and weight_shape[1] == Cin, (weight_shape, Cout, Cin)
    if WINOGRAD[0] and KH == 3 and KW == 3 and stride == 1 and padding == 1 and Cin % 64 == 0 and Cout % 64 == 0 and \
            x.dtype == torch.float32 and W / float((W + 15) // 16 * 16) >= WINO_MIN_FILL and H * W >= 1024 and \
            (in_ab is None or Cin <= 512):     # (20x20 maps: too few tiles per K slice against 16 frequencies of partials)
        return conv3x3_wino_wgrad(dy, x, weight_shape, in_ab, in_relu, grad, out)
    n = _lib.call('cpr_conv2d_wgrad_workspace', N, OH, OW, Cin, Cout, KH, KW, positive=True)
    ws = torch.empty((n,), device=x.device, dtype=torch.float32)
    acc = grad is not None
    if grad is None:
        grad = out if out is not None else torch.empty(tuple(weight_shape), device=x.device, dtype=torch.float32)
    assert tuple(grad.shape) == tuple(weight_shape) and grad.is_contiguous()
    a = b = None
    if in_ab is not None:
        a, b = in_ab
    _lib.call('cpr_conv2d_wgrad', _ptr(dy), _ptr(x), _ptr(a), _ptr(b), _ptr(grad), _ptr(ws), N, H, W, Cin, Cout, KH, KW,
              stride, padding, int(in_relu), int(acc), _stream())
    return grad


def gn_bwd(x, dz, a, b, mean, rstd, gamma, relu, dgamma=None, dbeta=None, slots=None, out_dgamma=None, out_dbeta=None,
           want16=False, want32=True):
    """GroupNorm(+ReLU) backward -> (dx, dgamma, dbeta); accumulates into dgamma/dbeta when given, writes into
    out_dgamma/out_dbeta when given.  x bf16 (the map the mixed-precision forward recorded): read as it is, and with want16 the
    bf16 rounding of dx is written by the same pass -> (dx | None, dgamma, dbeta, dx16)."""
    N, H, W, C = _check(x, ACT).shape
    HW, G = H * W, mean.shape[1]
    if slots is None:
        slots = max(1, min(256, HW // 256))
    acc = dgamma is not None
    if not acc:
        dgamma = out_dgamma if out_dgamma is not None else torch.empty((C,), device=x.device, dtype=torch.float32)
        dbeta = out_dbeta if out_dbeta is not None else torch.empty((C,), device=x.device, dtype=torch.float32)
    ws_part = torch.empty((N * slots * C * 2,), device=x.device, dtype=torch.float32)
    ws_k = torch.empty((2 * N * G + 2 * N * C,), device=x.device, dtype=torch.float32)
    if x.dtype == torch.bfloat16:
        assert want16 or want32
        dx = torch.empty(tuple(x.shape), device=x.device, dtype=torch.float32) if want32 else None
        dx16 = torch.empty_like(x) if want16 else None
        assert dz.dtype in (torch.float32, torch.bfloat16) and dz.is_contiguous()
        _lib.call('cpr_gn_bwd_bf16_dz16' if dz.dtype == torch.bfloat16 else 'cpr_gn_bwd_bf16', _ptr(x), _ptr(dz), _ptr(a), _ptr(b), _ptr(mean), _ptr(rstd), _ptr(_check(gamma)),
                  _ptr(dx), _ptr(dx16), _ptr(dgamma), _ptr(dbeta), _ptr(ws_part), _ptr(ws_k), N, HW, C, G, slots, int(relu),
                  int(acc), _stream())
        return dx, dgamma, dbeta, dx16
    assert want32 and not want16
    dx = torch.empty_like(x)
    _lib.call('cpr_gn_bwd', _ptr(x), _ptr(_check(dz)), _ptr(a), _ptr(b), _ptr(mean), _ptr(rstd), _ptr(_check(gamma)),
              _ptr(dx), _ptr(dgamma), _ptr(dbeta), _ptr(ws_part), _ptr(ws_k), N, HW, C, G, slots, int(relu), int(acc),
              _stream())
    return dx, dgamma, dbeta


def upsample_add_bwd(dfine, dcoarse_or_shape, accumulate=True):
    N, H, W, C = _check(dfine).shape
    if isinstance(dcoarse_or_shape, torch.Tensor):
        dc = dcoarse_or_shape
    else:
        dc, accumulate = torch.empty(dcoarse_or_shape, device=dfine.device, dtype=torch.float32), False
    _lib.call('cpr_upsample_add_bwd', _ptr(dfine), _ptr(dc), N, H, W, dc.shape[1], dc.shape[2], C, int(accumulate),
              _stream())
    return dc


def relu_bwd_colsum(dy, y=None, want_g=True, colsum=None, want16=False, add=None):
    """g = dy*(y>0) (y None: g = dy) and per-channel column sums of g -> (g|None, colsum (C)[, g16]).  y fp32 or the bf16 map the
    mixed-precision forward recorded (read as it is); want16: also the bf16 rounding of g, written by the same pass; add (fp32, same
    shape): g = (dy + add)*(y>0) -- a shortcut gradient joining in the same pass."""
    C = dy.shape[-1]
    M = dy.numel() // C
    acc = colsum is not None
    if not acc:
        colsum = torch.empty((C,), device=dy.device, dtype=torch.float32)
    g = torch.empty_like(dy) if want_g else None
    g16 = torch.empty(tuple(dy.shape), device=dy.device, dtype=torch.bfloat16) if want16 else None
    ws = torch.empty((_lib.call('cpr_relu_bwd_colsum_ws', M, C, positive=True),), device=dy.device, dtype=torch.float32)
    if y is not None:
        assert y.dtype in (torch.float32, torch.bfloat16) and y.is_contiguous() and y.numel() == dy.numel()
    if add is not None:
        assert add.dtype == torch.float32 and add.is_contiguous() and add.numel() == dy.numel() and want_g
    _lib.call('cpr_relu_bwd_colsum', _ptr(_check(dy)), _ptr(add), _ptr(y), int(y is not None and y.dtype == torch.bfloat16), _ptr(g), _ptr(g16),
              _ptr(colsum), _ptr(ws), M, C, int(acc), _stream())
    return (g, colsum, g16) if want16 else (g, colsum)


def bn_fold_bwd(Gw, weight, scale, mean, inv_sigma, colsum_g, want_affine=True, out_dgamma=None, out_dbeta=None):
    """In place dW = scale*Gw; returns (dgamma, dbeta) of the folded eval BatchNorm (or (None, None)).  colsum_g: the (Cout,) column
    sums of the output gradient, or the TilePartials a conv epilogue left (summed inside the kernel)."""
    Cout = Gw.shape[0]
    K = Gw.numel() // Cout
    dg, db = out_dgamma, out_dbeta
    if want_affine and dg is None:
        dg = torch.empty((Cout,), device=Gw.device, dtype=torch.float32)
        db = torch.empty((Cout,), device=Gw.device, dtype=torch.float32)
    if isinstance(colsum_g, TilePartials):     # the conv epilogue's partials as they are: the kernel adds its channel's column up
        assert colsum_g.C == Cout
        _lib.call('cpr_bn_fold_bwd_part', _ptr(Gw), _ptr(_check(weight)), _ptr(scale), _ptr(mean), _ptr(inv_sigma),
                  _ptr(colsum_g.part), colsum_g.tiles, _ptr(dg), _ptr(db), Cout, K, _stream())
        return dg, db
    _lib.call('cpr_bn_fold_bwd', _ptr(Gw), _ptr(_check(weight)), _ptr(scale), _ptr(mean), _ptr(inv_sigma),
              _ptr(colsum_g), _ptr(dg), _ptr(db), Cout, K, _stream())
    return dg, db


def concurrent_stream(device, tries=8, spin_us=150.0):
    """A new torch stream that runs CONCURRENTLY with the current one, i.e. sits on another hardware queue.  The HIP runtime
    multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues, so a fresh stream shares the current stream's queue
    with probability 1 / queues (and always, once RCCL has taken the queues: pointtinybenchmark_amd/__init__.py) -- work on it
    then simply runs after the current stream's.  Probe: one idle wave of ``spin_us`` on each of the two streams (csrc/pack.hip,
    cpr_spin); together they take ~spin_us on two queues and ~2 spin_us on one.  -> (stream, concurrent: bool); after ``tries``
    shared candidates the last one is returned with concurrent = False."""
    ticks = int(spin_us * 100)                                    # 100 MHz wall clock
    s, ok = None, False
    with torch.cuda.device(device):                                # the probe's launches belong to ``device`` whatever the caller's current device is
        cur = torch.cuda.current_stream(device)
        for _ in range(tries):
            s = torch.cuda.Stream(device=device)
            _lib.call('cpr_spin', 1, s.cuda_stream)                # first use of a stream creates its queue: keep that out of the timing
            _lib.call('cpr_spin', 1, cur.cuda_stream)
            torch.cuda.synchronize(device)
            best = float('inf')
            for _rep in range(3):                                  # a shared GPU can stretch one measurement: the minimum of three decides
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cur)
                s.wait_stream(cur)                                 # both spins start after e0
                _lib.call('cpr_spin', ticks, cur.cuda_stream)
                _lib.call('cpr_spin', ticks, s.cuda_stream)
                cur.wait_stream(s)
                e1.record(cur)
                e1.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3)
                if best < 1.6 * spin_us:
                    break
            if best < 1.6 * spin_us:
                ok = True
                break
    return s, ok


def axpby(y, x, alpha=1.0, beta=1.0):
    assert y.numel() == x.numel()
    _lib.call('cpr_axpby', _ptr(_check(y)), _ptr(_check(x)), float(alpha), float(beta), y.numel(), _stream())
    return y


def cpr_loss_bwd(lmap, neg_mask, out5, bag_logits, valid, labels, bag_ws, centers, gt_img, offsets, ins_off, num_classes,
                 stride, w_mil, w_gt, w_neg, Jd, gt_weight=None, eps=1e-6, upstream=None, radius_cells=None, gather=True):
    """-> dmap (N,H,W,Jd): gradient of gt_loss + pos_loss + neg_loss wrt the logit map (channels >= J are zero).
    radius_cells: the bag radius in grid cells (every offset lies within radius_cells * stride of its centre): the taps of a
    bag are gathered in a (2 * radius_cells + 3)^2 window -- deterministic, no float atomics.  gt_img must ascend.
    upstream (5,) fp32 device tensor: gradient of the caller's total wrt the forward's (gt_loss, pos_loss, bag_acc, neg_loss,
    num_sample) vector -- the autograd bridge passes what torch hands it; None = unit weights."""
    assert upstream is None or (upstream.numel() == 5 and upstream.dtype == torch.float32 and upstream.is_contiguous())
    N, H, W, J = _check(lmap).shape
    G, K, _ = bag_logits.shape
    dmap = torch.empty((N, H, W, Jd), device=lmap.device, dtype=torch.float32)
    dbag = torch.empty((G, K, J), device=lmap.device, dtype=torch.float32)
    if gather:
        assert radius_cells is not None and radius_cells >= 0, 'the bag radius (in cells) sizes the gather window'
        win = 2 * int(math.ceil(radius_cells)) + 3      # taps of a bag span floor(c - r) .. floor(c + r) + 1
        win_ws = torch.empty((G, win, win, J), device=lmap.device, dtype=torch.float32)
        win_org = torch.empty((G, 2), device=lmap.device, dtype=torch.int32)
    else:     # the bag logits were not sampled from ``lmap`` (num_cls_fcs > 0): dmap = the negative-grid term, dbag is returned as is
        win, win_ws, win_org = 0, None, None
    _lib.call('cpr_loss_bwd', _ptr(lmap), _ptr(neg_mask), _ptr(out5), _ptr(bag_logits), _ptr(valid), _ptr(labels),
              _ptr(gt_weight), _ptr(bag_ws), _ptr(centers), _ptr(gt_img), _ptr(offsets), _ptr(dbag), _ptr(dmap), _ptr(win_ws),
              _ptr(win_org), win, N, H, W, J, Jd, ins_off, G, K, num_classes, float(stride), float(eps), float(w_mil),
              float(w_gt), float(w_neg), _ptr(upstream), _stream())
    return dmap, dbag


def cpr_loss_bwd_general(lmap, neg_mask, out5, bag_logits, valid, labels, bag_ws, bags, centres, ins_off, num_cls_out, Jd,
                         w_mil, w_gt, w_neg, gt_weight=None, eps=1e-6, upstream=None, prob_type='sigmoid', norm_p=1.0,
                         binary_ins=False, allpos=False, neg_from_gt=False):
    """The loss gradients for any CPRHead loss option (csrc/backward.hip, general form): bags / centres are ops.mil_loss's
    geometry tuples.  -> (dmap (N,H,W,Jd): the negative-grid term alone, dbag: the gradient wrt every bag entry's logits in the
    layout of ``bag_logits`` -- not gathered onto the map)."""
    assert upstream is None or (upstream.numel() == 5 and upstream.dtype == torch.float32 and upstream.is_contiguous())
    N, H, W, J = _check(lmap).shape
    E = bag_logits.numel() // J
    nb, bstride, boff, blen = bags
    coff, cstride, ccount, cmod = centres
    assert nb * bstride == E and labels.numel() == nb, (nb, bstride, E)
    dmap = torch.empty((N, H, W, Jd), device=lmap.device, dtype=torch.float32)
    dbag = torch.empty(tuple(bag_logits.shape), device=lmap.device, dtype=torch.float32)
    _lib.call('cpr_loss_bwd_general', _ptr(lmap), _ptr(neg_mask), _ptr(out5), _ptr(_check(bag_logits)), _ptr(valid), _ptr(labels),
              _ptr(gt_weight), _ptr(bag_ws), _ptr(dbag), _ptr(dmap), N, H, W, J, Jd, int(ins_off), nb, bstride, boff, blen, coff,
              max(int(cstride), 1), ccount, cmod, int(num_cls_out), float(eps), PROB_TYPES[prob_type], float(norm_p), int(binary_ins),
              int(allpos), float(w_mil), float(w_gt), float(w_neg), int(neg_from_gt), _ptr(upstream), _stream())
    return dmap, dbag


def bag_gather_bwd(dsample, centers, gt_img, offsets, dmap, stride, radius_cells):
    """dsample (G,K,J): gradient wrt the bilinear samples ``bag_sample`` took from a (N,H,W,J') map, J <= J'; ADDED onto dmap
    (N,H,W,J') through the same taps (deterministic: per-bag windows, per-image gt order).  gt_img must ascend."""
    G, K, J = _check(dsample).shape
    N, H, W, Jd = _check(dmap).shape
    win = 2 * int(math.ceil(radius_cells)) + 3
    win_ws = torch.empty((G, win, win, J), device=dmap.device, dtype=torch.float32)
    win_org = torch.empty((G, 2), device=dmap.device, dtype=torch.int32)
    _lib.call('cpr_bag_gather_bwd', _ptr(dsample), J, _ptr(centers), _ptr(gt_img), _ptr(offsets), _ptr(win_ws), _ptr(win_org), win,
              _ptr(dmap), N, H, W, Jd, G, K, float(stride), _stream())
    return dmap


def bag_points_gather_bwd(dsample, pts, code, gt_img, dmap, stride, align_corners=False, want_bias=False):
    """The gather stage from the forward's point list (grid generators, align_corners=True): dsample (G,Kt,J) ADDED onto dmap
    (N,H,W,J'), J <= J'; pts (G,Kt,2); code (G,Kt) int32 from ``grid_bag(want_cell=True)`` or None (every entry a bilinear sample).
    want_bias: also return (J,) = what the projection's bias receives through padding slots / dropped taps."""
    G, Kt, J = _check(dsample).shape
    N, H, W, Jd = _check(dmap).shape
    assert pts.numel() == G * Kt * 2 and (code is None or (code.numel() == G * Kt and code.dtype == torch.int32))
    wout = torch.empty((G * Kt,), device=dmap.device, dtype=torch.float32) if want_bias else None
    dbias = torch.empty((J,), device=dmap.device, dtype=torch.float32) if want_bias else None
    _lib.call('cpr_bag_points_gather_bwd', _ptr(dsample), J, _ptr(_check(pts)), _ptr(code), _ptr(gt_img), _ptr(dmap), _ptr(wout),
              _ptr(dbias), N, H, W, Jd, G, Kt, float(stride), int(align_corners), _stream())
    return dbias


def p2p_loss_bwd(logits, pred, gt_inds, gt_pts, gt_labels, gt_start, alpha, gamma, beta, pos_w, neg_w, reg_norm, w_cls,
                 w_reg, gamma_p, Cp, Rp, upstream=None, cls_mode=0, reg_mode=0):
    """-> (dcls (B,M,Cp), dreg (B,M,Rp)): gradient of the summed P2P losses wrt class logits / regression output.
    upstream (B,2): gradient of the caller's total wrt each image's (loss_cls, loss_pts); None = unit weights."""
    B, M, C = _check(logits).shape
    assert upstream is None or (tuple(upstream.shape) == (B, 2) and upstream.dtype == torch.float32 and upstream.is_contiguous())
    npos = (gt_inds > 0).sum().to(torch.float32).reshape(1)          # device scalar, no host sync
    dcls = torch.empty((B, M, Cp), device=logits.device, dtype=torch.float32)
    dreg = torch.empty((B, M, Rp), device=logits.device, dtype=torch.float32)
    _lib.call('cpr_p2p_loss_bwd', _ptr(logits), _ptr(_check(pred)), _ptr(gt_inds), _ptr(gt_pts), _ptr(gt_labels),
              _ptr(gt_start), _ptr(npos), _ptr(dcls), _ptr(dreg), B, M, C, Cp, Rp, float(alpha), float(gamma), float(beta),
              float(pos_w), float(neg_w), float(reg_norm), float(w_cls), float(w_reg), float(gamma_p), _ptr(upstream),
              int(cls_mode), int(reg_mode), _stream())
    return dcls, dreg


def grad_sumsq(g, out, ws, accumulate):
    _lib.call('cpr_grad_sumsq', _ptr(_check(g)), g.numel(), _ptr(ws), _ptr(out), int(accumulate), _stream())


def sgd_step(p, grad, buf, norm2, lr, momentum, weight_decay, max_norm, grad_scale, first):
    _lib.call('cpr_sgd_step', _ptr(p), _ptr(grad), _ptr(buf), _ptr(norm2), p.numel(), float(lr), float(momentum),
              float(weight_decay), float(max_norm), float(grad_scale), int(first), _stream())
