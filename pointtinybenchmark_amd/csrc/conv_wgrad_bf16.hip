// Weight gradient of a stride-1 convolution (1x1, or 3x3 with padding 1) on the bf16 matrix cores -- the mixed-precision
// training step (pointtinybenchmark_amd/training.py; reference analogue: torch autograd under mmcv's Fp16OptimizerHook,
// T/mmdet/apis/train.py:116-119).
//
// Since round 6 this file is the SECOND choice: with both maps in bf16 the entry point runs the pixel-major kernel of
// conv_wgrad_bf16_tn.hip (no rewritten operands; stride 2, Cin % 64, small 1x1 layers included) and the rewriting path below takes
// what is left -- an fp32 map among the two (rounded on the way in), or CPR_WGRAD_TN=0 / cpr_wgrad_bf16_set_tn(0).
//
//   dW[co][kh][kw][ci] = sum over pixels of dy[n, y, x, co] * x[n, y + kh - p, x + kw - p, ci]
//
// is a GEMM whose reduction runs over PIXELS, the slow dimension of both NHWC operands, while v_mfma_f32_32x32x16_bf16 wants 8
// consecutive k per lane.  Instead of transposing inside the GEMM the operands are rewritten once, channel-major over a padded
// pixel axis q = (n (H + 2p) + y + p) Wp + x + p (Wp = W + 2p rounded up to 8; border cells are zeros):
//   dyT[co][q]            (bf16, from the fp32 gradient map)
//   xT_kw[ci][q] = x at cell q + kw - p   (one copy per kw: the DMA reads 16-byte units, a one-element shift cannot be an offset)
// and every tap is a plain NT GEMM with contiguous K:  dW_tap[co][ci] = sum_q dyT[co][q] * xT_kw[ci][q + (kh - p) Wp]
// (zero padding = the zero border cells; the row shift is a multiple of 8 elements).  The GEMM is conv_bf16_dma_kernel in its
// NT mode (256 x 256 x 64 tiles, both operands by LDS-DMA), split over the pixel axis into `splits` workgroup rows whose fp32
// partials are summed -- and laid out as the [Cout][Cin][k][k] gradient -- by wgrad_bf16_reduce_kernel.
#include "common.h"
#include <type_traits>
#include <cstdlib>

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

int conv_bf16_dma_nt_launch(const void* a, const void* b, float* part, int M, int N, long long rs, int k, int pad, int Wp,
                            long long copy, int splits, int chunks, hipStream_t stream);
// conv_wgrad_bf16_tn.hip: the same gradient straight from the NHWC maps (transposing LDS reads), no rewritten operands
int wgrad_bf16_tn_launch(const void* dy, const void* x, float* part, int N, int H, int W, int Cin, int Cout, int k, int stride, int splits,
                         int chunks, hipStream_t stream);

struct WgradBf16Plan {
    int pad, Hp, Wp, G, splits, chunks, taps;
    bool nt_ok;                   // the rewriting path can take this shape (Cin % 256 == 0, row length within the descriptors' range)
    int tn_splits, tn_chunks;     // the pixel-major kernel's split of the UNPADDED pixel axis
    long long Q, Qk, rs;          // real cells, cells covered by the K loop, row length (elements, guards included)
    long long off_dy, off_x, off_part, bytes;   // workspace layout
};

static bool wgrad_bf16_plan(int N, int H, int W, int Cin, int Cout, int k, WgradBf16Plan* pl, int stride = 1) {
    if (!(k == 1 || k == 3) || N <= 0 || H <= 0 || W <= 0 || Cin % 64 != 0 || Cout % 64 != 0 || !(stride == 1 || stride == 2)) return false;
    pl->pad = k / 2;
    pl->Hp = H + 2 * pl->pad;
    pl->Wp = (W + 2 * pl->pad + 7) / 8 * 8;
    pl->G = pl->Wp + 8;                                  // guard in front of and behind the cells: |shift| <= Wp + 1
    pl->Q = (long long)N * pl->Hp * pl->Wp;
    pl->taps = k * k;
    const long long tilesMN = (long long)((Cout + 255) / 256) * ((Cin + 255) / 256);
    const long long chunks_all = (pl->Q + 63) / 64;
    // equal workgroups: as many splits (a multiple of 8: one share per XCD) as fill four rounds of 256 CUs without starting a
    // fifth, each of at least 8 chunks
    long long splits = 1024 / (8 * pl->taps * tilesMN) * 8;
    if (splits > chunks_all / 8 / 8 * 8) splits = chunks_all / 8 / 8 * 8;
    if (splits < 8) splits = 8;
    pl->splits = (int)splits;
    pl->chunks = (int)((chunks_all + splits - 1) / splits);
    pl->Qk = (long long)pl->splits * pl->chunks * 64;
    pl->rs = pl->G + pl->Qk + pl->G;
    pl->rs = (pl->rs + 255) / 256 * 256;                 // (the transposing kernel writes 256 positions per workgroup)
    pl->nt_ok = stride == 1 && Cin % 256 == 0 &&
                !(pl->rs >= (1ll << 30) || (long long)Cout * pl->rs * 2 >= (1ll << 31) || (long long)Cin * pl->rs * 2 >= (1ll << 31));
    // the pixel-major kernel (conv_wgrad_bf16_tn.hip): the same rule on the unpadded pixel count
    const int OH = (H + 2 * pl->pad - k) / stride + 1, OW = (W + 2 * pl->pad - k) / stride + 1;      // (stride 2: the strided layers of a stage's first block)
    if (OH <= 0 || OW <= 0) return false;
    const long long P = (long long)N * OH * OW, pchunks = (P + 63) / 64;
    // (CPR_WGRAD_TN_WGS: the workgroup budget of the rule, default 512 = two rounds of 256 CUs: R50 640^2 B = 64 69.2 - 69.7 ms against 70.8 at 1024 and 72.7 at 2048, the configs[4] training line under torchrun 198 img/s against 196 (768) and 195 (1024) -- fewer slabs to write and sum; profiles/round6_wgrad_tn_splits_ab.txt)
    static const long long tn_wgs = []() { const char* e = getenv("CPR_WGRAD_TN_WGS"); const long long v = e ? atoll(e) : 0; return v >= 64 ? v : 512; }();
    long long ts = tn_wgs / (8 * pl->taps * tilesMN) * 8;
    if (ts > pchunks / 8 / 8 * 8) ts = pchunks / 8 / 8 * 8;
    if (ts < 8) ts = 8;
    pl->tn_splits = (int)ts;
    pl->tn_chunks = (int)((pchunks + ts - 1) / ts);
    const bool tn_ok = P * Cout * 2 < (1ll << 31) && (long long)N * H * W * Cin * 2 < (1ll << 31);
    if (!pl->nt_ok && !tn_ok) return false;
    pl->off_dy = 0;
    pl->off_x = pl->nt_ok ? (long long)Cout * pl->rs * 2 : 0;
    pl->off_part = pl->nt_ok ? pl->off_x + (long long)k * Cin * pl->rs * 2 : 0;
    const long long slabs = pl->splits > pl->tn_splits ? pl->splits : pl->tn_splits;
    pl->bytes = pl->off_part + slabs * pl->taps * Cout * Cin * 4;
    return true;
}

// src (N,H,W,C) NHWC (fp32 or bf16) -> NC copies dst_j[c][rs] bf16 (copy j at dst + j * copy): dst_j[c][G + q] = src at cell
// (q + j - NC / 2), zero at border / guard cells.
//
// Round 6 (this kernel; the round-3 one below stays for A/B, CPR_WGRAD_T64=1): one workgroup = 256 positions x 64 channels.  A
// thread owns an 8 x 8 block -- eight 16-byte loads along the channels of eight consecutive cells (a granule: granules never
// straddle a row of the padded grid, Wp % 8 == 0) -- transposes it in registers (one v_perm_b32 per output dword) and parks the
// eight channel granules (8 cells x 2 bytes) in LDS as [channel][granule], the granule column XORed with twice the channel group
// (16 lanes = 16 distinct bank quads).  The read-out runs along the cells: 32 lanes = 512 contiguous bytes of one channel row,
// whole lines; the +-1-cell copies of the 3x3 layers are funnel shifts (v_alignbit_b32) of three neighbouring granules.  Against
// the round-3 kernel (two-byte LDS writes and reads, 64-position tiles): 16 LDS instructions per 64 elements instead of 128.
template <bool SRC_BF16, int NC>
__global__ __launch_bounds__(256) void wgrad_bf16_transpose_kernel(const void* __restrict__ src, unsigned short* __restrict__ dst,
                                                                   int N, int H, int W, int C, int pad, int Hp, int Wp, int G,
                                                                   long long rs, long long copy) {
    constexpr int HALO = NC > 1 ? 1 : 0, NG = 32 + 2 * HALO;         // granules per channel row of the tile (halo: one per side)
    constexpr int ROWB = NC > 1 ? 768 : 512;                         // bytes per channel row (a multiple of 256: the swizzle's frame)
    __shared__ __attribute__((aligned(16))) unsigned char tile[64 * ROWB];
    const long long r0 = (long long)blockIdx.x * 256;                // first position of this block (0 = start of the leading guard)
    const int c0 = blockIdx.y * 64;
    const int tid = threadIdx.x;
    const long long Q = (long long)N * Hp * Wp;
    auto swz = [](int gi, int cg) { return (gi & ~15) | ((gi ^ (2 * cg)) & 15); };
    for (int item = tid; item < NG * 8; item += 256) {
        const int cg = item & 7, gi = item >> 3;
        const long long q = r0 - G + (long long)(gi - HALO) * 8;      // first cell of the granule (a multiple of 8)
        uint4 R[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) R[r] = make_uint4(0u, 0u, 0u, 0u);
        if (q >= 0 && q < Q) {
            const int n = (int)(q / ((long long)Hp * Wp));
            const int rem = (int)(q - (long long)n * Hp * Wp);
            const int yp = rem / Wp, xp = rem - yp * Wp;
            const int y = yp - pad;
            if ((unsigned)y < (unsigned)H) {
                const size_t e0 = (((size_t)n * H + y) * W) * C + c0 + cg * 8;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int x = xp + r - pad;
                    if ((unsigned)x < (unsigned)W) {
                        const size_t e = e0 + (size_t)x * C;
                        if (SRC_BF16) R[r] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(src) + e);
                        else {
                            const f32x4 f0 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(src) + e);
                            const f32x4 f1 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(src) + e + 4);
                            R[r].x = __builtin_bit_cast(unsigned, bf16x2{(__bf16)f0.x, (__bf16)f0.y});
                            R[r].y = __builtin_bit_cast(unsigned, bf16x2{(__bf16)f0.z, (__bf16)f0.w});
                            R[r].z = __builtin_bit_cast(unsigned, bf16x2{(__bf16)f1.x, (__bf16)f1.y});
                            R[r].w = __builtin_bit_cast(unsigned, bf16x2{(__bf16)f1.z, (__bf16)f1.w});
                        }
                    }
                }
            }
        }
        // channel e of the block: dword k = (cell 2k, cell 2k + 1) = the low (e even) or high halves of dword e / 2 of rows 2k, 2k + 1
        const unsigned* Rw = reinterpret_cast<const unsigned*>(R);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            uint4 o;
            unsigned* ow = reinterpret_cast<unsigned*>(&o);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                ow[k] = __builtin_amdgcn_perm(Rw[(2 * k + 1) * 4 + (e >> 1)], Rw[(2 * k) * 4 + (e >> 1)], (e & 1) ? 0x07060302u : 0x05040100u);
            *reinterpret_cast<uint4*>(tile + (cg * 8 + e) * ROWB + swz(gi, cg) * 16) = o;
        }
    }
    __syncthreads();
    const int og = tid & 31;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int ch = (tid >> 5) + 8 * i;
        const unsigned char* row = tile + ch * ROWB;
        const uint4 cur = *reinterpret_cast<const uint4*>(row + swz(og + HALO, ch >> 3) * 16);
        unsigned short* d = dst + (size_t)(c0 + ch) * rs + r0 + og * 8;
        if (NC == 1) {
            *reinterpret_cast<uint4*>(d) = cur;
        } else {
            const uint4 prv = *reinterpret_cast<const uint4*>(row + swz(og, ch >> 3) * 16);
            const uint4 nxt = *reinterpret_cast<const uint4*>(row + swz(og + 2, ch >> 3) * 16);
            // copy 0 at position i shows cell i - 1, copy 2 cell i + 1: 16-bit funnel shifts over the neighbouring granules
            const unsigned m01 = __builtin_amdgcn_alignbit(cur.y, cur.x, 16), m12 = __builtin_amdgcn_alignbit(cur.z, cur.y, 16),
                           m23 = __builtin_amdgcn_alignbit(cur.w, cur.z, 16);
            const uint4 lo = make_uint4(__builtin_amdgcn_alignbit(cur.x, prv.w, 16), m01, m12, m23);
            const uint4 hi = make_uint4(m01, m12, m23, __builtin_amdgcn_alignbit(nxt.x, cur.w, 16));
            *reinterpret_cast<uint4*>(d) = lo;
            *reinterpret_cast<uint4*>(d + copy) = cur;
            *reinterpret_cast<uint4*>(d + 2 * copy) = hi;
        }
    }
}

// The round-3 form of the kernel above (one workgroup = 64 positions x 64 channels through a two-byte-granular LDS tile): kept for
// A/B (CPR_WGRAD_T64=1) and as the second opinion of the tests.
template <bool SRC_BF16, int NC>
__global__ __launch_bounds__(256) void wgrad_bf16_transpose64_kernel(const void* __restrict__ src, unsigned short* __restrict__ dst,
                                                                   int N, int H, int W, int C, int pad, int Hp, int Wp, int G,
                                                                   long long rs, long long copy) {
    constexpr int ROWS = 64 + NC - 1;
    __shared__ unsigned short tile[ROWS][72];    // [cell][channel], rows padded against bank conflicts of the column reads
    const long long r0 = (long long)blockIdx.x * 64;     // first row position of this block (0 = start of the leading guard)
    const int c0 = blockIdx.y * 64;
    const int tid = threadIdx.x;
    {
        const int c4 = (tid & 15) * 4;
#pragma unroll
        for (int j = 0; j < (ROWS + 15) / 16; ++j) {
            const int t = (tid >> 4) + 16 * j;
            if (t >= ROWS) break;
            const int q = (int)(r0 + t) - G - NC / 2;           // padded cell of tile row t (the plan keeps rs < 2^30)
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (q >= 0 && q < N * Hp * Wp) {
                const int n = q / (Hp * Wp);
                const int rem = q - n * Hp * Wp;
                const int yp = rem / Wp, xp = rem - yp * Wp;
                const int y = yp - pad, x = xp - pad;
                if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
                    const size_t e = (((size_t)n * H + y) * W + x) * C + c0 + c4;
                    if (SRC_BF16) {
                        const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(src) + e);
                        v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
                        v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
                    } else {
                        const f32x4 f = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(src) + e);
                        v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[t][c4 + e] = __builtin_bit_cast(unsigned short, (__bf16)v[e]);
        }
    }
    __syncthreads();
    {
        const int q8 = (tid & 7) * 8;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ch = (tid >> 3) + 32 * j;
#pragma unroll
            for (int cp = 0; cp < NC; ++cp) {       // copy cp at position i shows tile row i + cp
                unsigned short o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = tile[q8 + e + cp][ch];
                uint4 w;
                w.x = o[0] | ((unsigned)o[1] << 16); w.y = o[2] | ((unsigned)o[3] << 16);
                w.z = o[4] | ((unsigned)o[5] << 16); w.w = o[6] | ((unsigned)o[7] << 16);
                *reinterpret_cast<uint4*>(dst + (size_t)cp * copy + (size_t)(c0 + ch) * rs + r0 + q8) = w;
            }
        }
    }
}

// part [splits][taps][Cout][Cin] fp32 -> grad [Cout][Cin][k][k] (the nn.Conv2d weight layout), summed over the splits: a block sums
// 256 consecutive elements (64 threads x 4), its four 64-thread rows take every fourth split.  Round 6: 16-byte loads, four splits
// requested before the first add (the round-3 form read 4 bytes per thread, one split at a time: 2.1 TB/s on the 64-slab layers of
// configs[4]) -- the additions and their order are the old kernel's, so the gradients are bit for bit the same.
__global__ __launch_bounds__(256) void wgrad_bf16_reduce_kernel(const float* __restrict__ part, float* __restrict__ grad, int splits,
                                                                int taps, int Cout, int Cin, int accumulate) {
    __shared__ f32x4 red[4][64];
    const long long per = (long long)Cout * Cin;                              // per % 256 == 0 (Cin % 256 == 0)
    const long long i = ((long long)blockIdx.x * 64 + (threadIdx.x & 63)) * 4;  // (tap, co, ci), ci fastest: four consecutive ci
    const int row = threadIdx.x >> 6;
    const int tap = (int)(i / per);
    const long long r = i - (long long)tap * per;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i < per * taps) {
        const float* src = part + (long long)tap * per + r;
        const long long step = (long long)taps * per;
        int sp = row;
        for (; sp + 12 < splits; sp += 16) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(src + sp * step), b = *reinterpret_cast<const f32x4*>(src + (sp + 4) * step);
            const f32x4 c = *reinterpret_cast<const f32x4*>(src + (sp + 8) * step), d = *reinterpret_cast<const f32x4*>(src + (sp + 12) * step);
            s += a; s += b; s += c; s += d;
        }
        for (; sp < splits; sp += 4) s += *reinterpret_cast<const f32x4*>(src + sp * step);
    }
    red[row][threadIdx.x & 63] = s;
    __syncthreads();
    if (row == 0 && i < per * taps) {
        const int t = threadIdx.x;
        s = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
        if (taps == 1) {
            f32x4* g = reinterpret_cast<f32x4*>(grad + r);
            *g = accumulate ? *g + s : s;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float* g = grad + (r + e) * taps + tap;
                *g = accumulate ? *g + s[e] : s[e];
            }
        }
    }
}

// which kernel takes the bf16 weight gradient when both maps are bf16: 0 = channel-major rewrites + NT GEMM, 1 = the pixel-major kernel
// (conv_wgrad_bf16_tn.hip).  Initial value: CPR_WGRAD_TN (default 1); cpr_wgrad_bf16_set_tn switches at run time (tests, A/B tools).
static int& wgrad_bf16_tn_mode() {
    static int mode = []() { const char* e = getenv("CPR_WGRAD_TN"); return (e && e[0] == '0') ? 0 : 1; }();
    return mode;
}
extern "C" int cpr_wgrad_bf16_set_tn(int on) {
    const int old = wgrad_bf16_tn_mode();
    if (on >= 0) wgrad_bf16_tn_mode() = on != 0;
    return old;
}

// workspace size in units of 256 bytes (the byte count of a B=64 head layer does not fit the int every entry point returns)
extern "C" int cpr_conv_wgrad_bf16_workspace_s(int N, int H, int W, int Cin, int Cout, int k, int stride) {
    WgradBf16Plan pl;
    if (!wgrad_bf16_plan(N, H, W, Cin, Cout, k, &pl, stride)) return CPR_ERR_UNSUPPORTED;
    const long long units = (pl.bytes + 255) / 256;
    return units < (1ll << 31) ? (int)units : CPR_ERR_UNSUPPORTED;
}
extern "C" int cpr_conv_wgrad_bf16_workspace(int N, int H, int W, int Cin, int Cout, int k) {
    return cpr_conv_wgrad_bf16_workspace_s(N, H, W, Cin, Cout, k, 1);
}

// dy (N,H,W,Cout) fp32 or bf16 (dy_bf16); x (N,H,W,Cin) fp32 or bf16 (x_bf16); grad [Cout][Cin][k][k] fp32 (accumulate: += ); ws: workspace of
// cpr_conv_wgrad_bf16_workspace x 256 bytes, 256-byte aligned.  k in {1, 3}, stride 1, padding k / 2, Cin % 256 == 0, Cout % 64 == 0.
// _s (round 6): + stride (1 or 2; dy is then (N,OH,OW,Cout)).  Stride 2, Cin % 256 != 0 and 1x1 layers are the pixel-major kernel's
// alone: both maps must be bf16 (else CPR_ERR_UNSUPPORTED).
extern "C" int cpr_conv_wgrad_bf16_s(const void* dy, int dy_bf16, const void* x, int x_bf16, float* grad, void* ws, int N, int H, int W,
                                     int Cin, int Cout, int k, int stride, int accumulate, hipStream_t stream);
extern "C" int cpr_conv_wgrad_bf16(const void* dy, int dy_bf16, const void* x, int x_bf16, float* grad, void* ws, int N, int H, int W,
                                   int Cin, int Cout, int k, int accumulate, hipStream_t stream) {
    return cpr_conv_wgrad_bf16_s(dy, dy_bf16, x, x_bf16, grad, ws, N, H, W, Cin, Cout, k, 1, accumulate, stream);
}
extern "C" int cpr_conv_wgrad_bf16_s(const void* dy, int dy_bf16, const void* x, int x_bf16, float* grad, void* ws, int N, int H, int W,
                                     int Cin, int Cout, int k, int stride, int accumulate, hipStream_t stream) {
    CPR_CHECK_ARG(dy && x && grad && ws);
    WgradBf16Plan pl;
    if (!wgrad_bf16_plan(N, H, W, Cin, Cout, k, &pl, stride)) return CPR_ERR_UNSUPPORTED;
    unsigned short* dyT = reinterpret_cast<unsigned short*>((char*)ws + pl.off_dy);
    unsigned short* xT = reinterpret_cast<unsigned short*>((char*)ws + pl.off_x);
    float* part = reinterpret_cast<float*>((char*)ws + pl.off_part);
    // both maps bf16 -> the pixel-major kernel (no dyT / xT rewrites) unless switched off (CPR_WGRAD_TN=0 / cpr_wgrad_bf16_set_tn(0))
    if (wgrad_bf16_tn_mode() != 0 && dy_bf16 && x_bf16) {
        const int rc = wgrad_bf16_tn_launch(dy, x, part, N, H, W, Cin, Cout, k, stride, pl.tn_splits, pl.tn_chunks, stream);
        if (rc == CPR_OK) {
            const long long n = (long long)pl.taps * Cout * Cin;
            hipLaunchKernelGGL(wgrad_bf16_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, part, grad, pl.tn_splits,
                               pl.taps, Cout, Cin, accumulate);
            CPR_LAUNCH_STATUS();
        }
        if (rc != CPR_ERR_UNSUPPORTED) return rc;
    }
    if (!pl.nt_ok) return CPR_ERR_UNSUPPORTED;
    const long long copy = (long long)Cin * pl.rs;
    static const bool t64 = []() { const char* e = getenv("CPR_WGRAD_T64"); return e && e[0] == '1'; }();     // A/B: the round-3 kernel
    auto rewrite = [&](auto bf, auto nc, const void* src, unsigned short* dst, int C, long long cp) {
        constexpr bool B = decltype(bf)::value;
        constexpr int NC = decltype(nc)::value;
        if (t64) hipLaunchKernelGGL((wgrad_bf16_transpose64_kernel<B, NC>), dim3((unsigned)(pl.rs / 64), C / 64), dim3(256), 0, stream, src, dst,
                                    N, H, W, C, pl.pad, pl.Hp, pl.Wp, pl.G, pl.rs, cp);
        else hipLaunchKernelGGL((wgrad_bf16_transpose_kernel<B, NC>), dim3((unsigned)(pl.rs / 256), C / 64), dim3(256), 0, stream, src, dst,
                                N, H, W, C, pl.pad, pl.Hp, pl.Wp, pl.G, pl.rs, cp);
    };
    using std::integral_constant;
    if (dy_bf16) rewrite(integral_constant<bool, true>{}, integral_constant<int, 1>{}, dy, dyT, Cout, 0ll);
    else rewrite(integral_constant<bool, false>{}, integral_constant<int, 1>{}, dy, dyT, Cout, 0ll);
    if (k == 3) {
        if (x_bf16) rewrite(integral_constant<bool, true>{}, integral_constant<int, 3>{}, x, xT, Cin, copy);
        else rewrite(integral_constant<bool, false>{}, integral_constant<int, 3>{}, x, xT, Cin, copy);
    } else {
        if (x_bf16) rewrite(integral_constant<bool, true>{}, integral_constant<int, 1>{}, x, xT, Cin, copy);
        else rewrite(integral_constant<bool, false>{}, integral_constant<int, 1>{}, x, xT, Cin, copy);
    }
    const int rc = conv_bf16_dma_nt_launch(dyT + pl.G, xT + pl.G, part, Cout, Cin, pl.rs, k, pl.pad, pl.Wp, copy, pl.splits, pl.chunks,
                                           stream);
    if (rc != CPR_OK) return rc;
    const long long n = (long long)pl.taps * Cout * Cin;
    hipLaunchKernelGGL(wgrad_bf16_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, part, grad, pl.splits,
                       pl.taps, Cout, Cin, accumulate);
    CPR_LAUNCH_STATUS();
}
