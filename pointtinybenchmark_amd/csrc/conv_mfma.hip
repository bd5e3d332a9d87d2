// Implicit-GEMM convolution on the CDNA4 matrix cores, exact fp32 (v_mfma_f32_32x32x2_f32).
//
// Replaces the cuDNN/ATen conv2d the reference reaches through torch for every conv on the hot path
// (T/mmdet/models/backbones/resnet.py:630-645, T/mmdet/models/necks/fpn.py:172-194,
//  T/mmdet/models/point/dense_heads/cpr_head.py:1033-1043, .../p2p_head.py:113-123).
//
// Layout (chosen for MI355X, not inherited from the reference's NCHW):
//   activations  NHWC fp32          -> the GEMM K dimension (kh, kw, cin) is contiguous per tap, so one
//                                      128-byte line = 32 input channels of one pixel = one row of a K-chunk
//   weights      [Cout][KH][KW][Cin] fp32, row stride Kpad (multiple of 32, zero padded)
//   GEMM         D[m = (n,oy,ox)][c = cout] = sum_k A[m][k] * Wt[c][k],  M = N*OH*OW
//
// Block = 256 threads = 4 waves (2x2), tile 128(M) x BN(cout) x 32(K) per step, double-buffered LDS,
// register-staged global loads (the A operand needs per-pixel zero padding and, optionally, the fused
// GroupNorm-apply+ReLU of the producing layer, so it cannot be a raw LDS-DMA copy).
// LDS rows are padded to 36 floats: ds_read_b128 of 32 consecutive rows at one k-offset is conflict free
// (slot = 9*row mod 16 is a bijection over each 16-lane service group).
// MFMA operand mapping (one f32 per lane): A[i = lane&31][k = lane>>5], B[k = lane>>5][j = lane&31];
// a lane reads 4 consecutive k (one ds_read_b128) and feeds 4 MFMAs: lanes <32 cover k0..k0+3 and lanes
// >=32 cover k0+4..k0+7 of every 8-wide k group -- the reduction order inside a K-chunk is permuted,
// which is immaterial for the fp32 sum.
// Epilogue (fused): y = acc*scale[c] + bias[c] (+ residual[m][c]) (ReLU)  -- eval-mode BatchNorm folded
// to a per-channel affine, the bottleneck shortcut add and the activation never touch HBM separately.
// Optional fused GroupNorm statistics: per (image, group) sum / sum-of-squares partials of the raw conv
// output (one slot per M-tile, reduced by gn_finalize -- deterministic, no float atomics).
#include <type_traits>
#include "common.h"

struct ConvParams {
    const float* in;
    const float* wgt;
    float* out;
    const float* scale;     // [Cout] or null
    const float* bias;      // [Cout] or null
    const float* residual;  // [M][Cout] or null
    const float* in_a;      // [N][Cin] or null: input transform x*a+b then ReLU (fused GN apply of the producer)
    const float* in_b;
    float* gn_part;         // [tilesM][Cout][2] per-M-tile per-channel (sum, sumsq) of the output, or null
    int N, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad, Kpad, M, relu, in_relu, out_bf16;
    int res_mask;           // residual is a ReLU mask source: out = residual > 0 ? v : 0 (backward of a fused ReLU)
    int tilesM, tilesN;
    // DUAL instances: a second 1x1 / unpadded GEMM source accumulated beside the first one, out = epi1(acc) + epi2(acc2):
    // the bottleneck's projection shortcut (T/mmdet/models/backbones/resnet.py:262-302: out = bn3(conv3(o2)) + bn_d(conv_d(x)))
    // in the launch of conv3 -- the 4x-wide identity map is never written to or read back from HBM.
    const float* in2;       // (N, H2, W2, Cin2); the output pixel (n, oy, ox) reads (n, oy * stride2, ox * stride2)
    const float* wgt2;      // [Cout][Kpad2]
    const float* scale2;    // [Cout] or null
    const float* bias2;     // [Cout] or null
    int H2, W2, Cin2, stride2, Kpad2;
};

constexpr int BK = 32;
constexpr int LDSW = 36;  // padded row (floats)

// BM x BN output tile (BM, BN in {64, 128}); MODE 0: Cin % 32 == 0, MODE 1: Cin == 4 (stem, one tap per float4);
// XF: fused per-(image, channel) affine (+ReLU) on the input = GroupNorm-apply of the producing layer.
// ABL (benchmark-only ablations of the pipelined loop, results are then WRONG): bit0 = no global loads / LDS writes,
// bit1 = no fragment reads, bit2 = no barrier.  ABL = 0 in every product launch.
template <int BM, int BN, int MODE, bool XF, int PIPE, int ABL = 0, bool DUAL = false>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(ConvParams p) {
    static_assert(!DUAL || (MODE == 0 && !XF && PIPE == 1 && ABL == 0), "the dual-source form exists for the plain pipelined instances");
    constexpr int WM = BM / 2;
    constexpr int WN = BN / 2;
    constexpr int MI = WM / 32;
    constexpr int NI = WN / 32;
    constexpr int AL = BM * 8 / 256;  // float4 A loads per thread
    constexpr int BL = BN * 8 / 256;  // float4 B loads per thread
    constexpr int XFMAX = 512;  // fused-GN launches keep the image's (a, b) table in LDS (Cin <= 512)
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LDSW + (XF ? 2 * XFMAX : 0)];
    float* As = smem;
    float* Bs = smem + 2 * BM * LDSW;
    float* ABs = smem + 2 * (BM + BN) * LDSW;  // [a(Cin) | b(Cin)] of this tile's image (XF only)

    // XCD-aware tile order: block b runs on XCD b%8; give each XCD a contiguous run of tiles (cout tiles
    // fastest) so the activations' 3x3 halo rows and the two cout tiles of one pixel tile share one L2.
    const int T = p.tilesM * p.tilesN;
    const int per = (T + 7) >> 3;
    const int tile = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (tile >= T) return;
    const int tm = tile / p.tilesN, tn = tile - tm * p.tilesN;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c4 = tid & 7, r0 = tid >> 3;

    // ---- per-thread A row descriptors (AL rows: r0 + 32 j)
    int iy0[AL], ix0[AL], nimg[AL], rowoff[AL];
    bool mok[AL];
    const int ohw = p.OH * p.OW;
    // 1x1 / stride 1 / unpadded launches are plain GEMMs: a row of A is pixel m itself, no (n, oy, ox) decomposition (two
    // integer divisions per row -- a tenth of the instructions of a short-K tile, which is issue-bound, not MFMA-bound)
    const bool gemm = (p.KH == 1) & (p.KW == 1) & (p.stride == 1) & (p.pad == 0) & (MODE == 0);
#pragma unroll
    for (int j = 0; j < AL; ++j) {
        int m = m0 + r0 + 32 * j;
        mok[j] = m < p.M;
        int mm = mok[j] ? m : 0;
        if (gemm) {       // wave-uniform branch
            nimg[j] = 0; iy0[j] = 0; ix0[j] = 0;
            rowoff[j] = mm * p.Cin + c4 * 4;
        } else {
            int n = mm / ohw;
            int rem = mm - n * ohw;
            int oy = rem / p.OW, ox = rem - oy * p.OW;
            nimg[j] = n;
            iy0[j] = oy * p.stride - p.pad;
            ix0[j] = ox * p.stride - p.pad;
            // 32-bit element offset of (n, iy0, ix0, c4*4); may be "negative" for padded rows/cols -- only used when in range
            rowoff[j] = ((n * p.H + iy0[j]) * p.W + ix0[j]) * p.Cin + c4 * 4;
        }
    }
    // Buffer descriptors: out-of-range offsets return 0, which IS the conv zero padding (no clamps, no selects)
    __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.in), 0, (int)((size_t)p.N * p.H * p.W * p.Cin * 4), 0x00020000);
    __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.wgt), 0, (int)((size_t)p.Cout * p.Kpad * 4), 0x00020000);
    int cin_cur = p.Cin;         // channels per tap of the source being read (DUAL: Cin2 in the second pass)
    bool second = false;         // DUAL: the second source is being read (one tap: its row offsets never change)
    // ---- B row pointers
    const float* wrow[BL];
    int woff[BL];  // byte offset of this thread's float4 in weight row j (-1 = row beyond Cout -> reads 0)
    bool wok[BL];
#pragma unroll
    for (int j = 0; j < BL; ++j) {
        int c = n0 + r0 + 32 * j;
        wok[j] = c < p.Cout;
        wrow[j] = p.wgt + (size_t)(wok[j] ? c : 0) * p.Kpad + c4 * 4;
        woff[j] = wok[j] ? (c * p.Kpad + c4 * 4) * 4 : -1;
    }

    // Global -> register staging.  The plain 64x64 instance keeps TWO register sets in flight: tile t+3 is requested as soon
    // as tile t+1 has been written to LDS, so a load has one whole iteration plus two k-steps (~6 k-steps instead of ~2) to
    // arrive.  Measured at B=64 (profiles/round2_two_register_sets.txt): +4 % on the long-K 3x3 layers that run this tile
    // (40x40x256 0.860 -> 0.829 ms, 20x20x512 0.889 -> 0.851), nothing on the short-K 1x1 layers (bounded by per-tile fixed
    // costs), and -3 % on the 128x128 instance (194 VGPRs, 16 MFMAs per k-step already cover its loads) -- hence 64x64 only.
    constexpr int NSET = (!XF && BM == 64 && BN == 64) ? 2 : 1;
    f32x4 ra[NSET][AL], rb[NSET][BL], xa, xb;
    unsigned okmask = 0;         // bit j: row j of the tile in flight is a real (not padded) pixel
    bool any_pad = true;         // wave-uniform: some lane of this wave has a padded row in the tile being stored
    int tapoff = 0;              // element offset of the current tap / channel chunk (wave-uniform)
    int kh = 0, kw = 0, c0 = 0;  // MODE 0 running tap state
    constexpr bool xform = XF;   // host guarantees OH*OW % BM == 0 then: one image per M-tile
    const float relu_floor = p.in_relu ? 0.f : -INFINITY;
    const int nblk = (m0 < p.M ? m0 : 0) / ohw;
    if (XF) {  // this tile's image: per-channel GroupNorm affine -> LDS once
        for (int i = tid; i < p.Cin; i += 256) {
            ABs[i] = p.in_a[nblk * p.Cin + i];
            ABs[XFMAX + i] = p.in_b[nblk * p.Cin + i];
        }
        __syncthreads();
    }

    // Loads are issued raw (from clamped, always-valid addresses) and stay in flight during the MFMA phase;
    // zero padding and the fused GN-apply+ReLU are applied when the registers are written to LDS.
    // Everything is split into per-row "pieces" so the pipelined K loop can drop one piece behind each MFMA.
    // MODE 0: per-row byte offsets (or -1 = zero padding) are invariant while the tap (kh, kw) stays the same, i.e. for
    // Cin/32 consecutive K-chunks; they are refreshed only when the tap changes.  Inside a tap the chunk offset is the
    // buffer instruction's SCALAR offset, so a load slot is a bare buffer_load (no per-lane address arithmetic).
    int voffA[AL];
    unsigned okcur = 0;
    auto refresh_rows = [&]() {
        okcur = 0;
        // the tap's pixel shift goes into the VECTOR offset (it must be a valid non-negative offset by itself: the
        // hardware range check looks at the vector offset only); the channel chunk is the scalar offset
        const int tapshift = (kh * p.W + kw) * p.Cin;
#pragma unroll
        for (int j = 0; j < AL; ++j) {
            const int iy = iy0[j] + kh, ix = ix0[j] + kw;
            const bool ok = mok[j] & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
            okcur |= (ok ? 1u : 0u) << j;
            voffA[j] = ok ? (rowoff[j] + tapshift) * 4 : -1;
        }
    };
    if (MODE == 0) refresh_rows();
    auto load_a = [&](auto set_c, int kt, int j) {
        constexpr int SET = decltype(set_c)::value;
        if (MODE == 0) {
            if (j == 0) {
                okmask = okcur;
                tapoff = c0 * 4;  // channel-chunk byte offset, wave-uniform (SALU)
            }
            ra[SET][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, voffA[j], tapoff, 0));
        } else {
            if (j == 0) okmask = 0;
            const int tap = kt * 8 + c4;
            const int th = (p.KW == 7) ? tap / 7 : tap / p.KW;   // the stem is 7x7: a constant divisor is a multiply + shift
            const int tw = tap - th * p.KW;
            const int iy = iy0[j] + th, ix = ix0[j] + tw;
            const bool ok = (tap < p.KH * p.KW) & mok[j] & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
            okmask |= (ok ? 1u : 0u) << j;
            const int voff = ok ? ((nimg[j] * p.H + iy) * p.W + ix) * 16 : -1;
            ra[SET][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, voff, 0, 0));
        }
    };
    auto load_x_advance = [&]() {  // after the last A row of a tile: GN affine of this K-chunk, then next tap
        if (MODE == 0) {
            if (xform) {
                xa = *reinterpret_cast<const f32x4*>(ABs + c0 + c4 * 4);
                xb = *reinterpret_cast<const f32x4*>(ABs + XFMAX + c0 + c4 * 4);
            }
            c0 += BK;
            if (c0 == cin_cur) {  // wave-uniform, once per Cin/32 chunks: next tap -> refresh the row offsets
                c0 = 0;
                if (!(DUAL && second)) {
                    if (++kw == p.KW) { kw = 0; ++kh; }
                    refresh_rows();
                }
            }
        }
    };
    int kt_last = p.Kpad / BK - 1;
    auto load_b = [&](auto set_c, int kt, int j) {  // tile indices past the end are clamped (the pipeline prefetches ahead)
        constexpr int SET = decltype(set_c)::value;
        rb[SET][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, woff[j], min(kt, kt_last) * (BK * 4), 0));
    };
    auto store_a = [&](auto set_c, int buf, int j) {
        constexpr int SET = decltype(set_c)::value;
        f32x4 v = ra[SET][j];
        if (MODE == 0 && xform) {
            v = v * xa + xb;
            v.x = fmaxf(v.x, relu_floor); v.y = fmaxf(v.y, relu_floor);
            v.z = fmaxf(v.z, relu_floor); v.w = fmaxf(v.w, relu_floor);
        }
        if (MODE == 0 && xform) {  // padded pixels must stay exactly 0 AFTER the affine (+ReLU)
            if (j == 0) any_pad = __builtin_amdgcn_ballot_w64(okmask != ((1u << AL) - 1u)) != 0;  // wave-uniform
            if (any_pad) {  // interior tiles (the vast majority) skip the selects with one scalar branch
                if (!((okmask >> j) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        *reinterpret_cast<f32x4*>(As + buf * BM * LDSW + (r0 + 32 * j) * LDSW + c4 * 4) = v;
    };
    auto store_b = [&](auto set_c, int buf, int j) {
        constexpr int SET = decltype(set_c)::value;
        *reinterpret_cast<f32x4*>(Bs + buf * BN * LDSW + (r0 + 32 * j) * LDSW + c4 * 4) = rb[SET][j];
    };
    auto load_tile = [&](auto set_c, int kt) {
#pragma unroll
        for (int j = 0; j < AL; ++j) load_a(set_c, kt, j);
        load_x_advance();
#pragma unroll
        for (int j = 0; j < BL; ++j) load_b(set_c, kt, j);
    };
    auto store_tile = [&](auto set_c, int buf) {
#pragma unroll
        for (int j = 0; j < AL; ++j) store_a(set_c, buf, j);
#pragma unroll
        for (int j = 0; j < BL; ++j) store_b(set_c, buf, j);
    };
    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, NSET - 1>;   // = Set0 when there is a single register set

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x16 acc2[DUAL ? MI : 1][DUAL ? NI : 1];   // DUAL: the second source's sums
    if (DUAL) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[DUAL ? i : 0][DUAL ? j : 0][r] = 0.f;
    }

    const int wm = wave & 1, wn = wave >> 1;
    const int arow = wm * WM + (lane & 31);
    const int brow = wn * WN + (lane & 31);
    const int koff = 4 * (lane >> 5);

    int KT = p.Kpad / BK;
    const float* a_lds = As + arow * LDSW + koff;
    const float* b_lds = Bs + brow * LDSW + koff;
    auto read_frags = [&](int buf, int kk, f32x4 (&fa)[MI], f32x4 (&fb)[NI]) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
            fa[i] = *reinterpret_cast<const f32x4*>(a_lds + buf * BM * LDSW + i * 32 * LDSW + kk * 8);
#pragma unroll
        for (int j = 0; j < NI; ++j)
            fb[j] = *reinterpret_cast<const f32x4*>(b_lds + buf * BN * LDSW + j * 32 * LDSW + kk * 8);
    };
    auto mfma_group = [&](const f32x4 (&fa)[MI], const f32x4 (&fb)[NI]) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][t], fb[j][t], acc[i][j], 0, 0, 0);
    };

    if (PIPE == 0) {
        // phase-separated schedule: [issue loads t+1] [MFMAs of t] [write t+1] barrier
        load_tile(Set0{}, 0);
        store_tile(Set0{}, 0);
        __syncthreads();
        for (int kt = 0; kt < KT; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < KT) load_tile(Set0{}, kt + 1);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                f32x4 fa[MI], fb[NI];
                read_frags(buf, kk, fa, fb);
                mfma_group(fa, fb);
            }
            if (kt + 1 < KT) store_tile(Set0{}, buf ^ 1);
            __syncthreads();
        }
    } else {
      auto kloop = [&](f32x16 (&accX)[MI][NI]) {
        // interleaved schedule: the two workgroups sharing a CU start together and run in lockstep, so a phase without
        // MFMAs idles the matrix pipe for BOTH (PMC: MFMA busy 70 % with the phase-separated loop).  Here every
        // non-MFMA instruction of an iteration sits in the shadow of that wave's own MFMAs:
        //   kk0: MFMAs | prefetch frags kk1
        //   kk1: MFMAs | prefetch frags kk2 | write tile t+1 (loaded during iteration t-1) to the other LDS buffer
        //   kk2: MFMAs | prefetch frags kk3 | issue global loads of tile t+2
        //   barrier (all reads of this buffer are complete, all writes of the other are visible)
        //   kk3: MFMAs | prefetch frags kk0 of tile t+1
        f32x4 fa0[MI], fb0[NI], fa1[MI], fb1[NI];
        // tile t lives in register set t & (NSET-1): with two sets, tiles 1 and 2 are both in flight when the loop starts
        load_tile(Set0{}, 0);
        store_tile(Set0{}, 0);
        load_tile(Set1{}, 1);
        if (NSET == 2) load_tile(Set0{}, 2);
        __syncthreads();
        read_frags(0, 0, fa0, fb0);
        constexpr int NM = 4 * MI * NI;  // MFMAs (= slots) per k-group
        constexpr int NF = MI + NI;      // fragment reads per k-group
        // pieces are dealt to the slots in contiguous runs so that they execute in program order (the tap-state
        // advance must follow the last A-row load of a tile)
        constexpr int P1 = NF + AL + BL, P2 = NF + AL + 1 + BL;
        constexpr int PPF = (NF + NM - 1) / NM, PP1 = (P1 + NM - 1) / NM, PP2 = (P2 + NM - 1) / NM;
        // one MFMA of a k-group: slot q -> (t, i, j)
#define MFMA_SLOT(FA, FB, q)                                                                                  \
    accX[((q) / NI) % MI][(q) % NI] = __builtin_amdgcn_mfma_f32_32x32x2f32(                                 \
        FA[((q) / NI) % MI][(q) / (MI * NI)], FB[(q) % NI][(q) / (MI * NI)], accX[((q) / NI) % MI][(q) % NI], 0, 0, 0)
        // fragment-read piece z (0..NF-1) of k-step kk from LDS buffer `buf` into (FA, FB)
#define FRAG_PIECE(FA, FB, buf, kk, z)                                                                        \
    do {                                                                                                      \
        if ((z) < MI)                                                                                         \
            FA[(z) < MI ? (z) : 0] = *reinterpret_cast<const f32x4*>(a_lds + (buf) * BM * LDSW +              \
                                                                     ((z) < MI ? (z) : 0) * 32 * LDSW + (kk) * 8); \
        else                                                                                                  \
            FB[(z) >= MI ? (z) - MI : 0] = *reinterpret_cast<const f32x4*>(                                   \
                b_lds + (buf) * BN * LDSW + ((z) >= MI ? (z) - MI : 0) * 32 * LDSW + (kk) * 8);                \
    } while (0)
        auto iteration = [&](auto nxt_c, int kt) {   // nxt_c: the register set of tile kt+1 (written to LDS here, then re-filled)
            const int buf = kt & 1;
            // ---- k-step 0: MFMAs on (fa0, fb0) | prefetch k-step 1 fragments
#pragma unroll
            for (int q = 0; q < NM; ++q) {
                MFMA_SLOT(fa0, fb0, q);
#pragma unroll
                for (int z = q * PPF; z < (q + 1) * PPF; ++z)
                    if (z < NF && !(ABL & 2)) FRAG_PIECE(fa1, fb1, buf, 1, z);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- k-step 1: MFMAs on (fa1, fb1) | prefetch k-step 2 | write tile kt+1 into the other LDS buffer
#pragma unroll
            for (int q = 0; q < NM; ++q) {
                MFMA_SLOT(fa1, fb1, q);
#pragma unroll
                for (int z = q * PP1; z < (q + 1) * PP1; ++z) {
                    if (z < NF) { if (!(ABL & 2)) FRAG_PIECE(fa0, fb0, buf, 2, z); }
                    else if (ABL & 1) {}
                    else if (ABL & 8) {  // keep the loaded registers alive, skip the LDS write
                        if (z < NF + AL) asm volatile("" ::"v"(ra[decltype(nxt_c)::value][z - NF < AL ? z - NF : 0]));
                        else if (z < P1) asm volatile("" ::"v"(rb[decltype(nxt_c)::value][z - NF - AL < BL ? z - NF - AL : 0]));
                    }
                    else if (z < NF + AL) store_a(nxt_c, buf ^ 1, z - NF);
                    else if (z < P1) store_b(nxt_c, buf ^ 1, z - NF - AL);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- k-step 2: MFMAs on (fa0, fb0) | prefetch k-step 3 | issue the global loads of tile kt+2
#pragma unroll
            for (int q = 0; q < NM; ++q) {
                MFMA_SLOT(fa0, fb0, q);
#pragma unroll
                for (int z = q * PP2; z < (q + 1) * PP2; ++z) {
                    if (z < NF) { if (!(ABL & 2)) FRAG_PIECE(fa1, fb1, buf, 3, z); }
                    else if (ABL & (1 | 16)) {}
                    else if (z < NF + AL) load_a(nxt_c, kt + 1 + NSET, z - NF);     // the set just written to LDS is free again
                    else if (z == NF + AL) load_x_advance();
                    else if (z < P2) load_b(nxt_c, kt + 1 + NSET, z - NF - AL - 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("" ::: "memory");       // no LDS access may be moved across the raw barrier by the compiler
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS reads/writes are done; loads stay in flight
            if (!(ABL & 4)) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            // ---- k-step 3: MFMAs on (fa1, fb1) | prefetch k-step 0 of tile kt+1 (other buffer, now complete)
#pragma unroll
            for (int q = 0; q < NM; ++q) {
                MFMA_SLOT(fa1, fb1, q);
#pragma unroll
                for (int z = q * PPF; z < (q + 1) * PPF; ++z)
                    if (z < NF && !(ABL & 2)) FRAG_PIECE(fa0, fb0, buf ^ 1, 0, z);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if constexpr (NSET == 2) {
            for (int kt = 0; kt < KT; kt += 2) {   // unrolled by the number of register sets: set indices stay compile-time
                iteration(Set1{}, kt);
                if (kt + 1 < KT) iteration(Set0{}, kt + 1);
            }
        } else {   // one register set: the plain loop (unrolling it by two cost the 128x128 instance 2 %)
            for (int kt = 0; kt < KT; ++kt) iteration(Set0{}, kt);
        }
#undef MFMA_SLOT
#undef FRAG_PIECE
      };
      kloop(acc);
      if constexpr (DUAL) {
        // ---- second source: 1x1, unpadded, stride2 over (N, H2, W2, Cin2); same M rows, same cout columns, its own accumulator
        // (acc * scale + bias is formed per source in the epilogue, exactly as the two separate launches would).  After the
        // first loop's last barrier no wave reads LDS data it still needs: the prologue below may overwrite both buffers.
        rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in2), 0,
                                                  (int)((size_t)p.N * p.H2 * p.W2 * p.Cin2 * 4), 0x00020000);
        rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wgt2), 0, (int)((size_t)p.Cout * p.Kpad2 * 4), 0x00020000);
        okcur = 0;
#pragma unroll
        for (int j = 0; j < AL; ++j) {
            const int mm = mok[j] ? m0 + r0 + 32 * j : 0;
            const int n = mm / ohw;
            const int rem = mm - n * ohw;
            const int oy = rem / p.OW, ox = rem - oy * p.OW;
            okcur |= (mok[j] ? 1u : 0u) << j;
            voffA[j] = mok[j] ? (((n * p.H2 + oy * p.stride2) * p.W2 + ox * p.stride2) * p.Cin2 + c4 * 4) * 4 : -1;
        }
#pragma unroll
        for (int j = 0; j < BL; ++j) woff[j] = wok[j] ? ((n0 + r0 + 32 * j) * p.Kpad2 + c4 * 4) * 4 : -1;
        cin_cur = p.Cin2; second = true;
        kh = 0; kw = 0; c0 = 0;
        KT = p.Kpad2 / BK; kt_last = KT - 1;
        kloop(acc2);
      }
    }

    // ---- epilogue.  D layout: col j = lane&31 (cout), row i = (r&3) + 8*(r>>2) + 4*(lane>>5) (pixel).
    // Branch-free: residual values are fetched in one batch from clamped addresses, stores are predicated.
    const int half = lane >> 5;
    const unsigned row_bytes = (unsigned)p.Cout * 4u;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
        p.out, 0, (int)((size_t)p.M * p.Cout * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.residual ? p.residual : p.out), 0, (int)((size_t)p.M * p.Cout * 4), 0x00020000);
    // (A 16-byte form of this epilogue -- affine in the accumulator layout, tile through LDS, buffer_*_dwordx4 residual loads
    // and stores, 4x fewer memory instructions -- was built and measured per layer at B=64: +-0 over the step, -7 % on the
    // HBM-bound 160x160 64->256 + residual layer; profiles/round3_wide_epilogue_ab.txt.  Not kept.)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int c = n0 + wn * WN + j * 32 + (lane & 31);
        const bool cok = c < p.Cout;
        const int cc = cok ? c : p.Cout - 1;
        const float sc = p.scale ? p.scale[cc] : 1.f;
        const float bi = p.bias ? p.bias[cc] : 0.f;
        const float sc2 = (DUAL && p.scale2) ? p.scale2[cc] : 1.f;
        const float bi2 = (DUAL && p.bias2) ? p.bias2[cc] : 0.f;
        float gsum = 0.f, gsq = 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int rbase = m0 + wm * WM + i * 32 + 4 * half;
            // 32-bit byte offsets through buffer descriptors whose range is the tensor: rows >= M and (offset 2^31)
            // channels >= Cout fall outside num_records -- loads return 0, stores are dropped, no predicates, and one
            // v_add per element instead of a 64-bit multiply-add
            const unsigned off0 = cok ? (unsigned)(rbase * p.Cout + c) * 4u : 0x80000000u;
            float res[16];
            if (p.residual) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    res[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                        rs_res, (int)(off0 + (unsigned)((r & 3) + 8 * (r >> 2)) * row_bytes), 0, 0));
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) res[r] = 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = rbase + (r & 3) + 8 * (r >> 2);
                float v = acc[i][j][r] * sc + bi;
                if (DUAL) {   // the shortcut branch's value, rounded to fp32 before the add as the separate launch stores it
                    const float idv = acc2[DUAL ? i : 0][DUAL ? j : 0][r] * sc2 + bi2;
                    v += idv;
                } else if (p.res_mask) v = res[r] > 0.f ? v : 0.f;
                else v += res[r];
                if (p.relu) v = fmaxf(v, 0.f);
                if (p.out_bf16) {  // bf16 compute mode: the fp32 stem hands a bf16 map to the bf16 layers
                    if (cok && m < p.M) {
                        const __bf16 hv = (__bf16)v;
                        reinterpret_cast<unsigned short*>(p.out)[(size_t)m * p.Cout + c] = __builtin_bit_cast(unsigned short, hv);
                    }
                } else {
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_out,
                                                          (int)(off0 + (unsigned)((r & 3) + 8 * (r >> 2)) * row_bytes), 0, 0);
                }
                if (p.gn_part) {     // statistics only count real pixels / channels
                    const float u = (cok && m < p.M) ? v : 0.f;
                    gsum += u;
                    gsq += u * u;
                }
            }
        }
        if (p.gn_part) {
            // per-channel partials of this block's 128 pixels: combine the two half-waves, then the two
            // M-waves through LDS (the A region is free: every wave is past the K loop's final barrier).
            gsum += __shfl_xor(gsum, 32, 64);
            gsq += __shfl_xor(gsq, 32, 64);
            const int cl = wn * WN + j * 32 + (lane & 31);
            if (half == 0) {
                smem[(wm * BN + cl) * 2 + 0] = gsum;
                smem[(wm * BN + cl) * 2 + 1] = gsq;
            }
        }
    }
    if (p.gn_part) {
        __syncthreads();
        if (tid < BN) {
            const int c = n0 + tid;
            if (c < p.Cout) {
                float s = smem[tid * 2] + smem[(BN + tid) * 2];
                float q = smem[tid * 2 + 1] + smem[(BN + tid) * 2 + 1];
                p.gn_part[((size_t)tm * p.Cout + c) * 2 + 0] = s;
                p.gn_part[((size_t)tm * p.Cout + c) * 2 + 1] = q;
            }
        }
    }
}

// C-ABI ------------------------------------------------------------------------------------------
// The product library holds no mutable state.  The benchmark hooks (forced tile, K-loop schedule A/B, loop ablations --
// the latter give WRONG results by design) exist only in builds with -DCPR_BENCH_HOOKS (python -m
// pointtinybenchmark_amd.build --bench-hooks -> libcprhip_bench.so, used by tools/); they are process-global and not
// thread-safe, which is acceptable for a single-threaded measurement script and for nothing else.
#ifdef CPR_BENCH_HOOKS
static int force_tile_bm = 0, force_tile_bn = 0;  // cpr_conv_force_tile, 0 = heuristic
static int conv_stream_on = 1;                     // cpr_conv_set_stream: 0 = the streamed 1x1 kernel is never chosen (A/B)
static int conv_pipeline = 1;                      // 1 = interleaved K loop (default), 0 = phase-separated (A/B reference)
static int conv_ablate = 0;                        // cpr_conv_set_ablation
static int conv_extra_lds = 0;                     // cpr_conv_set_extra_lds: dynamic LDS bytes added to every launch (occupancy probe)

extern "C" int cpr_conv_set_extra_lds(int bytes) {
    CPR_CHECK_ARG(bytes >= 0 && bytes <= 65536);
    conv_extra_lds = bytes;
    return CPR_OK;
}
extern "C" int cpr_conv_set_ablation(int mode) {
    CPR_CHECK_ARG(mode >= 0 && mode <= 16);
    conv_ablate = mode;
    return CPR_OK;
}
extern "C" int cpr_conv_set_stream(int on) {
    CPR_CHECK_ARG(on == 0 || on == 1);
    conv_stream_on = on;
    return CPR_OK;
}
extern "C" int cpr_conv_set_pipeline(int mode) {
    CPR_CHECK_ARG(mode == 0 || mode == 1);
    conv_pipeline = mode;
    return CPR_OK;
}
extern "C" int cpr_conv_force_tile(int bm, int bn) {
    CPR_CHECK_ARG((bm == 0 || bm == 64 || bm == 128) && (bn == 0 || bn == 64 || bn == 128));
    force_tile_bm = bm;
    force_tile_bn = bn;
    return CPR_OK;
}
#else
constexpr int force_tile_bm = 0, force_tile_bn = 0, conv_pipeline = 1, conv_ablate = 0, conv_extra_lds = 0, conv_stream_on = 1;
#endif

// csrc/conv1x1_stream.hip: the HBM-bound 1x1 shapes as a stream (CPR_ERR_UNSUPPORTED = not one of its shapes)
int conv1x1_stream_launch(const float* in, const float* wgt, float* out, const float* scale, const float* bias,
                          const float* residual, long long M, int Cin, int Cout, int relu, int res_mask, int min_tiles,
                          hipStream_t stream);
constexpr int STREAM_MIN_TILES = 1024;   // four tiles per persistent workgroup; smaller launches keep the 64x64 tiles (4 workgroups / CU)

// one launch: every tensor below 2 GiB.  bm_fix > 0: the M tile of an earlier chunk of the same call (column-sum slots must line up)
static int conv2d_fwd_launch(const float* in, const float* wgt, float* out, const float* scale, const float* bias,
                             const float* residual, const float* in_a, const float* in_b, float* gn_part, int N,
                             int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int Kpad,
                             int flags, int in_relu, int bm_fix, int* variant_out, hipStream_t stream) {
    CPR_CHECK_ARG(in && wgt && out);
    CPR_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0);
    CPR_CHECK_ARG(Kpad % BK == 0);
    ConvParams p;
    p.in = in; p.wgt = wgt; p.out = out; p.scale = scale; p.bias = bias; p.residual = residual;
    p.in_a = in_a; p.in_b = in_b; p.gn_part = gn_part;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad;
    CPR_CHECK_ARG((flags & ~15) == 0);
    p.Kpad = Kpad; p.relu = flags & CPR_CONV_RELU; p.in_relu = in_relu; p.out_bf16 = (flags & CPR_CONV_OUT_BF16) ? 1 : 0;
    p.res_mask = (flags & CPR_CONV_RES_MASK) ? 1 : 0;
    p.in2 = nullptr; p.wgt2 = nullptr; p.scale2 = nullptr; p.bias2 = nullptr; p.H2 = p.W2 = p.Cin2 = p.stride2 = p.Kpad2 = 0;
    const bool colsum_mode = (flags & CPR_CONV_COLSUM) != 0;   // gn_part partials are only summed over the whole tensor (any tile)
    if (p.res_mask) CPR_CHECK_ARG(residual != nullptr);
    p.OH = (H + 2 * pad - KH) / stride + 1;
    p.OW = (W + 2 * pad - KW) / stride + 1;
    CPR_CHECK_ARG(p.OH > 0 && p.OW > 0);
    long long M = (long long)N * p.OH * p.OW;
    // 32-bit byte offsets inside the buffer descriptors: < 2 GiB per tensor (B=64 at 160x160x256 is 1.7 GB)
    if ((long long)N * H * W * Cin * 4 >= (1ll << 31) || (long long)Cout * Kpad * 4 >= (1ll << 31) || M >= (1ll << 31) ||
        M * Cout * 4 >= (1ll << 31))
        return CPR_ERR_UNSUPPORTED;
    p.M = (int)M;
    const bool mode1 = (Cin == 4);
    if (!mode1) {
        CPR_CHECK_ARG(Cin % BK == 0 && Kpad == KH * KW * Cin);
    } else {
        CPR_CHECK_ARG(Kpad >= KH * KW * 4 && in_a == nullptr);
    }
    if ((gn_part && !colsum_mode) || in_a) CPR_CHECK_ARG((p.OH * p.OW) % 128 == 0);
    if (in_a) CPR_CHECK_ARG(in_b && p.OH == H && p.OW == W && Cin <= 512);
    // plain 1x1 GEMMs with 64 / 128 input channels are HBM-bound: streamed (bit-identical to the tiled kernel, see conv1x1_stream.hip)
    if (conv_stream_on && !mode1 && KH == 1 && KW == 1 && stride == 1 && pad == 0 && !in_a && !gn_part && !p.out_bf16 &&
        !force_tile_bm && !conv_ablate) {
        const int rc = conv1x1_stream_launch(in, wgt, out, scale, bias, residual, M, Cin, Cout, p.relu, p.res_mask,
                                             STREAM_MIN_TILES, stream);
        if (rc != CPR_ERR_UNSUPPORTED) {
            // BM, BN of conv1x1_stream_launch (K = 64: 128 x 256, K = 128: 64 x 128, K = 256: 128 x 64 in four k chunks), 3 = streamed
            if (variant_out) *variant_out = (Cin == 256 ? 128 : 8192 / Cin) * 1000000 + (Cin == 256 ? 64 : 16384 / Cin) * 1000 + 3;
            return rc;
        }
    }
    // tile selection (measured per layer on MI355X, profiles/round1_tile_sweep.txt): 64x64 tiles run 4 workgroups per CU
    // (36.9 KB LDS, 74 VGPRs) and win on everything except very large, long-K problems -- finer granularity against
    // tile quantisation and 4 waves/SIMD to hide the prologue/epilogue latency of short-K 1x1 convs.  128x128 is kept for
    // the long, wide launches; GN-fused launches keep 128-pixel tiles (one statistics slot per tile).
    int bm = 128, bn = (Cout <= 64) ? 64 : 128;
    if ((!gn_part || colsum_mode) && !in_a && !mode1) {
        const long long t128 = (long long)((p.M + 127) / 128) * ((Cout + 127) / 128);
        const int kt = Kpad / BK;
        if (!(kt >= 16 && t128 >= 4096 && Cout > 64)) { bm = 64; bn = 64; }
    }
    if (force_tile_bm > 0 && (!gn_part || colsum_mode) && !in_a && !mode1) { bm = force_tile_bm; }
    if (bm_fix > 0) bm = bm_fix;      // later chunk of a split batch: same M tile as the first one
    if (force_tile_bn > 0 && Cout > 64 && !mode1) { bn = force_tile_bn; }
    if (variant_out) *variant_out = bm * 1000000 + bn * 1000 + (mode1 ? 100 : 0) + (in_a ? 10 : 0) + conv_pipeline;
    p.tilesM = (p.M + bm - 1) / bm;
    p.tilesN = (Cout + bn - 1) / bn;
    const int T = p.tilesM * p.tilesN;
    const int grid = ((T + 7) / 8) * 8;
#ifdef CPR_BENCH_HOOKS
#define LAUNCH(BM_, BN_, MODE_, XF_)                                                                               \
    do {                                                                                                           \
        if (conv_pipeline == 0)                                                                                    \
            hipLaunchKernelGGL((conv_mfma_kernel<BM_, BN_, MODE_, XF_, 0>), dim3(grid), dim3(256), conv_extra_lds, stream, p); \
        else                                                                                                       \
            hipLaunchKernelGGL((conv_mfma_kernel<BM_, BN_, MODE_, XF_, 1>), dim3(grid), dim3(256), conv_extra_lds, stream, p); \
    } while (0)
#else
#define LAUNCH(BM_, BN_, MODE_, XF_) \
    hipLaunchKernelGGL((conv_mfma_kernel<BM_, BN_, MODE_, XF_, 1>), dim3(grid), dim3(256), 0, stream, p)
#endif
#ifdef CPR_BENCH_HOOKS
    if (conv_ablate && !mode1 && !in_a && bm == 128 && bn == 128) {
        switch (conv_ablate) {
            case 1: hipLaunchKernelGGL((conv_mfma_kernel<128, 128, 0, false, 1, 1>), dim3(grid), dim3(256), 0, stream, p); break;
            case 2: hipLaunchKernelGGL((conv_mfma_kernel<128, 128, 0, false, 1, 2>), dim3(grid), dim3(256), 0, stream, p); break;
            case 3: hipLaunchKernelGGL((conv_mfma_kernel<128, 128, 0, false, 1, 3>), dim3(grid), dim3(256), 0, stream, p); break;
            case 4: hipLaunchKernelGGL((conv_mfma_kernel<128, 128, 0, false, 1, 4>), dim3(grid), dim3(256), 0, stream, p); break;
            case 8: hipLaunchKernelGGL((conv_mfma_kernel<128, 128, 0, false, 1, 8>), dim3(grid), dim3(256), 0, stream, p); break;
            case 16: hipLaunchKernelGGL((conv_mfma_kernel<128, 128, 0, false, 1, 16>), dim3(grid), dim3(256), 0, stream, p); break;
            default: hipLaunchKernelGGL((conv_mfma_kernel<128, 128, 0, false, 1, 7>), dim3(grid), dim3(256), 0, stream, p); break;
        }
    } else
#endif
    if (mode1) {
        if (bn == 64) LAUNCH(128, 64, 1, false); else LAUNCH(128, 128, 1, false);
    } else if (in_a) {
        if (bn == 64) LAUNCH(128, 64, 0, true); else LAUNCH(128, 128, 0, true);
    } else if (bm == 128) {
        if (bn == 64) LAUNCH(128, 64, 0, false); else LAUNCH(128, 128, 0, false);
    } else {
        if (bn == 64) LAUNCH(64, 64, 0, false); else LAUNCH(64, 128, 0, false);
    }
#undef LAUNCH
    CPR_LAUNCH_STATUS();
}

extern "C" int cpr_conv2d_fwd(const float* in, const float* wgt, float* out, const float* scale, const float* bias,
                              const float* residual, const float* in_a, const float* in_b, float* gn_part, int N,
                              int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int Kpad,
                              int flags, int in_relu, int* variant_out, hipStream_t stream) {
    CPR_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0);
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    CPR_CHECK_ARG(OH > 0 && OW > 0);
    const long long in_img = (long long)H * W * Cin * 4, out_img = (long long)OH * OW * Cout * ((flags & CPR_CONV_OUT_BF16) ? 2 : 4);
    const long long res_img = (long long)OH * OW * Cout * 4;
    // column-sum partials are indexed by M tile across images: chunks must start on a 128-row boundary
    int align = 1;
    if (flags & CPR_CONV_COLSUM) { const long long r = (long long)OH * OW; align = 1; while ((r * align) % 128 != 0) align *= 2; }
    const int per = cpr_images_per_launch(N, cpr_max2(cpr_max2(in_img, out_img), residual ? res_img : 0), align);
    if (per <= 0) return CPR_ERR_UNSUPPORTED;
    if (per >= N) return conv2d_fwd_launch(in, wgt, out, scale, bias, residual, in_a, in_b, gn_part, N, H, W, Cin, Cout, KH, KW,
                                           stride, pad, Kpad, flags, in_relu, 0, variant_out, stream);
    int bm_fix = 0, variant = 0;
    const size_t oe = (flags & CPR_CONV_OUT_BF16) ? 2 : 4;
    for (int n0 = 0; n0 < N; n0 += per) {
        const int n = N - n0 < per ? N - n0 : per;
        const size_t rows = (size_t)n0 * OH * OW;
        // partial slots: one per 128-pixel tile of an image (GroupNorm statistics) or one per bm-row tile (column sums)
        float* part = nullptr;
        if (gn_part) part = gn_part + ((flags & CPR_CONV_COLSUM) ? (bm_fix ? rows / bm_fix : 0) : rows / 128) * Cout * 2;
        const int rc = conv2d_fwd_launch(in + (size_t)n0 * H * W * Cin, wgt, (float*)((char*)out + rows * Cout * oe), scale, bias,
                                         residual ? residual + rows * Cout : nullptr, in_a ? in_a + (size_t)n0 * Cin : nullptr,
                                         in_b ? in_b + (size_t)n0 * Cin : nullptr, part, n, H, W, Cin, Cout, KH, KW, stride, pad,
                                         Kpad, flags, in_relu, bm_fix, &variant, stream);
        if (rc != CPR_OK) return rc;
        // only the column-sum slots tie later chunks to the first chunk's M tile (never a streamed launch: those carry no partials)
        if (n0 == 0) { if (flags & CPR_CONV_COLSUM) bm_fix = variant / 1000000; if (variant_out) *variant_out = variant; }
    }
    return CPR_OK;
}

// ---- dual-source launch: out = relu?((conv(in, wgt) * scale + bias) + (conv1x1_stride2(in2, wgt2) * scale2 + bias2)) -----------
// The first block of every ResNet stage (T/mmdet/models/backbones/resnet.py:262-302, 564-610): conv3 + bn3 and the projection
// shortcut (downsample conv + bn) share their output pixels and channels, so both GEMMs run in one launch with two
// accumulators and the shortcut map (4x the bottleneck width: 1.68 GB at 160x160x256, B=64) never exists in HBM.  The result
// is bit-identical to the two-launch form (same K order per source, same epilogue roundings).
static int conv2d_dual_launch(const float* in, const float* wgt, const float* in2, const float* wgt2, float* out,
                              const float* scale, const float* bias, const float* scale2, const float* bias2, int N, int H,
                              int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int Kpad, int H2, int W2,
                              int Cin2, int stride2, int Kpad2, int flags, int* variant_out, hipStream_t stream) {
    ConvParams p;
    p.in = in; p.wgt = wgt; p.out = out; p.scale = scale; p.bias = bias; p.residual = nullptr;
    p.in_a = nullptr; p.in_b = nullptr; p.gn_part = nullptr;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad;
    p.Kpad = Kpad; p.relu = flags & CPR_CONV_RELU; p.in_relu = 0; p.out_bf16 = 0; p.res_mask = 0;
    p.OH = (H + 2 * pad - KH) / stride + 1;
    p.OW = (W + 2 * pad - KW) / stride + 1;
    p.in2 = in2; p.wgt2 = wgt2; p.scale2 = scale2; p.bias2 = bias2;
    p.H2 = H2; p.W2 = W2; p.Cin2 = Cin2; p.stride2 = stride2; p.Kpad2 = Kpad2;
    p.tilesM = p.tilesN = 0;
    const long long M = (long long)N * p.OH * p.OW;
    if ((long long)N * H * W * Cin * 4 >= (1ll << 31) || (long long)N * H2 * W2 * Cin2 * 4 >= (1ll << 31) ||
        (long long)Cout * Kpad * 4 >= (1ll << 31) || (long long)Cout * Kpad2 * 4 >= (1ll << 31) || M * Cout * 4 >= (1ll << 31))
        return CPR_ERR_UNSUPPORTED;
    p.M = (int)M;
    // same tile rule as the single-source launcher, on the summed K
    int bm = 64, bn = 64;
    const long long t128 = (long long)((p.M + 127) / 128) * ((Cout + 127) / 128);
    if ((Kpad + Kpad2) / BK >= 16 && t128 >= 4096 && Cout > 64) { bm = 128; bn = 128; }
    if (force_tile_bm > 0) { bm = bn = force_tile_bm; }
    if (variant_out) *variant_out = bm * 1000000 + bn * 1000 + 2;
    p.tilesM = (p.M + bm - 1) / bm;
    p.tilesN = (Cout + bn - 1) / bn;
    const int grid = ((p.tilesM * p.tilesN + 7) / 8) * 8;
    if (bm == 64) hipLaunchKernelGGL((conv_mfma_kernel<64, 64, 0, false, 1, 0, true>), dim3(grid), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((conv_mfma_kernel<128, 128, 0, false, 1, 0, true>), dim3(grid), dim3(256), 0, stream, p);
    CPR_LAUNCH_STATUS();
}

extern "C" int cpr_conv2d_dual_fwd(const float* in, const float* wgt, const float* in2, const float* wgt2, float* out,
                                   const float* scale, const float* bias, const float* scale2, const float* bias2, int N,
                                   int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int Kpad, int H2,
                                   int W2, int Cin2, int stride2, int Kpad2, int flags, int* variant_out, hipStream_t stream) {
    CPR_CHECK_ARG(in && wgt && in2 && wgt2 && out);
    CPR_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0);
    CPR_CHECK_ARG(H2 > 0 && W2 > 0 && Cin2 > 0 && stride2 > 0 && (flags & ~CPR_CONV_RELU) == 0);
    CPR_CHECK_ARG(Cin % BK == 0 && Kpad == KH * KW * Cin && Cin2 % BK == 0 && Kpad2 == Cin2);
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    CPR_CHECK_ARG(OH > 0 && OW > 0 && OH == (H2 - 1) / stride2 + 1 && OW == (W2 - 1) / stride2 + 1);
    const long long img = cpr_max2(cpr_max2((long long)H * W * Cin, (long long)H2 * W2 * Cin2), (long long)OH * OW * Cout) * 4;
    const int per = cpr_images_per_launch(N, img);
    if (per <= 0) return CPR_ERR_UNSUPPORTED;
    for (int n0 = 0; n0 < N; n0 += per) {
        const int n = N - n0 < per ? N - n0 : per;
        const int rc = conv2d_dual_launch(in + (size_t)n0 * H * W * Cin, wgt, in2 + (size_t)n0 * H2 * W2 * Cin2, wgt2,
                                          out + (size_t)n0 * OH * OW * Cout, scale, bias, scale2, bias2, n, H, W, Cin, Cout,
                                          KH, KW, stride, pad, Kpad, H2, W2, Cin2, stride2, Kpad2, flags,
                                          n0 == 0 ? variant_out : nullptr, stream);
        if (rc != CPR_OK) return rc;
    }
    return CPR_OK;
}
