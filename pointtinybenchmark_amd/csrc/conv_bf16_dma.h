// Shared by the bf16 LDS-DMA convolution kernels (conv_bf16_dma.hip: the lock-step 256 x 256 / 128 x 128 instances; conv_bf16_pp.hip: the
// two-group ping-pong 256 x 256 instance of round 6): launch parameters, tile geometry and the three epilogues.
#pragma once
#include "common.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef int i32x4v __attribute__((ext_vector_type(4)));

struct ConvDmaParams {
    const unsigned short* in;
    const unsigned short* wgt;
    const unsigned short* wfrag;   // BD instances: the weights in MFMA-fragment order (see conv_bf16_dma_kernel<.., BD = true>), else null
    void* out;
    const float* scale;
    const float* bias;
    const unsigned short* residual;
    float* gn_part;
    // TR instances only (round 6, the block-boundary data gradient of the mixed-precision backward): add32 = an fp32 map of the
    // output's shape summed in BEFORE the mask (the shortcut gradient), out16 = a second, bf16 copy of the fp32 result
    const float* add32;
    unsigned short* out16;
    int N, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad, Kpad, M, relu, out_fp32;
    // relu: 0 none, 1 ReLU behind the residual add, 2 = MASK mode (round 6, the mixed-precision backward): `residual` is the bf16 map a
    // forward ReLU produced and the result is v where that map is positive, 0 elsewhere (nothing is added); gn_part then carries
    // per-channel SUMS of the masked result (element 0 of each pair; [slot][Cout][2], slot = 128 pixels), i.e. the column sums the
    // folded-BN bias gradient needs -- data gradient, ReLU backward, bf16 rounding and column sums in one launch.
    int tilesM, tilesN;
    int ablate;     // measurement builds only (results are then WRONG): 1 no wait for the DMA, 2 no barrier, 4 no DMA requests, 8 no fragment reads,
                    // 16 / 32 / 64 epilogue forms, 128 no weight-fragment loads (BD instance), 256 request slots staggered by wave (BD)
    // NT-GEMM mode (the bf16 weight gradient, csrc/conv_wgrad_bf16.hip): out[split][tap][m][n] = sum over the split's K range of
    // in[m][k] * wgt_tap[n][k], both operands rows of Kpad (= Cin) elements.  Tiles = splits x taps x tilesM x tilesN.
    int nt_taps;        // 0 = convolution mode
    int nt_k;           // tap grid is nt_k x nt_k: tap t reads copy t % nt_k of the right operand, shifted by (t / nt_k - nt_pad) rows of nt_Wp
    int nt_pad, nt_Wp;
    int nt_chunks;      // 64-element K chunks per split
    long long nt_copy;  // elements between the copies of the right operand
};

constexpr int DBK = 64;
// Tile instances <MI, NJ>: a wave owns MI x NJ blocks of 32 x 32, the workgroup (2 x 4 waves) 64 MI pixels x 128 NJ couts.
//   <4, 2> = 256 x 256 (the round-3 kernel: 32 MFMAs per wave per chunk against 8 requests + 24 fragment reads, one workgroup
//            per CU): the MFMA-bound layers -- 3x3 with K >= 1024 on maps of >= 384 tiles;
//   <2, 1> = 128 x 128 (round 4): 8 MFMAs against 4 requests + 12 reads per chunk -- poor food for the matrix pipe, but 64 KB of
//            LDS and 91 VGPRs = TWO workgroups per CU, and the layers it takes are not MFMA-bound: the bottleneck 1x1s move
//            more bytes than flops (conv3 256 -> 1024 + residual on 32 768 pixels: 151 MB against 17 GFLOP), and with one
//            workgroup per CU all 256 CUs run K loop, residual read and store phase in lock step -- HBM idles during the K loops
//            and is the only thing working during the epilogues.  Two resident workgroups drift apart and overlap the phases;
//            and R101's layer3 at 1024^2 B = 8 (M = 32 768: 128 tiles of 256 x 256 for 256 CUs) fills the chip.
//   <2, 2> = 128 x 256 and <4, 1> = 256 x 128 were built and measured too (profiles/round4_bf16_tiles_and_epilogue.txt): never
//            ahead of <2, 1>; not instantiated.
//   <4, 4, 2> = 256 x 256 on FOUR waves (WN = 2 wave columns; wave = 128 x 128 = 256 accumulator registers in AGPRs, one wave per
//            SIMD): 8 fragment reads per 16 MFMAs instead of 6 per 8, i.e. a third less LDS read traffic -- built to test whether
//            LDS bandwidth is what holds <4, 2> at 0.42 (its reads + DMA writes are 256 KB per chunk against 262 KB of LDS
//            cycles).  Bit-equal to <4, 2> and 10-12 % SLOWER on every shape tried: one wave per SIMD has nobody
//            to cover its waits.  Measurement build only.
template <int MI, int NJ, int WN = 4, bool BD = false> struct DmaTile {
    static constexpr int NT = 128 * WN;                                // threads: 2 wave rows x WN wave columns
    static constexpr int RP = NT / 8;                                  // rows one request piece covers (a wave instruction = 8 rows)
    static constexpr int PIECE = NT * 16;                              // bytes per piece
    static constexpr int BM = 64 * MI, BN = WN * NJ * 32;
    static constexpr int NPA = BM / RP, NPW = BD ? 0 : BN / RP, NP = NPA + NPW; // request pieces per chunk: activations, weights (BD: none)
    static constexpr int NM = MI * NJ, NF = BD ? MI : MI + NJ;         // MFMAs / LDS fragment reads per wave per k-step
    static constexpr int STAGE = (BM + (BD ? 0 : BN)) * DBK * 2;       // bytes per stage
    // the chunk's pieces over three k-steps, in their LAST MFMA slots: C3 behind the barrier (k-step 3), C0 in k-step 0, C1 in k-step 1
    static constexpr int C3 = (NP + 2) / 3, C0 = (NP - C3 + 1) / 2, C1 = NP - C3 - C0;
    static_assert(C3 <= NM && C0 <= NM && C1 <= NM, "one request per MFMA slot at most");
};

__device__ __forceinline__ float bf16f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }

// Epilogue shared by both kernels: y = acc * scale + bias (+ residual) (ReLU) -> bf16 (cout pairs packed) or fp32; GroupNorm
// statistics from the fp32 values, one slot per 128 pixels = per (tile, wm): a wave owns its slot's 64 channels outright.
template <int MI, int NJ>
__device__ __forceinline__ void dma_epilogue(const ConvDmaParams& p, f32x16 (&acc)[MI][NJ], int tm, int m0, int n0, int wm,
                                             int wn, int lane) {
    const int l31 = lane & 31, half = lane >> 5;
    if (p.ablate & 32) return;
    // D layout of a 32 x 32 block: col = lane & 31 (cout), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (pixel)
    unsigned short* out16 = reinterpret_cast<unsigned short*>(p.out);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
        p.out, 0, (p.ablate & 16) ? 0 : (int)((size_t)p.M * p.Cout * (p.out_fp32 ? 4 : 2)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(p.residual ? p.residual : (const unsigned short*)p.out), 0,
        (int)((size_t)p.M * p.Cout * 2), 0x00020000);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = n0 + wn * (NJ * 32) + j * 32 + l31;
        const bool cok = c < p.Cout;
        const int cc = cok ? c : p.Cout - 1;
        const float sc = p.scale ? p.scale[cc] : 1.f;
        const float bi = p.bias ? p.bias[cc] : 0.f;
        float gsum = 0.f, gsq = 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int rbase = m0 + wm * (MI * 32) + i * 32 + 4 * half;
            const unsigned e0 = (unsigned)(rbase * p.Cout + c);
            float res[16];
            if (p.residual) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    res[r] = bf16f(__builtin_amdgcn_raw_buffer_load_b16(
                        rs_res, (int)(cok ? (e0 + (unsigned)((r & 3) + 8 * (r >> 2)) * (unsigned)p.Cout) * 2u : 0x80000000u), 0, 0));
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) res[r] = 0.f;
            }
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = rbase + (r & 3) + 8 * (r >> 2);
                float x = acc[i][j][r] * sc + bi;
                if (p.relu == 2) x = res[r] > 0.f ? x : 0.f;
                else {
                    x += res[r];
                    if (p.relu) x = fmaxf(x, 0.f);
                }
                v[r] = x;
                if (p.gn_part) {
                    const float u = (cok && m < p.M) ? x : 0.f;
                    gsum += u;
                    gsq += u * u;
                }
            }
            if (p.out_fp32) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(
                        __builtin_bit_cast(unsigned, v[r]), rs_out,
                        (int)(cok ? (e0 + (unsigned)((r & 3) + 8 * (r >> 2)) * (unsigned)p.Cout) * 4u : 0x80000000u), 0, 0);
            } else {
                // cout pairs packed into one dword: even lanes write the even registers, odd lanes the odd ones (Cout is even)
                const unsigned ep = (unsigned)(rbase * p.Cout + (c & ~1));
                const bool pok = (c & ~1) < p.Cout;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float nb = __shfl_xor(v[r], 1, 64);
                    const bool mine = ((r & 1) == (lane & 1)) && pok;
                    const bf16x2 pk = (lane & 1) ? bf16x2{(__bf16)nb, (__bf16)v[r]} : bf16x2{(__bf16)v[r], (__bf16)nb};
                    __builtin_amdgcn_raw_buffer_store_b32(
                        __builtin_bit_cast(unsigned, pk), rs_out,
                        (int)(mine ? (ep + (unsigned)((r & 3) + 8 * (r >> 2)) * (unsigned)p.Cout) * 2u : 0x80000000u), 0, 0);
                }
            }
        }
        if (p.gn_part) {
            gsum += __shfl_xor(gsum, 32, 64);
            gsq += __shfl_xor(gsq, 32, 64);
            if (half == 0 && cok) {
                float* dst = p.gn_part + ((size_t)(tm * 2 + wm) * p.Cout + c) * 2;
                dst[0] = gsum;
                dst[1] = gsq;
            }
        }
    }
    (void)out16;
}

// Direct epilogue of the instances with TWO cout blocks per wave (<4, 2>; round 4).  The K loop of those instances lays the weight
// rows out INTERLEAVED in LDS -- row 64 g + 32 j + l of the tile holds cout 64 g + 2 l + j (the permutation sits in the rows the DMA
// requests, see woff in the kernel) -- so lane l of the wave holds couts 2 l (block 0) and 2 l + 1 (block 1): one dword per row
// WITHOUT a lane exchange, all 64 lanes storing, and a wave instruction writes two whole 128-byte lines (rows m, m + 4) where the
// shared epilogue above writes four 64-byte halves with half its lanes idle -- PMC had 96 MB written per 67 MB map on the head
// layer (profiles/round4_pmc_bf16_big_tile.json).  Residual: one 4-byte load per row instead of two 2-byte ones; fp32 output: 8 bytes.
// PRE (round 5, the weights-direct instance: its 219 registers leave room for a second set of 16): the residual of pixel block
// i + 1 is requested BEFORE block i is scaled and stored -- un-prefetched, every block's 16 loads are issued and awaited in
// turn, and on the short-K bottleneck layers (1x1 256 -> 1024 + residual on 8 x 128^2: four K chunks) the epilogue is 60 % of the
// launch (0.196 ms with, 0.076 ms without it; 0.168 with the stores dropped: profiles/round5_bf16_epilogue_ablation.txt).
// TR (round 6): the block-boundary data gradient of the mixed-precision backward in one launch --
//   g = mask > 0 ? acc * scale + bias + add32 : 0   -> out (fp32: the shortcut chain stays fp32) AND out16 (its bf16 rounding),
// column sums of g into gn_part (element 0 of each slot pair).  It replaces an fp32 store + a streaming pass that read it back
// with the shortcut gradient and the mask (20 bytes per element -> 12).
template <int MI, bool PRE = false, bool TR = false>
__device__ __forceinline__ void dma_epilogue_pairs(const ConvDmaParams& p, f32x16 (&acc)[MI][2], int tm, int m0, int n0, int wm,
                                                   int wn, int lane) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const int l31 = lane & 31, half = lane >> 5;
    if (p.ablate & 32) return;
    if constexpr (TR) {
        static_assert(PRE, "the training epilogue prefetches its mask");
        const __amdgpu_buffer_rsrc_t rs32 = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)((size_t)p.M * p.Cout * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs16 = __builtin_amdgcn_make_buffer_rsrc(p.out16, 0, (int)((size_t)p.M * p.Cout * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_add = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.add32), 0, (int)((size_t)p.M * p.Cout * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_msk = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.residual), 0, (int)((size_t)p.M * p.Cout * 2), 0x00020000);
        const int c0 = n0 + wn * 64 + 2 * l31;
        const bool cok = c0 < p.Cout;
        const int cc = cok ? c0 : p.Cout - 2;
        const float sc0 = p.scale ? p.scale[cc] : 1.f, sc1 = p.scale ? p.scale[cc + 1] : 1.f;
        const float bi0 = p.bias ? p.bias[cc] : 0.f, bi1 = p.bias ? p.bias[cc + 1] : 0.f;
        float gs0 = 0.f, gs1 = 0.f;
        unsigned msk[16];
        u32x2 addv[16];
        auto row_e = [&](int i, int r) { return (unsigned)((m0 + wm * (MI * 32) + i * 32 + 4 * half + (r & 3) + 8 * (r >> 2)) * p.Cout + c0); };
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            // both operands of pixel block i are requested here and awaited below: the partner wave of the SIMD covers the wait, and
            // prefetch sets beside the 128 accumulators spilled (61 registers with the mask of block i + 1 in flight as in PRE)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned e = row_e(i, r);
                msk[r] = __builtin_amdgcn_raw_buffer_load_b32(rs_msk, (int)(cok ? e * 2u : 0x80000000u), 0, 0);
                addv[r] = __builtin_amdgcn_raw_buffer_load_b64(rs_add, (int)(cok ? e * 4u : 0x80000000u), 0, 0);
            }
            const int rbase = m0 + wm * (MI * 32) + i * 32 + 4 * half;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2);
                const unsigned mk = msk[r];
                float x0 = acc[i][0][r] * sc0 + bi0 + __uint_as_float(addv[r][0]);
                float x1 = acc[i][1][r] * sc1 + bi1 + __uint_as_float(addv[r][1]);
                // rows past M / couts past Cout read a zero mask: they add nothing to the sums and their stores are dropped
                x0 = __uint_as_float(mk << 16) > 0.f ? x0 : 0.f;
                x1 = __uint_as_float(mk & 0xffff0000u) > 0.f ? x1 : 0.f;
                gs0 += x0;
                gs1 += x1;
                const bool ok = cok && rbase + rr < p.M;
                const unsigned e = row_e(i, r);
                const u32x2 v = {__builtin_bit_cast(unsigned, x0), __builtin_bit_cast(unsigned, x1)};
                __builtin_amdgcn_raw_buffer_store_b64(v, rs32, (int)(ok ? e * 4u : 0x80000000u), 0, 0);
                const bf16x2 pk = {(__bf16)x0, (__bf16)x1};
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pk), rs16, (int)(ok ? e * 2u : 0x80000000u), 0, 0);
            }
        }
        gs0 += __shfl_xor(gs0, 32, 64);
        gs1 += __shfl_xor(gs1, 32, 64);
        if (half == 0 && cok) {
            const f32x4 v = {gs0, 0.f, gs1, 0.f};
            *reinterpret_cast<f32x4*>(p.gn_part + ((size_t)(tm * 2 + wm) * p.Cout + c0) * 2) = v;
        }
        return;
    }
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
        p.out, 0, (p.ablate & 16) ? 0 : (int)((size_t)p.M * p.Cout * (p.out_fp32 ? 4 : 2)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(p.residual ? p.residual : (const unsigned short*)p.out), 0,
        (int)((size_t)p.M * p.Cout * 2), 0x00020000);
    const int c0 = n0 + wn * 64 + 2 * l31;                       // couts c0 (block 0) and c0 + 1 (block 1); Cout is even
    const bool cok = c0 < p.Cout;
    const int cc = cok ? c0 : p.Cout - 2;
    const float sc0 = p.scale ? p.scale[cc] : 1.f, sc1 = p.scale ? p.scale[cc + 1] : 1.f;
    const float bi0 = p.bias ? p.bias[cc] : 0.f, bi1 = p.bias ? p.bias[cc + 1] : 0.f;
    float gs0 = 0.f, gq0 = 0.f, gs1 = 0.f, gq1 = 0.f;
    unsigned resbuf[PRE ? 2 : 1][16];
    auto load_res = [&](int i, unsigned (&dst)[16]) {
        const unsigned e0 = (unsigned)((m0 + wm * (MI * 32) + i * 32 + 4 * half) * p.Cout + c0);
        if (p.residual) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                dst[r] = __builtin_amdgcn_raw_buffer_load_b32(
                    rs_res, (int)(cok ? (e0 + (unsigned)((r & 3) + 8 * (r >> 2)) * (unsigned)p.Cout) * 2u : 0x80000000u), 0, 0);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[r] = 0u;
        }
    };
    if (PRE) load_res(0, resbuf[0]);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int rbase = m0 + wm * (MI * 32) + i * 32 + 4 * half;
        const unsigned e0 = (unsigned)(rbase * p.Cout + c0);
        if (PRE) { if (i + 1 < MI) load_res(i + 1, resbuf[PRE ? (i + 1) & 1 : 0]); }
        else load_res(i, resbuf[0]);
        unsigned (&res)[16] = resbuf[PRE ? i & 1 : 0];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = (r & 3) + 8 * (r >> 2);
            float x0 = acc[i][0][r] * sc0 + bi0, x1 = acc[i][1][r] * sc1 + bi1;
            const float r0 = __uint_as_float(res[r] << 16), r1 = __uint_as_float(res[r] & 0xffff0000u);
            if (p.relu == 2) {
                x0 = r0 > 0.f ? x0 : 0.f;
                x1 = r1 > 0.f ? x1 : 0.f;
            } else {
                x0 += r0;
                x1 += r1;
                if (p.relu) {
                    x0 = fmaxf(x0, 0.f);
                    x1 = fmaxf(x1, 0.f);
                }
            }
            if (p.gn_part) {
                const bool ok = cok && rbase + rr < p.M;
                const float u0 = ok ? x0 : 0.f, u1 = ok ? x1 : 0.f;
                gs0 += u0; gq0 += u0 * u0;
                gs1 += u1; gq1 += u1 * u1;
            }
            const unsigned e = e0 + (unsigned)rr * (unsigned)p.Cout;
            if (p.out_fp32) {
                const u32x2 v = {__builtin_bit_cast(unsigned, x0), __builtin_bit_cast(unsigned, x1)};
                __builtin_amdgcn_raw_buffer_store_b64(v, rs_out, (int)(cok ? e * 4u : 0x80000000u), 0, 0);
            } else {
                const bf16x2 pk = {(__bf16)x0, (__bf16)x1};
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pk), rs_out, (int)(cok ? e * 2u : 0x80000000u), 0, 0);
            }
        }
    }
    if (p.gn_part) {
        gs0 += __shfl_xor(gs0, 32, 64); gq0 += __shfl_xor(gq0, 32, 64);
        gs1 += __shfl_xor(gs1, 32, 64); gq1 += __shfl_xor(gq1, 32, 64);
        if (half == 0 && cok) {
            const f32x4 v = {gs0, gq0, gs1, gq1};
            *reinterpret_cast<f32x4*>(p.gn_part + ((size_t)(tm * 2 + wm) * p.Cout + c0) * 2) = v;
        }
    }
}

// bf16 output through LDS (round 4).  The direct epilogue above stores cout PAIRS per lane: one buffer_store_b32 moves two
// 64-byte row segments, 128 of them (+ 128 two-byte residual loads) per wave -- measured on R101's layer3 conv3 (1x1 256 -> 1024
// on 32 768 pixels): 0.044 ms with, 0.016 ms without the epilogue, 0.038 ms with every store dropped by the range check, i.e. the
// INSTRUCTIONS, not the bytes (profiles/round4_bf16_epilogue_ablation.txt).  Here the fp32 results (acc * scale + bias) of half a
// tile go to the LDS the K loop has finished with (row pitch = BN floats: a wave's D-layout write is two conflict-free 128-byte
// runs), and the workgroup reads them back ROW-wise: lane l owns couts [4 l, 4 l + 4) of a row -- one ds_read_b128, one 8-byte
// residual load (requested before the tile's writes, consumed behind the barrier), fp32 add + ReLU, one 8-byte store; a wave
// instruction covers whole 512-byte (BN = 256) or 2 x 256-byte rows.  16 x fewer vector-memory instructions, every line written
// once and whole.  Same arithmetic and rounding as the direct form (fp32 through the residual add, one RNE rounding).
template <int MI, int NJ>
__device__ __forceinline__ void dma_epilogue_lds(const ConvDmaParams& p, f32x16 (&acc)[MI][NJ], unsigned char* smem, int tm,
                                                 int m0, int n0, int wm, int wn, int wave, int lane) {
    constexpr int BN = 128 * NJ, HI = MI / 2;                  // HI pixel blocks per wave per half tile
    constexpr int LPR = BN / 4, RPI = 64 / LPR;                // lanes per row, rows per wave instruction of the read-out
    constexpr int HROWS = 32 * MI, PASSES = HROWS / (8 * RPI); // rows per half tile, read-out instructions per wave per half
    static_assert(MI % 2 == 0 && HROWS * BN * 4 <= 2 * DmaTile<MI, NJ, 4>::STAGE, "half a tile of fp32 must fit the two stages");
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const int l31 = lane & 31, half = lane >> 5;
    float* tile = reinterpret_cast<float*>(smem);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
        p.out, 0, (p.ablate & 16) ? 0 : (int)((size_t)p.M * p.Cout * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(p.residual ? p.residual : (const unsigned short*)p.out), 0,
        (int)((size_t)p.M * p.Cout * 2), 0x00020000);
    float sc[NJ], bi[NJ], gsum[NJ], gsq[NJ];
    bool cok[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = n0 + wn * (NJ * 32) + j * 32 + l31;
        cok[j] = c < p.Cout;
        const int cc = cok[j] ? c : p.Cout - 1;
        sc[j] = p.scale ? p.scale[cc] : 1.f;
        bi[j] = p.bias ? p.bias[cc] : 0.f;
        gsum[j] = 0.f;
        gsq[j] = 0.f;
    }
    const bool relu_here = p.relu && !p.residual;              // with a residual the ReLU follows the add, in the read-out
    const bool mask_mode = p.relu == 2;                        // (the residual is a ReLU mask source; column sums in the read-out)
    float csum[4] = {0.f, 0.f, 0.f, 0.f};
    const int rl = lane / LPR, cl = (lane % LPR) * 4;
    __syncthreads();                                           // every wave's last fragment reads are done: the stages are free
    // (the row arithmetic hangs on an opaque copy of the wave's first row: computed where it is used -- hoisted above the tile's
    // LDS writes, the 2 x PASSES offsets of both halves were spilled and re-read one by one: 10 us per tile)
    // read-out role: local row lr = wm' (32 HI) + ii 32 + rr of half h  ->  pixel m0 + wm' (32 MI) + (h HI + ii) 32 + rr
    auto row_offset = [&](int h, int ps, int wrow) {
        const int lr = ps * 8 * RPI + wrow;
        const int m = m0 + (lr / (HI * 32)) * (MI * 32) + h * (HI * 32) + (lr % (HI * 32));
        return (m < p.M && n0 + cl < p.Cout) ? (int)(((unsigned)m * (unsigned)p.Cout + (unsigned)(n0 + cl)) * 2u) : (int)0x80000000;
    };
    auto write_half = [&](auto hc) {
        constexpr int h = decltype(hc)::value;
#pragma unroll
        for (int ii = 0; ii < HI; ++ii) {
            const int i = h * HI + ii;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float* dst = tile + (wm * (HI * 32) + ii * 32 + 4 * half) * BN + wn * (NJ * 32) + j * 32 + l31;
                const int mb = m0 + wm * (MI * 32) + i * 32 + 4 * half;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = (r & 3) + 8 * (r >> 2);
                    float x = acc[i][j][r] * sc[j] + bi[j];
                    if (relu_here) x = fmaxf(x, 0.f);
                    if (p.gn_part) {
                        const float u = (cok[j] && mb + rr < p.M) ? x : 0.f;
                        gsum[j] += u;
                        gsq[j] += u * u;
                    }
                    dst[rr * BN] = x;
                }
            }
        }
    };
    auto read_half = [&](int h, u32x2 (&rv)[PASSES]) {
        int wrow = wave * RPI + rl;
        asm volatile("" : "+v"(wrow));
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int lr = ps * 8 * RPI + wrow;
            f32x4 v = *reinterpret_cast<const f32x4*>(tile + lr * BN + cl);
            if (p.residual) {
                const float r0 = __uint_as_float(rv[ps][0] << 16), r1 = __uint_as_float(rv[ps][0] & 0xffff0000u);
                const float r2 = __uint_as_float(rv[ps][1] << 16), r3 = __uint_as_float(rv[ps][1] & 0xffff0000u);
                if (mask_mode) {            // rows past M / couts past Cout read a zero mask: they add nothing to the sums
                    v[0] = r0 > 0.f ? v[0] : 0.f;
                    v[1] = r1 > 0.f ? v[1] : 0.f;
                    v[2] = r2 > 0.f ? v[2] : 0.f;
                    v[3] = r3 > 0.f ? v[3] : 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) csum[e] += v[e];
                } else {
                    v[0] += r0;
                    v[1] += r1;
                    v[2] += r2;
                    v[3] += r3;
                    if (p.relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                }
            }
            u32x2 pk;
            pk[0] = __builtin_bit_cast(unsigned, bf16x2{(__bf16)v[0], (__bf16)v[1]});
            pk[1] = __builtin_bit_cast(unsigned, bf16x2{(__bf16)v[2], (__bf16)v[3]});
            __builtin_amdgcn_raw_buffer_store_b64(pk, rs_out, row_offset(h, ps, wrow), 0, 0);
        }
    };
    u32x2 rv0[PASSES], rv1[PASSES];
    write_half(std::integral_constant<int, 0>{});
    // the residual of BOTH halves is requested here: the first half's accumulators have just died (their registers carry the
    // 2 x PASSES x 8 bytes), and the second half's requests fly during the first half's read-out
    if (p.residual) {
        int wrow = wave * RPI + rl;
        asm volatile("" : "+v"(wrow));
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) rv0[ps] = __builtin_amdgcn_raw_buffer_load_b64(rs_res, row_offset(0, ps, wrow), 0, 0);
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) rv1[ps] = __builtin_amdgcn_raw_buffer_load_b64(rs_res, row_offset(1, ps, wrow), 0, 0);
    }
    __syncthreads();
    read_half(0, rv0);
    __syncthreads();
    write_half(std::integral_constant<int, 1>{});
    __syncthreads();
    read_half(1, rv1);
    if (p.gn_part && mask_mode) {
        // column sums of the masked tile, in a fixed order (the step stays bit-repeatable): lane pairs, then the eight waves through
        // the LDS behind the half tile, one slot per tile (BM pixels)
        static_assert(HROWS * BN * 4 + 8 * BN * 4 <= 2 * DmaTile<MI, NJ, 4>::STAGE, "room for the waves' column sums");
        float* sums = tile + HROWS * BN;
        if (RPI == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) csum[e] += __shfl_xor(csum[e], 32, 64);
        }
        if (lane < LPR) *reinterpret_cast<f32x4*>(sums + wave * BN + cl) = f32x4{csum[0], csum[1], csum[2], csum[3]};
        __syncthreads();
        const int t = wave * 64 + lane;
        if (t < BN && n0 + t < p.Cout) {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) a += sums[w * BN + t];
            p.gn_part[((size_t)tm * p.Cout + n0 + t) * 2] = a;
        }
    } else if (p.gn_part) {     // (MI == 4 only: the launcher keeps statistics layers off the 128-pixel tile)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float a = gsum[j], b = gsq[j];
            a += __shfl_xor(a, 32, 64);
            b += __shfl_xor(b, 32, 64);
            const int c = n0 + wn * (NJ * 32) + j * 32 + l31;
            if (half == 0 && cok[j]) {
                float* dst = p.gn_part + ((size_t)(tm * 2 + wm) * p.Cout + c) * 2;
                dst[0] = a;
                dst[1] = b;
            }
        }
    }
}
