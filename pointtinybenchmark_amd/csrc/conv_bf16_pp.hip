// bf16 implicit-GEMM convolution, 256 x 256 x 64 tiles, TWO WAVE GROUPS IN PING-PONG (round 6).
//
// conv_bf16_dma_kernel<4, 2, 4> (conv_bf16_dma.hip) runs its eight waves in lock step: every wave interleaves its 32 MFMAs of a
// K chunk with 8 LDS-DMA requests and 24 fragment reads, and its time is MFMA time PLUS request time (round 5: 1.75 ms on the head
// layer against 1.09 ms with no vector-memory request at all).  Here the two waves of a SIMD never do the same thing at the same time:
//   * group g = wave >> 2 (waves w and w + 4 share a SIMD) owns the tile's pixels [128 g, +128); wave wq = wave & 3 its couts
//     [64 wq, +64): 4 x 2 blocks of 32 x 32 as before (same accumulators, same k order: results are BIT-EQUAL to <4, 2, 4>);
//   * a K chunk is four INTERVALS separated by workgroup barriers.  In each interval one group runs a cluster of 16 MFMAs whose
//     operands are already in registers while the other one does its memory part, and group 1 runs one interval behind group 0
//     (one extra barrier up front):
//         interval        4 kt            4 kt + 1        4 kt + 2        4 kt + 3
//         group 0         La(kt)          Ma(kt)          Lb(kt)          Mb(kt)
//         group 1         Mb(kt - 1)      La(kt)          Ma(kt)          Lb(kt)
//     The clusters split the chunk by K: Ma = k-steps 0, 1 of all 4 x 2 blocks, Mb = k-steps 2, 3; both memory parts are
//     12 ds_read_b128 (4 pixel + 2 cout fragments per k-step) + 4 requests.
// Four schedules of this were built (profiles/round6_bf16_pp_development.txt); the first three -- requests and their address
// arithmetic in the memory parts, or requests between the MFMAs of the clusters -- were no faster than the lock-step kernel, and
// two microbenchmarks said why (tools/diag/mfma_dma_mix.hip, dma_stream.hip; profiles/round6_mfma_dma_mix.txt, round6_dma_stream.txt):
// a wave that issues 4 LDS-DMA requests + 12 ds_read_b128 per iteration beside a partner streaming MFMAs needs 447 cycles per
// iteration and costs the partner NOTHING (32.0 cycles per MFMA; the request stream alone runs at the texture path's 64 B/clk/CU) --
// but every VECTOR-ALU instruction in that wave waits ~430 cycles: the MFMA stream holds the SIMD's vector issue port (twelve
// four-wide v_add turned the 447 cycles into 20 800).  Tap cursor, row offsets, "past the last chunk" selects and read addresses
// (~50 VALU per chunk) had sat in the memory parts, which therefore ran AFTER the partner's cluster, not beside it.  So here
//   * the memory parts hold NO vector-ALU instruction (checked in the ISA: SALU, DS and VMEM only between the barriers):
//         La(kt)  12 reads (k-steps 0, 1), 4 requests B(kt + 1)                                  | barrier
//         Ma(kt)  16 MFMAs                                                                      | barrier
//         Lb(kt)  12 reads (k-steps 2, 3), 4 requests A_g(kt + 2), vmcnt(4)                     | barrier
//         Mb(kt)  16 MFMAs + ALL vector arithmetic of the chunk between its own MFMAs: the row offsets of A_g(kt + 3) and the eight
//                 read addresses of chunk kt + 1 (pinned there with empty asm: the compiler otherwise sinks them to their use) | barrier
//     reads before requests: the reads' latency then runs under the requests' issue (measured 6 % against the other order);
//   * "past the last chunk" is a SCALAR select of the buffer descriptor's size word (0 records: every lane out of range: zeros,
//     no traffic), so the request stream of a wave is the same 8 requests per chunk in a fixed order and one counted wait per chunk
//     retires exactly what the next chunk's reads need; every request has 2 (cout rows, L2-resident) to 4 intervals to land;
//   * LDS: pixel rows 3 stages (A_g(kt + 2) is requested while Lb(kt) still reads A_g(kt)), cout rows 2 stages = all 160 KB;
//     source-side swizzle and row layout as the lock-step kernel (0 bank-conflict cycles).
// Measured (same box, head layer 3x3 256 -> 256 + GN statistics on (64,160,160,256)): 1.56 .. 1.61 ms = 1200 .. 1240 TF against
// 1.72 .. 1.77 ms of the lock-step and weights-direct instances; ablations of THIS schedule: no requests 1.24, no reads 1.28, neither
// 1.14, requests that hit the vector L1 1.33 -- what is left is the memory system behind the requests (issue back-pressure, not
// landing: dropping the counted wait changes nothing) and 4 barriers per chunk.  Short-K layers (< 4 chunks) stay on the
// weights-direct instance, whose epilogue-side wins carry them.
#include "conv_bf16_dma.h"

namespace {
constexpr int PP_A_BYTES = 256 * DBK * 2;          // one stage of pixel (or cout) rows: 256 rows of 128 bytes
}

template <bool PRIO, bool PRE, int ABL = 0, bool STAMPS = false, bool TR = false>
__global__ __launch_bounds__(512, 2) void conv_bf16_pp_kernel(ConvDmaParams p) {
    constexpr int MI = 4, NJ = 2, DBM = 256, DBN = 256;
    constexpr int B_BASE = 3 * PP_A_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[5 * PP_A_BYTES];

    const bool nt = p.nt_taps > 0;
    const int TMN = p.tilesM * p.tilesN;
    int tile, st = 0, nt_split = 0, nt_tap = 0;
    if (nt) {
        const int idx = blockIdx.x >> 3, per_split = p.nt_taps * TMN;
        nt_split = (idx / per_split) * 8 + (blockIdx.x & 7);
        const int rem = idx % per_split;
        nt_tap = rem / TMN;
        tile = rem - nt_tap * TMN;
        st = nt_split * p.nt_taps + nt_tap;
    } else {
        const int per = (TMN + 7) >> 3;
        tile = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
        if (tile >= TMN) return;
    }
    const int tm = tile / p.tilesN, tn = tile - tm * p.tilesN;
    const int m0 = tm * DBM, n0 = tn * DBN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 2, wq = wave & 3;

    // ---- staging role: pixel and cout pieces z = 0..3 = tile rows 128 g + 32 wq + 8 z + (lane >> 3), swizzle by piece parity
    const int prow = lane >> 3;
    const int sunit0 = (lane & 7) ^ (lane >> 4), sunit1 = sunit0 ^ 4;
    const int ohw = p.OH * p.OW;
    const bool gemm = (p.KH == 1) & (p.KW == 1) & (p.stride == 1) & (p.pad == 0);
    int iy0[4], ix0[4], rowoff[4];
    bool mok[4];
#pragma unroll
    for (int z = 0; z < 4; ++z) {
        const int m = m0 + 128 * g + 32 * wq + 8 * z + prow;
        mok[z] = m < p.M;
        const int mm = mok[z] ? m : 0;
        if (gemm) {
            iy0[z] = 0; ix0[z] = 0;
            rowoff[z] = mm * p.Cin;
        } else {
            const int n = mm / ohw;
            const int rem = mm - n * ohw;
            const int oy = rem / p.OW, ox = rem - oy * p.OW;
            iy0[z] = oy * p.stride - p.pad;
            ix0[z] = ox * p.stride - p.pad;
            rowoff[z] = ((n * p.H + iy0[z]) * p.W + ix0[z]) * p.Cin;
        }
    }
    const size_t in_addr = (size_t)p.in;
    const size_t w_addr = (size_t)(p.wgt + (nt ? (long long)(nt_tap % p.nt_k) * p.nt_copy + (nt_tap / p.nt_k - p.nt_pad) * p.nt_Wp : 0));
    const int in_bytes = (int)((size_t)p.N * p.H * p.W * p.Cin * 2), w_bytes = (int)((size_t)p.Cout * p.Kpad * 2);
    int woff[4];
#pragma unroll
    for (int z = 0; z < 4; ++z) {
        const int rr = 128 * g + 32 * wq + 8 * z + prow;
        const int c = n0 + (rr & ~63) + 2 * (rr & 31) + ((rr >> 5) & 1);        // interleaved cout layout of dma_epilogue_pairs
        woff[z] = c < p.Cout ? (c * p.Kpad) * 2 + ((z & 1) ? sunit1 : sunit0) * 16 : (int)0x80000000;
    }
    const int kbase = nt ? nt_split * p.nt_chunks * DBK : 0;
    int kh = 0, kw = 0, c0 = kbase;         // pixel cursor: tap / channel chunk of the chunk whose row offsets voffA holds
    int btap = 0, bc0 = kbase;              // cout cursor
    int voffA[4];
    auto row_voff = [&](int z, int kh_, int kw_) {
        const int iy = iy0[z] + kh_, ix = ix0[z] + kw_;
        const bool ok = mok[z] & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
        return ok ? (rowoff[z] + (kh_ * p.W + kw_) * p.Cin) * 2 + ((z & 1) ? sunit1 : sunit0) * 16 : (int)0x80000000;
    };
    auto next_a = [&]() { if (++kw == p.KW) { kw = 0; if (++kh == p.KH) { kh = 0; c0 += DBK; } } };      // scalar
    auto next_b = [&]() { if (++btap == p.KH * p.KW) { btap = 0; bc0 += DBK; } };
    const int lds0 = (int)(unsigned)(size_t)smem;
    // one request: everything but the per-lane offset register is scalar; `bytes` = the descriptor's size word (0 = past the last chunk)
    auto dma = [&](size_t base, int bytes, int voff, int soff, int lds_byte) {
        const i32x4v rs = {(int)(unsigned)base, (int)(unsigned)(base >> 32) & 0xffff, bytes, 0x00020000};
        asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:0 lds"
                     :: "s"(lds_byte), "v"(voff), "s"(rs), "s"(soff) : "memory");
    };
    const int KT = nt ? p.nt_chunks : p.Kpad / DBK;
    const int ldsA = lds0 + (128 * g + 32 * wq) * 128, ldsB = lds0 + B_BASE + (128 * g + 32 * wq) * 128;
    const int same_kb = lane * 16;
    auto req_a = [&](int stage, int z, bool ok) {
        if (ABL & 4) return;
        if (ABL & 128) { dma(in_addr, in_bytes, same_kb, 0, ldsA + stage * PP_A_BYTES + z * 1024); return; }
        dma(in_addr, ok ? in_bytes : 0, voffA[z], c0 * 2, ldsA + stage * PP_A_BYTES + z * 1024);
    };
    auto req_b = [&](int stage, int z, bool ok) {
        if (ABL & 4) return;
        if (ABL & 128) { dma(w_addr, w_bytes, same_kb, 0, ldsB + stage * PP_A_BYTES + z * 1024); return; }
        dma(w_addr, ok ? w_bytes : 0, woff[z], (btap * p.Cin + bc0) * 2, ldsB + stage * PP_A_BYTES + z * 1024);
    };

    // ---- MFMA role
    const int l31 = lane & 31, half = lane >> 5;
    const int rswz = (l31 >> 1) & 7;
    int koff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) koff[kk] = l31 * 128 + (((2 * kk + half) ^ rswz) * 16);
    const int a_rd = (128 * g) * 128, b_rd = B_BASE + (64 * wq) * 128;          // byte offsets of the wave's fragment rows in a stage

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 fa[2][4], fb[2][2];           // [k-step of the cluster][block]
    int ra[4], rb[4];                   // LDS byte offsets of the lane's fragment rows for the four k-steps of the chunk read next

    // STAMPS (measurement build, its own instance): phase clocks of chunk 8 of workgroup 16, written over that workgroup's own
    // output rows AFTER its epilogue (the accumulators stay live: a clock instance without the epilogue loses its MFMAs)
    unsigned long long ts[STAMPS ? 12 : 1];
#pragma unroll
    for (int i = 0; i < (STAMPS ? 12 : 1); ++i) ts[i] = 0;
    const bool stamping = STAMPS && blockIdx.x == 16;
#undef PP_STAMP
#define PP_STAMP(i) do { if constexpr (STAMPS) { if (stamping && kt == 8) ts[i] = __builtin_amdgcn_s_memtime(); } } while (0)
    auto reads = [&](int h) {            // the 12 fragment reads of k-steps 2 h, 2 h + 1
        if (ABL & 8) return;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[kk][j] = *reinterpret_cast<const f32x4*>(smem + rb[2 * h + kk] + j * 4096);
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[kk][i] = *reinterpret_cast<const f32x4*>(smem + ra[2 * h + kk] + i * 4096);
        }
    };
#define PP_MFMA(q)                                                                                                     \
    acc[((q) >> 1) & 3][(q) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                                             \
        __builtin_bit_cast(bf16x8, fa[(q) >> 3][((q) >> 1) & 3]), __builtin_bit_cast(bf16x8, fb[(q) >> 3][(q) & 1]),    \
        acc[((q) >> 1) & 3][(q) & 1], 0, 0, 0)
#define PP_MEM_END(N, s0)                                                                                              \
    do {                                                                                                                \
        PP_STAMP(s0);                                                                                                   \
        if ((N) >= 0 && !(ABL & 64)) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((N) >= 0 ? (N) : 0) : "memory");        \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        PP_STAMP(s0 + 1);                                                                                               \
        __builtin_amdgcn_s_barrier();                                                                                   \
        asm volatile("" ::: "memory");                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
    } while (0)
#define PP_CLUSTER_END(cond)                                                                                           \
    do {                                                                                                                \
        if (PRIO) __builtin_amdgcn_s_setprio(0);                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        if (cond) __builtin_amdgcn_s_barrier();                                                                         \
        asm volatile("" ::: "memory");                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
    } while (0)

    // prologue: A(0), B(0), A(1) (La(0) requests B(1)); afterwards the pixel cursor and voffA point at chunk 2, the cout cursor at chunk 1
#pragma unroll
    for (int z = 0; z < 4; ++z) voffA[z] = row_voff(z, 0, 0);
#pragma unroll
    for (int z = 0; z < 4; ++z) req_a(0, z, true);
#pragma unroll
    for (int z = 0; z < 4; ++z) req_b(0, z, true);
    next_b();
    next_a();
#pragma unroll
    for (int z = 0; z < 4; ++z) voffA[z] = row_voff(z, kh, kw);
#pragma unroll
    for (int z = 0; z < 4; ++z) req_a(1, z, KT > 1);
    next_a();
#pragma unroll
    for (int z = 0; z < 4; ++z) voffA[z] = row_voff(z, kh, kw);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { ra[kk] = a_rd + koff[kk]; rb[kk] = b_rd + koff[kk]; }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) asm volatile("" : "+v"(ra[kk]), "+v"(rb[kk]), "+v"(voffA[kk]));
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __syncthreads();
    if (g == 1) __builtin_amdgcn_s_barrier();        // group 1 runs one interval behind
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    int sa = 0, sa_req = 2;            // pixel-row stage (of three) of the chunk being read / of the chunk requested next
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < KT, more2 = kt + 2 < KT;        // wave-uniform
        // La: no vector ALU instruction from here to the barrier
        PP_STAMP(0);
        if (!(ABL & 32)) reads(0);                                   // reads first: their latency runs under the requests' issue (ABL & 32: the other order)
#pragma unroll
        for (int z = 0; z < 4; ++z) req_b(buf ^ 1, z, more);
        next_b();
        if (ABL & 32) reads(0);
        PP_MEM_END(-1, 1);
        // Ma
        PP_STAMP(3);
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int q = 0; q < 16; ++q) PP_MFMA(q);
        PP_STAMP(4);
        PP_CLUSTER_END(true);
        PP_STAMP(5);
        // Lb: no vector ALU instruction from here to the barrier
        if (!(ABL & 32)) reads(1);
#pragma unroll
        for (int z = 0; z < 4; ++z) req_a(sa_req, z, more2);
        next_a();
        if (ABL & 32) reads(1);
        PP_MEM_END(4, 6);
        PP_STAMP(8);
        // Mb: + the row offsets of A_g(kt + 3) and the read addresses of chunk kt + 1
        if (PRIO) __builtin_amdgcn_s_setprio(1);
        sa = sa == 2 ? 0 : sa + 1;
        sa_req = sa_req == 2 ? 0 : sa_req + 1;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            PP_MFMA(q);
            if ((q & 3) == 1) {
                const int z = q >> 2;
                if (!(ABL & 16)) voffA[z] = row_voff(z, kh, kw);         // (a 1x1 layer recomputes the same value: no branch inside the cluster)
                ra[z] = a_rd + sa * PP_A_BYTES + koff[z];
                rb[z] = b_rd + (buf ^ 1) * PP_A_BYTES + koff[z];
                asm volatile("" : "+v"(ra[z]), "+v"(rb[z]), "+v"(voffA[z]));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        PP_STAMP(9);
        PP_CLUSTER_END(g == 0 || more);                              // group 1's last cluster has no partner left to wait for
        PP_STAMP(10);
    }
#undef PP_MFMA
#undef PP_MEM_END
#undef PP_CLUSTER_END
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the requests past the last chunk must land before the LDS is released
    if (nt) {   // this (split, tap)'s fp32 partial
        ConvDmaParams q = p;
        q.out = (float*)p.out + (size_t)st * p.M * p.Cout;
        dma_epilogue_pairs<MI, PRE>(q, acc, tm, m0, n0, g, wq, lane);
        return;
    }
    dma_epilogue_pairs<MI, PRE, TR>(p, acc, tm, m0, n0, g, wq, lane);       // (TR: the training epilogue, conv_bf16_dma.h)
    if constexpr (STAMPS) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (stamping && (wave & 3) == 0 && lane == 0) {
            unsigned long long* dst = reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned short*>(p.out) + (size_t)m0 * p.Cout) + g * 16;
#pragma unroll
            for (int i = 0; i < 12; ++i) dst[i] = ts[i];
        }
    }
}

// called by conv_bf16_dma_launch (shape 5).  Measurement build: p.ablate selects a compile-time ablated instance (results are then
// WRONG except for bit 8): bit 2 no requests, 3 no fragment reads, 7 no row-offset arithmetic, 8 requests before the reads, 0 no
// counted wait, 1 every request the same KB (vector-L1 hits), 10 phase clocks of one chunk over the workgroup's own output rows
void conv_bf16_pp_launch(const ConvDmaParams& p, unsigned grid, hipStream_t stream) {
#ifdef CPR_BENCH_HOOKS
#define PP_GO(A_, S_) hipLaunchKernelGGL((conv_bf16_pp_kernel<true, true, A_, S_>), dim3(grid), dim3(512), 0, stream, p)
    const int abl = ((p.ablate >> 2) & 3) * 4 | ((p.ablate >> 7) & 3) * 16 | ((p.ablate & 1) ? 64 : 0) | ((p.ablate & 2) ? 128 : 0);
    const bool stamps = (p.ablate & 1024) != 0;
    if (stamps) { if (abl == 4) PP_GO(4, true); else if (abl == 8) PP_GO(8, true); else if (abl == 12) PP_GO(12, true); else PP_GO(0, true); return; }
    switch (abl) {
    case 4: PP_GO(4, false); return;
    case 8: PP_GO(8, false); return;
    case 12: PP_GO(12, false); return;
    case 16: PP_GO(16, false); return;
    case 32: PP_GO(32, false); return;
    case 64: PP_GO(64, false); return;
    case 128: PP_GO(128, false); return;
    default: break;
    }
#undef PP_GO
#endif
    if (p.add32) hipLaunchKernelGGL((conv_bf16_pp_kernel<true, true, 0, false, true>), dim3(grid), dim3(512), 0, stream, p);
    else hipLaunchKernelGGL((conv_bf16_pp_kernel<true, true>), dim3(grid), dim3(512), 0, stream, p);
}
