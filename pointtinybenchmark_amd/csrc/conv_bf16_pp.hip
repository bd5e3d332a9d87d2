// bf16 implicit-GEMM convolution, 256 x 256 x 64 tiles, TWO WAVE GROUPS IN PING-PONG (round 6).
//
// conv_bf16_dma_kernel<4, 2, 4> (conv_bf16_dma.hip) runs its eight waves in lock step: every wave interleaves its 32 MFMAs of a
// K chunk with 8 LDS-DMA requests and 24 fragment reads, and the round-5 ablations showed the kernel's time to be MFMA time PLUS
// request time (1.75 ms on the head layer against 1.09 ms with no vector-memory request at all): the two waves of a SIMD hand their
// requests to the texture path in the same slots and the matrix pipe waits behind them.  Here the two waves of a SIMD never do the
// same thing at the same time:
//   * group g = wave >> 2 (waves w and w + 4 share a SIMD) owns the tile's pixels [128 g, +128); wave wq = wave & 3 its couts
//     [64 wq, +64): 4 x 2 blocks of 32 x 32 as before (same accumulators, same k order: results are bit-equal to <4, 2, 4>);
//   * a K chunk is four INTERVALS separated by workgroup barriers.  In each interval one group runs a pure cluster of 16 MFMAs
//     (all operands already in registers, s_setprio 1) while the other one does its memory part -- fragment reads for its next
//     cluster + LDS-DMA requests for later chunks -- and group 1 runs one interval behind group 0 (one extra barrier up front):
//         interval        4 kt            4 kt + 1        4 kt + 2        4 kt + 3
//         group 0         La(kt)          Ma(kt)          Lb(kt)          Mb(kt)
//         group 1         Mb(kt - 1)      La(kt)          Ma(kt)          Lb(kt)
//     La = read the cout fragments of all four k-steps (8 ds_read_b128, kept for both clusters) + pixel blocks 0, 1 (8);
//     Ma = blocks {0, 1} x {0, 1} x 4 k-steps; Lb = pixel blocks 2, 3 (8 reads); Mb = blocks {2, 3} x {0, 1} x 4 k-steps.
//   * staging keeps the two 64 KB stages and the source-side swizzle of the lock-step kernel; what makes two stages enough is
//     that the pieces of a stage die at different times.  The cout rows B(kt) are last read in interval 4 kt + 1, a group's own
//     pixel rows A_g,lo(kt) (blocks 0, 1) in its La(kt), A_g,hi(kt) in its Lb(kt) -- so
//         Lb(kt) requests B(kt + 2) (the group's half of the cout rows: 4 pieces per wave) and A_g,lo(kt + 2) (2 pieces),
//         La(kt) requests A_g,hi(kt + 1) (2 pieces),
//     every request has four intervals (~1 us) to land, a wave's request stream is the same 8 requests per chunk in a fixed order
//     (past the last chunk with an out-of-range vector offset: zeros, no traffic), and ONE counted wait, vmcnt(8) at the end of
//     every memory part, retires exactly what the reads two intervals later need.
// One workgroup per CU (128 KB of LDS), 512 threads, two waves per SIMD.
#include "conv_bf16_dma.h"

#ifdef CPR_BENCH_HOOKS
#define PP_ABL(b) (p.ablate & (b))      // measurement build: loop ablations (results are then WRONG), as in conv_bf16_dma_kernel
#else
#define PP_ABL(b) 0
#endif
namespace {
constexpr int PP_A_BYTES = 256 * DBK * 2;          // the pixel rows of a stage (rows of 128 bytes), then the cout rows
constexpr int PP_STAGE = 2 * PP_A_BYTES;
}

// KSPLIT = false: the schedule described above.  KSPLIT = true (built second, same visit): the clusters split the chunk by K
// instead of by pixel block -- Ma = k-steps 0, 1 of all 4 x 2 blocks, Mb = k-steps 2, 3 -- so that BOTH memory parts are 12
// fragment reads + 4 requests (above: 16 + 2 and 8 + 6).  Every row is then read in both memory parts, the cout rows B(kt) die
// only in interval 4 kt + 3, and two stages of pixel rows would leave A(kt + 2) one interval to land; with THREE stages of the
// group-private pixel rows (3 x 32 KB) + two of the cout rows (2 x 32 KB) = all 160 KB of LDS:
//     La(kt) requests B(kt + 1) (4 pieces per wave; two intervals to land -- the weights are L2-resident),
//     Lb(kt) requests A_g(kt + 2) (4 pieces; six intervals), one counted wait, vmcnt(4) at the end of Lb.
// ORDER: 0 = requests first, then the fragment reads; 1 = reads first.  PRE: residual prefetch in the epilogue (as the BD instance).
template <bool KSPLIT, int ORDER, bool PRE>
__global__ __launch_bounds__(512, 2) void conv_bf16_pp_kernel(ConvDmaParams p) {
    constexpr int MI = 4, NJ = 2, DBM = 256, DBN = 256;
    constexpr int A_STAGES = KSPLIT ? 3 : 2;
    // KSPLIT: [3][pixel rows] then [2][cout rows]; else [2][pixel rows | cout rows]
    constexpr int A_STRIDE = KSPLIT ? PP_A_BYTES : PP_STAGE, B_STRIDE = KSPLIT ? PP_A_BYTES : PP_STAGE;
    constexpr int B_BASE = KSPLIT ? 3 * PP_A_BYTES : PP_A_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[(A_STAGES + 2) * PP_A_BYTES];

    // tile order: as conv_bf16_dma_kernel (block b runs on XCD b % 8; an XCD walks a contiguous run of tiles, cout tiles fastest)
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

    // ---- staging role.  A request piece = 8 rows of 128 bytes (lane = 16-byte unit lane & 7 of row lane >> 3); every piece starts
    // at a multiple of 8 rows, so the swizzle (row >> 1) & 7 of a lane's row is (lane >> 4) + 4 (piece parity).
    //   pixel pieces z = 0..3: tile rows 128 g + 64 (z >> 1) + 16 wq + 8 (z & 1) + (lane >> 3)   (z < 2: blocks 0, 1 = "lo");
    //                          KSPLIT: 128 g + 32 wq + 8 z + (lane >> 3)
    //   cout  pieces z = 0..3: tile rows 128 g + 32 wq + 8 z + (lane >> 3)
    const int prow = lane >> 3;
    const int sunit0 = (lane & 7) ^ (lane >> 4), sunit1 = sunit0 ^ 4;
    const int ohw = p.OH * p.OW;
    const bool gemm = (p.KH == 1) & (p.KW == 1) & (p.stride == 1) & (p.pad == 0);
    auto arow = [&](int z) { return KSPLIT ? 32 * wq + 8 * z : 64 * (z >> 1) + 16 * wq + 8 * (z & 1); };   // first row of piece z within the group's 128
    int iy0[4], ix0[4], rowoff[4];
    bool mok[4];
#pragma unroll
    for (int z = 0; z < 4; ++z) {
        const int m = m0 + 128 * g + arow(z) + prow;
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
    const i32x4v rs_in = {(int)(unsigned)in_addr, (int)(unsigned)(in_addr >> 32) & 0xffff,
                          (int)((size_t)p.N * p.H * p.W * p.Cin * 2), 0x00020000};
    const i32x4v rs_w = {(int)(unsigned)w_addr, (int)(unsigned)(w_addr >> 32) & 0xffff,
                         (int)((size_t)p.Cout * p.Kpad * 2), 0x00020000};
    int woff[4];
#pragma unroll
    for (int z = 0; z < 4; ++z) {
        // interleaved cout layout of dma_epilogue_pairs: tile row 64 c + 32 j + l holds cout 64 c + 2 l + j
        const int rr = 128 * g + 32 * wq + 8 * z + prow;
        const int c = n0 + (rr & ~63) + 2 * (rr & 31) + ((rr >> 5) & 1);
        woff[z] = c < p.Cout ? (c * p.Kpad) * 2 + ((z & 1) ? sunit1 : sunit0) * 16 : (int)0x80000000;
    }
    // K order: channel chunk OUTER, tap INNER (see conv_bf16_dma_kernel); the weights are stored tap-major.  Two cursors: the
    // pixel rows' (kh, kw, c0 + the per-lane row offsets) and the cout rows' byte offset inside a weight row (KSPLIT requests
    // them for different chunks)
    const int kbase = nt ? nt_split * p.nt_chunks * DBK : 0;
    int kh = 0, kw = 0, c0 = kbase;
    int btap = 0, bc0 = kbase;
    int voffA[4];
    auto refresh_rows = [&]() {
        const int tapshift = (kh * p.W + kw) * p.Cin;
#pragma unroll
        for (int z = 0; z < 4; ++z) {
            const int iy = iy0[z] + kh, ix = ix0[z] + kw;
            const bool ok = mok[z] & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
            voffA[z] = ok ? (rowoff[z] + tapshift) * 2 + ((z & 1) ? sunit1 : sunit0) * 16 : (int)0x80000000;
        }
    };
    auto next_a = [&]() {
        if (++kw == p.KW) {
            kw = 0;
            if (++kh == p.KH) { kh = 0; c0 += DBK; }
        }
        if (p.KH * p.KW > 1) refresh_rows();
    };
    auto next_b = [&]() {
        if (++btap == p.KH * p.KW) { btap = 0; bc0 += DBK; }
    };
    refresh_rows();
    const int lds0 = (int)(unsigned)(size_t)smem;
    auto dma = [&](const i32x4v& rs, int voff, int soff, int lds_byte) {
        asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:0 lds"
                     :: "s"(lds_byte), "v"(voff), "s"(rs), "s"(soff) : "memory");
    };
    const int KT = nt ? p.nt_chunks : p.Kpad / DBK;
    const int ldsA = lds0 + (128 * g) * 128, ldsB = lds0 + B_BASE + (128 * g + 32 * wq) * 128;
    // ok = false: a request of the fixed schedule past the last chunk -- issued all the same (the counted wait needs it) with a
    // vector offset beyond the buffer range: zeros, no memory traffic, into a stage nobody reads again
    auto req_a = [&](int stage, int z, bool ok) {     // pixel piece z of the chunk the pixel cursor points at
        if (PP_ABL(4)) return;
        dma(rs_in, ok ? voffA[z] : (int)0x80000000, c0 * 2, ldsA + stage * A_STRIDE + arow(z) * 128);
    };
    auto req_b = [&](int stage, int z, bool ok) {
        if (PP_ABL(4)) return;
        dma(rs_w, ok ? woff[z] : (int)0x80000000, (btap * p.Cin + bc0) * 2, ldsB + stage * B_STRIDE + z * 1024);
    };

    // ---- MFMA role
    const int l31 = lane & 31, half = lane >> 5;
    const int rswz = (l31 >> 1) & 7;
    int koff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) koff[kk] = l31 * 128 + (((2 * kk + half) ^ rswz) * 16);
    const unsigned char* a_base = smem + (128 * g) * 128;
    const unsigned char* b_base = smem + B_BASE + (64 * wq) * 128;

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // KSPLIT: fa[k-step of the cluster][pixel block 0..3], fb[k-step of the cluster][cout block]
    // else:   fa[k-step 0..3][pixel block of the cluster], fb[k-step 0..3][cout block] (kept for both clusters of a chunk)
    f32x4 fa[KSPLIT ? 2 : 4][KSPLIT ? 4 : 2], fb[KSPLIT ? 2 : 4][2];

#ifdef CPR_BENCH_HOOKS
    // phase clocks of one chunk of one workgroup (ablate bit 10; the epilogue is then skipped and the stamps go to p.out)
    unsigned long long ts[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) ts[i] = 0;
    const bool stamping = PP_ABL(1024) && blockIdx.x == 16;
#define PP_STAMP(i) do { if (stamping && kt == 8) ts[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PP_STAMP(i) do { } while (0)
#endif

    auto read_a = [&](int off, int h) {           // !KSPLIT: pixel blocks h, h + 1, all k-steps; KSPLIT: k-steps 2 h, 2 h + 1, all blocks
        if (PP_ABL(8)) return;
        if constexpr (KSPLIT) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    fa[kk][i] = *reinterpret_cast<const f32x4*>(a_base + off + i * 4096 + koff[2 * h + kk]);
        } else {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    fa[kk][i] = *reinterpret_cast<const f32x4*>(a_base + off + (h + i) * 4096 + koff[kk]);
        }
    };
    auto read_b = [&](int off, int h) {
        if (PP_ABL(8)) return;
#pragma unroll
        for (int kk = 0; kk < (KSPLIT ? 2 : 4); ++kk)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                fb[kk][j] = *reinterpret_cast<const f32x4*>(b_base + off + j * 4096 + koff[KSPLIT ? 2 * h + kk : kk]);
    };
    // 16 MFMAs on registers only
#define PP_CLUSTER(i0)                                                                                                  \
    do {                                                                                                                \
        __builtin_amdgcn_s_setprio(1);                                                                                  \
        if constexpr (KSPLIT) {                                                                                         \
            _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                            \
                _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                           \
                    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                       \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                                            \
                            __builtin_bit_cast(bf16x8, fa[kk][i]), __builtin_bit_cast(bf16x8, fb[kk][j]), acc[i][j], 0, 0, 0); \
        } else {                                                                                                        \
            _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                                            \
                _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                           \
                    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                       \
                        acc[(i0) + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                                     \
                            __builtin_bit_cast(bf16x8, fa[kk][i]), __builtin_bit_cast(bf16x8, fb[kk][j]), acc[(i0) + i][j], 0, 0, 0); \
        }                                                                                                               \
        __builtin_amdgcn_s_setprio(0);                                                                                  \
    } while (0)
    // end of a memory part: the wave's requests older than its newest N have landed, its fragment reads are done; the barrier
    // makes both true for every wave
#define PP_MEM_END(N, s0)                                                                                               \
    do {                                                                                                                \
        PP_STAMP(s0);                                                                                                   \
        if ((N) >= 0 && !PP_ABL(1)) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((N) >= 0 ? (N) : 0) : "memory");          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        PP_STAMP(s0 + 1);                                                                                               \
        if (!PP_ABL(2)) __builtin_amdgcn_s_barrier();                                                                   \
        asm volatile("" ::: "memory");                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
    } while (0)
#define PP_CLUSTER_END(cond)                                                                                            \
    do {                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        if (!PP_ABL(2) && (cond)) __builtin_amdgcn_s_barrier();                                                         \
        asm volatile("" ::: "memory");                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
    } while (0)

    if constexpr (KSPLIT) {
        // prologue: A(0), B(0), A(1)
#pragma unroll
        for (int z = 0; z < 4; ++z) req_a(0, z, true);
        next_a();
#pragma unroll
        for (int z = 0; z < 4; ++z) req_b(0, z, true);
        next_b();
#pragma unroll
        for (int z = 0; z < 4; ++z) req_a(1, z, KT > 1);
        next_a();
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        // prologue: chunk 0 completely, chunk 1 but its A_hi pieces (La(0) requests them)
#pragma unroll
        for (int z = 0; z < 4; ++z) req_b(0, z, true);
#pragma unroll
        for (int z = 0; z < 4; ++z) req_a(0, z, true);
        next_a();
        next_b();
#pragma unroll
        for (int z = 0; z < 4; ++z) req_b(1, z, KT > 1);
        req_a(1, 0, KT > 1);
        req_a(1, 1, KT > 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    }
    __syncthreads();
    if (g == 1 && !PP_ABL(2)) __builtin_amdgcn_s_barrier();        // group 1 runs one interval behind
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    int sa = 0, sa_req = 2;            // KSPLIT: pixel-row stage of the chunk being read / of the chunk requested next (mod 3)
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < KT, more2 = kt + 2 < KT;        // wave-uniform
        if constexpr (KSPLIT) {
            const int aoff = sa * A_STRIDE, boff = buf * B_STRIDE;
            // La
            PP_STAMP(0);
            if (ORDER == 1) { read_b(boff, 0); read_a(aoff, 0); }
#pragma unroll
            for (int z = 0; z < 4; ++z) req_b(buf ^ 1, z, more);
            next_b();
            if (ORDER == 0) { read_b(boff, 0); read_a(aoff, 0); }
            PP_MEM_END(-1, 1);
            PP_STAMP(3);
            PP_CLUSTER(0);
            PP_STAMP(4);
            PP_CLUSTER_END(true);
            // Lb
            PP_STAMP(5);
            if (ORDER == 1) { read_b(boff, 1); read_a(aoff, 1); }
#pragma unroll
            for (int z = 0; z < 4; ++z) req_a(sa_req, z, more2);
            next_a();
            if (ORDER == 0) { read_b(boff, 1); read_a(aoff, 1); }
            PP_MEM_END(4, 6);
            PP_STAMP(8);
            PP_CLUSTER(0);
            PP_STAMP(9);
            PP_CLUSTER_END(g == 0 || more);
            PP_STAMP(10);
            sa = sa == 2 ? 0 : sa + 1;
            sa_req = sa_req == 2 ? 0 : sa_req + 1;
        } else {
            const int off = buf * PP_STAGE;
            // La
            PP_STAMP(0);
            if (ORDER == 1) { read_b(off, 0); read_a(off, 0); }
            req_a(buf ^ 1, 2, more);
            req_a(buf ^ 1, 3, more);
            next_a();                                                    // the cursors now point at chunk kt + 2
            next_b();
            if (ORDER == 0) { read_b(off, 0); read_a(off, 0); }
            PP_MEM_END(8, 1);
            // Ma
            PP_STAMP(3);
            PP_CLUSTER(0);
            PP_STAMP(4);
            PP_CLUSTER_END(true);
            // Lb
            PP_STAMP(5);
            if (ORDER == 1) read_a(off, 2);
#pragma unroll
            for (int z = 0; z < 4; ++z) req_b(buf, z, more2);
            req_a(buf, 0, more2);
            req_a(buf, 1, more2);
            if (ORDER == 0) read_a(off, 2);
            PP_MEM_END(8, 6);
            // Mb
            PP_STAMP(8);
            PP_CLUSTER(2);
            PP_STAMP(9);
            PP_CLUSTER_END(g == 0 || more);                              // group 1's last cluster has no partner left to wait for
            PP_STAMP(10);
        }
    }
#undef PP_CLUSTER
#undef PP_MEM_END
#undef PP_CLUSTER_END
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the requests past the last chunk must land before the LDS is released
#ifdef CPR_BENCH_HOOKS
    if (PP_ABL(1024)) {
        if (stamping && (wave & 3) == 0 && lane == 0) {
            unsigned long long* dst = reinterpret_cast<unsigned long long*>(p.out) + g * 16;
#pragma unroll
            for (int i = 0; i < 12; ++i) dst[i] = ts[i];
        }
        return;
    }
#endif

    if (nt) {   // this (split, tap)'s fp32 partial
        ConvDmaParams q = p;
        q.out = (float*)p.out + (size_t)st * p.M * p.Cout;
        dma_epilogue_pairs<MI, PRE>(q, acc, tm, m0, n0, g, wq, lane);
        return;
    }
    dma_epilogue_pairs<MI, PRE>(p, acc, tm, m0, n0, g, wq, lane);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Schedule 2 (what the phase clocks of the kernel above asked for, profiles/round6_bf16_pp_phase_clocks.txt): a memory part
// that also carries the requests and their address arithmetic takes 700 .. 1000 cycles against 560 .. 600 of a cluster -- the
// VALU work (tap cursor, row offsets, read addresses: ~50 instructions) waits behind the partner's MFMAs, which hold the SIMD's
// vector issue port, and an in-order wave pays request issue (~85 cycles each with four waves asking) PLUS read issue (~25 each).
// Here the memory part is NOTHING but 12 ds_read_b128 from addresses computed earlier, and everything else rides in the
// cluster of the SAME wave, where issue slots are free (an MFMA occupies the pipe for 32 cycles, the wave for 4):
//     La(kt)  12 reads (k-steps 0, 1: 4 pixel blocks + 2 cout blocks each)              | barrier
//     Ma(kt)  16 MFMAs + the 4 requests of B(kt + 2), one per four MFMAs                 | barrier
//     Lb(kt)  12 reads (k-steps 2, 3), vmcnt(4): everything but Ma's requests has landed | barrier
//     Mb(kt)  16 MFMAs + the 4 requests of A_g(kt + 2), each followed by that piece's row offsets for the next chunk; the read
//             addresses of the next two memory parts                                    | barrier
// Only four waves ask at any time, one request per ~128 cycles each: the texture path (16 cycles per request at its 64 B/clk)
// keeps up and a request is an issue slot, not a stall.  LDS: pixel rows 2 stages (A_g(kt) dies in Lb(kt), just before Mb(kt)
// re-requests its stage), cout rows 3 stages (B(kt) dies in interval 4 kt + 3 = group 1's Lb(kt); with three stages B(kt + 2)
// goes to the stage B(kt - 1) left in interval 4 kt - 1 and has six intervals to land) = 64 + 96 = all 160 KB.
// ABL (measurement build): compile-time loop ablations -- 4 no requests, 8 no fragment reads, 16 no row-offset arithmetic, 32 the
// requests and their arithmetic not pinned to their MFMA slots (results are WRONG for 4 / 8 / 16)
template <bool PRIO, bool PRE, bool STAMPS, int ABL = 0>
__global__ __launch_bounds__(512, 2) void conv_bf16_pp2_kernel(ConvDmaParams p) {
    constexpr int MI = 4, NJ = 2, DBM = 256, DBN = 256;
    constexpr int B_BASE = 2 * PP_A_BYTES;
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
    const i32x4v rs_in = {(int)(unsigned)in_addr, (int)(unsigned)(in_addr >> 32) & 0xffff,
                          (int)((size_t)p.N * p.H * p.W * p.Cin * 2), 0x00020000};
    const i32x4v rs_w = {(int)(unsigned)w_addr, (int)(unsigned)(w_addr >> 32) & 0xffff,
                         (int)((size_t)p.Cout * p.Kpad * 2), 0x00020000};
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
    auto next_b = [&]() {
        if (++btap == p.KH * p.KW) { btap = 0; bc0 += DBK; }
    };
    const int lds0 = (int)(unsigned)(size_t)smem;
    auto dma = [&](const i32x4v& rs, int voff, int soff, int lds_byte) {
        asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:0 lds"
                     :: "s"(lds_byte), "v"(voff), "s"(rs), "s"(soff) : "memory");
    };
    // the weight rows with a cache policy that keeps them out of the CU's vector L1 (measurement: ABL & 64 = nt, & 256 = sc1, both = sc1 nt)
    auto dma_w = [&](const i32x4v& rs, int voff, int soff, int lds_byte) {
        if constexpr ((ABL & 320) == 64)
            asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:0 nt lds"
                         :: "s"(lds_byte), "v"(voff), "s"(rs), "s"(soff) : "memory");
        else if constexpr ((ABL & 320) == 256)
            asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:0 sc1 lds"
                         :: "s"(lds_byte), "v"(voff), "s"(rs), "s"(soff) : "memory");
        else if constexpr ((ABL & 320) == 320)
            asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:0 sc1 nt lds"
                         :: "s"(lds_byte), "v"(voff), "s"(rs), "s"(soff) : "memory");
        else dma(rs, voff, soff, lds_byte);
    };
    const int KT = nt ? p.nt_chunks : p.Kpad / DBK;
    const int ldsA = lds0 + (128 * g + 32 * wq) * 128, ldsB = lds0 + B_BASE + (128 * g + 32 * wq) * 128;
    auto req_a = [&](int stage, int z, bool ok) {
        if (ABL & 4) return;
        if (ABL & 128) { dma(rs_in, lane * 16, 0, ldsA + stage * PP_A_BYTES + z * 1024); return; }     // every request the same KB: L1 hits
        dma(rs_in, ok ? voffA[z] : (int)0x80000000, c0 * 2, ldsA + stage * PP_A_BYTES + z * 1024);
    };
    auto req_b = [&](int stage, int z, bool ok) {
        if (ABL & 4) return;
        if (ABL & 128) { dma(rs_w, lane * 16, 0, ldsB + stage * PP_A_BYTES + z * 1024); return; }
        dma_w(rs_w, ok ? woff[z] : (int)0x80000000, (btap * p.Cin + bc0) * 2, ldsB + stage * PP_A_BYTES + z * 1024);
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

    // STAMPS (measurement build, its own instance): phase clocks of chunk 8 of workgroup 16 instead of results
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
#define PP2_MFMA(q)                                                                                                     \
    acc[((q) >> 1) & 3][(q) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                                             \
        __builtin_bit_cast(bf16x8, fa[(q) >> 3][((q) >> 1) & 3]), __builtin_bit_cast(bf16x8, fb[(q) >> 3][(q) & 1]),    \
        acc[((q) >> 1) & 3][(q) & 1], 0, 0, 0)
#define PP2_MEM_END(N, s0)                                                                                              \
    do {                                                                                                                \
        PP_STAMP(s0);                                                                                                   \
        if ((N) >= 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((N) >= 0 ? (N) : 0) : "memory");                        \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        PP_STAMP(s0 + 1);                                                                                               \
        __builtin_amdgcn_s_barrier();                                                                                   \
        asm volatile("" ::: "memory");                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
    } while (0)
#define PP2_CLUSTER_END(cond)                                                                                           \
    do {                                                                                                                \
        if (PRIO) __builtin_amdgcn_s_setprio(0);                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        if (cond) __builtin_amdgcn_s_barrier();                                                                         \
        asm volatile("" ::: "memory");                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
    } while (0)

    // prologue: chunks 0 and 1 completely; the cursors then point at chunk 2
#pragma unroll
    for (int z = 0; z < 4; ++z) voffA[z] = row_voff(z, 0, 0);
#pragma unroll
    for (int z = 0; z < 4; ++z) req_a(0, z, true);
#pragma unroll
    for (int z = 0; z < 4; ++z) req_b(0, z, true);
    next_b();
    if (++kw == p.KW) { kw = 0; if (++kh == p.KH) { kh = 0; c0 += DBK; } }
#pragma unroll
    for (int z = 0; z < 4; ++z) voffA[z] = row_voff(z, kh, kw);
#pragma unroll
    for (int z = 0; z < 4; ++z) req_a(1, z, KT > 1);
#pragma unroll
    for (int z = 0; z < 4; ++z) req_b(1, z, KT > 1);
    next_b();
    if (++kw == p.KW) { kw = 0; if (++kh == p.KH) { kh = 0; c0 += DBK; } }
#pragma unroll
    for (int z = 0; z < 4; ++z) voffA[z] = row_voff(z, kh, kw);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { ra[kk] = a_rd + koff[kk]; rb[kk] = b_rd + koff[kk]; }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __syncthreads();
    if (g == 1) __builtin_amdgcn_s_barrier();        // group 1 runs one interval behind
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    int sb = 0, sb_req = 2;            // cout-row stage (of three) of the chunk being read / of the chunk requested next
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        const bool more2 = kt + 2 < KT;        // wave-uniform
        // La
        PP_STAMP(0);
        reads(0);
        PP2_MEM_END(-1, 1);
        // Ma: + the requests of B(kt + 2)
        PP_STAMP(3);
        if (PRIO) __builtin_amdgcn_s_setprio(1);
        auto cluster_a = [&](auto sc) {
            constexpr int S = decltype(sc)::value;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                PP2_MFMA(q);
                if ((q & 3) == S) {
                    req_b(sb_req, q >> 2, more2);
                    if (!(ABL & 32)) __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        cluster_a(std::integral_constant<int, 1>{});
        next_b();
        PP_STAMP(4);
        PP2_CLUSTER_END(true);
        // Lb
        PP_STAMP(5);
        reads(1);
        PP2_MEM_END(4, 6);
        // Mb: + the requests of A_g(kt + 2), each followed by the piece's row offsets for chunk kt + 3; the read addresses of chunk kt + 1
        PP_STAMP(8);
        if (PRIO) __builtin_amdgcn_s_setprio(1);
        int nkh = kh, nkw = kw, nc0 = c0;
        if (++nkw == p.KW) { nkw = 0; if (++nkh == p.KH) { nkh = 0; nc0 += DBK; } }
        sb = sb == 2 ? 0 : sb + 1;
        sb_req = sb_req == 2 ? 0 : sb_req + 1;
        auto cluster_b = [&](auto sc) {
            constexpr int S = decltype(sc)::value;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                PP2_MFMA(q);
                if ((q & 3) == S) {
                    req_a(buf, q >> 2, more2);
                    if (!(ABL & 16) && p.KH * p.KW > 1) voffA[q >> 2] = row_voff(q >> 2, nkh, nkw);
                    ra[q >> 2] = a_rd + (buf ^ 1) * PP_A_BYTES + koff[q >> 2];
                    rb[q >> 2] = b_rd + sb * PP_A_BYTES + koff[q >> 2];
                    if (!(ABL & 32)) __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        cluster_b(std::integral_constant<int, 1>{});
        kh = nkh; kw = nkw; c0 = nc0;
        PP_STAMP(9);
        PP2_CLUSTER_END(g == 0 || kt + 1 < KT);                              // group 1's last cluster has no partner left to wait for
        PP_STAMP(10);
    }
#undef PP2_MFMA
#undef PP2_MEM_END
#undef PP2_CLUSTER_END
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the requests past the last chunk must land before the LDS is released
    if constexpr (STAMPS) {
        if (stamping && (wave & 3) == 0 && lane == 0) {
            unsigned long long* dst = reinterpret_cast<unsigned long long*>(p.out) + g * 16;
#pragma unroll
            for (int i = 0; i < 12; ++i) dst[i] = ts[i];
        }
        return;
    }
    if (nt) {   // this (split, tap)'s fp32 partial
        ConvDmaParams q = p;
        q.out = (float*)p.out + (size_t)st * p.M * p.Cout;
        dma_epilogue_pairs<MI, PRE>(q, acc, tm, m0, n0, g, wq, lane);
        return;
    }
    dma_epilogue_pairs<MI, PRE>(p, acc, tm, m0, n0, g, wq, lane);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Schedule 3: the memory part holds NO vector-ALU instruction.  tools/diag/mfma_dma_mix.hip (profiles/round6_mfma_dma_mix.txt)
// priced what the two waves of a SIMD share: a wave that issues 4 LDS-DMA requests + 12 ds_read_b128 per iteration beside a
// partner streaming MFMAs needs 447 cycles per iteration and costs the partner NOTHING (32.0 cycles per MFMA) -- but every VALU
// instruction in that wave waits ~430 cycles (the MFMA stream holds the SIMD's vector issue port: 12 four-wide v_add turned the
// 447 cycles into 20 800).  That is what serialised every earlier schedule: tap cursor, row offsets, "past the last chunk"
// selects and read addresses (~50 VALU per chunk) sat in the memory parts, which therefore ran AFTER the partner's cluster, not
// beside it.  Here
//     La(kt)  4 requests B(kt + 1) + 12 reads (k-steps 0, 1)                      | barrier      -- SALU, DS and VMEM only
//     Ma(kt)  16 MFMAs                                                            | barrier
//     Lb(kt)  4 requests A_g(kt + 2) + 12 reads (k-steps 2, 3), vmcnt(4)          | barrier      -- SALU, DS and VMEM only
//     Mb(kt)  16 MFMAs + all vector arithmetic of the chunk, between its own MFMAs: the row offsets of A_g(kt + 3), the eight read
//             addresses of chunk kt + 1 (pinned there: the compiler otherwise sinks them next to their use)  | barrier
// "Past the last chunk" is a scalar select of the buffer descriptor's size word (0 records: every lane out of range, zeros, no
// traffic).  LDS: pixel rows 3 stages (A_g(kt + 2) is requested while Lb(kt) still reads A_g(kt)), cout rows 2 stages = 160 KB.
template <bool PRIO, bool PRE, int ABL = 0, bool STAMPS = false>
__global__ __launch_bounds__(512, 2) void conv_bf16_pp3_kernel(ConvDmaParams p) {
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
#define PP3_MFMA(q)                                                                                                     \
    acc[((q) >> 1) & 3][(q) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                                             \
        __builtin_bit_cast(bf16x8, fa[(q) >> 3][((q) >> 1) & 3]), __builtin_bit_cast(bf16x8, fb[(q) >> 3][(q) & 1]),    \
        acc[((q) >> 1) & 3][(q) & 1], 0, 0, 0)
#define PP3_MEM_END(N, s0)                                                                                              \
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
#define PP3_CLUSTER_END(cond)                                                                                           \
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
        PP3_MEM_END(-1, 1);
        // Ma
        PP_STAMP(3);
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int q = 0; q < 16; ++q) PP3_MFMA(q);
        PP_STAMP(4);
        PP3_CLUSTER_END(true);
        PP_STAMP(5);
        // Lb: no vector ALU instruction from here to the barrier
        if (!(ABL & 32)) reads(1);
#pragma unroll
        for (int z = 0; z < 4; ++z) req_a(sa_req, z, more2);
        next_a();
        if (ABL & 32) reads(1);
        PP3_MEM_END(4, 6);
        PP_STAMP(8);
        // Mb: + the row offsets of A_g(kt + 3) and the read addresses of chunk kt + 1
        if (PRIO) __builtin_amdgcn_s_setprio(1);
        sa = sa == 2 ? 0 : sa + 1;
        sa_req = sa_req == 2 ? 0 : sa_req + 1;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            PP3_MFMA(q);
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
        PP3_CLUSTER_END(g == 0 || more);                              // group 1's last cluster has no partner left to wait for
        PP_STAMP(10);
    }
#undef PP3_MFMA
#undef PP3_MEM_END
#undef PP3_CLUSTER_END
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the requests past the last chunk must land before the LDS is released
    if (nt) {   // this (split, tap)'s fp32 partial
        ConvDmaParams q = p;
        q.out = (float*)p.out + (size_t)st * p.M * p.Cout;
        dma_epilogue_pairs<MI, PRE>(q, acc, tm, m0, n0, g, wq, lane);
        return;
    }
    dma_epilogue_pairs<MI, PRE>(p, acc, tm, m0, n0, g, wq, lane);
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

// called by conv_bf16_dma_launch (shape 5) and conv_bf16_dma_nt_launch; variant: bit 0 = ORDER, bit 1 = KSPLIT; bit 2 = schedule 2
// (bit 0 then = no s_setprio around the clusters)
void conv_bf16_pp_launch(const ConvDmaParams& p, unsigned grid, int variant, hipStream_t stream) {
    if (variant & 8) {          // schedule 3 (bit 0: no s_setprio)
#ifdef CPR_BENCH_HOOKS
        const int abl = ((p.ablate >> 2) & 3) * 4 | ((p.ablate >> 7) & 3) * 16 | ((p.ablate & 1) ? 64 : 0) | ((p.ablate & 2) ? 128 : 0);
        if (abl == 64) hipLaunchKernelGGL((conv_bf16_pp3_kernel<true, true, 64>), dim3(grid), dim3(512), 0, stream, p);
        else if (abl == 128) hipLaunchKernelGGL((conv_bf16_pp3_kernel<true, true, 128>), dim3(grid), dim3(512), 0, stream, p);
        else if (abl == 16) hipLaunchKernelGGL((conv_bf16_pp3_kernel<true, true, 16>), dim3(grid), dim3(512), 0, stream, p);
        else if (p.ablate & 1024) {
            if (abl == 4) hipLaunchKernelGGL((conv_bf16_pp3_kernel<true, true, 4, true>), dim3(grid), dim3(512), 0, stream, p);
            else if (abl == 8) hipLaunchKernelGGL((conv_bf16_pp3_kernel<true, true, 8, true>), dim3(grid), dim3(512), 0, stream, p);
            else if (abl == 12) hipLaunchKernelGGL((conv_bf16_pp3_kernel<true, true, 12, true>), dim3(grid), dim3(512), 0, stream, p);
            else hipLaunchKernelGGL((conv_bf16_pp3_kernel<true, true, 0, true>), dim3(grid), dim3(512), 0, stream, p);
        }
        else if (abl == 32) hipLaunchKernelGGL((conv_bf16_pp3_kernel<true, true, 32>), dim3(grid), dim3(512), 0, stream, p);
        else if (abl == 4) hipLaunchKernelGGL((conv_bf16_pp3_kernel<false, true, 4>), dim3(grid), dim3(512), 0, stream, p);
        else if (abl == 8) hipLaunchKernelGGL((conv_bf16_pp3_kernel<false, true, 8>), dim3(grid), dim3(512), 0, stream, p);
        else if (abl == 12) hipLaunchKernelGGL((conv_bf16_pp3_kernel<false, true, 12>), dim3(grid), dim3(512), 0, stream, p);
        else if (abl == 16) hipLaunchKernelGGL((conv_bf16_pp3_kernel<false, true, 16>), dim3(grid), dim3(512), 0, stream, p);
        else
#endif
        if (variant & 1) hipLaunchKernelGGL((conv_bf16_pp3_kernel<false, true>), dim3(grid), dim3(512), 0, stream, p);
        else hipLaunchKernelGGL((conv_bf16_pp3_kernel<true, true>), dim3(grid), dim3(512), 0, stream, p);
        return;
    }
    switch (variant & 7) {
    case 0: hipLaunchKernelGGL((conv_bf16_pp_kernel<false, 0, true>), dim3(grid), dim3(512), 0, stream, p); break;
    case 1: hipLaunchKernelGGL((conv_bf16_pp_kernel<false, 1, true>), dim3(grid), dim3(512), 0, stream, p); break;
    case 2: hipLaunchKernelGGL((conv_bf16_pp_kernel<true, 0, true>), dim3(grid), dim3(512), 0, stream, p); break;
    case 3: hipLaunchKernelGGL((conv_bf16_pp_kernel<true, 1, true>), dim3(grid), dim3(512), 0, stream, p); break;
#ifdef CPR_BENCH_HOOKS
    case 4: case 6: case 5: case 7: {
        const int abl = ((p.ablate >> 2) & 3) * 4 | ((p.ablate >> 7) & 3) * 16 | ((p.ablate & 1) ? 64 : 0) | ((p.ablate & 2) ? 128 : 0) | ((p.ablate & 64) ? 256 : 0);     // ablate bits 2, 3 -> ABL 4, 8; bits 7, 8 -> ABL 16, 32; bit 0 -> 64 (staggered slots), bit 1 -> 128 (same-address requests)
#define PP2_GO(PRIO_, ST_, A_) hipLaunchKernelGGL((conv_bf16_pp2_kernel<PRIO_, true, ST_, A_>), dim3(grid), dim3(512), 0, stream, p)
        const bool prio = !(variant & 1);
        if (p.ablate & 1024) { if (prio) PP2_GO(true, true, 0); else PP2_GO(false, true, 0); }
        else if (abl == 0) { if (prio) PP2_GO(true, false, 0); else PP2_GO(false, false, 0); }
        else if (abl == 4) PP2_GO(true, false, 4);
        else if (abl == 8) PP2_GO(true, false, 8);
        else if (abl == 12) PP2_GO(true, false, 12);
        else if (abl == 16) PP2_GO(true, false, 16);
        else if (abl == 32) { if (prio) PP2_GO(true, false, 32); else PP2_GO(false, false, 32); }
        else if (abl == 48) PP2_GO(true, false, 48);
        else if (abl == 64) PP2_GO(true, false, 64);
        else if (abl == 128) PP2_GO(true, false, 128);
        else if (abl == 256) PP2_GO(true, false, 256);
        else if (abl == 320) PP2_GO(true, false, 320);
        else if (abl == 136) PP2_GO(true, false, 136);
        else PP2_GO(true, false, 28);
#undef PP2_GO
        break;
    }
#else
    case 4: case 6: hipLaunchKernelGGL((conv_bf16_pp2_kernel<true, true, false>), dim3(grid), dim3(512), 0, stream, p); break;
    default: hipLaunchKernelGGL((conv_bf16_pp2_kernel<false, true, false>), dim3(grid), dim3(512), 0, stream, p); break;
#endif
    }
}
