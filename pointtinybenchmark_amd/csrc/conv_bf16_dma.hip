// bf16 implicit-GEMM convolution, 256 x 256 output tiles staged by LDS-DMA (round 3; the big layers of the bf16 compute mode).
//
// Same call sites and arithmetic as conv_mfma_bf16.hip (v_mfma_f32_32x32x16_bf16, fp32 accumulate, fused scale / bias /
// residual / ReLU, GroupNorm partial statistics from the fp32 accumulators); what changes is how the operands reach the
// matrix cores.  conv_mfma_bf16_kernel<128,128> moves every byte global -> VGPR -> ds_write -> ds_read: at bf16 MFMA rates
// (16x the fp32 rate) its four waves issue one staging instruction per MFMA and the matrix pipe is busy 43 % of the time
// (profiles/round2_pmc_bf16_kernel.json: 0.35 of the 2.5 PFLOP/s peak).  Here
//   * the tile is 256 (pixels) x 256 (couts) x 64 (K): 32 MFMAs per wave per K chunk against 8 LDS-DMA requests and 24
//     ds_read_b128 (a 128 x 128 tile needs twice the operand bytes per MFMA);
//   * both operands go global -> LDS directly (buffer_load_dwordx4 ... lds, 1 KB per wave instruction, no staging registers, no
//     ds_write); zero padding = the buffer range check, exactly as in the register-staged kernels;
//   * the LDS image of a DMA is lane-linear (wave base + 16 lane), rows are 128 bytes and every lane of a fragment read
//     wants the SAME 16-byte column of 32 different rows: un-swizzled that is an 8-way bank conflict.  The swizzle therefore
//     sits on the SOURCE side: the lane that fills 16-byte unit u of row r fetches k-unit u ^ ((r >> 1) & 7), and the fragment
//     read of k-unit q goes to unit q ^ ((r >> 1) & 7) -- two rows share a 256-byte bank row, so (r & 1, unit) is distinct
//     over the 16 rows of every ds_read_b128 service group: conflict free, and each row's 128 bytes are still one full line;
//   * two 64 KB stages, one wait + one barrier per chunk, placed before the LAST k-step: behind that k-step's MFMAs the wave
//     reads the next chunk's first fragments and requests the chunk after it into the stage that has just become free
//     (3.5 k-steps, ~0.8 us, to land).
// 512 threads = 8 waves: wave (wm, wn) = (wave & 1, wave >> 1) owns pixels [128 wm, +128) x couts [64 wn, +64): 4 x 2 blocks of
// 32 x 32 = 128 accumulator registers.  One workgroup per CU (128 KB of LDS), two waves per SIMD.
// Round 4: the kernel is a template over the wave's block grid and the wave grid (DmaTile below): the 256 x 256 instance described
// here (<4, 2, 4>), a 128 x 128 instance with two workgroups per CU and a row-wise LDS epilogue (<2, 1, 4>) for launches that
// cannot fill the chip with big tiles, and -- in the 256 x 256 instance -- an interleaved cout layout that lets every lane store
// adjacent couts (dma_epilogue_pairs).  Which layer takes which instance: bf16_dma_shape in conv_mfma_bf16.hip.
#include "conv_bf16_dma.h"
#include <cstdlib>

// BD = true (round 5, <4, 2, 4, true> only): the WEIGHT operand goes global -> VGPR directly, only the activations go through LDS.
// The host hands over a second image of the weights in fragment order -- wfrag[g = cout / 64][ks = k / 16][j][lane][8]: lane
// (l31 + 32 half) of block j holds k = 16 ks + 8 half .. + 8 of cout 64 g + 2 l31 + j (the interleaved cout layout of
// dma_epilogue_pairs), i.e. exactly the 16 bytes that lane feeds to the MFMA -- so a wave's B operand of one k-step is two
// coalesced 1 KB loads with no LDS round trip.  Per chunk a wave then issues 4 LDS-DMA requests + 8 register loads (before: 8
// requests) and 16 ds_read_b128 (before: 24); the stages shrink to 32 KB.  The register loads are inline asm like the DMA
// requests (the compiler would otherwise count its own loads against a vmcnt that also holds the DMA requests it cannot see and
// drain them with every wait): four k-steps of B fragments live in a ring of 32 registers, slot kk is re-requested for the NEXT
// chunk one k-step after the MFMAs of k-step kk have been issued, and the loop's two waits (the barrier's vmcnt(0), one counted
// wait in front of k-step 2) cover them.
template <int MI, int NJ, int WN, bool BD = false, bool STAG = false, bool TR = false>
__global__ __launch_bounds__(128 * WN, (WN == 2 ? 1 : MI * NJ <= 2 ? 4 : 2)) void conv_bf16_dma_kernel   // (threads, waves per SIMD)
(ConvDmaParams p) {
    static_assert(!STAG || BD, "staggered request slots exist for the weights-direct instance");
    static_assert(!TR || BD, "the training epilogue (dma_epilogue_pairs<.., TR>) is instantiated for the weights-direct instance");
    static_assert(!BD || (MI == 4 && NJ == 2 && WN == 4), "weights direct to registers: the 256 x 256 eight-wave instance");
    using T = DmaTile<MI, NJ, WN, BD>;
    constexpr int DBM = T::BM, DBN = T::BN, DSTAGE = T::STAGE, NM = T::NM, NF = T::NF, NP = T::NP, NPA = T::NPA, NPW = T::NPW;
    constexpr int RP = T::RP, PIECE = T::PIECE;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * DSTAGE];

    // XCD-aware tile order (block b runs on XCD b % 8; an XCD walks a contiguous run of tiles, cout tiles fastest)
    const bool nt = p.nt_taps > 0;
    const int TMN = p.tilesM * p.tilesN;
    int tile, st = 0, nt_split = 0, nt_tap = 0;
    if (nt) {
        // splits are dealt to the XCDs (block b runs on XCD b % 8; the host makes the split count a multiple of 8): the taps and
        // tiles of one split are neighbours on one XCD and read the same K range of both operands within one L2
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

    // ---- staging role: piece j of an operand = rows RP j + (tid >> 3), 16-byte unit (tid & 7) of the 128-byte row; the
    // lane fetches the k-unit (tid & 7) ^ swz, swz = (row >> 1) & 7 = (tid >> 4) & 7 for every j
    const int srow = tid >> 3;
    const int sunit = (tid & 7) ^ ((tid >> 4) & 7);
    const int ohw = p.OH * p.OW;
    const bool gemm = (p.KH == 1) & (p.KW == 1) & (p.stride == 1) & (p.pad == 0);
    int iy0[NPA], ix0[NPA], rowoff[NPA];
    bool mok[NPA];
#pragma unroll
    for (int j = 0; j < NPA; ++j) {
        const int m = m0 + srow + RP * j;
        mok[j] = m < p.M;
        const int mm = mok[j] ? m : 0;
        if (gemm) {
            iy0[j] = 0; ix0[j] = 0;
            rowoff[j] = mm * p.Cin;
        } else {
            const int n = mm / ohw;
            const int rem = mm - n * ohw;
            const int oy = rem / p.OW, ox = rem - oy * p.OW;
            iy0[j] = oy * p.stride - p.pad;
            ix0[j] = ox * p.stride - p.pad;
            rowoff[j] = ((n * p.H + iy0[j]) * p.W + ix0[j]) * p.Cin;      // element offset, may be "negative" when padded
        }
    }
    // buffer descriptors as four SGPR words (base, stride 0, num_records, raw dword format): out-of-range offsets read 0
    const size_t in_addr = (size_t)p.in;
    const size_t w_addr = (size_t)(p.wgt + (nt ? (long long)(nt_tap % p.nt_k) * p.nt_copy + (nt_tap / p.nt_k - p.nt_pad) * p.nt_Wp : 0));
    const i32x4v rs_in = {(int)(unsigned)in_addr, (int)(unsigned)(in_addr >> 32) & 0xffff,
                          (int)((size_t)p.N * p.H * p.W * p.Cin * 2), 0x00020000};
    const i32x4v rs_w = {(int)(unsigned)w_addr, (int)(unsigned)(w_addr >> 32) & 0xffff,
                         (int)((size_t)p.Cout * p.Kpad * 2), 0x00020000};
    int woff[NPW > 0 ? NPW : 1];
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
        // <*, 2, 4> (two cout blocks per wave): tile row 64 g + 32 jj + l holds cout 64 g + 2 l + jj, see dma_epilogue_pairs
        const int c = (NJ == 2 && WN == 4) ? n0 + 64 * j + 2 * (srow & 31) + (srow >> 5) : n0 + srow + RP * j;
        woff[j] = c < p.Cout ? (c * p.Kpad) * 2 + sunit * 16 : (int)0x80000000;
    }
    // K order: channel chunk OUTER, tap INNER.  Consecutive chunks then read the same 128-byte segments of pixels one tap
    // apart, i.e. the lines the previous chunk brought into the XCD's L2 (2 MB of traffic ago).  Tap-major order re-touches a
    // line only four chunks = 8 MB of L2 traffic later: measured on the head layer 4.8 GB of L2 misses per launch against
    // 0.84 GB of input, TCC hit rate 64 % (gpurun_out/r3j PMC).  The weights are stored tap-major: chunk (c0, tap) sits at
    // K offset tap * Cin + c0.
    int kh = 0, kw = 0, c0 = nt ? nt_split * p.nt_chunks * DBK : 0;     // tap / channel chunk of the NEXT chunk to be requested (wave-uniform)
    int wsoff = 0;                  // byte offset of that chunk inside a weight row
    int voffA[NPA];
    auto refresh_rows = [&]() {
        const int tapshift = (kh * p.W + kw) * p.Cin;
#pragma unroll
        for (int j = 0; j < NPA; ++j) {
            const int iy = iy0[j] + kh, ix = ix0[j] + kw;
            const bool ok = mok[j] & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
            voffA[j] = ok ? (rowoff[j] + tapshift) * 2 + sunit * 16 : (int)0x80000000;
        }
    };
    refresh_rows();
    const int lds0 = (int)(unsigned)(size_t)smem;                 // LDS byte address of stage 0
    // one LDS-DMA request: 1 KB of this wave (lane * 16 bytes from M0).  Inline asm: the compiler neither counts nor waits for
    // it; every global load of the K loop is of this kind and the loop counts them by hand (dma_wait)
    auto dma = [&](const i32x4v& rs, int voff, int soff, int lds_byte) {
        asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:0 lds"
                     :: "s"(lds_byte), "v"(voff), "s"(rs), "s"(soff) : "memory");
    };
    const int KT = nt ? p.nt_chunks : p.Kpad / DBK;
    // piece z = 0..NP-1 of chunk kt (z < NPA: activations, else weights) into stage `buf`; pieces are issued in order
    auto stage_piece = [&](int kt, int buf, int z) {
        if (p.ablate & 4) return;
        const int base = lds0 + buf * DSTAGE + wave * 1024;
        if (z < NPA) {
            if (z == 0) wsoff = ((kh * p.KW + kw) * p.Cin + c0) * 2;
            dma(rs_in, voffA[z < NPA ? z : 0], c0 * 2, base + z * PIECE);
            if (z == NPA - 1) {      // after the last activation piece: advance the tap / channel state to the next chunk
                if (++kw == p.KW) {
                    kw = 0;
                    if (++kh == p.KH) { kh = 0; c0 += DBK; }
                }
                if (p.KH * p.KW > 1) refresh_rows();
            }
        } else {
            dma(rs_w, woff[z >= NPA ? z - NPA : 0], wsoff, base + DBM * DBK * 2 + (z - NPA) * PIECE);
        }
    };

    // ---- MFMA role
    const int wm = wave & 1, wn = wave >> 1;
    const int l31 = lane & 31, half = lane >> 5;
    const int rswz = (l31 >> 1) & 7;                              // (row >> 1) & 7 of the lane's fragment rows (all blocks alike)
    // byte offset of k-unit 2 kk + half of the lane's row inside a row block: per kk, XORed with the row's swizzle
    int koff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) koff[kk] = l31 * 128 + (((2 * kk + half) ^ rswz) * 16);
    const unsigned char* a_base = smem + (wm * MI * 32) * 128;
    const unsigned char* b_base = smem + DBM * DBK * 2 + (wn * NJ * 32) * 128;
    // BD: the B fragments of the four k-steps of a chunk, requested straight from the fragment-order image.  One request =
    // buffer_load_dwordx4 of 1 KB: wave-uniform byte offset of (cout group, k-step) in an SGPR + lane * 16 (+ 1 KB for block 1)
    f32x4 fbr[BD ? 4 : 1][NJ];
    const size_t wf_addr = (size_t)p.wfrag;
    const i32x4v rs_wf = {(int)(unsigned)wf_addr, (int)(unsigned)(wf_addr >> 32) & 0xffff,
                          (int)((size_t)p.Cout * p.Kpad * 2), 0x00020000};
    const int wf_lane = lane * 16;
    const int wf_group = BD ? ((n0 >> 6) + wn) * (p.Kpad >> 4) * 2048 : 0;          // byte offset of this wave's 64-cout group
    int bkh = 0, bkw = 0, bc0 = 0;            // tap / channel chunk of the NEXT chunk whose B fragments are requested (wave-uniform)
    int bks = 0;                              // byte offset of that chunk's first k-step inside the group: (K offset / 16) * 2048
    auto b_advance = [&]() {                  // the K order of the activations: channel chunk outer, tap inner
        if (++bkw == p.KW) {
            bkw = 0;
            if (++bkh == p.KH) { bkh = 0; bc0 += DBK; }
        }
        bks = (((bkh * p.KW + bkw) * p.Cin + bc0) >> 4) * 2048;
    };
    // ok = false: a request of the fixed schedule that lies past the last chunk -- issued all the same (the counted waits need
    // it), but with a vector offset beyond the buffer range (the range check looks at the vector offset): zeros, no memory traffic
#define BLOAD_AT(kk, off, ok)                                                                                         \
    do {                                                                                                              \
        if (p.ablate & 128) break;                                                                                    \
        asm volatile("buffer_load_dwordx4 %0, %2, %3, %4 offen offset:0\n\tbuffer_load_dwordx4 %1, %2, %3, %4 offen offset:1024" \
                     : "=&v"(fbr[kk][0]), "=&v"(fbr[kk][1]) : "v"((ok) ? wf_lane : (int)0x80000000), "s"(rs_wf), "s"(wf_group + (off) + (kk) * 2048) : "memory"); \
    } while (0)

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 fa0[MI], fb0[NJ], fa1[MI], fb1[NJ];
    // fragment piece z of a k-step: z < NJ = the cout blocks, then the MI pixel blocks.  MFMA q of the NEXT k-step uses pixel
    // block q / NJ and cout block q % NJ: with the cout blocks read first every operand of the 256 x 256 instance is requested
    // >= 6 MFMA slots (~200 cycles of this wave, as many of its partner) before its first use
    constexpr int NJR = BD ? 0 : NJ;        // cout-block fragments read from LDS (BD: none, they come from global memory)
#define DFRAG(FA, FB, buf, kk, z)                                                                                     \
    do {                                                                                                              \
        if (p.ablate & 8) break;                                                                                      \
        if ((z) >= NJR) FA[(z) >= NJR ? (z) - NJR : 0] = *reinterpret_cast<const f32x4*>(a_base + (buf) * DSTAGE + ((z) >= NJR ? (z) - NJR : 0) * 4096 + koff[kk]); \
        else FB[(z) < NJR ? (z) : 0] = *reinterpret_cast<const f32x4*>(b_base + (buf) * DSTAGE + ((z) < NJR ? (z) : 0) * 4096 + koff[kk]); \
    } while (0)
    // the NF reads of a k-step over its NM MFMA slots: slot q reads piece q (NF <= NM), else pieces [q NF / NM, (q + 1) NF / NM)
#define DFRAGS(FA, FB, buf, kk, q)                                                                    \
    do {                                                                                              \
        if (NF <= NM) { if ((q) < NF) DFRAG(FA, FB, buf, kk, (q) < NF ? (q) : 0); }                     \
        else {                                                                                        \
            _Pragma("unroll") for (int z_ = ((q) * NF) / NM; z_ < (((q) + 1) * NF) / NM; ++z_) DFRAG(FA, FB, buf, kk, z_); \
        }                                                                                             \
    } while (0)
#define DMFMA(FA, FB, q)                                                                              \
    acc[(q) / NJ][(q) % NJ] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                                \
        __builtin_bit_cast(bf16x8, FA[(q) / NJ]), __builtin_bit_cast(bf16x8, FB[(q) % NJ]), acc[(q) / NJ][(q) % NJ], 0, 0, 0)

    constexpr int C3 = T::C3, C0 = T::C0, C1 = T::C1;
    if constexpr (BD) {
        // Weights direct to registers.  EVERY request of the schedule is issued in every chunk -- past the last chunk with a
        // vector offset beyond the buffer range (zeros, no memory traffic), landing in a stage / a ring slot nobody reads again --
        // so the number of requests between any two points of the loop is a constant and the waits can be counted:
        //   per chunk   k-step 0: B3 (this chunk, 2 loads) .. P2 | k-step 1: B0' (next chunk) .. P3 | k-step 2: B1' |
        //               wait + barrier | k-step 3: B2' .. P0'', P1'' (the chunk after next -> the stage just freed)
        //   before k-step 1: B1 landed = vmcnt(7) (B2 B2 P0 P1 B3 B3 P2 younger); before k-step 2: B2 landed = vmcnt(8);
        //   barrier: P3 and everything older (B3, B0') landed = vmcnt(2) (B1' B1' younger).  A request has three k-steps to land.
        static_assert(NPA == 4 && C3 == 2 && C0 == 1 && C1 == 1, "the counted waits below are written for four activation pieces");
        // an activation piece = one LDS-DMA request; the tap / channel state moves on ONCE per chunk (after its fourth piece), at
        // a fixed point of the schedule -- the conditional request slots below then hold nothing but the request itself
        auto piece = [&](int buf, int z, bool ok) {       // ok = false: past the last chunk (see BLOAD_AT)
            if (p.ablate & 4) return;
            dma(rs_in, ok ? voffA[z] : (int)0x80000000, c0 * 2, lds0 + buf * DSTAGE + wave * 1024 + z * PIECE);
        };
        auto next_chunk_rows = [&]() {
            if (++kw == p.KW) {
                kw = 0;
                if (++kh == p.KH) { kh = 0; c0 += DBK; }
            }
            if (p.KH * p.KW > 1) refresh_rows();
        };
#pragma unroll
        for (int z = 0; z < NP; ++z) piece(0, z, true);
        next_chunk_rows();
        BLOAD_AT(0, 0, true); BLOAD_AT(1, 0, true); BLOAD_AT(2, 0, true);
        int bks_cur = bks;
        b_advance();
        piece(1, 0, KT > 1);
        piece(1, 1, KT > 1);
        asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(fbr[0][0]), "+v"(fbr[0][1]), "+v"(fbr[1][0]), "+v"(fbr[1][1]), "+v"(fbr[2][0]),
                     "+v"(fbr[2][1]) : [n] "n"(C3) : "memory");
        __syncthreads();
#pragma unroll
        for (int z = 0; z < NF; ++z) DFRAG(fa0, fb0, 0, 0, z);
        // Request slots.  All eight waves leave the chunk's barrier together and walk the same schedule: with every wave's
        // requests in the same MFMA slot the texture path sees bursts of 16 requests and every wave queues behind them.
        // STAG (measurement build) spreads them: wave w asks for its weight fragments in slot bsl = 6 w / 8 (k-step 2, which
        // carries no activation piece: slot w) and for its activation pieces two slots later -- the ORDER of a wave's requests,
        // and with it every counted wait, is unchanged.
        const int bsl0 = STAG ? (wave * 6) >> 3 : 0, bsl20 = STAG ? wave : 0, psl0 = STAG ? bsl0 + 2 : NM - 1;
        for (int kt = 0; kt < KT; ++kt) {
            const int buf = kt & 1;
            const bool more = kt + 1 < KT, more2 = kt + 2 < KT;        // wave-uniform: do chunks kt + 1 / kt + 2 exist?
            int bsl = bsl0, bsl2 = bsl20, psl = psl0;
            // (STAG: opaque copies -- otherwise the 40 loop-invariant slot comparisons are hoisted as 40 live SGPR pairs and spill)
            if constexpr (STAG) asm volatile("" : "+s"(bsl), "+s"(bsl2), "+s"(psl));
#pragma unroll
            for (int q = 0; q < NM; ++q) {
                DMFMA(fa0, fbr[0], q);
                DFRAGS(fa1, fb1, buf, 1, q);
                if (q == bsl) BLOAD_AT(3, bks_cur, true);
                if (q == psl) piece(buf ^ 1, 2, more);
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt vmcnt(7)" : "+v"(fbr[1][0]), "+v"(fbr[1][1]) :: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < NM; ++q) {
                DMFMA(fa1, fbr[1], q);
                DFRAGS(fa0, fb0, buf, 2, q);
                if (q == bsl) BLOAD_AT(0, bks, more);
                if (q == psl) piece(buf ^ 1, 3, more);
                __builtin_amdgcn_sched_barrier(0);
            }
            next_chunk_rows();                 // chunk kt + 1 is requested completely: the rows of chunk kt + 2
            asm volatile("s_waitcnt vmcnt(8)" : "+v"(fbr[2][0]), "+v"(fbr[2][1]) :: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < NM; ++q) {
                DMFMA(fa0, fbr[2], q);
                DFRAGS(fa1, fb1, buf, 3, q);
                if (q == bsl2) BLOAD_AT(1, bks, more);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!(p.ablate & 1)) asm volatile("s_waitcnt vmcnt(2)" : "+v"(fbr[3][0]), "+v"(fbr[3][1]), "+v"(fbr[0][0]), "+v"(fbr[0][1]) :: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (!(p.ablate & 2)) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < NM; ++q) {
                DMFMA(fa1, fbr[3], q);
                DFRAGS(fa0, fb0, buf ^ 1, 0, q);
                if (q == bsl) BLOAD_AT(2, bks, more);
                if (q == psl - 1) piece(buf, 0, more2);
                if (q == psl) piece(buf, 1, more2);
                __builtin_amdgcn_sched_barrier(0);
            }
            bks_cur = bks;
            b_advance();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the requests past the last chunk must land before the LDS is released
        dma_epilogue_pairs<MI, true, TR>(p, acc, tm, m0, n0, wm, wn, lane);
        return;
    }
    // prologue: chunk 0 -> stage 0 completely, the first C3 pieces of chunk 1 -> stage 1
#pragma unroll
    for (int z = 0; z < NP; ++z) stage_piece(0, 0, z);
    if (KT > 1) {
#pragma unroll
        for (int z = 0; z < C3; ++z) stage_piece(1, 1, z);
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(C3) : "memory");         // chunk 0 has landed, chunk 1's pieces stay in flight
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
#pragma unroll
    for (int z = 0; z < NF; ++z) DFRAG(fa0, fb0, 0, 0, z);

    // One chunk kt (stage buf = kt & 1).  A burst of LDS-DMA requests stalls the wave AT the requests (each costs 100+ cycles
    // of issue when eight follow each other, against 32 cycles per MFMA), so the pieces of a chunk are spread over three
    // k-steps, one piece per MFMA slot at most, in the slots that carry no (or the last) fragment read (256 x 256: 3 + 3 + 2):
    //   k-step 0: MFMAs | the fragments of k-step 1 | pieces C3.. of chunk kt+1 -> stage buf^1 (free since the last barrier)
    //   k-step 1: MFMAs | the fragments of k-step 2 | the last C1 pieces of chunk kt+1
    //   k-step 2: MFMAs | the fragments of k-step 3 (the last reads of stage buf)
    //   wait: this wave's pieces of chunk kt+1 have landed; barrier: true for every wave, and all reads of stage buf are done
    //   k-step 3: MFMAs | the first fragments of chunk kt+1 | pieces 0 .. C3-1 of chunk kt+2 -> stage buf
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < KT, more2 = kt + 2 < KT;        // wave-uniform
#pragma unroll
        for (int q = 0; q < NM; ++q) {
            DMFMA(fa0, fb0, q);
            DFRAGS(fa1, fb1, buf, 1, q);
            if (more && q >= NM - C0) stage_piece(kt + 1, buf ^ 1, C3 + q - (NM - C0));
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = 0; q < NM; ++q) {
            DMFMA(fa1, fb1, q);
            DFRAGS(fa0, fb0, buf, 2, q);
            if (more && q >= NM - C1) stage_piece(kt + 1, buf ^ 1, C3 + C0 + q - (NM - C1));
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = 0; q < NM; ++q) {
            DMFMA(fa0, fb0, q);
            DFRAGS(fa1, fb1, buf, 3, q);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!(p.ablate & 1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!(p.ablate & 2)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NM; ++q) {
            DMFMA(fa1, fb1, q);
            if (more) DFRAGS(fa0, fb0, buf ^ 1, 0, q);
            if (more2 && q >= NM - C3) stage_piece(kt + 2, buf, q - (NM - C3));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef DFRAGS
#undef DFRAG
#undef DMFMA
#undef BLOAD_AT

    if (nt) {   // this (split, tap)'s fp32 partial
        ConvDmaParams q = p;
        q.out = (float*)p.out + (size_t)st * p.M * p.Cout;
        if constexpr (NJ == 2 && WN == 4) dma_epilogue_pairs<MI>(q, acc, tm, m0, n0, wm, wn, lane);
        else dma_epilogue<MI, NJ>(q, acc, tm, m0, n0, wm, wn, lane);
        return;
    }
    static_assert(WN == 4 || MI * NJ > 2, "the LDS epilogue is written for eight waves");
    // the 256 x 256 instance keeps the direct epilogue: its layers are MFMA-bound 3x3s whose statistics epilogue measured 12 %
    // SLOWER through LDS (1.80 -> 2.06 ms on the head layer at B = 64; profiles/round4_bf16_tiles_and_epilogue.txt)
    if constexpr (NJ == 2 && WN == 4) dma_epilogue_pairs<MI>(p, acc, tm, m0, n0, wm, wn, lane);
    else if constexpr (MI * NJ > 2) dma_epilogue<MI, NJ>(p, acc, tm, m0, n0, wm, wn, lane);
    else {
        if (p.out_fp32 || (p.ablate & 64)) dma_epilogue<MI, NJ>(p, acc, tm, m0, n0, wm, wn, lane);
        else dma_epilogue_lds<MI, NJ>(p, acc, smem, tm, m0, n0, wm, wn, wave, lane);
    }
}

// (A variant with the two wave groups staggered by half a chunk -- activation halves A_g[2] + a ring of three weight stages =
// all 160 KB of LDS, a barrier every two k-steps, each wave waiting down to its newest batch of four requests in front of every
// barrier -- was built in round 3: bit-equal to this kernel on six shapes x six launches, and SLOWER (1.99 vs 1.84 ms on the
// head layer).  With full-workgroup barriers between phases the group that reads its first fragments after a barrier is the
// slower one of that phase and its partner waits at the next barrier: the exposed latency is not hidden, only moved, and the
// barrier count doubles.  What the guide's 8-phase template does instead is a ping-pong -- memory part | barrier | MFMA cluster |
// barrier, the partner one barrier behind -- i.e. a different kernel, not a re-timing of this one.  Removed; see git history and
// profiles/round3_bf16_dma_kernel_development.txt.)

void conv_bf16_pp_launch(const ConvDmaParams& p, unsigned grid, hipStream_t stream);     // conv_bf16_pp.hip

// The launcher of conv_mfma_bf16.hip calls this for the layers that fit the tile; returns CPR_ERR_UNSUPPORTED otherwise.
// shape: 0 = 256 x 256, 1 = 128 (pixels) x 256 (couts), 2 = 256 x 128, 3 = 128 x 128 (64 KB of LDS: two workgroups per CU),
// 4 = 256 x 256 on four waves (wave = 128 x 128), 5 = 256 x 256 in two-group ping-pong (conv_bf16_pp.hip; shape 0 takes it by itself for >= 4 K chunks).
int conv_bf16_dma_launch(const void* in, const void* wgt, void* out, const float* scale, const float* bias,
                         const void* residual, float* gn_part, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                         int stride, int pad, int Kpad, int relu, int out_fp32, int* variant_out, hipStream_t stream,
                         int ablate, int shape, const void* wfrag, const float* add32, void* out16) {
    const int bm = (shape == 1 || shape == 3) ? 128 : 256, bn = (shape == 2 || shape == 3) ? 128 : 256;
    if (Cout % bn != 0 || Cin % DBK != 0 || Kpad != KH * KW * Cin) return CPR_ERR_UNSUPPORTED;
    ConvDmaParams p;
    p.in = (const unsigned short*)in; p.wgt = (const unsigned short*)wgt; p.out = out; p.scale = scale; p.bias = bias;
    p.wfrag = (const unsigned short*)wfrag;
    p.residual = (const unsigned short*)residual; p.gn_part = gn_part;
    p.add32 = add32; p.out16 = (unsigned short*)out16;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad;
    p.Kpad = Kpad; p.relu = relu; p.out_fp32 = out_fp32; p.ablate = ablate;
    p.nt_taps = 0; p.nt_k = 1; p.nt_pad = 0; p.nt_Wp = 0; p.nt_chunks = 0; p.nt_copy = 0;
    p.OH = (H + 2 * pad - KH) / stride + 1;
    p.OW = (W + 2 * pad - KW) / stride + 1;
    const long long M = (long long)N * p.OH * p.OW;
    if ((long long)N * H * W * Cin * 2 >= (1ll << 31) || (long long)Cout * Kpad * 2 >= (1ll << 31) || M >= (1ll << 31) ||
        M * Cout * (out_fp32 ? 4 : 2) >= (1ll << 31))
        return CPR_ERR_UNSUPPORTED;
    p.M = (int)M;
    const bool stats = gn_part && relu != 2;                      // GroupNorm statistics (mask mode: plain column sums, any slot shape)
    if (stats && (p.OH * p.OW) % 128 != 0) return CPR_ERR_UNSUPPORTED;
    if (stats && M % 256 != 0) return CPR_ERR_UNSUPPORTED;        // both 128-pixel slots of every tile must exist
    if (stats && bm != 256) return CPR_ERR_UNSUPPORTED;           // a wave of the 128-pixel tile owns half a statistics slot
    if (relu == 2 && !residual) return CPR_ERR_UNSUPPORTED;
    p.tilesM = (int)((M + bm - 1) / bm);
    p.tilesN = Cout / bn;
    // the 256 x 256 tile has three instances: both operands through LDS in lock step (no fragment image given: the reference the
    // other two are bit-equal to), two-group ping-pong (conv_bf16_pp.hip: >= 4 K chunks), weights direct to registers (short K)
    if (shape == 0 && wfrag != nullptr && Kpad / DBK >= 4 && !(ablate & 512)) shape = 5;      // (ablate bit 9, measurement build: keep shape 0 off the ping-pong instance)
    const bool bd = shape == 0 && wfrag != nullptr;      // the 256 x 256 instance with the weights direct to registers
    // the training epilogue (shortcut sum + mask + two outputs + column sums) exists for the two pair-epilogue instances
    const bool tr = add32 != nullptr;
    if (tr && !((bd || shape == 5) && relu == 2 && out_fp32 && out16 && gn_part && residual)) return CPR_ERR_UNSUPPORTED;
    if (variant_out) *variant_out = bm * 1000 + bn + (shape == 3 ? 1000000 : shape == 4 ? 2000000 : shape == 5 ? 5000000 : bd ? 3000000 : 0);       // 128128 alone is the register-staged <128, 128>
    const int T = p.tilesM * p.tilesN;
    const int grid = ((T + 7) / 8) * 8;
    if (shape == 1 || shape == 2) return CPR_ERR_UNSUPPORTED;      // <2, 2> and <4, 1> were measured and dropped
    if (shape == 5) conv_bf16_pp_launch(p, (unsigned)grid, stream);
    else if (bd && tr) hipLaunchKernelGGL((conv_bf16_dma_kernel<4, 2, 4, true, false, true>), dim3(grid), dim3(512), 0, stream, p);
    else if (shape == 3) hipLaunchKernelGGL((conv_bf16_dma_kernel<2, 1, 4>), dim3(grid), dim3(512), 0, stream, p);
#ifdef CPR_BENCH_HOOKS   // measured 10-12 % slower than the eight-wave instance (profiles/round4_bf16_four_wave_tile.txt): measurement build only
    else if (shape == 4) hipLaunchKernelGGL((conv_bf16_dma_kernel<4, 4, 2>), dim3(grid), dim3(256), 0, stream, p);
#else
    else if (shape == 4) return CPR_ERR_UNSUPPORTED;
#endif
#ifdef CPR_BENCH_HOOKS
    else if (bd && (ablate & 256)) hipLaunchKernelGGL((conv_bf16_dma_kernel<4, 2, 4, true, true>), dim3(grid), dim3(512), 0, stream, p);
#endif
    else if (bd) hipLaunchKernelGGL((conv_bf16_dma_kernel<4, 2, 4, true>), dim3(grid), dim3(512), 0, stream, p);
    else hipLaunchKernelGGL((conv_bf16_dma_kernel<4, 2, 4>), dim3(grid), dim3(512), 0, stream, p);
    CPR_LAUNCH_STATUS();
}

// NT-GEMM launch for the bf16 weight gradient (csrc/conv_wgrad_bf16.hip): part[split][tap][m][n] (fp32) = sum over the split's
// K range of a[m][k] * b_tap[n][k]; a: M rows of `rs` bf16 elements (k = 0 at a), b: `k` copies (`copy` elements apart) of N rows
// of `rs` elements, tap t = (t / k, t % k) reads copy t % k shifted by (t / k - pad) * Wp elements.  N % 256 == 0.
int conv_bf16_dma_nt_launch(const void* a, const void* b, float* part, int M, int N, long long rs, int k, int pad, int Wp,
                            long long copy, int splits, int chunks, hipStream_t stream) {
    constexpr int DBM = 256, DBN = 256;
    if (N % DBN != 0 || rs % 8 != 0 || M <= 0 || splits <= 0 || splits % 8 != 0 || chunks <= 0) return CPR_ERR_UNSUPPORTED;
    if ((long long)M * rs * 2 >= (1ll << 31) || (long long)N * rs * 2 >= (1ll << 31) || rs >= (1ll << 30)) return CPR_ERR_UNSUPPORTED;
    ConvDmaParams p;
    p.in = (const unsigned short*)a; p.wgt = (const unsigned short*)b; p.wfrag = nullptr; p.out = part; p.scale = nullptr; p.bias = nullptr;
    p.residual = nullptr; p.gn_part = nullptr; p.add32 = nullptr; p.out16 = nullptr;
    p.N = 1; p.H = 1; p.W = M; p.Cin = (int)rs; p.Cout = N; p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0;
    p.Kpad = (int)rs; p.relu = 0; p.out_fp32 = 1; p.ablate = 0;
    p.OH = 1; p.OW = M; p.M = M;
    p.tilesM = (M + DBM - 1) / DBM; p.tilesN = N / DBN;
    p.nt_taps = k * k; p.nt_k = k; p.nt_pad = pad; p.nt_Wp = Wp; p.nt_chunks = chunks; p.nt_copy = copy;
    const long long blocks = (long long)splits * p.nt_taps * p.tilesM * p.tilesN;
    if (blocks >= (1ll << 31)) return CPR_ERR_UNSUPPORTED;
    // the ping-pong instance (conv_bf16_pp.hip) takes this mode too (bit-equal partials).  First measured SLOWER (configs[4] training
    // step 55.3 ms with it, 54.3 without: profiles/round6_mixed_train_ab.txt); after the backward lost its streaming passes the same
    // A/B reads 47.7 vs 48.8 ms on configs[4] and 95.0 vs 96.9 ms on R50 640^2 B = 64 (profiles/round6_mixed_backward_ab.txt, two
    // boxes alike): the weight gradients share the chip with fewer HBM-bound kernels of the main stream.  Default: wherever it can
    // run (>= 4 chunks per split); CPR_BF16_NT_PP=0 keeps the lock-step instance, =n asks for splits of >= n chunks.
    static const int nt_pp_min = []() { const char* e = getenv("CPR_BF16_NT_PP"); const int v = e ? atoi(e) : 1; return v <= 0 ? (1 << 30) : v < 4 ? 4 : v; }();
    if (chunks >= nt_pp_min) conv_bf16_pp_launch(p, (unsigned)blocks, stream);
    else hipLaunchKernelGGL((conv_bf16_dma_kernel<4, 2, 4>), dim3((unsigned)blocks), dim3(512), 0, stream, p);
    CPR_LAUNCH_STATUS();
}
