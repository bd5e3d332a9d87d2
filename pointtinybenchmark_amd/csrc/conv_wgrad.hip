// Weight gradient of the implicit-GEMM convolution, exact fp32 on the matrix cores (backward).
//   dW[co][kh][kw][ci] = sum_m dY[m][co] * X[(n, oy*s + kh - pad, ox*s + kw - pad)][ci],   m = (n, oy, ox)
// GEMM view: rows = cout, cols = (tap, cin), reduction over the M output pixels.  One workgroup owns a
// 128(cout) x 128(cin of ONE tap) tile and a slab of the pixel range (split-K); slabs are summed by wgrad_reduce, which
// also converts to the parameter layout [Cout][Cin][KH][KW] (deterministic: no float atomics).
// Both operands are pixel-major (NHWC), i.e. the reduction index is the slow one: tiles are staged in LDS k-major
// ([32 pixels][128 channels]) and MFMA fragments are read with ds_read_b32 (32 consecutive channels per half-wave:
// conflict-free).  The X operand takes the conv zero padding from the buffer-load range check and, optionally, the fused
// GroupNorm-apply + ReLU of the producing layer (the forward never materialised that tensor).
#include "common.h"

struct WgradParams {
    const float* dy;   // (N, OH, OW, Cout)
    const float* x;    // (N, H, W, Cin)
    const float* in_a; // (N, Cin) or null
    const float* in_b;
    float* part;       // [S][Cout][KH*KW][Cin]
    int N, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad, M, in_relu;
    int tilesCo, tilesCi, S, chunks_per_slab;
};

constexpr int WG_BK = 32;     // pixels per chunk
constexpr int WG_T = 128;     // tile edge (cout and cin)
constexpr int WG_LD = 132;    // LDS row stride in floats (132 = 128 + 4 keeps float4 writes 16-byte aligned)

// ABL: benchmark-only ablations of the main loop (cpr_wgrad_set_ablation; results are wrong by construction):
//   1 no global loads, 2 no LDS stores, 4 no barrier, 8 no fragment reads, 16 loads without address advance,
//   32 address advance without loads
// GEO: geometry class of the launch, fixed at compile time so the per-row state update carries no branch and as few VALU
// instructions as the geometry allows (measured: every VALU instruction between two MFMAs costs ~12 cycles of matrix-pipe
// time -- the two workgroups of a CU run in lockstep, so nothing else fills the pipe):
//   2  1x1, stride 1, no padding: both operands advance linearly, no validity besides m < M
//   1  'same' convolutions (stride 1, OH == H, OW == W): the input row of (pixel m, tap) is m + const, only the border
//      masks need the pixel's (ox, oy)
//   0  general (strided): full input-coordinate tracking
//  -1  images with fewer than 32 output pixels (unit-test sizes): decode from scratch every chunk
template <bool XF, int ABL = 0, int GEO = 0>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(WgradParams p) {
    __shared__ __attribute__((aligned(16))) float smem[2 * 2 * WG_BK * WG_LD];
    float* As = smem;                         // [2][32][132]  dY tile (k = pixel, i = cout)
    float* Bs = smem + 2 * WG_BK * WG_LD;     // [2][32][132]  X tile  (k = pixel, j = cin)
    const int KK = p.KH * p.KW;
    // XCD-aware order: workgroup b runs on XCD b % 8.  All tiles of one pixel slab (they stream the same dY / X rows) are
    // consecutive workgroups of ONE XCD, so the slab is fetched from HBM once and re-read from that XCD's L2 as a sliding
    // window by the other tiles; slabs are dealt round-robin to the XCDs (S % 8 == 0).
    const int xcd = blockIdx.x & 7;
    const int local = blockIdx.x >> 3;
    const int tiles = p.tilesCo * KK * p.tilesCi;
    const int slab = (local / tiles) * 8 + xcd;
    int b = local % tiles;
    const int tci = b % p.tilesCi; b /= p.tilesCi;
    const int tap = b % KK; b /= KK;
    const int tco = b;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    const int co0 = tco * WG_T, ci0 = tci * WG_T;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c4 = tid & 31, kr = tid >> 5;   // 32 threads (x float4) cover a 128-channel row; 8 rows per pass

    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.dy), 0, (int)((size_t)p.M * p.Cout * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (int)((size_t)p.N * p.H * p.W * p.Cin * 4), 0x00020000);
    const int tab_bytes = XF ? p.N * p.Cin * 4 : 0;
    const __amdgpu_buffer_rsrc_t rs_ta = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in_a), 0, tab_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_tb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in_b), 0, tab_bytes, 0x00020000);
    const bool co_ok = co0 + c4 * 4 < p.Cout;   // Cout % 4 == 0 is checked by the launcher
    const bool ci_ok = ci0 + c4 * 4 < p.Cin;

    const int m_begin = slab * p.chunks_per_slab * WG_BK;
    const int nchunks = min(p.chunks_per_slab, (p.M - m_begin + WG_BK - 1) / WG_BK);
    const int ohw = p.OH * p.OW;
    // per-thread pixel rows kr + 8j of the current chunk: decoded once, then advanced by 32 pixels per chunk with adds and
    // selects of wave-uniform constants only (integer multiplies are quarter rate and sit in the MFMA shadow otherwise)
    const int q32 = WG_BK / p.OW, r32 = WG_BK - q32 * p.OW;
    constexpr bool big_map = GEO >= 0;      // 32 pixels cross at most one image boundary
    // byte offsets of this thread's channel group; a group beyond the tensor's channels gets 2^31, which puts every address
    // built on it outside num_records (< 2^31) -- no per-load mask for the channel-tile overhang
    const int cho = co_ok ? (co0 + c4 * 4) * 4 : (int)0x80000000, chi = ci_ok ? (ci0 + c4 * 4) * 4 : (int)0x80000000;
    const int pxb = p.Cin * 4, rowb = p.W * pxb;                    // bytes per input pixel / input row
    const int dX0 = r32 * p.stride, dX1 = (r32 - p.OW) * p.stride;  // input-x step without / with a row wrap
    const int dY0 = q32 * p.stride, dY1 = (q32 + 1) * p.stride, dYw = p.OH * p.stride;
    const int bX0 = dX0 * pxb, bX1 = dX1 * pxb, bY0 = dY0 * rowb, bY1 = dY1 * rowb;
    const int bN = (p.H - dYw) * rowb;                              // image wrap: +H rows, -OH*stride rows
    const int va_step = WG_BK * p.Cout * 4, vb_step = WG_BK * pxb;
    const int dkh = kh - p.pad, dkw = kw - p.pad;
    int rm[4], rox[4], roy[4], rix[4], riy[4], va[4], vb[4], vt[4];
    auto decode_row = [&](int j) {
        const int m = rm[j];
        const int n = m / ohw;
        const int rem = m - n * ohw;
        roy[j] = rem / p.OW;
        rox[j] = rem - roy[j] * p.OW;
        riy[j] = roy[j] * p.stride + kh - p.pad;
        rix[j] = rox[j] * p.stride + kw - p.pad;
        va[j] = m * p.Cout * 4 + cho;
        vb[j] = ((n * p.H + riy[j]) * p.W + rix[j]) * pxb + chi;
        vt[j] = n * pxb + chi;
        if (GEO == 2) rox[j] = rem;           // pixel index inside the image (see advance_row)
    };
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        rm[j] = m_begin + kr + 8 * j;
        decode_row(j);
    }

    // Global-load register sets.  The plain variant keeps TWO chunks in flight (loads are issued 2.5 chunks = ~7000 cycles
    // before their LDS store: first-touch HBM latency under load exceeds the ~2800 cycles of a single-chunk distance and
    // every stalled wave holds its workgroup at the barrier); the XF variant has no registers left for a second set.
    constexpr int NSET = XF ? 1 : 2;
    f32x4 ra[NSET][4], rb[NSET][4], xa[4], xb[4];
    // Row pieces.  issue_row(j): the loads of row j at its current state.  advance_rows(stage): the state update of ALL
    // four rows, cut into three short stages that go behind three different MFMAs -- measured (cpr_wgrad_set_ablation 32):
    // one row's update as a single dependent VALU chain behind one MFMA costs 16 % of the kernel (the chain's
    // issue-to-use latencies add up to several MFMA times); four independent chains per stage hide each other.
    auto issue_row = [&](int j, f32x4 (&RA)[4], f32x4 (&RB)[4]) {
        // No m < M test: rows past the last pixel lie beyond num_records of dY (and of X for the linear geometries), and
        // whatever X row a negative tap shift still reaches there is multiplied by a dY row that read as zero.
        bool xok = true;
        if (GEO == 1) xok = xok & ((unsigned)(roy[j] + dkh) < (unsigned)p.H) & ((unsigned)(rox[j] + dkw) < (unsigned)p.W);
        else if (GEO <= 0) xok = xok & ((unsigned)riy[j] < (unsigned)p.H) & ((unsigned)rix[j] < (unsigned)p.W);
        // out-of-range rows get voffset -1 (= beyond num_records: the load returns 0); OR with an all-ones mask instead
        // of a select so the compiler cannot turn the address arithmetic into divergent control flow
        const int xmask = -(int)(!xok);
        if (!(ABL & 32)) {
            RA[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, va[j], 0, 0));
            RB[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, vb[j] | xmask, 0, 0));
        } else {   // address math only: keep it alive without issuing the loads
            asm volatile("" ::"v"(va[j]), "v"(vb[j] | xmask));
        }
        if (XF) {   // rows that must read as zero get a = b = 0 from the range check as well (branch-free)
            xa[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_ta, vt[j] | xmask, 0, 0));
            xb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_tb, vt[j] | xmask, 0, 0));
        }
    };
    bool wcx[4], wcy[4];
    // one micro-piece of the state update: (row j, stage) -- 12 per chunk, each behind its own MFMA.  Non-MFMA work has
    // to be spread evenly: a slot holding more than ~10 VALU instructions outlasts its MFMA (64 cycles), and since the two
    // workgroups of a CU run in lockstep nothing fills the matrix pipe meanwhile (cpr_wgrad_set_ablation 32 measured the
    // whole update packed into 4 slots at 16 % of the kernel).
    auto advance_row = [&](int j, int stage) {
        if (ABL & 16) return;      // loads from a fixed address every chunk (cache hits, no address arithmetic)
        if (!big_map) {            // tiny maps (unit-test sizes): decode from scratch, once
            if (stage == 0) { rm[j] += WG_BK; decode_row(j); }
            return;
        }
        if (GEO == 2) {            // 1x1 stride 1: linear; the image index only matters for the fused GroupNorm table
            if (stage == 0) {
                rm[j] += WG_BK;
                va[j] += va_step;
                vb[j] += vb_step;
            } else if (stage == 1 && XF) {
                const int rp = rox[j] + WG_BK;          // rox doubles as the pixel index inside the image
                wcy[j] = rp >= ohw;
                rox[j] = wcy[j] ? rp - ohw : rp;
                vt[j] += wcy[j] ? pxb : 0;
            }
            return;
        }
        if (GEO == 1) {            // 'same' conv: linear addresses, (ox, oy) only for the border masks
            if (stage == 0) {
                rm[j] += WG_BK;
                va[j] += va_step;
                vb[j] += vb_step;
                const int ox = rox[j] + r32;
                wcx[j] = ox >= p.OW;
                rox[j] = wcx[j] ? ox - p.OW : ox;
            } else if (stage == 1) {
                const int oy = roy[j] + q32 + (wcx[j] ? 1 : 0);
                wcy[j] = oy >= p.OH;
                roy[j] = wcy[j] ? oy - p.OH : oy;
                if (XF) vt[j] += wcy[j] ? pxb : 0;
            }
            return;
        }
        if (stage == 0) {          // wrap flags and the output-pixel coordinates
            rm[j] += WG_BK;
            va[j] += va_step;
            const int ox = rox[j] + r32;
            wcx[j] = ox >= p.OW;
            rox[j] = wcx[j] ? ox - p.OW : ox;
            const int oy = roy[j] + q32 + (wcx[j] ? 1 : 0);
            wcy[j] = oy >= p.OH;
            roy[j] = wcy[j] ? oy - p.OH : oy;
        } else if (stage == 1) {   // input coordinates
            rix[j] += wcx[j] ? dX1 : dX0;
            riy[j] += (wcx[j] ? dY1 : dY0) - (wcy[j] ? dYw : 0);
        } else {                   // byte offsets
            vb[j] += (wcx[j] ? bX1 + bY1 : bX0 + bY0) + (wcy[j] ? bN : 0);
            vt[j] += wcy[j] ? pxb : 0;
        }
    };
    auto load_row = [&](int j, f32x4 (&RA)[4], f32x4 (&RB)[4]) {   // prologue form: issue + the whole advance
        issue_row(j, RA, RB);
        advance_row(j, 0); advance_row(j, 1); advance_row(j, 2);
    };
    const float relu_floor = p.in_relu ? 0.f : -INFINITY;
    // store piece z: z < 4 -> dY row z, else X row z-4 (with the fused GroupNorm affine + ReLU)
    auto store_piece = [&](int buf, int z, const f32x4 (&RA)[4], const f32x4 (&RB)[4]) {
        if (z < 4) {
            *reinterpret_cast<f32x4*>(As + (buf * WG_BK + kr + 8 * z) * WG_LD + c4 * 4) = RA[z];
        } else {
            const int j = z - 4;
            f32x4 v = RB[j];
            if (XF) {  // padded / out-of-range entries have a = b = 0 -> max(0, floor) = 0 for either floor
                v = v * xa[j] + xb[j];
                v.x = fmaxf(v.x, relu_floor); v.y = fmaxf(v.y, relu_floor);
                v.z = fmaxf(v.z, relu_floor); v.w = fmaxf(v.w, relu_floor);
            }
            *reinterpret_cast<f32x4*>(Bs + (buf * WG_BK + kr + 8 * j) * WG_LD + c4 * 4) = v;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int wm = wave & 1, wn = wave >> 1;
    const int half = lane >> 5;
    // Four base pointers, each opaque to the compiler: fragment reads then compile to ds_read_b32 with a 16-bit immediate
    // offset and NO per-read address arithmetic.  (With the two 32-channel sub-tiles addressed off one pointer the compiler
    // pairs them into ds_read2_b32, whose 8-bit dword offsets cannot reach the next k row (264 dwords), so every read
    // needed a v_add -- and each VALU instruction between two MFMAs costs ~12 cycles of matrix-pipe time here.)
    int ia0 = half * WG_LD + wm * 64 + (lane & 31), ia1 = ia0 + 32;                       // dword indices into smem
    int ib0 = 2 * WG_BK * WG_LD + half * WG_LD + wn * 64 + (lane & 31), ib1 = ib0 + 32;
    asm volatile("" : "+v"(ia0), "+v"(ia1), "+v"(ib0), "+v"(ib1));

    // A "k-group" is 4 MFMA k-steps (8 pixels: lanes < 32 take the even pixel of a step, lanes >= 32 the odd one) =
    // 16 MFMAs.  Fragment sets: F[step][0..1] = A values of the two 32-row sub-tiles, F[step][2..3] = B values.
    // Same interleaved schedule as the forward kernel (conv_mfma.hip): every non-MFMA instruction of a chunk sits
    // behind one of that wave's own MFMAs, because the two workgroups sharing a CU run in lockstep:
    //   g0: MFMAs(set0) | read set1 <- group 1
    //   g1: MFMAs(set1) | read set0 <- group 2 | write chunk c+1 (registers loaded during chunk c-1) to the other buffer
    //   g2: MFMAs(set0) | read set1 <- group 3 | issue the global loads of chunk c+2
    //   barrier
    //   g3: MFMAs(set1) | read set0 <- group 0 of chunk c+1 (other buffer, now complete)
    float f0[4][4], f1[4][4];
#define WG_FRAG(F, buf, g, z)                                                                         \
    do {                                                                                              \
        const int off_ = ((buf) * WG_BK + (g) * 8 + ((z) >> 1) * 2) * WG_LD;                          \
        F[(z) >> 1][((z) & 1) * 2] = smem[((z) & 1 ? ib0 : ia0) + off_];                              \
        F[(z) >> 1][((z) & 1) * 2 + 1] = smem[((z) & 1 ? ib1 : ia1) + off_];                          \
    } while (0)
#define WG_MFMA(F, q)                                                                                 \
    acc[((q) >> 1) & 1][(q) & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(                              \
        F[(q) >> 2][((q) >> 1) & 1], F[(q) >> 2][2 + ((q) & 1)], acc[((q) >> 1) & 1][(q) & 1], 0, 0, 0)
    // one chunk; SS = register set holding chunk c+1 (stored in g1) and receiving the next loads (g2)
#define WG_CHUNK(c, SS)                                                                               \
    do {                                                                                              \
        const int buf = (c) & 1;                                                                      \
        _Pragma("unroll") for (int q = 0; q < 16; ++q) {                                              \
            WG_MFMA(f0, q);                                                                           \
            if (q < 8 && !(ABL & 8)) WG_FRAG(f1, buf, 1, q);                                          \
            __builtin_amdgcn_sched_barrier(0);                                                        \
        }                                                                                             \
        _Pragma("unroll") for (int q = 0; q < 16; ++q) {                                              \
            WG_MFMA(f1, q);                                                                           \
            if (q < 8) { if (!(ABL & 8)) WG_FRAG(f0, buf, 2, q); }                                    \
            else if (!(ABL & 2)) store_piece(buf ^ 1, q - 8, ra[SS], rb[SS]);                         \
            __builtin_amdgcn_sched_barrier(0);                                                        \
        }                                                                                             \
        _Pragma("unroll") for (int q = 0; q < 16; ++q) {                                              \
            WG_MFMA(f0, q);                                                                           \
            if (q < 8) { if (!(ABL & 8)) WG_FRAG(f1, buf, 3, q); }                                    \
            else if (q < 12) { if (!(ABL & 1)) issue_row(q - 8, ra[SS], rb[SS]); }                    \
            else { if (!(ABL & 1)) advance_row(q - 12, 0); }                                          \
            __builtin_amdgcn_sched_barrier(0);                                                        \
        }                                                                                             \
        asm volatile("" ::: "memory"); /* no LDS access may be moved across the raw barrier */        \
        __builtin_amdgcn_s_waitcnt(0xc07f); /* lgkmcnt(0) only: global loads stay in flight */        \
        if (!(ABL & 4)) __builtin_amdgcn_s_barrier();                                                 \
        asm volatile("" ::: "memory");                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        _Pragma("unroll") for (int q = 0; q < 16; ++q) {                                              \
            WG_MFMA(f1, q);                                                                           \
            if (q < 8) { if (!(ABL & 8)) WG_FRAG(f0, buf ^ 1, 0, q); }                                \
            else if (!(ABL & 1)) advance_row(q & 3, q < 12 ? 1 : 2);                                  \
            __builtin_amdgcn_sched_barrier(0);                                                        \
        }                                                                                             \
    } while (0)
    if (nchunks > 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) load_row(j, ra[0], rb[0]);                 // chunk 0
#pragma unroll
        for (int z = 0; z < 8; ++z) store_piece(0, z, ra[0], rb[0]);
#pragma unroll
        for (int j = 0; j < 4; ++j) load_row(j, ra[NSET - 1], rb[NSET - 1]);   // chunk 1
        if (NSET == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) load_row(j, ra[0], rb[0]);             // chunk 2
        }
        __syncthreads();
#pragma unroll
        for (int z = 0; z < 8; ++z) WG_FRAG(f0, 0, 0, z);
        if (NSET == 2) {   // chunk c stores chunk c+1 from set (c+1)&1 and refills it with chunk c+3
            int c = 0;
            for (; c + 1 < nchunks; c += 2) {
                WG_CHUNK(c, NSET - 1);
                WG_CHUNK(c + 1, 0);
            }
            if (c < nchunks) WG_CHUNK(c, NSET - 1);
        } else {
            for (int c = 0; c < nchunks; ++c) WG_CHUNK(c, 0);
        }
    }
#undef WG_CHUNK
#undef WG_FRAG
#undef WG_MFMA
    // partial[slab][co][tap][ci]: col j = lane&31 (cin), row i = (r&3) + 8*(r>>2) + 4*half (cout)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ci = ci0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (co < p.Cout && ci < p.Cin)
                    p.part[(((size_t)slab * p.Cout + co) * KK + tap) * p.Cin + ci] = acc[i][j][r];
            }
        }
    }
}

// sum the slabs; write (or accumulate into) the parameter-layout gradient [Cout][Cin][KH][KW]
__global__ void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ grad, int S, int Cout, int KK,
                                    int Cin, int accumulate) {
    const long long total = (long long)Cout * KK * Cin;
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= total) return;
    float s = 0.f;
    int k = 0;
    for (; k + 8 <= S; k += 8) {   // 8 loads in flight (the kernel is latency-bound otherwise); fixed order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(k + u) * total + i];
        s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    for (; k < S; ++k) s += part[(size_t)k * total + i];
    const int ci = (int)(i % Cin);
    const long long r = i / Cin;
    const int tap = (int)(r % KK), co = (int)(r / KK);
    float* dst = grad + ((size_t)co * Cin + ci) * KK + tap;
    *dst = accumulate ? *dst + s : s;
}

// workspace floats needed by cpr_conv2d_wgrad for these shapes (the split factor is chosen here, once)
static int wgrad_split(long long M, int Cout, int Cin, int KK) {
    const long long tiles = (long long)((Cout + WG_T - 1) / WG_T) * ((Cin + WG_T - 1) / WG_T) * KK;
    const long long chunks = (M + WG_BK - 1) / WG_BK;
    // S % 8 == 0 (slabs are dealt to the 8 XCDs).  The chip holds 512 workgroups at a time (2 per CU): pick the slab count
    // whose grid fills whole rounds of 512 best (a 3.4-round grid runs at 3.4/4 of peak), between ~2 and ~8 rounds.
    long long smax = (4096 + tiles - 1) / tiles;
    smax = (smax + 7) / 8 * 8;
    if (smax > 256) smax = 256;
    while (smax > 8 && smax > chunks) smax -= 8;       // no more slabs than chunks (empty slabs still write zeros)
    long long smin = (1024 + tiles - 1) / tiles;
    smin = (smin + 7) / 8 * 8;
    if (smin > smax) smin = smax;
    long long S = smin;
    double best = 0.0;
    for (long long c = smin; c <= smax; c += 8) {
        const double rounds = (double)(c * tiles) / 512.0;
        const double eff = rounds / (double)((c * tiles + 511) / 512);
        if (eff > best + 0.02) { best = eff; S = c; }  // prefer fewer slabs unless clearly better
    }
    return (int)S;
}
#ifdef CPR_BENCH_HOOKS   // benchmark-only loop ablations (results are wrong by construction): tools builds only
static int wgrad_ablate = 0;
extern "C" int cpr_wgrad_set_ablation(int mode) {
    CPR_CHECK_ARG(mode >= 0 && mode <= 32);
    wgrad_ablate = mode;
    return CPR_OK;
}
#endif
// (the slab count never grows when M shrinks, so the workspace of the whole batch also serves every chunk of a split batch)
extern "C" int cpr_conv2d_wgrad_workspace(int N, int OH, int OW, int Cin, int Cout, int KH, int KW) {
    CPR_CHECK_ARG(N > 0 && OH > 0 && OW > 0 && Cin > 0 && Cout > 0 && KH > 0 && KW > 0);
    const int S = wgrad_split((long long)N * OH * OW, Cout, Cin, KH * KW);
    const long long n = (long long)S * Cout * KH * KW * Cin;
    return n < (1ll << 31) ? (int)n : CPR_ERR_UNSUPPORTED;
}

static int conv2d_wgrad_launch(const float* dy, const float* x, const float* in_a, const float* in_b, float* grad_w,
                               float* ws, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                               int in_relu, int accumulate, hipStream_t stream) {
    CPR_CHECK_ARG(dy && x && grad_w && ws);
    CPR_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0);
    CPR_CHECK_ARG(Cin % 4 == 0 && Cout % 4 == 0);
    if (in_a) CPR_CHECK_ARG(in_b != nullptr);
    WgradParams p;
    p.dy = dy; p.x = x; p.in_a = in_a; p.in_b = in_b; p.part = ws;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad;
    p.in_relu = in_relu;
    p.OH = (H + 2 * pad - KH) / stride + 1;
    p.OW = (W + 2 * pad - KW) / stride + 1;
    CPR_CHECK_ARG(p.OH > 0 && p.OW > 0);
    const long long M = (long long)N * p.OH * p.OW;
    if (M * Cout * 4 >= (1ll << 31) || (long long)N * H * W * Cin * 4 >= (1ll << 31)) return CPR_ERR_UNSUPPORTED;
    p.M = (int)M;
    const int KK = KH * KW;
    p.tilesCo = (Cout + WG_T - 1) / WG_T;
    p.tilesCi = (Cin + WG_T - 1) / WG_T;
    p.S = wgrad_split(M, Cout, Cin, KK);
    const int chunks = (int)((M + WG_BK - 1) / WG_BK);
    p.chunks_per_slab = (chunks + p.S - 1) / p.S;
    const long long grid = (long long)p.tilesCo * KK * p.tilesCi * p.S;
    CPR_CHECK_ARG(grid < (1ll << 31));
    const int geo = (p.OH * p.OW < WG_BK) ? -1 : (KK == 1 && stride == 1 && pad == 0) ? 2
                    : (stride == 1 && p.OH == H && p.OW == W) ? 1 : 0;
#define WG_LAUNCH(XF_, ABL_, GEO_) \
    hipLaunchKernelGGL((conv_wgrad_kernel<XF_, ABL_, GEO_>), dim3((unsigned)grid), dim3(256), 0, stream, p)
#ifdef CPR_BENCH_HOOKS
    if (wgrad_ablate && !in_a && geo == 1) {
        switch (wgrad_ablate) {
            case 1: WG_LAUNCH(false, 1, 1); break;
            case 2: WG_LAUNCH(false, 2, 1); break;
            case 4: WG_LAUNCH(false, 4, 1); break;
            case 8: WG_LAUNCH(false, 8, 1); break;
            case 16: WG_LAUNCH(false, 16, 1); break;
            case 32: WG_LAUNCH(false, 32, 1); break;
            default: WG_LAUNCH(false, 15, 1); break;
        }
    } else
#endif
    if (in_a) {
        if (geo == 2) WG_LAUNCH(true, 0, 2); else if (geo == 1) WG_LAUNCH(true, 0, 1);
        else if (geo == 0) WG_LAUNCH(true, 0, 0); else WG_LAUNCH(true, 0, -1);
    } else {
        if (geo == 2) WG_LAUNCH(false, 0, 2); else if (geo == 1) WG_LAUNCH(false, 0, 1);
        else if (geo == 0) WG_LAUNCH(false, 0, 0); else WG_LAUNCH(false, 0, -1);
    }
#undef WG_LAUNCH
    const long long total = (long long)Cout * KK * Cin;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)cdivll(total, 256)), dim3(256), 0, stream, ws, grad_w, p.S, Cout,
                       KK, Cin, accumulate);
    CPR_LAUNCH_STATUS();
}

// >= 2 GiB maps: balanced chunks of whole images, every chunk after the first accumulates into grad_w (deterministic; the
// summation is grouped per chunk, so the result equals the unsplit one up to fp32 rounding, not bit for bit).
extern "C" int cpr_conv2d_wgrad(const float* dy, const float* x, const float* in_a, const float* in_b, float* grad_w,
                                float* ws, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                                int in_relu, int accumulate, hipStream_t stream) {
    CPR_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0);
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    CPR_CHECK_ARG(OH > 0 && OW > 0);
    const int per = cpr_images_per_launch(N, cpr_max2((long long)OH * OW * Cout * 4, (long long)H * W * Cin * 4));
    if (per <= 0) return CPR_ERR_UNSUPPORTED;
    for (int n0 = 0; n0 < N; n0 += per) {
        const int n = N - n0 < per ? N - n0 : per;
        const int rc = conv2d_wgrad_launch(dy + (size_t)n0 * OH * OW * Cout, x + (size_t)n0 * H * W * Cin,
                                           in_a ? in_a + (size_t)n0 * Cin : nullptr, in_b ? in_b + (size_t)n0 * Cin : nullptr,
                                           grad_w, ws, n, H, W, Cin, Cout, KH, KW, stride, pad, in_relu,
                                           n0 == 0 ? accumulate : 1, stream);
        if (rc != CPR_OK) return rc;
    }
    return CPR_OK;
}
