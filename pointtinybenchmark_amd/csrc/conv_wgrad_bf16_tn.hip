// Weight gradient of a stride-1 convolution (1x1, or 3x3 with padding 1) on the bf16 matrix cores WITHOUT rewritten operands
// (round 6; the mixed-precision training step, pointtinybenchmark_amd/training.py; reference analogue: torch autograd under mmcv's
// Fp16OptimizerHook, T/mmdet/apis/train.py:116-119).
//
//   dW[co][kh][kw][ci] = sum over output pixels q = (n, oy, ox) of dy[q][co] * x[n, s oy + kh - p, s ox + kw - p][ci]      (stride s = 1, 2)
//
// The reduction runs over PIXELS, the slow dimension of both NHWC maps, and v_mfma_f32_32x32x16_bf16 wants 8 consecutive k per
// lane.  conv_wgrad_bf16.hip therefore rewrites both maps channel-major (dyT, three shifted xT copies) and runs an NT GEMM -- the
// rewrites are 4.2 of the 48.5 kernel-ms of a configs[4] step and 7.9 of R50's 98.6 (profiles/round6_train_*_one_stream_kernel_stats.csv).
// This kernel reads the maps AS THEY ARE: a K chunk is 64 pixels x 256 channels of each map, fetched by LDS-DMA exactly as the
// forward's pixel rows are (512-byte rows, 16 bytes per lane, zero rows from the buffer range check = the conv's zero padding and
// the overhang of the last chunk), and the MFMA fragments come out of LDS through gfx950's transposing read: ds_read_b64_tr_b16
// hands lane i of a 16-lane group column i of a [4 pixels][16 channels] block (tools/diag/tr_read_probe.hip pins the mapping), so
// two of them are a lane's 8 consecutive k (pixels) of its channel.  Bank conflicts: the four pixel rows of a block are 512 bytes
// apart (the same banks), so the DMA stores channel group s of pixel row r at slot s ^ 4 (r & 3) (source-side swizzle, the LDS image
// of a request is lane-linear) and a wave's tr read touches 32 distinct 8-byte units of a 256-byte bank frame.
//
// Tile 256 (cout) x 256 (cin) x one tap, eight waves (2 x 4, a wave = 4 x 2 blocks of 32 x 32), two 64 KB stages, the request
// pieces of a chunk spread over the MFMA slots of three k-steps as in conv_bf16_dma_kernel (its schedule, counted waits and all);
// split over the pixel axis into `splits` workgroup rows whose fp32 partials part[split][tap][Cout][Cin] wgrad_bf16_reduce_kernel sums.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 tn_bf16x8;
typedef short tn_s16x4 __attribute__((ext_vector_type(4)));
typedef short tn_s16x8 __attribute__((ext_vector_type(8)));
typedef int tn_i32x4 __attribute__((ext_vector_type(4)));

struct WgradTnParams {
    const unsigned short* dy;   // (P, Cout) bf16: the gradient map, NHWC rows
    const unsigned short* x;    // (P, Cin) bf16: the recorded input map
    float* part;                // [splits][taps][Cout][Cin]
    int H, W, OH, OW, Cin, Cout, k, pad, stride;
    int P;                      // N * OH * OW output pixels (the K axis)
    int XP;                     // N * H * W input pixels
    int tilesM, tilesN, taps, chunks;
};

constexpr int TN_STAGE = 65536, TN_PIECE = 8192, TN_B = 32768;

__global__ __launch_bounds__(512, 2) void wgrad_bf16_tn_kernel(WgradTnParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TN_STAGE];
    // splits are dealt to the XCDs (block b runs on XCD b % 8; the split count is a multiple of 8): the taps and tiles of one split
    // are neighbours on one XCD and read the same pixel range of both maps within one L2
    const int TMN = p.tilesM * p.tilesN;
    const int idx = blockIdx.x >> 3, per_split = p.taps * TMN;
    const int split = (idx / per_split) * 8 + (blockIdx.x & 7);
    const int rem = idx % per_split;
    const int tap = rem / TMN, tile = rem - tap * TMN;
    const int tm = tile / p.tilesN, tn = tile - tm * p.tilesN;
    const int m0 = tm * 256, n0 = tn * 256;
    const int kh = tap / p.k, kw = tap - kh * p.k;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- staging role: piece z of a map = pixel rows 16 z + (tid >> 5) of the chunk, 16-byte unit tid & 31 of the 512-byte row;
    // the lane fetches channel group (tid & 31) ^ 4 (row & 3)
    const int prow = tid >> 5;
    const int cgrp = (tid & 31) ^ (4 * (prow & 3));
    const int ca = m0 + 8 * cgrp, cb = n0 + 8 * cgrp;
    const bool aok = ca < p.Cout, bok = cb < p.Cin;
    const size_t dy_addr = (size_t)p.dy, x_addr = (size_t)p.x;
    const tn_i32x4 rs_dy = {(int)(unsigned)dy_addr, (int)(unsigned)(dy_addr >> 32) & 0xffff, (int)((size_t)p.P * p.Cout * 2), 0x00020000};
    const tn_i32x4 rs_x = {(int)(unsigned)x_addr, (int)(unsigned)(x_addr >> 32) & 0xffff, (int)((size_t)p.XP * p.Cin * 2), 0x00020000};
    const int dkh = kh - p.pad, dkw = kw - p.pad;
    const bool linear = p.k == 1 && p.stride == 1;           // the input row of output pixel q is q itself
    const int ihw = p.H * p.W;
    // pixel state of this lane's row in each of the four pieces: output pixel index and -- unless linear -- its (oy, ox) and the
    // first input pixel of its image
    int pq[4], py[4], px[4], pimg[4], voffA[4], voffB[4];
    const int a64 = 64 / p.OW, b64 = 64 - a64 * p.OW;        // a chunk advances a pixel by a64 rows and b64 columns
    auto offsets = [&](int z) {
        const bool in = pq[z] < p.P;
        voffA[z] = (in && aok) ? (int)(((unsigned)pq[z] * (unsigned)p.Cout + (unsigned)ca) * 2u) : (int)0x80000000;
        if (linear) {
            voffB[z] = (in && bok) ? (int)(((unsigned)pq[z] * (unsigned)p.Cin + (unsigned)cb) * 2u) : (int)0x80000000;
        } else {
            const int iy = py[z] * p.stride + dkh, ix = px[z] * p.stride + dkw;
            const bool ok = in && bok && ((unsigned)iy < (unsigned)p.H) && ((unsigned)ix < (unsigned)p.W);
            voffB[z] = ok ? (int)(((unsigned)(pimg[z] + iy * p.W + ix) * (unsigned)p.Cin + (unsigned)cb) * 2u) : (int)0x80000000;
        }
    };
#pragma unroll
    for (int z = 0; z < 4; ++z) {
        pq[z] = split * p.chunks * 64 + 16 * z + prow;
        const int r = pq[z] / p.OW;
        px[z] = pq[z] - r * p.OW;
        const int n = r / p.OH;
        py[z] = r - n * p.OH;
        pimg[z] = n * ihw;
        offsets(z);
    }
    auto advance = [&]() {           // the rows of the next chunk: 64 output pixels on
#pragma unroll
        for (int z = 0; z < 4; ++z) {
            pq[z] += 64;
            if (!linear) {
                px[z] += b64;
                py[z] += a64;
                if (px[z] >= p.OW) { px[z] -= p.OW; py[z] += 1; }
                while (py[z] >= p.OH) { py[z] -= p.OH; pimg[z] += ihw; }
            }
            offsets(z);
        }
    };
    const int lds0 = (int)(unsigned)(size_t)smem;
    // one LDS-DMA request: 1 KB of this wave (lane * 16 bytes from M0); inline asm: counted by hand (the waits below)
    auto dma = [&](const tn_i32x4& rs, int voff, int lds_byte) {
        asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen offset:0 lds"
                     :: "s"(lds_byte), "v"(voff), "s"(rs) : "memory");
    };
    const int KT = p.chunks;
    // piece z = 0..7 of the chunk whose row state is current (z < 4: gradient map, else input map) into stage `buf`
    auto stage_piece = [&](int buf, int z) {
        const int base = lds0 + buf * TN_STAGE + wave * 1024;
        if (z < 4) dma(rs_dy, voffA[z < 4 ? z : 0], base + z * TN_PIECE);
        else dma(rs_x, voffB[z >= 4 ? z - 4 : 0], base + TN_B + (z - 4) * TN_PIECE);
    };

    // ---- MFMA role
    const int wm = wave & 1, wn = wave >> 1;
    const int i16 = lane & 15, g1 = (lane >> 4) & 1, half = lane >> 5, r3 = (i16 >> 2) & 3;
    const int rowpart = (half * 8 + (i16 >> 2)) * 512 + (i16 & 1) * 8;
    int addrA[4], addrB[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) addrA[i] = rowpart + (((wm * 16 + i * 4 + g1 * 2 + ((i16 >> 1) & 1)) ^ (4 * r3)) * 16);
#pragma unroll
    for (int j = 0; j < 2; ++j) addrB[j] = TN_B + rowpart + (((wn * 8 + j * 4 + g1 * 2 + ((i16 >> 1) & 1)) ^ (4 * r3)) * 16);
    typedef __attribute__((address_space(3))) tn_s16x4* lds_s16x4;
    auto frag = [&](int buf, int kk, int addr) {        // a lane's 8 consecutive k (pixels 16 kk + 8 half ..) of its channel
        const unsigned char* q = smem + buf * TN_STAGE + addr + kk * (16 * 512);
        const tn_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(q));
        const tn_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(q + 4 * 512));
        return __builtin_bit_cast(tn_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    tn_bf16x8 fa0[4], fb0[2], fa1[4], fb1[2];
    // fragment piece z of a k-step: z < 2 = the cin blocks, then the four cout blocks
#define TFRAG(FA, FB, buf, kk, z)                                                       \
    do {                                                                                \
        if ((z) >= 2) FA[(z) >= 2 ? (z) - 2 : 0] = frag(buf, kk, addrA[(z) >= 2 ? (z) - 2 : 0]); \
        else FB[(z) < 2 ? (z) : 0] = frag(buf, kk, addrB[(z) < 2 ? (z) : 0]);           \
    } while (0)
#define TMFMA(FA, FB, q) acc[(q) / 2][(q) % 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[(q) / 2], FB[(q) % 2], acc[(q) / 2][(q) % 2], 0, 0, 0)

    // prologue: chunk 0 -> stage 0 completely, the first three pieces of chunk 1 -> stage 1
#pragma unroll
    for (int z = 0; z < 8; ++z) stage_piece(0, z);
    advance();
    if (KT > 1) {
#pragma unroll
        for (int z = 0; z < 3; ++z) stage_piece(1, z);
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
#pragma unroll
    for (int z = 0; z < 6; ++z) TFRAG(fa0, fb0, 0, 0, z);

    // one chunk kt (stage kt & 1): k-step 0: MFMAs | fragments of k-step 1 | pieces 3..5 of chunk kt+1; k-step 1: MFMAs | fragments
    // of k-step 2 | pieces 6, 7; row state -> chunk kt+2; k-step 2: MFMAs | fragments of k-step 3; wait + barrier; k-step 3: MFMAs |
    // first fragments of chunk kt+1 | pieces 0..2 of chunk kt+2
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < KT, more2 = kt + 2 < KT;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            TMFMA(fa0, fb0, q);
            if (q < 6) TFRAG(fa1, fb1, buf, 1, q < 6 ? q : 0);
            if (more && q >= 5) stage_piece(buf ^ 1, 3 + q - 5);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            TMFMA(fa1, fb1, q);
            if (q < 6) TFRAG(fa0, fb0, buf, 2, q < 6 ? q : 0);
            if (more && q >= 6) stage_piece(buf ^ 1, 6 + q - 6);
            __builtin_amdgcn_sched_barrier(0);
        }
        advance();
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            TMFMA(fa0, fb0, q);
            if (q < 6) TFRAG(fa1, fb1, buf, 3, q < 6 ? q : 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): every fragment read of stage buf has returned
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            TMFMA(fa1, fb1, q);
            if (more && q < 6) TFRAG(fa0, fb0, buf ^ 1, 0, q < 6 ? q : 0);
            if (more2 && q >= 5) stage_piece(buf, q - 5);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef TFRAG
#undef TMFMA

    // this (split, tap)'s fp32 partial: D layout of a 32 x 32 block: col = lane & 31 (cin), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    float* out = p.part + ((size_t)(split * p.taps + tap) * p.Cout) * p.Cin;
    const int l31 = lane & 31;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = n0 + wn * 64 + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 128 + i * 32 + 4 * half + (r & 3) + 8 * (r >> 2);
                if (m < p.Cout && c < p.Cin) out[(size_t)m * p.Cin + c] = acc[i][j][r];
            }
        }
}

// dy (N,OH,OW,Cout) bf16, x (N,H,W,Cin) bf16, part [splits][k*k][Cout][Cin] fp32 (splits % 8 == 0, `chunks` 64-pixel chunks per
// split of the OUTPUT pixel axis).  k in {1, 3}, padding k / 2, stride 1 or 2 (OH = (H + 2 p - k) / s + 1).  Both maps below 2 GiB;
// Cout % 8 == 0, Cin % 8 == 0 (16-byte channel groups).
int wgrad_bf16_tn_launch(const void* dy, const void* x, float* part, int N, int H, int W, int Cin, int Cout, int k, int stride, int splits,
                         int chunks, hipStream_t stream) {
    if (!(k == 1 || k == 3) || !(stride == 1 || stride == 2) || Cout % 8 != 0 || Cin % 8 != 0 || splits <= 0 || splits % 8 != 0 || chunks <= 0)
        return CPR_ERR_UNSUPPORTED;
    const int pad = k / 2, OH = (H + 2 * pad - k) / stride + 1, OW = (W + 2 * pad - k) / stride + 1;
    if (OH <= 0 || OW <= 0) return CPR_ERR_UNSUPPORTED;
    const long long P = (long long)N * OH * OW, XP = (long long)N * H * W;
    if (P * Cout * 2 >= (1ll << 31) || XP * Cin * 2 >= (1ll << 31) || P + 64ll * splits * chunks >= (1ll << 30)) return CPR_ERR_UNSUPPORTED;
    WgradTnParams p;
    p.dy = (const unsigned short*)dy; p.x = (const unsigned short*)x; p.part = part;
    p.H = H; p.W = W; p.OH = OH; p.OW = OW; p.Cin = Cin; p.Cout = Cout; p.k = k; p.pad = pad; p.stride = stride; p.P = (int)P; p.XP = (int)XP;
    p.tilesM = (Cout + 255) / 256; p.tilesN = (Cin + 255) / 256; p.taps = k * k; p.chunks = chunks;
    const long long blocks = (long long)splits * p.taps * p.tilesM * p.tilesN;
    if (blocks >= (1ll << 31)) return CPR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(wgrad_bf16_tn_kernel, dim3((unsigned)blocks), dim3(512), 0, stream, p);
    CPR_LAUNCH_STATUS();
}
