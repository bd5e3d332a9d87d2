// ResNet stem in one kernel, exact fp32 on the matrix cores (round 4): conv 7x7 / stride 2 / pad 3, 3 -> 64 channels + folded
// BatchNorm + ReLU + max-pool 3x3 / stride 2 / pad 1 (T/mmdet/models/backbones/resnet.py:630-637).  fp32 NHWC4 image in, fp32
// NHWC (N, H/4, W/4, 64) map out.
//
// The implicit-GEMM kernel ran this layer in its stem mode (conv_mfma_kernel<128, 64, 1>: K rows of 7 x 8 x 4 = 224 floats for
// 147 real ones, every 16-byte pixel gathered by its own load: 0.43 of the matrix pipe busy, a third of that on padding) and wrote
// the (N, H/2, W/2, 64) conv map -- 1.68 GB at 640^2 B = 64 -- for maxpool3x3s2_kernel to read back.  Here
//   * a workgroup owns 8 x 16 POOLED pixels x 64 couts: it reads the 39 x 71 input patch under them ONCE (coalesced 16-byte
//     pixels) and keeps it in LDS as 3-channel pixels (12 bytes), so the 21 floats (7 taps x 3 channels) of a kernel row are
//     CONTIGUOUS for every output pixel; K = 7 rows x 22 (21 + one zero-weight slot, so that an MFMA's k pair never straddles a
//     row) = 154 = 77 steps of v_mfma_f32_32x32x2_f32 for 147 real products: 95 % of the multiplies are real;
//   * the weights ([64][7][22] fp32, 39 KB) sit in LDS next to the patch (34 KB): 73 KB and <= 128 registers = TWO workgroups per
//     CU, one's patch load / pooling / stores under the other's MFMAs;
//   * 17 x 33 conv outputs (the 16 x 32 under the pooled tile + one halo row / column, 12.5 % recomputed) = 17 row blocks of 32
//     pixels + 1 block for the left halo column; wave w owns cout block w & 1 and pixel blocks (w >> 1) + 4 i;
//   * epilogue per cout half: BN + ReLU -> LDS (over the patch and the weights, dead by then) -> 3x3 / 2 max over the tile ->
//     16-byte stores.  Out-of-map conv positions hold 0 = the pool's -inf padding behind a ReLU (every window has its valid centre).
// Same products as the implicit GEMM, summed kernel row by kernel row in fp32 (no reduced precision anywhere).
#include "common.h"
#include <type_traits>

constexpr int SF_PH = 39, SF_PW = 72;                             // patch rows / columns (71 real + one zero column)
constexpr int SF_PROW = SF_PW * 3;                                // floats per patch row: 216
constexpr int SF_PATCH_FLOATS = SF_PH * SF_PROW;                  // 8424  (33 696 bytes)
constexpr int SF_WROW = 154;                                      // floats per cout: 7 x 22
constexpr int SF_W_FLOATS = 64 * SF_WROW;                         // 9856  (39 424 bytes)
constexpr int SF_CT_FLOATS = 17 * 33 * 32;                        // conv tile of one cout half: 17 952 floats (71 808 bytes)
constexpr int SF_LDS_FLOATS = SF_PATCH_FLOATS + SF_W_FLOATS > SF_CT_FLOATS ? SF_PATCH_FLOATS + SF_W_FLOATS : SF_CT_FLOATS;

struct StemF32Params {
    const float* in;       // layout 0: (N, H, W, 4) fp32, 4th channel ignored; layout 1: (N, 3, H, W) fp32 planes (NCHW)
    const float* wgt;      // (64, 154): [cout][kh][kw * 3 + c], slot 21 of every kernel row zero
    const float* scale;    // folded BatchNorm (64) or null
    const float* bias;
    float* out;            // (N, PH, PW, 64)
    int N, H, W, OH, OW, PH, PW, tilesY, tilesX, layout;
};

__global__ __launch_bounds__(512, 4) void stem_pool_f32_kernel(StemF32Params p) {
    __shared__ __attribute__((aligned(16))) float smem[SF_LDS_FLOATS];
    float* patch = smem;
    float* wl = smem + SF_PATCH_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b = blockIdx.x;
    const int tx = b % p.tilesX;
    b /= p.tilesX;
    const int ty = b % p.tilesY;
    const int n = b / p.tilesY;
    const int r0 = 16 * ty - 1, c0 = 32 * tx - 1;                 // first conv row / column of the tile (may be -1)
    const int iy0 = 2 * r0 - 3, ix0 = 2 * c0 - 3;

    for (int u = tid; u < SF_W_FLOATS / 4; u += 512)              // 154 floats per row: rows are 8-byte, the image 16-byte aligned
        *reinterpret_cast<f32x4*>(wl + u * 4) = *reinterpret_cast<const f32x4*>(p.wgt + u * 4);
    // (layout 1: the network input as torch hands it over, three planes -- consecutive threads read consecutive floats of a
    // plane row; saves the nchw_to_nhwc4 pass and its 0.4 GB at 640^2 B = 64)
    const size_t plane = (size_t)p.H * p.W;
    const float* img = p.in + (size_t)n * plane * (p.layout ? 3 : 4);
    for (int u = tid; u < SF_PH * SF_PW; u += 512) {
        const int py = u / SF_PW, px = u - py * SF_PW;
        const int iy = iy0 + py, ix = ix0 + px;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (px < SF_PW - 1 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
            const size_t o = (size_t)iy * p.W + ix;
            if (p.layout) { v[0] = img[o]; v[1] = img[plane + o]; v[2] = img[2 * plane + o]; }
            else v = *reinterpret_cast<const f32x4*>(img + o * 4);
        }
        float* d = patch + u * 3;
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2];
    }
    __syncthreads();

    // blocks: q < 17 = conv row r0 + q, columns c0 + 1 + m; q = 17 = column c0, rows r0 + min(m, 16)
    const int l31 = lane & 31, half = lane >> 5;
    const int jb = wave & 1, qb = wave >> 1;
    f32x16 acc[5];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    int abase[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int q = qb + 4 * i;
        const int m16 = l31 < 16 ? l31 : 16;
        abase[i] = (q < 17 ? (2 * q) * SF_PROW + (2 * l31 + 2) * 3 : (2 * m16) * SF_PROW) + half;
    }
    const bool five = qb < 2;                                     // wave-uniform: blocks 16 and 17 exist
    const float* bptr = wl + (jb * 32 + l31) * SF_WROW + half;
    // operands of step t + 1 are requested before the MFMAs of step t; the scheduling barrier keeps the compiler from hoisting a
    // whole kernel row of LDS reads (66 registers) above them.  Two copies of the loop (5 / 4 blocks): no branch inside it
    auto kloop = [&](auto nbc) {
        constexpr int NB = decltype(nbc)::value;
        float av[2][NB], bv[2];
#pragma unroll 1
        for (int kh = 0; kh < 7; ++kh) {
            const float* prow = patch + kh * SF_PROW;
            const float* brow = bptr + kh * 22;
            bv[0] = brow[0];
#pragma unroll
            for (int i = 0; i < NB; ++i) av[0][i] = prow[abase[i]];
#pragma unroll
            for (int t = 0; t < 11; ++t) {
                if (t < 10) {
                    bv[(t + 1) & 1] = brow[2 * t + 2];
#pragma unroll
                    for (int i = 0; i < NB; ++i) av[(t + 1) & 1][i] = prow[abase[i] + 2 * t + 2];
                }
#pragma unroll
                for (int i = 0; i < NB; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t & 1][i], bv[t & 1], acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    if (five) kloop(std::integral_constant<int, 5>{});
    else kloop(std::integral_constant<int, 4>{});

    // D layout of a 32 x 32 block: col = lane & 31 (cout of the wave's half), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (pixel)
    const int c = jb * 32 + l31;
    const float sc = p.scale ? p.scale[c] : 1.f;
    const float bi = p.bias ? p.bias[c] : 0.f;
    float* ct = smem;                                             // [17][33][32] of the current cout half
    // BN + ReLU in place, once (computed inside the half loop the 80 results sat next to the 80 accumulators: 130 spills)
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = fmaxf(acc[i][r] * sc + bi, 0.f);
    // (two copies, each hanging on an opaque copy of the lane's column: as one loop the 80 addresses and 80 in-map predicates were
    // hoisted out of it and spilled)
    auto write_tile = [&]() {
        int lx = l31, hx = half;
        asm volatile("" : "+v"(lx), "+v"(hx));
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            if (i == 4 && !five) continue;
            const int q = qb + 4 * i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * hx;
                const int row = q < 17 ? q : m, col = q < 17 ? m + 1 : 0;
                const int cr = r0 + row, cc = c0 + col;
                const bool ok = (unsigned)cr < (unsigned)p.OH && (unsigned)cc < (unsigned)p.OW;
                if (q < 17 || m < 17) ct[(row * 33 + col) * 32 + lx] = ok ? acc[i][r] : 0.f;
            }
        }
    };
    // pooled pixel (py, px) of the tile, cout quad g of half hb: rows 2 py .. 2 py + 2, columns 2 px .. 2 px + 2 of the tile
    auto pool_half = [&](int hb) {
        for (int u = tid; u < 8 * 16 * 8; u += 512) {
            const int g = u & 7, px = (u >> 3) & 15, py = u >> 7;
            const int gy = 8 * ty + py, gx = 16 * tx + px;
            if (gy >= p.PH || gx >= p.PW) continue;
            f32x4 mx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(ct + ((2 * py + dy) * 33 + 2 * px + dx) * 32 + g * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) mx[e] = fmaxf(mx[e], v[e]);
                }
            *reinterpret_cast<f32x4*>(p.out + (((size_t)n * p.PH + gy) * p.PW + gx) * 64 + hb * 32 + g * 4) = mx;
        }
    };
    __syncthreads();                                              // all fragment reads of the K loop are done
    if (jb == 0) write_tile();
    __syncthreads();
    pool_half(0);
    __syncthreads();
    if (jb == 1) write_tile();
    __syncthreads();
    pool_half(1);
}

// in (N,H,W,4) fp32 -> out (N,PH,PW,64) fp32 = maxpool3x3/2/pad 1 (ReLU(conv7x7/2/pad 3 (in) * scale + bias)); OH = (H-1)/2+1,
// PH = (OH-1)/2+1; wgt = the (64, 154) image described above.
extern "C" int cpr_stem7x7s2_pool_f32(const float* in, const float* wgt, const float* scale, const float* bias, float* out, int N,
                                      int H, int W, int layout, hipStream_t stream) {
    CPR_CHECK_ARG(in && wgt && out && N > 0 && H > 0 && W > 0 && (layout == 0 || layout == 1));
    StemF32Params p;
    p.in = in; p.wgt = wgt; p.scale = scale; p.bias = bias; p.out = out;
    p.N = N; p.H = H; p.W = W; p.layout = layout;
    p.OH = (H - 1) / 2 + 1;
    p.OW = (W - 1) / 2 + 1;
    p.PH = (p.OH - 1) / 2 + 1;
    p.PW = (p.OW - 1) / 2 + 1;
    p.tilesY = (p.PH + 7) / 8;
    p.tilesX = (p.PW + 15) / 16;
    const long long blocks = (long long)N * p.tilesY * p.tilesX;
    if (blocks >= (1ll << 31)) return CPR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(stem_pool_f32_kernel, dim3((unsigned)blocks), dim3(512), 0, stream, p);
    CPR_LAUNCH_STATUS();
}
