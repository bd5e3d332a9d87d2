// ResNet stem for the bf16 compute mode (round 4): conv 7x7 / stride 2 / pad 3, 3 -> 64 channels, + folded BatchNorm + ReLU
// (T/mmdet/models/backbones/resnet.py:630-636) on the bf16 matrix cores, fp32 NHWC4 image in, bf16 NHWC map out.
//
// Until round 3 the bf16 mode ran its stem on the fp32 implicit-GEMM kernel (conv_mfma_kernel<128, 64, 1>: 0.8 ms of the 10.9 ms
// step at R101 1024^2 B = 8 -- its K = 7 x 7 x 4 rows are gathered 16 bytes at a time and the fp32 matrix pipe needs 0.25 ms for
// the multiplies alone).  At bf16 rates the layer is pure data movement: 134 MB of image in, 268 MB of map out.  So:
//   * workgroup = 16 x 32 output pixels x 64 couts; its (2 x 16 + 5) x (2 x 32 + 5) input patch is read ONCE, coalesced, rounded to
//     bf16 (RNE) and laid down in LDS as 8-byte pixels (4 channels, the 4th zero), rows padded to 70 pixels = 35 x 16 bytes;
//   * K order (kh, kw padded to 8, channel): one 16-byte fragment = two horizontally adjacent pixels = 8 consecutive k, so the A
//     operand of v_mfma_f32_32x32x16_bf16 is a single ds_read_b128 at ((2 oy + kh) 70 + 2 ox + 2 kwpair) 8 -- output pixels along x
//     are 16 bytes apart: conflict free.  The weights ([64][7][8][4] bf16, kw = 7 and channel 3 zero: 224 k = 14 MFMA k-steps) sit
//     in LDS too (rows padded to 464 bytes);
//   * 8 waves: wave w owns output rows 2 w, 2 w + 1 (two 32-pixel blocks) x both 32-cout blocks = 64 accumulator registers;
//     50 KB of LDS, 93 registers -> two workgroups per CU (the second __launch_bounds__ argument is waves per SIMD), so one's
//     patch load and store phase run under the other's MFMAs;
//   * epilogue as conv_mfma_bf16.hip: y = ReLU(acc * scale + shift), cout pairs packed into one dword.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

struct StemParams {
    const float* in;             // (N, H, W, 4) fp32
    const unsigned short* wgt;   // (64, 224) bf16: k = (kh * 8 + kw) * 4 + c
    const float* scale;          // (64) folded BatchNorm, or null
    const float* bias;
    unsigned short* out;         // (N, OH, OW, 64) bf16
    int N, H, W, OH, OW, tilesY, tilesX, relu, layout;     // layout 0: NHWC4 input, 1: (N, 3, H, W) planes
};

constexpr int ST_TY = 16, ST_TX = 32;
constexpr int ST_PH = 2 * ST_TY + 5, ST_PW = 2 * ST_TX + 6;       // 37 x 70 pixels (69 real columns + the kw = 7 slot)
constexpr int ST_WROW = 464;                                      // bytes per cout row of the weight image in LDS (448 + 16)
constexpr int ST_PATCH_BYTES = ST_PH * ST_PW * 8;                 // 20 720
constexpr int ST_LDS = ST_PATCH_BYTES + 64 * ST_WROW;             // 50 416

__global__ __launch_bounds__(512, 4) void stem_bf16_kernel(StemParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[ST_LDS];
    unsigned char* patch = smem;
    unsigned char* wl = smem + ST_PATCH_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b = blockIdx.x;
    const int tx = b % p.tilesX;
    b /= p.tilesX;
    const int ty = b % p.tilesY;
    const int n = b / p.tilesY;
    const int oy0 = ty * ST_TY, ox0 = tx * ST_TX;
    const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;

    // weights: 64 rows x 448 bytes = 28 x 16 bytes
    for (int u = tid; u < 64 * 28; u += 512) {
        const int r = u / 28, q = u - r * 28;
        *reinterpret_cast<f32x4*>(wl + r * ST_WROW + q * 16) = *reinterpret_cast<const f32x4*>(
            reinterpret_cast<const unsigned char*>(p.wgt) + (size_t)r * 448 + q * 16);
    }
    // patch: fp32 NHWC4 -> bf16, zero outside the image and in the padded column
    const size_t plane = (size_t)p.H * p.W;
    const float* img = p.in + (size_t)n * plane * (p.layout ? 3 : 4);
    for (int u = tid; u < ST_PH * ST_PW; u += 512) {
        const int py = u / ST_PW, px = u - py * ST_PW;
        const int iy = iy0 + py, ix = ix0 + px;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (px < ST_PW - 1 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
        {
            const size_t o = (size_t)iy * p.W + ix;
            if (p.layout) { v[0] = img[o]; v[1] = img[plane + o]; v[2] = img[2 * plane + o]; }
            else v = *reinterpret_cast<const f32x4*>(img + o * 4);
        }
        const bf16x4 h = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)0.f};
        *reinterpret_cast<bf16x4*>(patch + (size_t)u * 8) = h;
    }
    __syncthreads();

    const int l31 = lane & 31, half = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // A: pixel (row 2 wave + i, column l31) of the tile; B: cout 32 j + l31
    const unsigned char* a0 = patch + ((2 * (2 * wave)) * ST_PW + 2 * l31) * 8;
    const unsigned char* b0 = wl + l31 * ST_WROW;
#pragma unroll
    for (int s = 0; s < 14; ++s) {
        const int t = 2 * s + half;                 // tap pair: kh = t >> 2, kw = 2 (t & 3), 2 (t & 3) + 1
        const int aoff = ((t >> 2) * ST_PW + 2 * (t & 3)) * 8;
        f32x4 fa[2], fb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const f32x4*>(a0 + i * (2 * ST_PW * 8) + aoff);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const f32x4*>(b0 + j * 32 * ST_WROW + t * 16);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i]),
                                                                    __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
    }

    // D layout of a 32 x 32 block: col = lane & 31 (cout), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (pixel = ox)
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
        p.out, 0, (int)((size_t)p.N * p.OH * p.OW * 64 * 2), 0x00020000);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = j * 32 + l31;
        const float sc = p.scale ? p.scale[c] : 1.f;
        const float bi = p.bias ? p.bias[c] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int oy = oy0 + 2 * wave + i;
            const unsigned rowbase = (unsigned)((n * p.OH + oy) * p.OW) * 64u + (unsigned)(c & ~1);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ox = ox0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                float x = acc[i][j][r] * sc + bi;
                if (p.relu) x = fmaxf(x, 0.f);
                const float nb = __shfl_xor(x, 1, 64);
                const bool mine = ((r & 1) == (lane & 1)) && oy < p.OH && ox < p.OW;
                const bf16x2 pk = (lane & 1) ? bf16x2{(__bf16)nb, (__bf16)x} : bf16x2{(__bf16)x, (__bf16)nb};
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pk), rs_out,
                                                      (int)(mine ? (rowbase + (unsigned)ox * 64u) * 2u : 0x80000000u), 0, 0);
            }
        }
    }
}

// ---- stem conv + BN + ReLU + max-pool 3x3 / stride 2 / pad 1 in one kernel (resnet.py:630-637) ------------------------------------
// The unfused pair writes the (N, H/2, W/2, 64) conv map (268 MB at 1024^2 B = 8) and reads it back to pool it.  Here a workgroup
// owns 8 x 16 POOLED pixels: it computes the 17 x 33 conv outputs under them (rows 16 ty - 1 .., columns 32 tx - 1 ..: 17 row blocks
// of 32 columns + one block for the 17 pixels of the left halo column = 18 MFMA blocks, 12.5 % recomputed), rounds them to bf16 into
// LDS (over the patch and weight images, which are dead by then) and pools from there.  Bit-identical to conv -> bf16 -> pool:
// rounding is monotonic, so max(bf16(a), bf16(b)) = bf16(max(a, b)); conv positions outside the map hold 0, which equals the
// pool's -inf padding because every window contains its valid centre and all values are >= 0 after the ReLU (required here).
constexpr int SP_PH = 39, SP_PW = 72;                            // input patch: 2 x 16 + 7 rows, 2 x 32 + 7 columns + the kw = 7 slot
constexpr int SP_PATCH_BYTES = SP_PH * SP_PW * 8;                // 22 464
constexpr int SP_CONV_BYTES = 17 * 33 * 128;                     // conv tile, bf16 NHWC: 71 808
constexpr int SP_LDS = SP_CONV_BYTES > SP_PATCH_BYTES + 64 * ST_WROW ? SP_CONV_BYTES : SP_PATCH_BYTES + 64 * ST_WROW;

struct StemPoolParams {
    const float* in;
    const unsigned short* wgt;
    const float* scale;
    const float* bias;
    unsigned short* out;         // (N, PH, PW, 64) bf16
    int N, H, W, OH, OW, PH, PW, tilesY, tilesX, layout;
};

__global__ __launch_bounds__(512, 4) void stem_pool_bf16_kernel(StemPoolParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[SP_LDS];
    unsigned char* patch = smem;
    unsigned char* wl = smem + SP_PATCH_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b = blockIdx.x;
    const int tx = b % p.tilesX;
    b /= p.tilesX;
    const int ty = b % p.tilesY;
    const int n = b / p.tilesY;
    const int r0 = 16 * ty - 1, c0 = 32 * tx - 1;               // first conv row / column of the tile (may be -1)
    const int iy0 = 2 * r0 - 3, ix0 = 2 * c0 - 3;

    for (int u = tid; u < 64 * 28; u += 512) {
        const int r = u / 28, q = u - r * 28;
        *reinterpret_cast<f32x4*>(wl + r * ST_WROW + q * 16) = *reinterpret_cast<const f32x4*>(
            reinterpret_cast<const unsigned char*>(p.wgt) + (size_t)r * 448 + q * 16);
    }
    const size_t plane = (size_t)p.H * p.W;
    const float* img = p.in + (size_t)n * plane * (p.layout ? 3 : 4);
    for (int u = tid; u < SP_PH * SP_PW; u += 512) {
        const int py = u / SP_PW, px = u - py * SP_PW;
        const int iy = iy0 + py, ix = ix0 + px;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (px < SP_PW - 1 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
        {
            const size_t o = (size_t)iy * p.W + ix;
            if (p.layout) { v[0] = img[o]; v[1] = img[plane + o]; v[2] = img[2 * plane + o]; }
            else v = *reinterpret_cast<const f32x4*>(img + o * 4);
        }
        const bf16x4 h = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)0.f};
        *reinterpret_cast<bf16x4*>(patch + (size_t)u * 8) = h;
    }
    __syncthreads();

    // blocks: q < 17 = conv row r0 + q, columns c0 + 1 + m; q = 17 = column c0, rows r0 + min(m, 16).  Wave w: q = w, w + 8 (, w + 16)
    const int l31 = lane & 31, half = lane >> 5;
    const int nblk = wave < 2 ? 3 : 2;
    f32x16 acc[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int abase[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int q = wave + 8 * i;
        const int m16 = l31 < 16 ? l31 : 16;
        abase[i] = q < 17 ? ((2 * q) * SP_PW + 2 * l31 + 2) * 8 : ((2 * m16) * SP_PW) * 8;
    }
    const unsigned char* b0 = wl + l31 * ST_WROW;
#pragma unroll
    for (int s = 0; s < 14; ++s) {
        const int t = 2 * s + half;
        const int aoff = ((t >> 2) * SP_PW + 2 * (t & 3)) * 8;
        f32x4 fa[3], fb[2];
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i < 2 || nblk == 3) fa[i] = *reinterpret_cast<const f32x4*>(patch + abase[i] + aoff);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const f32x4*>(b0 + j * 32 * ST_WROW + t * 16);
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i < 2 || nblk == 3) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i]),
                                                                        __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
            }
    }
    __syncthreads();                                             // patch and weights are dead: the conv tile takes their place

    // conv tile in LDS: [row 0..16][column 0..32][64 couts] bf16; positions outside the conv map hold 0
    unsigned short* ct = reinterpret_cast<unsigned short*>(smem);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = j * 32 + l31;
        const float sc = p.scale ? p.scale[c] : 1.f;
        const float bi = p.bias ? p.bias[c] : 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (i == 2 && nblk < 3) continue;
            const int q = wave + 8 * i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
                const int row = q < 17 ? q : m, col = q < 17 ? m + 1 : 0;
                const bool in_tile = q < 17 || m < 17;
                const int cr = r0 + row, cc = c0 + col;
                const bool ok = (unsigned)cr < (unsigned)p.OH && (unsigned)cc < (unsigned)p.OW;
                const float x = ok ? fmaxf(acc[i][j][r] * sc + bi, 0.f) : 0.f;
                if (in_tile) ct[(row * 33 + col) * 64 + c] = __builtin_bit_cast(unsigned short, (__bf16)x);
            }
        }
    }
    __syncthreads();

    // pooled pixel (py, px) of the tile, cout group g (8 couts = 16 bytes): rows 2 py .. 2 py + 2, columns 2 px .. 2 px + 2 of the tile
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    for (int u = tid; u < 8 * 16 * 8; u += 512) {
        const int g = u & 7, px = (u >> 3) & 15, py = u >> 7;
        const int gy = 8 * ty + py, gx = 16 * tx + px;
        if (gy >= p.PH || gx >= p.PW) continue;
        float mx[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) mx[e] = 0.f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(ct + ((2 * py + dy) * 33 + 2 * px + dx) * 64 + g * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    mx[2 * e] = fmaxf(mx[2 * e], __uint_as_float(v[e] << 16));
                    mx[2 * e + 1] = fmaxf(mx[2 * e + 1], __uint_as_float(v[e] & 0xffff0000u));
                }
            }
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (__float_as_uint(mx[2 * e]) >> 16) | (__float_as_uint(mx[2 * e + 1]) & 0xffff0000u);
        *reinterpret_cast<u32x4*>(p.out + (((size_t)n * p.PH + gy) * p.PW + gx) * 64 + g * 8) = o;
    }
}

// in (N,H,W,4) fp32 -> out (N,PH,PW,64) bf16 = maxpool3x3/2/pad 1 (ReLU(conv7x7/2/pad 3 (in) * scale + bias)); OH = (H-1)/2+1,
// PH = (OH-1)/2+1.
extern "C" int cpr_stem7x7s2_pool_bf16(const float* in, const void* wgt, const float* scale, const float* bias, void* out, int N,
                                       int H, int W, int layout, hipStream_t stream) {
    CPR_CHECK_ARG(in && wgt && out && N > 0 && H > 0 && W > 0 && (layout == 0 || layout == 1));
    StemPoolParams p;
    p.in = in; p.wgt = (const unsigned short*)wgt; p.scale = scale; p.bias = bias; p.out = (unsigned short*)out;
    p.N = N; p.H = H; p.W = W; p.layout = layout;
    p.OH = (H - 1) / 2 + 1;
    p.OW = (W - 1) / 2 + 1;
    p.PH = (p.OH - 1) / 2 + 1;
    p.PW = (p.OW - 1) / 2 + 1;
    p.tilesY = (p.PH + 7) / 8;
    p.tilesX = (p.PW + 15) / 16;
    const long long blocks = (long long)N * p.tilesY * p.tilesX;
    if (blocks >= (1ll << 31)) return CPR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(stem_pool_bf16_kernel, dim3((unsigned)blocks), dim3(512), 0, stream, p);
    CPR_LAUNCH_STATUS();
}

// in (N,H,W,4) fp32 -> out (N,OH,OW,64) bf16, OH = (H - 1) / 2 + 1; wgt = the (64, 224) bf16 image described above.
extern "C" int cpr_stem7x7s2_bf16(const float* in, const void* wgt, const float* scale, const float* bias, void* out, int N,
                                  int H, int W, int relu, int layout, hipStream_t stream) {
    CPR_CHECK_ARG(in && wgt && out && N > 0 && H > 0 && W > 0 && (layout == 0 || layout == 1));
    StemParams p;
    p.in = in; p.wgt = (const unsigned short*)wgt; p.scale = scale; p.bias = bias; p.out = (unsigned short*)out;
    p.N = N; p.H = H; p.W = W; p.relu = relu; p.layout = layout;
    p.OH = (H - 1) / 2 + 1;
    p.OW = (W - 1) / 2 + 1;
    if ((long long)N * p.OH * p.OW * 64 * 2 >= (1ll << 31)) return CPR_ERR_UNSUPPORTED;
    p.tilesY = (p.OH + ST_TY - 1) / ST_TY;
    p.tilesX = (p.OW + ST_TX - 1) / ST_TX;
    const long long blocks = (long long)N * p.tilesY * p.tilesX;
    if (blocks >= (1ll << 31)) return CPR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(stem_bf16_kernel, dim3((unsigned)blocks), dim3(512), 0, stream, p);
    CPR_LAUNCH_STATUS();
}
