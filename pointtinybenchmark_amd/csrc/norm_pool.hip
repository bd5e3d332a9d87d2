// HBM-bound helper kernels around the conv stack (all NHWC fp32, float4 = 4 channels per lane):
//   cpr_nchw_to_nhwc4     network input (N,3,H,W) -> (N,H,W,4) zero-padded 4th channel (stem conv K-order)
//   cpr_maxpool3x3s2      ResNet stem pool            (T/mmdet/models/backbones/resnet.py:610,637)
//   cpr_gn_stats          GroupNorm partial sums      (mmcv ConvModule norm in fpn.py:124-144, cpr_head.py:990)
//   cpr_gn_finalize       partials -> per (image, channel) affine a = rstd*gamma, b = beta - mean*a
//   cpr_gn_apply          y = x*a + b [ReLU] [+ nearest-upsampled coarser level]   (fpn.py:179-188 fused)
//   cpr_nhwc_to_nchw      optional export of a feature map in the reference's dense NCHW layout
// Each is a single streaming pass: 16 B per lane, fully coalesced, no LDS except the block reductions.
#include "common.h"

// ------------------------------------------------------------------------------------------------
__global__ void nchw_to_nhwc4_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int C, int HW) {
    const long long total = (long long)N * HW;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / HW, p = i - n * HW;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const float* src = in + n * C * HW + p;
        v.x = src[0];
        if (C > 1) v.y = src[HW];
        if (C > 2) v.z = src[2ll * HW];
        if (C > 3) v.w = src[3ll * HW];
        *reinterpret_cast<f32x4*>(out + i * 4) = v;
    }
}

extern "C" int cpr_nchw_to_nhwc4(const float* in, float* out, int N, int C, int H, int W, hipStream_t stream) {
    CPR_CHECK_ARG(in && out && N > 0 && C >= 1 && C <= 4 && H > 0 && W > 0);
    const long long total = (long long)N * H * W;
    const int grid = (int)(cdivll(total, 256) < 8192 ? cdivll(total, 256) : 8192);
    hipLaunchKernelGGL(nchw_to_nhwc4_kernel, dim3(grid), dim3(256), 0, stream, in, out, N, C, H * W);
    CPR_LAUNCH_STATUS();
}

// NHWC -> NCHW through a 32x32 LDS tile (both sides coalesced)
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int HW) {
    __shared__ float t[32][33];
    const int n = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 8 rows per pass
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + tx;
        t[r][tx] = (p < HW && c < C) ? in[((size_t)n * HW + p) * C + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + tx;
        if (p < HW && c < C) out[((size_t)n * C + c) * HW + p] = t[tx][r];
    }
}

extern "C" int cpr_nhwc_to_nchw(const float* in, float* out, int N, int C, int H, int W, hipStream_t stream) {
    CPR_CHECK_ARG(in && out && N > 0 && C > 0 && H > 0 && W > 0);
    const int HW = H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(cdiv(HW, 32), cdiv(C, 32), N), dim3(256), 0, stream, in, out, C, HW);
    CPR_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------
__global__ void maxpool3x3s2_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int H, int W,
                                    int C4, int OH, int OW) {
    const long long total = (long long)N * OH * OW * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        long long r = i / C4;
        const int ox = (int)(r % OW);
        r /= OW;
        const int oy = (int)(r % OH);
        const int n = (int)(r / OH);
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = oy * 2 - 1 + dy;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = ox * 2 - 1 + dx;
                if ((unsigned)ix >= (unsigned)W) continue;
                const f32x4 v = *reinterpret_cast<const f32x4*>(in + (((size_t)n * H + iy) * W + ix) * C4 * 4 + c * 4);
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        *reinterpret_cast<f32x4*>(out + i * 4) = m;
    }
}

extern "C" int cpr_maxpool3x3s2(const float* in, float* out, int N, int H, int W, int C, hipStream_t stream) {
    CPR_CHECK_ARG(in && out && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0);
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)N * OH * OW * (C / 4);
    const int grid = (int)(cdivll(total, 256) < 16384 ? cdivll(total, 256) : 16384);
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(grid), dim3(256), 0, stream, in, out, N, H, W, C / 4, OH, OW);
    CPR_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------
// GroupNorm statistics, stage 1: per (image, slot) per-channel (sum, sumsq) over a slab of pixels.
// part layout [N][P][C][2] -- the same layout the conv epilogue emits (P = OH*OW/128 there).
// Block = 256 threads; Q = C/4 lanes cover one pixel's channels, 256/Q pixels per pass.
__global__ void gn_stats_kernel(const float* __restrict__ x, float* __restrict__ part, int HW, int C, int P) {
    __shared__ float red[256 * 8];
    const int n = blockIdx.y, slot = blockIdx.x;
    const int Q = C >> 2;
    const int q = threadIdx.x % Q, pl = threadIdx.x / Q, PP = 256 / Q;
    const int per = (HW + P - 1) / P;
    const int p0 = slot * per, p1 = min(HW, p0 + per);
    float s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    for (int p = p0 + pl; p < p1; p += PP) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((size_t)n * HW + p) * C + q * 4);
        s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
        ss[0] += v.x * v.x; ss[1] += v.y * v.y; ss[2] += v.z * v.z; ss[3] += v.w * v.w;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        red[threadIdx.x * 8 + k] = s[k];
        red[threadIdx.x * 8 + 4 + k] = ss[k];
    }
    __syncthreads();
    if (threadIdx.x < Q) {
        float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int r = 0; r < PP; ++r)
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] += red[(r * Q + q) * 8 + k];
        float* dst = part + (((size_t)n * P + slot) * C + q * 4) * 2;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            dst[k * 2] = a[k];
            dst[k * 2 + 1] = a[4 + k];
        }
    }
}

extern "C" int cpr_gn_stats(const float* x, float* part, int N, int HW, int C, int P, hipStream_t stream) {
    CPR_CHECK_ARG(x && part && N > 0 && HW > 0 && P > 0);
    CPR_CHECK_ARG(C % 4 == 0 && C / 4 <= 256 && 256 % (C / 4) == 0);
    hipLaunchKernelGGL(gn_stats_kernel, dim3(P, N), dim3(256), 0, stream, x, part, HW, C, P);
    CPR_LAUNCH_STATUS();
}

// stage 2: one block per (image, group): 256 threads reduce the group's P x cpg partials in double, then emit the
// per (image, channel) affine of torch's GroupNorm:  a = rstd*gamma, b = beta - mean*a   (y = x*a + b)
__global__ void gn_finalize_kernel(const float* __restrict__ part, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ a_out,
                                   float* __restrict__ b_out, float* __restrict__ mean_out,
                                   float* __restrict__ rstd_out, int P, int C, int G, double count, float eps) {
    __shared__ double red[2][4];
    __shared__ double s_mean, s_rstd;
    const int n = blockIdx.y, g = blockIdx.x;
    const int cpg = C / G;
    double s = 0, q = 0;
    for (int i = threadIdx.x; i < P * cpg; i += blockDim.x) {
        const int t = i / cpg, k = i - t * cpg;
        const float* src = part + (((size_t)n * P + t) * C + g * cpg + k) * 2;
        s += (double)src[0];
        q += (double)src[1];
    }
    s = wave_sum_d(s);
    q = wave_sum_d(q);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double ts = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        const double tq = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        const double mean = ts / count;
        double var = tq / count - mean * mean;
        if (var < 0) var = 0;
        s_mean = mean;
        s_rstd = 1.0 / sqrt(var + (double)eps);
        if (mean_out) mean_out[n * G + g] = (float)mean;
        if (rstd_out) rstd_out[n * G + g] = (float)s_rstd;
    }
    __syncthreads();
    if (threadIdx.x < cpg) {
        const int c = g * cpg + threadIdx.x;
        const float a = (float)s_rstd * gamma[c];
        a_out[n * C + c] = a;
        b_out[n * C + c] = beta[c] - (float)s_mean * a;
    }
}

extern "C" int cpr_gn_finalize(const float* part, const float* gamma, const float* beta, float* a_out, float* b_out,
                               float* mean_out, float* rstd_out, int N, int P, int C, int G, int HW, float eps,
                               hipStream_t stream) {
    CPR_CHECK_ARG(part && gamma && beta && a_out && b_out && N > 0 && P > 0 && C > 0 && G > 0 && C % G == 0 && HW > 0);
    CPR_CHECK_ARG(C / G <= 256);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(G, N), dim3(256), 0, stream, part, gamma, beta, a_out, b_out, mean_out,
                       rstd_out, P, C, G, (double)HW * (C / G), eps);
    CPR_LAUNCH_STATUS();
}

// apply: y[n,p,c] = x*a[n,c] + b[n,c]; optional ReLU; optional += up[n, sy(p), sx(p), c] (nearest, the
// index rule of F.interpolate(mode='nearest'): src = min(floor(dst * in/out), in-1)).  In-place safe.
__global__ void gn_apply_kernel(const float* __restrict__ x, const float* __restrict__ a, const float* __restrict__ b,
                                const float* __restrict__ up, float* __restrict__ y, int N, int H, int W, int C4,
                                int UH, int UW, int relu) {
    const long long total = (long long)N * H * W * C4;
    const float sy = (float)UH / (float)H, sx = (float)UW / (float)W;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        long long r = i / C4;
        const int px = (int)(r % W);
        r /= W;
        const int py = (int)(r % H);
        const int n = (int)(r / H);
        f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
        const f32x4 av = *reinterpret_cast<const f32x4*>(a + ((size_t)n * C4 + c) * 4);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(b + ((size_t)n * C4 + c) * 4);
        v = v * av + bv;
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (up) {
            const int uy = min((int)floorf(py * sy), UH - 1), ux = min((int)floorf(px * sx), UW - 1);
            const f32x4 u = *reinterpret_cast<const f32x4*>(up + ((((size_t)n * UH + uy) * UW + ux) * C4 + c) * 4);
            v = v + u;
        }
        *reinterpret_cast<f32x4*>(y + i * 4) = v;
    }
}

extern "C" int cpr_gn_apply(const float* x, const float* a, const float* b, const float* up, float* y, int N, int H,
                            int W, int C, int UH, int UW, int relu, hipStream_t stream) {
    CPR_CHECK_ARG(x && a && b && y && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0);
    if (up) CPR_CHECK_ARG(UH > 0 && UW > 0);
    const long long total = (long long)N * H * W * (C / 4);
    const int grid = (int)(cdivll(total, 256) < 32768 ? cdivll(total, 256) : 32768);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(grid), dim3(256), 0, stream, x, a, b, up, y, N, H, W, C / 4, UH, UW, relu);
    CPR_LAUNCH_STATUS();
}

// channel-blocked [N][C/8][H][W][8] (the Winograd conv's fast layout, conv_wino.hip) -> NHWC, optionally through the same
// affine (+ReLU) as gn_apply: one 32 x 32-channel LDS tile per step so both sides move whole 128-byte runs.
__global__ __launch_bounds__(256) void gn_apply_b8_kernel(const float* __restrict__ x, const float* __restrict__ a,
                                                          const float* __restrict__ b, float* __restrict__ y, int HW, int C,
                                                          int relu) {
    __shared__ float t[32][33];
    const int n = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tid = threadIdx.x;
    {   // read: 4 blocks of 8 channels x 32 pixels; a block row is 32 pixels x 32 bytes contiguous
        const int k = tid & 7, px = (tid >> 3) & 31;
        for (int blk = 0; blk < 4; ++blk) {
            const int p = p0 + px, c = c0 + blk * 8 + k;
            t[px][blk * 8 + k] = (p < HW && c < C) ? x[(((size_t)n * (C >> 3) + (c >> 3)) * HW + p) * 8 + k] : 0.f;
        }
    }
    __syncthreads();
    const int cx = tid & 31;
    for (int r = tid >> 5; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + cx;
        if (p < HW && c < C) {
            float v = t[r][cx];
            if (a) v = v * a[(size_t)n * C + c] + b[(size_t)n * C + c];
            if (relu) v = fmaxf(v, 0.f);
            y[((size_t)n * HW + p) * C + c] = v;
        }
    }
}
extern "C" int cpr_gn_apply_b8(const float* x, const float* a, const float* b, float* y, int N, int H, int W, int C,
                               int relu, hipStream_t stream) {
    CPR_CHECK_ARG(x && y && x != y && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && (a == nullptr) == (b == nullptr));
    hipLaunchKernelGGL(gn_apply_b8_kernel, dim3(cdiv(H * W, 32), cdiv(C, 32), N), dim3(256), 0, stream, x, a, b, y, H * W, C, relu);
    CPR_LAUNCH_STATUS();
}

// ================================================================================================
// bf16 variants (bf16 compute mode, BASELINE.json configs[4]): same kernels with 4 bf16 (8 bytes) per lane, fp32 math.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ f32x4 ld_bf16x4(const unsigned short* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    f32x4 v;
    v.x = __uint_as_float(u.x << 16); v.y = __uint_as_float(u.x & 0xffff0000u);
    v.z = __uint_as_float(u.y << 16); v.w = __uint_as_float(u.y & 0xffff0000u);
    return v;
}
__device__ __forceinline__ void st_bf16x4(unsigned short* p, f32x4 v) {
    const bf16x2_t lo = {(__bf16)v.x, (__bf16)v.y}, hi = {(__bf16)v.z, (__bf16)v.w};
    uint2 u;
    u.x = __builtin_bit_cast(unsigned, lo);
    u.y = __builtin_bit_cast(unsigned, hi);
    *reinterpret_cast<uint2*>(p) = u;
}

__global__ void maxpool3x3s2_bf16_kernel(const unsigned short* __restrict__ in, unsigned short* __restrict__ out, int N,
                                         int H, int W, int C4, int OH, int OW) {
    const long long total = (long long)N * OH * OW * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        long long r = i / C4;
        const int ox = (int)(r % OW);
        r /= OW;
        const int oy = (int)(r % OH);
        const int n = (int)(r / OH);
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = oy * 2 - 1 + dy;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = ox * 2 - 1 + dx;
                if ((unsigned)ix >= (unsigned)W) continue;
                const f32x4 v = ld_bf16x4(in + (((size_t)n * H + iy) * W + ix) * C4 * 4 + c * 4);
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        st_bf16x4(out + i * 4, m);
    }
}
extern "C" int cpr_maxpool3x3s2_bf16(const void* in, void* out, int N, int H, int W, int C, hipStream_t stream) {
    CPR_CHECK_ARG(in && out && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0);
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)N * OH * OW * (C / 4);
    const int grid = (int)(cdivll(total, 256) < 16384 ? cdivll(total, 256) : 16384);
    hipLaunchKernelGGL(maxpool3x3s2_bf16_kernel, dim3(grid), dim3(256), 0, stream, (const unsigned short*)in,
                       (unsigned short*)out, N, H, W, C / 4, OH, OW);
    CPR_LAUNCH_STATUS();
}

__global__ void gn_stats_bf16_kernel(const unsigned short* __restrict__ x, float* __restrict__ part, int HW, int C,
                                     int P) {
    __shared__ float red[256 * 8];
    const int n = blockIdx.y, slot = blockIdx.x;
    const int Q = C >> 2;
    const int q = threadIdx.x % Q, pl = threadIdx.x / Q, PP = 256 / Q;
    const int per = (HW + P - 1) / P;
    const int p0 = slot * per, p1 = min(HW, p0 + per);
    float s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    for (int p = p0 + pl; p < p1; p += PP) {
        const f32x4 v = ld_bf16x4(x + ((size_t)n * HW + p) * C + q * 4);
        s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
        ss[0] += v.x * v.x; ss[1] += v.y * v.y; ss[2] += v.z * v.z; ss[3] += v.w * v.w;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        red[threadIdx.x * 8 + k] = s[k];
        red[threadIdx.x * 8 + 4 + k] = ss[k];
    }
    __syncthreads();
    if (threadIdx.x < Q) {
        float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int r = 0; r < PP; ++r)
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] += red[(r * Q + q) * 8 + k];
        float* dst = part + (((size_t)n * P + slot) * C + q * 4) * 2;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            dst[k * 2] = a[k];
            dst[k * 2 + 1] = a[4 + k];
        }
    }
}
extern "C" int cpr_gn_stats_bf16(const void* x, float* part, int N, int HW, int C, int P, hipStream_t stream) {
    CPR_CHECK_ARG(x && part && N > 0 && HW > 0 && P > 0);
    CPR_CHECK_ARG(C % 4 == 0 && C / 4 <= 256 && 256 % (C / 4) == 0);
    hipLaunchKernelGGL(gn_stats_bf16_kernel, dim3(P, N), dim3(256), 0, stream, (const unsigned short*)x, part, HW, C, P);
    CPR_LAUNCH_STATUS();
}

__global__ void gn_apply_bf16_kernel(const unsigned short* __restrict__ x, const float* __restrict__ a,
                                     const float* __restrict__ b, const unsigned short* __restrict__ up,
                                     unsigned short* __restrict__ y, int N, int H, int W, int C4, int UH, int UW,
                                     int relu) {
    const long long total = (long long)N * H * W * C4;
    const float sy = (float)UH / (float)H, sx = (float)UW / (float)W;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        long long r = i / C4;
        const int px = (int)(r % W);
        r /= W;
        const int py = (int)(r % H);
        const int n = (int)(r / H);
        f32x4 v = ld_bf16x4(x + i * 4);
        const f32x4 av = *reinterpret_cast<const f32x4*>(a + ((size_t)n * C4 + c) * 4);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(b + ((size_t)n * C4 + c) * 4);
        v = v * av + bv;
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (up) {
            const int uy = min((int)floorf(py * sy), UH - 1), ux = min((int)floorf(px * sx), UW - 1);
            v = v + ld_bf16x4(up + ((((size_t)n * UH + uy) * UW + ux) * C4 + c) * 4);
        }
        st_bf16x4(y + i * 4, v);
    }
}
// The common case of the bf16 mode (C % 8 == 0, 256 % (C / 8) == 0): 16 bytes per lane, a thread keeps its channel group and walks
// pixels, so the per-image affine sits in registers and is reloaded only when the image changes; 32-bit index arithmetic.  (The
// generic kernel above moved 8 bytes per lane behind three 64-bit divisions and two 16-byte table loads: 1.5 TB/s, 0.8 ms = 8 %
// of the configs[4] step over nine launches; profiles/round4_cfg4_kernel_stats.csv.)
__global__ __launch_bounds__(256) void gn_apply_bf16_wide_kernel(const unsigned short* __restrict__ x, const float* __restrict__ a,
                                                                 const float* __restrict__ b, const unsigned short* __restrict__ up,
                                                                 unsigned short* __restrict__ y, int NP, int HW, int W, int C8,
                                                                 int H, int UH, int UW, int relu) {
    const int cg = threadIdx.x % C8, prow = threadIdx.x / C8, PP = 256 / C8;
    const int C = C8 * 8;
    const float sy = (float)UH / (float)H, sx = (float)UW / (float)W;
    int ncur = -1;
    f32x4 a0, a1, b0, b1;
    for (int pix = blockIdx.x * PP + prow; pix < NP; pix += gridDim.x * PP) {
        const int n = pix / HW;
        if (n != ncur) {
            ncur = n;
            const float* ap = a + (size_t)n * C + cg * 8;
            const float* bp = b + (size_t)n * C + cg * 8;
            a0 = *reinterpret_cast<const f32x4*>(ap); a1 = *reinterpret_cast<const f32x4*>(ap + 4);
            b0 = *reinterpret_cast<const f32x4*>(bp); b1 = *reinterpret_cast<const f32x4*>(bp + 4);
        }
        const size_t off = (size_t)pix * C + cg * 8;
        const uint4 u = *reinterpret_cast<const uint4*>(x + off);
        f32x4 v0 = {__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
        f32x4 v1 = {__uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u), __uint_as_float(u.w << 16), __uint_as_float(u.w & 0xffff0000u)};
        v0 = v0 * a0 + b0;
        v1 = v1 * a1 + b1;
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { v0[e] = fmaxf(v0[e], 0.f); v1[e] = fmaxf(v1[e], 0.f); }
        }
        if (up) {
            const int rem = pix - n * HW;
            const int py = rem / W, px = rem - py * W;
            const int uy = min((int)floorf(py * sy), UH - 1), ux = min((int)floorf(px * sx), UW - 1);
            const uint4 w = *reinterpret_cast<const uint4*>(up + (((size_t)n * UH + uy) * UW + ux) * C + cg * 8);
            v0[0] += __uint_as_float(w.x << 16); v0[1] += __uint_as_float(w.x & 0xffff0000u);
            v0[2] += __uint_as_float(w.y << 16); v0[3] += __uint_as_float(w.y & 0xffff0000u);
            v1[0] += __uint_as_float(w.z << 16); v1[1] += __uint_as_float(w.z & 0xffff0000u);
            v1[2] += __uint_as_float(w.w << 16); v1[3] += __uint_as_float(w.w & 0xffff0000u);
        }
        uint4 o;
        o.x = __builtin_bit_cast(unsigned, bf16x2_t{(__bf16)v0[0], (__bf16)v0[1]});
        o.y = __builtin_bit_cast(unsigned, bf16x2_t{(__bf16)v0[2], (__bf16)v0[3]});
        o.z = __builtin_bit_cast(unsigned, bf16x2_t{(__bf16)v1[0], (__bf16)v1[1]});
        o.w = __builtin_bit_cast(unsigned, bf16x2_t{(__bf16)v1[2], (__bf16)v1[3]});
        *reinterpret_cast<uint4*>(y + off) = o;
    }
}
extern "C" int cpr_gn_apply_bf16(const void* x, const float* a, const float* b, const void* up, void* y, int N, int H,
                                 int W, int C, int UH, int UW, int relu, hipStream_t stream) {
    CPR_CHECK_ARG(x && a && b && y && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0);
    if (up) CPR_CHECK_ARG(UH > 0 && UW > 0);
    const long long np = (long long)N * H * W;
    if (C % 8 == 0 && C / 8 <= 256 && 256 % (C / 8) == 0 && np < (1ll << 31)) {
        const int C8 = C / 8, PP = 256 / C8;
        const long long blocks = cdivll(np, PP);
        const int grid = (int)(blocks < 8192 ? blocks : 8192);
        hipLaunchKernelGGL(gn_apply_bf16_wide_kernel, dim3(grid), dim3(256), 0, stream, (const unsigned short*)x, a, b,
                           (const unsigned short*)up, (unsigned short*)y, (int)np, H * W, W, C8, H, UH, UW, relu);
        CPR_LAUNCH_STATUS();
    }
    const long long total = (long long)N * H * W * (C / 4);
    const int grid = (int)(cdivll(total, 256) < 32768 ? cdivll(total, 256) : 32768);
    hipLaunchKernelGGL(gn_apply_bf16_kernel, dim3(grid), dim3(256), 0, stream, (const unsigned short*)x, a, b,
                       (const unsigned short*)up, (unsigned short*)y, N, H, W, C / 4, UH, UW, relu);
    CPR_LAUNCH_STATUS();
}
