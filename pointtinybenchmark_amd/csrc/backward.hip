// Backward kernels of the CPR training step, everything except the conv data/weight gradients:
//   GroupNorm(+ReLU) backward as two streaming passes, eval-BatchNorm(+residual)+ReLU backward, FPN top-down add backward,
//   the CPR loss gradients (negative grid, MIL bags, gt centre) and the bilinear scatter back onto the logit map,
//   SGD-momentum with global-norm gradient clipping.
// References for the math: torch autograd on oracle/cpr_oracle.py (tests/test_gpu_backward.py compares against it).
#include "common.h"

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

// ------------------------------------------------------------------------------------------------ GroupNorm backward
// forward: y = x*a[n,c] + b[n,c] (a = rstd*gamma, b = beta - mean*a), z = relu?(y).  Given dz:
//   dy = dz * (y > 0 | !relu);  xhat = (x - mean) * rstd
//   dgamma[c] = sum dy*xhat, dbeta[c] = sum dy
//   dx = dy*a + x*k2[n,g] + k3[n,g],  k2 = -rstd^2 * m2, k3 = -rstd*m1 + mean*rstd^2*m2,
//        m1 = mean_g(dy*gamma), m2 = mean_g(dy*gamma*xhat)
// pass 1: per (image, slot, channel) partial (sum dy, sum dy*xhat)   -- same [N][P][C][2] layout as the forward stats
// (round 5) XB: x is the bf16 map the mixed-precision forward recorded, widened in registers (exact) -- the fp32 copy the step
// used to make first (and the bf16 copy of dx it made afterwards: OUT of the apply pass) cost 9 GB of traffic per head layer.
typedef __attribute__((ext_vector_type(2))) __bf16 gnb_bf16x2_t;
template <bool XB> __device__ __forceinline__ f32x4 gnb_load4(const void* __restrict__ x, size_t o) {
    if constexpr (XB) {
        const uint2 u = *reinterpret_cast<const uint2*>(static_cast<const unsigned short*>(x) + o);
        f32x4 v;
        v.x = __uint_as_float(u.x << 16); v.y = __uint_as_float(u.x & 0xffff0000u);
        v.z = __uint_as_float(u.y << 16); v.w = __uint_as_float(u.y & 0xffff0000u);
        return v;
    } else {
        return *reinterpret_cast<const f32x4*>(static_cast<const float*>(x) + o);
    }
}
// DZB (round 6): the upstream gradient dz is a bf16 map too (the bf16 data gradient of the layer above wrote it in bf16: 2 bytes
// less per element in its store and in both passes here); widened exactly on load.
template <bool XB, bool DZB = false>
__global__ void gn_bwd_stats_kernel(const void* __restrict__ x, const void* __restrict__ dz,
                                    const float* __restrict__ a, const float* __restrict__ b,
                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                    float* __restrict__ part, int HW, int C, int G, int P, int relu) {
    __shared__ float red[256 * 8];
    const int n = blockIdx.y, slot = blockIdx.x;
    const int Q = C >> 2;
    const int q = threadIdx.x % Q, pl = threadIdx.x / Q, PP = 256 / Q;
    const int per = (HW + P - 1) / P;
    const int p0 = slot * per, p1 = min(HW, p0 + per);
    const int cpg = C / G;
    const f32x4 av = *reinterpret_cast<const f32x4*>(a + (size_t)n * C + q * 4);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(b + (size_t)n * C + q * 4);
    float mu[4], rs[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int g = (q * 4 + k) / cpg;
        mu[k] = mean[n * G + g];
        rs[k] = rstd[n * G + g];
    }
    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    for (int p = p0 + pl; p < p1; p += PP) {
        const size_t o = ((size_t)n * HW + p) * C + q * 4;
        const f32x4 xv = gnb_load4<XB>(x, o);
        const f32x4 dv = gnb_load4<DZB>(dz, o);
        const f32x4 y = xv * av + bv;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float dy = (relu && !(y[k] > 0.f)) ? 0.f : dv[k];
            s1[k] += dy;
            s2[k] += dy * (xv[k] - mu[k]) * rs[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        red[threadIdx.x * 8 + k] = s1[k];
        red[threadIdx.x * 8 + 4 + k] = s2[k];
    }
    __syncthreads();
    if (threadIdx.x < Q) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int r = 0; r < PP; ++r)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += red[(r * Q + q) * 8 + k];
        float* dst = part + (((size_t)n * P + slot) * C + q * 4) * 2;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            dst[k * 2] = acc[k];
            dst[k * 2 + 1] = acc[4 + k];
        }
    }
}

// pass 1b: one block per image: per-channel sums over slots (double), group means, k2/k3; per-image dgamma/dbeta
// contributions go to dgb_part [N][C][2] and are summed over images by the caller's second launch (gn_bwd_params).
__global__ void gn_bwd_finalize_kernel(const float* __restrict__ part, const float* __restrict__ gamma,
                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                       float* __restrict__ k2, float* __restrict__ k3, float* __restrict__ dgb_part,
                                       int P, int C, int G, double count) {
    extern __shared__ double sh[];  // [C][2]
    const int n = blockIdx.x;
    const int cpg = C / G;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        double s1 = 0, s2 = 0;
        for (int t = 0; t < P; ++t) {
            const float* src = part + (((size_t)n * P + t) * C + c) * 2;
            s1 += (double)src[0];
            s2 += (double)src[1];
        }
        sh[c * 2] = s1;
        sh[c * 2 + 1] = s2;
        dgb_part[((size_t)n * C + c) * 2] = (float)s1;       // dbeta contribution
        dgb_part[((size_t)n * C + c) * 2 + 1] = (float)s2;   // dgamma contribution
    }
    __syncthreads();
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        double A = 0, B = 0;
        for (int k = 0; k < cpg; ++k) {
            const double gm = (double)gamma[g * cpg + k];
            A += gm * sh[(g * cpg + k) * 2];
            B += gm * sh[(g * cpg + k) * 2 + 1];
        }
        const double m1 = A / count, m2 = B / count;
        const double r = (double)rstd[n * G + g], mu = (double)mean[n * G + g];
        k2[n * G + g] = (float)(-r * r * m2);
        k3[n * G + g] = (float)(-r * m1 + mu * r * r * m2);
    }
}
__global__ void gn_bwd_params_kernel(const float* __restrict__ dgb_part, float* __restrict__ dgamma,
                                     float* __restrict__ dbeta, int N, int C, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double sb = 0, sg = 0;
    for (int n = 0; n < N; ++n) {
        sb += (double)dgb_part[((size_t)n * C + c) * 2];
        sg += (double)dgb_part[((size_t)n * C + c) * 2 + 1];
    }
    dbeta[c] = accumulate ? dbeta[c] + (float)sb : (float)sb;
    dgamma[c] = accumulate ? dgamma[c] + (float)sg : (float)sg;
}

// pass 2: dx = dy*a + x*k2 + k3   (dx and / or its bf16 rounding dx16: whichever pointer is given)
template <bool XB, bool DZB = false>
__global__ void gn_bwd_apply_kernel(const void* __restrict__ x, const void* __restrict__ dz,
                                    const float* __restrict__ a, const float* __restrict__ b,
                                    const float* __restrict__ k2, const float* __restrict__ k3, float* __restrict__ dx,
                                    unsigned short* __restrict__ dx16, int N, int HW, int C4, int G, int relu) {
    const long long total = (long long)N * HW * C4;
    const int cpg4 = (C4 * 4) / G;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        const int n = (int)(i / ((long long)HW * C4));
        const f32x4 xv = gnb_load4<XB>(x, (size_t)i * 4);
        const f32x4 dv = gnb_load4<DZB>(dz, (size_t)i * 4);
        const f32x4 av = *reinterpret_cast<const f32x4*>(a + ((size_t)n * C4 + c) * 4);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(b + ((size_t)n * C4 + c) * 4);
        const f32x4 y = xv * av + bv;
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int g = (c * 4 + k) / cpg4;
            const float dy = (relu && !(y[k] > 0.f)) ? 0.f : dv[k];
            o[k] = dy * av[k] + xv[k] * k2[n * G + g] + k3[n * G + g];
        }
        if (dx) *reinterpret_cast<f32x4*>(dx + i * 4) = o;
        if (dx16) {
            const gnb_bf16x2_t lo = {(__bf16)o.x, (__bf16)o.y}, hi = {(__bf16)o.z, (__bf16)o.w};   // round to nearest even, as torch's .to(bfloat16)
            uint2 u;
            u.x = __builtin_bit_cast(unsigned, lo);
            u.y = __builtin_bit_cast(unsigned, hi);
            *reinterpret_cast<uint2*>(dx16 + i * 4) = u;
        }
    }
}

template <bool XB, bool DZB = false>
static int gn_bwd_launch(const void* x, const void* dz, const float* a, const float* b, const float* mean, const float* rstd,
                         const float* gamma, float* dx, unsigned short* dx16, float* dgamma, float* dbeta, float* ws_part,
                         float* ws_k, int N, int HW, int C, int G, int P, int relu, int accumulate, hipStream_t stream) {
    // ws_part: N*P*C*2 floats; ws_k: 2*N*G + 2*N*C floats (k2 | k3 | per-image dgamma/dbeta contributions)
    CPR_CHECK_ARG(x && dz && a && b && mean && rstd && gamma && (dx || dx16) && dgamma && dbeta && ws_part && ws_k);
    CPR_CHECK_ARG(N > 0 && HW > 0 && G > 0 && P > 0 && C % G == 0 && (C / G) % 4 == 0 || (C / G) >= 1);
    CPR_CHECK_ARG(C % 4 == 0 && C / 4 <= 256 && 256 % (C / 4) == 0);
    float* k2 = ws_k;
    float* k3 = ws_k + (size_t)N * G;
    float* dgb = ws_k + (size_t)2 * N * G;
    hipLaunchKernelGGL((gn_bwd_stats_kernel<XB, DZB>), dim3(P, N), dim3(256), 0, stream, x, dz, a, b, mean, rstd, ws_part, HW, C, G,
                       P, relu);
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(N), dim3(256), (size_t)2 * C * sizeof(double), stream, ws_part, gamma,
                       mean, rstd, k2, k3, dgb, P, C, G, (double)HW * (C / G));
    hipLaunchKernelGGL(gn_bwd_params_kernel, dim3(cdiv(C, 256)), dim3(256), 0, stream, dgb, dgamma, dbeta, N, C,
                       accumulate);
    const long long total = (long long)N * HW * (C / 4);
    const int grid = (int)(cdivll(total, 256) < 32768 ? cdivll(total, 256) : 32768);
    hipLaunchKernelGGL((gn_bwd_apply_kernel<XB, DZB>), dim3(grid), dim3(256), 0, stream, x, dz, a, b, k2, k3, dx, dx16, N, HW, C / 4, G,
                       relu);
    CPR_LAUNCH_STATUS();
}
extern "C" int cpr_gn_bwd(const float* x, const float* dz, const float* a, const float* b, const float* mean,
                          const float* rstd, const float* gamma, float* dx, float* dgamma, float* dbeta, float* ws_part,
                          float* ws_k, int N, int HW, int C, int G, int P, int relu, int accumulate, hipStream_t stream) {
    CPR_CHECK_ARG(dx);
    return gn_bwd_launch<false>(x, dz, a, b, mean, rstd, gamma, dx, nullptr, dgamma, dbeta, ws_part, ws_k, N, HW, C, G, P, relu,
                                accumulate, stream);
}
// The mixed-precision step's form: x is the bf16 recorded map; dx (fp32) and / or dx16 (its bf16 rounding) are written -- the same
// values cpr_gn_bwd gives on the widened map, followed by a round-to-nearest-even narrowing.
extern "C" int cpr_gn_bwd_bf16(const void* x_bf16, const float* dz, const float* a, const float* b, const float* mean,
                               const float* rstd, const float* gamma, float* dx, void* dx_bf16, float* dgamma, float* dbeta,
                               float* ws_part, float* ws_k, int N, int HW, int C, int G, int P, int relu, int accumulate,
                               hipStream_t stream) {
    return gn_bwd_launch<true>(x_bf16, dz, a, b, mean, rstd, gamma, dx, (unsigned short*)dx_bf16, dgamma, dbeta, ws_part, ws_k, N,
                               HW, C, G, P, relu, accumulate, stream);
}
// ... and with the upstream gradient dz in bf16 as well (round 6)
extern "C" int cpr_gn_bwd_bf16_dz16(const void* x_bf16, const void* dz_bf16, const float* a, const float* b, const float* mean,
                                    const float* rstd, const float* gamma, float* dx, void* dx_bf16, float* dgamma, float* dbeta,
                                    float* ws_part, float* ws_k, int N, int HW, int C, int G, int P, int relu, int accumulate,
                                    hipStream_t stream) {
    return gn_bwd_launch<true, true>(x_bf16, dz_bf16, a, b, mean, rstd, gamma, dx, (unsigned short*)dx_bf16, dgamma, dbeta, ws_part,
                                     ws_k, N, HW, C, G, P, relu, accumulate, stream);
}

// ------------------------------------------------------------------------------------------------ FPN top-down add
// forward: fine = gn(fine_raw) + up_nearest(coarse).  backward wrt coarse: dcoarse[u] (+)= sum of dfine over its children
__global__ void upsample_add_bwd_kernel(const float* __restrict__ dfine, float* __restrict__ dcoarse, int N, int H, int W,
                                        int UH, int UW, int C4, int accumulate) {
    const long long total = (long long)N * UH * UW * C4;
    const float sy = (float)UH / (float)H, sx = (float)UW / (float)W;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        long long r = i / C4;
        const int ux = (int)(r % UW);
        r /= UW;
        const int uy = (int)(r % UH);
        const int n = (int)(r / UH);
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        const int y0 = max(0, (int)(uy / sy) - 1), y1 = min(H - 1, (int)((uy + 1) / sy) + 1);
        const int x0 = max(0, (int)(ux / sx) - 1), x1 = min(W - 1, (int)((ux + 1) / sx) + 1);
        for (int y = y0; y <= y1; ++y) {
            if (min((int)floorf(y * sy), UH - 1) != uy) continue;
            for (int x = x0; x <= x1; ++x) {
                if (min((int)floorf(x * sx), UW - 1) != ux) continue;
                s = s + *reinterpret_cast<const f32x4*>(dfine + ((((size_t)n * H + y) * W + x) * C4 + c) * 4);
            }
        }
        f32x4* dst = reinterpret_cast<f32x4*>(dcoarse + i * 4);
        *dst = accumulate ? *dst + s : s;
    }
}
extern "C" int cpr_upsample_add_bwd(const float* dfine, float* dcoarse, int N, int H, int W, int UH, int UW, int C,
                                    int accumulate, hipStream_t stream) {
    CPR_CHECK_ARG(dfine && dcoarse && N > 0 && H > 0 && W > 0 && UH > 0 && UW > 0 && C % 4 == 0);
    const long long total = (long long)N * UH * UW * (C / 4);
    const int grid = (int)(cdivll(total, 256) < 32768 ? cdivll(total, 256) : 32768);
    hipLaunchKernelGGL(upsample_add_bwd_kernel, dim3(grid), dim3(256), 0, stream, dfine, dcoarse, N, H, W, UH, UW, C / 4,
                       accumulate);
    CPR_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------ column sums
// out[c] (+)= sum_t src[t*row_stride + c*col_stride], t < rows.  32 channels x 8 row lanes per workgroup, 4 loads in
// flight per thread, blockIdx.y splits the rows; a second pass of the same kernel folds the splits.  Fixed summation
// order (deterministic).
__global__ void strided_colsum_kernel(const float* __restrict__ src, float* __restrict__ out, int rows, int C,
                                      long long row_stride, int col_stride, int rows_per_split, int accumulate) {
    __shared__ double red[8][33];
    const int cl = threadIdx.x & 31, r = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    const int t0 = blockIdx.y * rows_per_split, t1 = min(rows, t0 + rows_per_split);
    double s = 0;
    if (c < C) {
        const float* p = src + (size_t)c * col_stride;
        int t = t0 + r;
        for (; t + 24 < t1; t += 32) {
            const float a0 = p[(size_t)t * row_stride], a1 = p[(size_t)(t + 8) * row_stride];
            const float a2 = p[(size_t)(t + 16) * row_stride], a3 = p[(size_t)(t + 24) * row_stride];
            s += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
        }
        for (; t < t1; t += 8) s += (double)p[(size_t)t * row_stride];
    }
    red[r][cl] = s;
    __syncthreads();
    if (r == 0 && c < C) {
        double a = 0;
        for (int k = 0; k < 8; ++k) a += red[k][cl];
        float* dst = out + (size_t)blockIdx.y * C + c;
        *dst = accumulate ? *dst + (float)a : (float)a;
    }
}
// tmp: 64*C floats (only touched when rows > 256)
static void launch_colsum(const float* src, float* out, float* tmp, int rows, int C, long long row_stride, int col_stride,
                          int accumulate, hipStream_t stream) {
    int nsplit = (rows + 255) / 256;
    if (nsplit > 64) nsplit = 64;
    if (nsplit <= 1) {
        hipLaunchKernelGGL(strided_colsum_kernel, dim3(cdiv(C, 32), 1), dim3(256), 0, stream, src, out, rows, C, row_stride,
                           col_stride, rows, accumulate);
        return;
    }
    const int per = (rows + nsplit - 1) / nsplit;
    nsplit = (rows + per - 1) / per;
    hipLaunchKernelGGL(strided_colsum_kernel, dim3(cdiv(C, 32), nsplit), dim3(256), 0, stream, src, tmp, rows, C, row_stride,
                       col_stride, per, 0);
    hipLaunchKernelGGL(strided_colsum_kernel, dim3(cdiv(C, 32), 1), dim3(256), 0, stream, tmp, out, nsplit, C, (long long)C,
                       1, nsplit, accumulate);
}

// ------------------------------------------------------------------------------------------------ BN(eval)+add+ReLU
// forward (conv epilogue): y = relu?(conv*s[c] + t[c] (+ identity)).  Given dy:  g = dy * (y > 0 | !relu) is at once the
// gradient of the shortcut and (times s, folded into the data-gradient weights / applied to the weight gradient by
// bn_fold_bwd) of the conv output.  This kernel writes g and the per-channel column sums of g (= dshift).
// (M, C) row-major, C % 4 == 0.  y == nullptr: no mask (g = dy; g_out may be null -> column sums only).
// Round 6 (mixed precision): YB = the mask source y is the bf16 map the forward recorded (read as it is: no widened copy), g16_out
// (optional) = the bf16 rounding of g, written by the same pass (the bf16 weight / data gradients read it: no torch narrowing pass).
// add (optional, fp32): g = (dy + add) masked -- the shortcut gradient joins in the same pass (it was an axpby launch of its own).
template <bool YB>
__global__ void relu_bwd_colsum_kernel(const float* __restrict__ dy, const float* __restrict__ add, const void* __restrict__ yv_,
                                       float* __restrict__ g_out, unsigned short* __restrict__ g16_out, float* __restrict__ part,
                                       long long M, int C, int rows_per_block) {
    __shared__ float red[256 * 4];
    typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
    const float* y = reinterpret_cast<const float*>(yv_);
    const unsigned short* y16 = reinterpret_cast<const unsigned short*>(yv_);
    const int c0 = blockIdx.y * 1024;               // blockIdx.y: 1024-channel column group
    const int Q = min(1024, C - c0) >> 2;
    const int PP = 256 / Q;                         // rows handled per pass; threads >= PP*Q idle
    const int q = threadIdx.x % Q, pl = threadIdx.x / Q;
    const long long r0 = (long long)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float s1[4] = {0, 0, 0, 0};
    if (pl < PP) {
        // 4 rows in flight per thread: the kernel is pure streaming, so memory-level parallelism is what sets its speed
        for (long long r = r0 + pl; r < r1; r += 4 * PP) {
            f32x4 g[4], yv[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long ru = r + (long long)u * PP;
                ok[u] = ru < r1;
                const size_t o = (size_t)(ok[u] ? ru : r) * C + c0 + q * 4;
                g[u] = *reinterpret_cast<const f32x4*>(dy + o);
                if (add) g[u] += *reinterpret_cast<const f32x4*>(add + o);
                if (yv_) {
                    if (YB) {
                        const u16x4 h = *reinterpret_cast<const u16x4*>(y16 + o);
#pragma unroll
                        for (int k = 0; k < 4; ++k) yv[u][k] = __uint_as_float(((unsigned)h[k]) << 16);
                    } else yv[u] = *reinterpret_cast<const f32x4*>(y + o);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (!ok[u]) continue;
                if (yv_) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) g[u][k] = (yv[u][k] > 0.f) ? g[u][k] : 0.f;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) s1[k] += g[u][k];
                const size_t o = (size_t)(r + (long long)u * PP) * C + c0 + q * 4;
                if (g_out) *reinterpret_cast<f32x4*>(g_out + o) = g[u];
                if (g16_out) {
                    u16x4 h;
#pragma unroll
                    for (int k = 0; k < 4; ++k) h[k] = __builtin_bit_cast(unsigned short, (__bf16)g[u][k]);      // round to nearest even, as torch's .to(bfloat16)
                    *reinterpret_cast<u16x4*>(g16_out + o) = h;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) red[threadIdx.x * 4 + k] = s1[k];
    __syncthreads();
    if (threadIdx.x < Q) {
        float acc[4] = {0, 0, 0, 0};
        for (int r = 0; r < PP; ++r)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] += red[(r * Q + q) * 4 + k];
        float* dst = part + (size_t)blockIdx.x * C + c0 + q * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[k] = acc[k];
    }
}

// rows per workgroup: 128, halved until the launch has 2048 workgroups (down to 16).  A workgroup is 4 waves and the pass is pure
// streaming: layer3 of R101 at 1024^2 B = 8 (32 768 rows x 1024 channels) was 256 workgroups = one per CU = 4 waves per CU and ran at
// 3.2 TB/s (profiles/round6_train_cfg4_one_stream_after_mask_mode_kernel_stats.csv: 168 us for 537 MB).
static int relu_bwd_rows_per_block(long long M) {
    int rows = 128;
    while (rows > 16 && cdivll(M, rows) < 2048) rows >>= 1;
    return rows;
}
// floats of workspace cpr_relu_bwd_colsum needs for an (M, C) map
extern "C" int cpr_relu_bwd_colsum_ws(long long M, int C) {
    if (M <= 0 || C <= 0) return CPR_ERR_ARG;
    const long long n = (cdivll(M, relu_bwd_rows_per_block(M)) + 64) * C;
    return n < (1ll << 31) ? (int)n : CPR_ERR_UNSUPPORTED;
}
extern "C" int cpr_relu_bwd_colsum(const float* dy, const float* add, const void* y, int y_bf16, float* g_out, unsigned short* g16_out,
                                   float* colsum, float* ws_part, long long M, int C, int accumulate, hipStream_t stream) {
    // ws_part: cpr_relu_bwd_colsum_ws(M, C) floats
    CPR_CHECK_ARG(dy && colsum && ws_part && M > 0 && C > 0 && C % 4 == 0);
    const int rows_per_block = relu_bwd_rows_per_block(M);
    const int blocks = (int)cdivll(M, rows_per_block);
    if (y_bf16) hipLaunchKernelGGL(relu_bwd_colsum_kernel<true>, dim3(blocks, cdiv(C, 1024)), dim3(256), 0, stream, dy, add, y, g_out,
                                   g16_out, ws_part, M, C, rows_per_block);
    else hipLaunchKernelGGL(relu_bwd_colsum_kernel<false>, dim3(blocks, cdiv(C, 1024)), dim3(256), 0, stream, dy, add, y, g_out, g16_out,
                            ws_part, M, C, rows_per_block);
    launch_colsum(ws_part, colsum, ws_part + (size_t)blocks * C, blocks, C, (long long)C, 1, accumulate, stream);
    CPR_LAUNCH_STATUS();
}
// column sums from the conv epilogue's per-tile partials [tiles][C][2] (element 0 = sum); ws: 64*C floats
extern "C" int cpr_part_colsum(const float* part, float* out, float* ws, int tiles, int C, hipStream_t stream) {
    CPR_CHECK_ARG(part && out && ws && tiles > 0 && C > 0);
    launch_colsum(part, out, ws, tiles, C, (long long)2 * C, 2, 0, stream);
    CPR_LAUNCH_STATUS();
}
// parameter side of the folded BatchNorm: Gw = wgrad(g, x) [Cout][K] (unscaled), W [Cout][K], scale = gamma*inv_sigma:
//   dscale[c] = <W[c], Gw[c]> (= sum_p g*conv),  dshift[c] = colsum_g[c]
//   dgamma = inv_sigma*(dscale - mean*dshift),  dbeta = dshift,  dW[c] = scale[c]*Gw[c]  (in place)
// One block per output channel.
// tiles > 0 (round 6): colsum_g is not the (C) vector but the conv epilogue's partials [tiles][C][2] (element 0 = sum; mask mode /
// TR instances of the bf16 data gradient, the fp32 epilogue's CPR_CONV_COLSUM) -- the block adds its channel's column up itself,
// in a fixed order, instead of one or two strided_colsum launches in front of every one of these (~115 per configs[4] step).
__global__ void bn_fold_bwd_kernel(float* __restrict__ Gw, const float* __restrict__ Wt, const float* __restrict__ scale,
                                   const float* __restrict__ mean, const float* __restrict__ inv_sigma,
                                   const float* __restrict__ colsum_g, float* __restrict__ dgamma,
                                   float* __restrict__ dbeta, int K, int tiles) {
    __shared__ double red[4];
    __shared__ double redc[4];
    const int c = blockIdx.x;
    const float sc = scale[c];
    double csum = 0;
    if (tiles > 0) {
        const int C = gridDim.x;
        for (int t = threadIdx.x; t < tiles; t += blockDim.x) csum += (double)colsum_g[((size_t)t * C + c) * 2];
        csum = wave_sum_d(csum);
        if ((threadIdx.x & 63) == 0) redc[threadIdx.x >> 6] = csum;
    }
    double dot = 0;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const float gv = Gw[(size_t)c * K + k];
        dot += (double)gv * (double)Wt[(size_t)c * K + k];
        Gw[(size_t)c * K + k] = gv * sc;
    }
    dot = wave_sum_d(dot);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double dscale = red[0] + red[1] + red[2] + red[3];
        const double dshift = tiles > 0 ? (double)(float)(redc[0] + redc[1] + redc[2] + redc[3]) : (double)colsum_g[c];
        if (dgamma) dgamma[c] = (float)((double)inv_sigma[c] * (dscale - (double)mean[c] * dshift));
        if (dbeta) dbeta[c] = (float)dshift;
    }
}
extern "C" int cpr_bn_fold_bwd(float* Gw, const float* weight, const float* scale, const float* mean,
                               const float* inv_sigma, const float* colsum_g, float* dgamma, float* dbeta, int Cout, int K,
                               hipStream_t stream) {
    CPR_CHECK_ARG(Gw && weight && scale && mean && inv_sigma && colsum_g && Cout > 0 && K > 0);
    hipLaunchKernelGGL(bn_fold_bwd_kernel, dim3(Cout), dim3(256), 0, stream, Gw, weight, scale, mean, inv_sigma, colsum_g,
                       dgamma, dbeta, K, 0);
    CPR_LAUNCH_STATUS();
}
// the same with the column sums still in the conv epilogue's partials [tiles][Cout][2] (element 0 = sum): no cpr_part_colsum in front
extern "C" int cpr_bn_fold_bwd_part(float* Gw, const float* weight, const float* scale, const float* mean, const float* inv_sigma,
                                    const float* part, int tiles, float* dgamma, float* dbeta, int Cout, int K, hipStream_t stream) {
    CPR_CHECK_ARG(Gw && weight && scale && mean && inv_sigma && part && tiles > 0 && Cout > 0 && K > 0);
    hipLaunchKernelGGL(bn_fold_bwd_kernel, dim3(Cout), dim3(256), 0, stream, Gw, weight, scale, mean, inv_sigma, part, dgamma, dbeta, K,
                       tiles);
    CPR_LAUNCH_STATUS();
}

// elementwise helpers on flat fp32 buffers
__global__ void axpby_kernel(float* __restrict__ y, const float* __restrict__ x, float alpha, float beta, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        y[i] = alpha * x[i] + beta * y[i];
}
extern "C" int cpr_axpby(float* y, const float* x, float alpha, float beta, long long n, hipStream_t stream) {
    CPR_CHECK_ARG(n >= 0);
    if (n == 0) return CPR_OK;
    CPR_CHECK_ARG(x && y);
    const int grid = (int)(cdivll(n, 256) < 32768 ? cdivll(n, 256) : 32768);
    hipLaunchKernelGGL(axpby_kernel, dim3(grid), dim3(256), 0, stream, y, x, alpha, beta, n);
    CPR_LAUNCH_STATUS();
}
// phase decomposition of a strided data gradient: dst[n, s*i+py, s*j+px, :] += src[n, i+sh, j+sw, :] for every (i, j) whose
// target lies inside (H, W).  src (N,Hs,Ws,C) is the stride-1 sub-convolution of one output-parity class.
__global__ void phase_scatter_add_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int Hs, int Ws,
                                         int C4, int H, int W, int py, int px, int sh, int sw, int s) {
    const int nh = (H - py + s - 1) / s, nw = (W - px + s - 1) / s;
    const long long total = (long long)N * nh * nw * C4;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C4);
        long long r = idx / C4;
        const int j = (int)(r % nw);
        r /= nw;
        const int i = (int)(r % nh);
        const int n = (int)(r / nh);
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + ((((size_t)n * Hs + i + sh) * Ws + j + sw) * C4 + c) * 4);
        f32x4* d = reinterpret_cast<f32x4*>(dst + ((((size_t)n * H + s * i + py) * W + s * j + px) * C4 + c) * 4);
        *d = *d + v;
    }
}
extern "C" int cpr_phase_scatter_add(const float* src, float* dst, int N, int Hs, int Ws, int C, int H, int W, int py,
                                     int px, int sh, int sw, int s, hipStream_t stream) {
    CPR_CHECK_ARG(src && dst && N > 0 && Hs > 0 && Ws > 0 && C % 4 == 0 && H > 0 && W > 0 && s >= 1);
    CPR_CHECK_ARG(py >= 0 && py < s && px >= 0 && px < s && sh >= 0 && sw >= 0);
    const int nh = (H - py + s - 1) / s, nw = (W - px + s - 1) / s;
    CPR_CHECK_ARG(nh + sh <= Hs && nw + sw <= Ws);
    const long long total = (long long)N * nh * nw * (C / 4);
    if (total <= 0) return CPR_OK;
    const int grid = (int)(cdivll(total, 256) < 32768 ? cdivll(total, 256) : 32768);
    hipLaunchKernelGGL(phase_scatter_add_kernel, dim3(grid), dim3(256), 0, stream, src, dst, N, Hs, Ws, C / 4, H, W, py, px,
                       sh, sw, s);
    CPR_LAUNCH_STATUS();
}
// stride-2 data gradient helper: out (N,2H',2W',C)-like zero-inserted copy of dy: out[n, y*s, x*s, :] = dy[n,y,x,:]
__global__ void zero_insert_kernel(const float* __restrict__ dy, float* __restrict__ out, int N, int OH, int OW, int C4,
                                   int H, int W, int s) {
    const long long total = (long long)N * H * W * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        long long r = i / C4;
        const int x = (int)(r % W);
        r /= W;
        const int y = (int)(r % H);
        const int n = (int)(r / H);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (y % s == 0 && x % s == 0 && y / s < OH && x / s < OW)
            v = *reinterpret_cast<const f32x4*>(dy + ((((size_t)n * OH + y / s) * OW + x / s) * C4 + c) * 4);
        *reinterpret_cast<f32x4*>(out + i * 4) = v;
    }
}
extern "C" int cpr_zero_insert(const float* dy, float* out, int N, int OH, int OW, int C, int H, int W, int s,
                               hipStream_t stream) {
    CPR_CHECK_ARG(dy && out && N > 0 && OH > 0 && OW > 0 && C % 4 == 0 && H > 0 && W > 0 && s >= 1);
    const long long total = (long long)N * H * W * (C / 4);
    const int grid = (int)(cdivll(total, 256) < 32768 ? cdivll(total, 256) : 32768);
    hipLaunchKernelGGL(zero_insert_kernel, dim3(grid), dim3(256), 0, stream, dy, out, N, OH, OW, C / 4, H, W, s);
    CPR_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------ CPR loss backward
// d(total loss)/d(logit map) for the negative term (cpr_head.py:1216-1228): f = -p^2 log(1-p+eps) on valid (pixel,class)
// entries, scaled by w_neg/num_sample (num_sample = out5[4] of the forward).  Also zero-fills the remaining channels so the
// bag scatter can accumulate on top.  dmap (N,HW,Jd), Jd >= J (padded so the conv data/weight-gradient kernels can eat it).
__global__ void neg_loss_bwd_kernel(const float* __restrict__ logit, const unsigned char* __restrict__ mask,
                                    const float* __restrict__ out5, float* __restrict__ dmap, long long NP, int J, int Jd,
                                    int C, float eps, float w_neg, const float* __restrict__ up) {
    if (up) w_neg *= up[3];          // upstream gradient of neg_loss (autograd bridge; 1.0 multiplies exactly)
    const float scale = w_neg / out5[4];
    const long long total = NP * Jd;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(i % Jd);
        const long long p = i / Jd;
        float g = 0.f;
        if (j < C && mask[p * C + j]) {
            const float s = sigm(logit[p * J + j]);
            const float om = 1.f - s + eps;
            g = (-2.f * s * logf(om) + s * s / om) * s * (1.f - s) * scale;
        }
        dmap[i] = g;
    }
}

// gfocal derivative wrt the probability:  L = -(p-q)^2 [q log(p+eps) + (1-q) log(1-p+eps)]
__device__ __forceinline__ float gfocal_dp(float p, float q, float eps) {
    const float l2 = q * logf(p + eps) + (1.f - q) * logf(1.f - p + eps);
    return -(2.f * (p - q) * l2 + (p - q) * (p - q) * (q / (p + eps) - (1.f - q) / (1.f - p + eps)));
}

// gradient of (pos_loss + gt_loss) wrt the sampled bag logits dbag (G,K,J)
//   P = sum_k s_k w_k, w_k = softmax_k(ins)*valid / sum;  dP/dcls_k = w_k s_k (1-s_k);  dP/dins_k = w_k (s_k - P)
// CLS == false: one wave per bag, the classes one after the other.  CLS == true (round 5; C >= 8, e.g. the 80 classes of
// BASELINE.json configs[2]: 0.92 ms per step on 48 workgroups there): one 8-wave workgroup per bag, wave w takes classes w, w + 8, ...
// -- every class's lanes-over-the-bag passes and wave reductions are the same code, and a class owns its own columns of dbag:
// BIT-identical outputs (the forward's mil_bag_cls_kernel makes the same split).
template <bool CLS>
__global__ void bag_loss_bwd_kernel(const float* __restrict__ logits, int J, int ins_off,
                                    const unsigned char* __restrict__ valid, const int* __restrict__ labels,
                                    const float* __restrict__ gt_weight, const float* __restrict__ bag,
                                    float* __restrict__ dbag, int G, int K, int C, float eps, float w_mil, float w_gt,
                                    const float* __restrict__ up) {
    constexpr int NWV = CLS ? 8 : 4;
    __shared__ double red[2][NWV];
    if (up) { w_gt *= up[0]; w_mil *= up[1]; }     // upstream gradients of gt_loss / pos_loss
    // num_sample / num_pos_gt: deterministic block-local recount of the forward's per-bag flags
    double ns = 0, ng = 0;
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        ns += (double)bag[(size_t)g * 5 + 2];
        ng += (double)bag[(size_t)g * 5 + 3];
    }
    ns = wave_sum_d(ns);
    ng = wave_sum_d(ng);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = ns;
        red[1][threadIdx.x >> 6] = ng;
    }
    __syncthreads();
    // (counts of bags held in double: exact however the partial sums are grouped, so the 4- and the 8-wave form agree)
    double ns_t = 0, ng_t = 0;
#pragma unroll
    for (int w = 0; w < NWV; ++w) { ns_t += red[0][w]; ng_t += red[1][w]; }
    const double num_sample = fmax(ns_t, 1.0);
    const double num_pos_gt = fmax(ng_t, 1.0);
    const float k_mil = (float)((double)w_mil / num_sample), k_gt = (float)((double)w_gt / num_pos_gt);

    const int wave = threadIdx.x >> 6;
    const int g = CLS ? (int)blockIdx.x : (int)(blockIdx.x * (blockDim.x >> 6)) + wave;
    const int lane = threadIdx.x & 63;
    if (g >= G) return;
    const float* L = logits + (size_t)g * K * J;
    float* D = dbag + (size_t)g * K * J;
    const unsigned char* V = valid + (size_t)g * K;
    const int label = labels[g];
    const float wg = gt_weight ? gt_weight[g] : 1.f;
    if (CLS) {
        for (int i = threadIdx.x; i < K * J; i += 64 * NWV) D[i] = 0.f;
        __syncthreads();             // (workgroup-scope release / acquire: the zero fill is ordered before the other waves' stores)
    } else {
        for (int i = lane; i < K * J; i += 64) D[i] = 0.f;
    }
    float nvalid = 0.f;
    for (int k = lane; k < K; k += 64) nvalid += V[k] ? 1.f : 0.f;
    nvalid = wave_sum(nvalid);
    const float lw = (nvalid * wg > 0.f) ? 1.f : 0.f;
    const float gtv = V[K - 1] ? wg : 0.f;
    __builtin_amdgcn_s_waitcnt(0);   // the zero fill above is ordered before the accumulating stores (same lanes/addresses differ)
    __builtin_amdgcn_wave_barrier();
    for (int c = CLS ? wave : 0; c < C; c += CLS ? NWV : 1) {
        float m = -INFINITY;
        for (int k = lane; k < K; k += 64) m = fmaxf(m, L[(size_t)k * J + ins_off + c]);
        m = wave_max(m);
        float se = 0.f;
        for (int k = lane; k < K; k += 64) se += expf(L[(size_t)k * J + ins_off + c] - m);
        se = wave_sum(se);
        float sv = 0.f, sp = 0.f;
        for (int k = lane; k < K; k += 64) {
            const float pi = expf(L[(size_t)k * J + ins_off + c] - m) / se * (V[k] ? wg : 0.f);
            sv += pi;
            sp += sigm(L[(size_t)k * J + c]) * pi;
        }
        sv = wave_sum(sv);
        sp = wave_sum(sp);
        const float den = fmaxf(sv, 1e-12f);
        const float P = sp / den;
        const float q = (c == label) ? 1.f : 0.f;
        const float dP = gfocal_dp(P, q, eps) * lw * k_mil;
        const bool norm_live = sv > 1e-12f;   // F.normalize: below eps the denominator is the constant eps
        for (int k = lane; k < K; k += 64) {
            const float sig = expf(L[(size_t)k * J + ins_off + c] - m) / se;
            const float pi = sig * (V[k] ? wg : 0.f);
            const float w = pi / den;
            const float s = sigm(L[(size_t)k * J + c]);
            float dcls = dP * w * s * (1.f - s);
            // d/dins_k of sum_j s_j pi_j / den:  norm live: w_k (s_k - P);  clamped: (pi_k s_k - sig_k * sp) / den
            const float dins = norm_live ? dP * w * (s - P) : dP * (pi * s - sig * sp) / den;
            if (k == K - 1) {
                const float gp = s;
                dcls += gfocal_dp(gp, q, eps) * gtv * k_gt * gp * (1.f - gp);
            }
            if (ins_off == 0) {   // ins_share_head_classifier: one logit plays both roles
                D[(size_t)k * J + c] = dcls + dins;
            } else {
                D[(size_t)k * J + c] = dcls;
                D[(size_t)k * J + ins_off + c] = dins;
            }
        }
    }
}

// dbag (G,K,J) back through the bilinear taps of bag_sample onto dmap (N,H,W,Jd) -- DETERMINISTIC (round 4; the round-1..3
// form scattered with float atomics, so two runs of the same step differed in the last bits and nothing downstream of the loss
// could be held bit-equal between the trainer and the autograd bridge).  Two launches:
//   bag_window_kernel      one workgroup per bag.  The K points of a bag lie within `radius` cells of their centre, so all their
//                          taps fall into a WIN x WIN window (WIN = 2*radius + 3) whose origin is the smallest tap cell of the
//                          bag.  Tap cells / weights of the K points go to LDS once; thread (cell, j) then GATHERS: it walks the
//                          points in k order and sums the taps that land on its cell -- one owner per output, fixed order.
//   bag_window_add_kernel  one workgroup per image: adds the windows of the image's bags onto dmap bag after bag (barrier in
//                          between: overlapping bags are summed in gt order).
// The tap arithmetic is bag_sample's (grid_sample coordinate round trip, border clamp, align_corners = False).
// (round 5) cap > 0: before the gather every window cell collects, once, the list of points whose 2 x 2 taps cover it (ascending k,
// at most `cap`; a longer list falls back to the full walk) -- the (cell, channel) threads then walk ~5 points instead of all K.
// With 80 classes (J = 160, K = 289 at radius 8: BASELINE.json configs[2]) the full walk was 16.7 M tap tests per bag, 4.3 ms per
// step; same points in the same order, same arithmetic: bit-identical windows.
__global__ void bag_window_kernel(const float* __restrict__ dbag, int J, const float* __restrict__ ctr,
                                  const float* __restrict__ offs, float* __restrict__ win, int* __restrict__ win_org, int WIN,
                                  int K, int H, int W, float stride, int cap) {
    extern __shared__ unsigned char smem_raw[];
    int* tx0 = reinterpret_cast<int*>(smem_raw);                 // [K] tap cell x0
    int* ty0 = tx0 + K;                                           // [K] tap cell y0
    float* tww = reinterpret_cast<float*>(ty0 + K);               // [K] weight of x0 + 1
    float* twn = tww + K;                                         // [K] weight of y0 + 1
    int* hcnt = reinterpret_cast<int*>(twn + K);                  // [WIN * WIN] points covering the cell (cap > 0)
    unsigned short* hits = reinterpret_cast<unsigned short*>(hcnt + WIN * WIN);   // [WIN * WIN][cap]
    __shared__ int org[2];
    const int g = blockIdx.x;
    if (threadIdx.x == 0) { org[0] = 0x7fffffff; org[1] = 0x7fffffff; }
    __syncthreads();
    const float cx = ctr[g * 2], cy = ctr[g * 2 + 1];
    const float fw = (float)W, fh = (float)H;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const float px = (k < K - 1) ? __fadd_rn(offs[k * 2], cx) : cx;
        const float py = (k < K - 1) ? __fadd_rn(offs[k * 2 + 1], cy) : cy;
        float gx = __fsub_rn(__fdiv_rn(__fadd_rn(2.f * __fdiv_rn(px, stride), 1.f), fw), 1.f);
        float gy = __fsub_rn(__fdiv_rn(__fadd_rn(2.f * __fdiv_rn(py, stride), 1.f), fh), 1.f);
        float ix = __fsub_rn(__fmul_rn(__fadd_rn(gx, 1.f), fw), 1.f) * 0.5f;
        float iy = __fsub_rn(__fmul_rn(__fadd_rn(gy, 1.f), fh), 1.f) * 0.5f;
        ix = fminf(fw - 1.f, fmaxf(ix, 0.f));
        iy = fminf(fh - 1.f, fmaxf(iy, 0.f));
        const float x0f = floorf(ix), y0f = floorf(iy);
        tx0[k] = (int)x0f;
        ty0[k] = (int)y0f;
        tww[k] = ix - x0f;
        twn[k] = iy - y0f;
        atomicMin(&org[0], (int)x0f);        // integer minimum: order-independent
        atomicMin(&org[1], (int)y0f);
    }
    __syncthreads();
    const int ox = org[0], oy = org[1];
    if (threadIdx.x == 0) { win_org[g * 2] = ox; win_org[g * 2 + 1] = oy; }
    if (cap > 0) {
        for (int cell = threadIdx.x; cell < WIN * WIN; cell += blockDim.x) {
            const int x = ox + cell % WIN, y = oy + cell / WIN;
            int n = 0;
            for (int k = 0; k < K; ++k) {
                const unsigned dx = (unsigned)(x - tx0[k]), dy = (unsigned)(y - ty0[k]);
                if (dx > 1u || dy > 1u) continue;
                if (n < cap) hits[cell * cap + n] = (unsigned short)k;
                ++n;
            }
            hcnt[cell] = n;
        }
        __syncthreads();
    }
    const float* D = dbag + (size_t)g * K * J;
    float* Wg = win + (size_t)g * WIN * WIN * J;
    const int total = WIN * WIN * J;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        const int j = i % J, cell = i / J;
        const int x = ox + cell % WIN, y = oy + cell / WIN;
        float acc = 0.f;
        if (x < W && y < H) {
            const int n = cap > 0 ? hcnt[cell] : K;
            const bool listed = cap > 0 && n <= cap;
            for (int t = 0; t < (listed ? n : K); ++t) {
                const int k = listed ? (int)hits[cell * cap + t] : t;
                const unsigned dx = (unsigned)(x - tx0[k]), dy = (unsigned)(y - ty0[k]);
                if (dx > 1u || dy > 1u) continue;
                const float d = D[(size_t)k * J + j];
                if (d == 0.f) continue;
                const float ww = tww[k], wn_ = twn[k];
                const float wy = dy ? wn_ : 1.f - wn_, wx = dx ? ww : 1.f - ww;
                acc += d * wy * wx;
            }
        }
        Wg[i] = acc;
    }
}
// (round 5: blockIdx.y takes a slice of BWA_SLICE channels -- with 160 logit channels five times the workgroups per image; an
// output element still meets its bags in gt order)
#define BWA_SLICE 32
__global__ void __launch_bounds__(1024) bag_window_add_kernel(const float* __restrict__ win, const int* __restrict__ win_org, int WIN, int J,
                                      const int* __restrict__ gt_img, float* __restrict__ dmap, int Jd, int G, int H, int W) {
    __shared__ int range[2];
    const int n = blockIdx.x;
    const int j0 = blockIdx.y * BWA_SLICE, jn = min(BWA_SLICE, J - j0);
    if (threadIdx.x == 0) {        // the bags of image n (gt_img ascends: CSR order)
        int lo = 0;
        while (lo < G && gt_img[lo] < n) ++lo;
        int hi = lo;
        while (hi < G && gt_img[hi] == n) ++hi;
        range[0] = lo;
        range[1] = hi;
    }
    __syncthreads();
    const int lo = range[0], hi = range[1];
    float* base = dmap + (size_t)n * H * W * Jd;
    const int total = WIN * WIN * jn;
    for (int g = lo; g < hi; ++g) {
        const int ox = win_org[g * 2], oy = win_org[g * 2 + 1];
        const float* Wg = win + (size_t)g * WIN * WIN * J;
        for (int i = threadIdx.x; i < total; i += blockDim.x) {
            const int j = j0 + i % jn, cell = i / jn;
            const int x = ox + cell % WIN, y = oy + cell / WIN;
            const float v = Wg[(size_t)cell * J + j];
            if (v != 0.f && x < W && y < H) base[((size_t)y * W + x) * Jd + j] += v;
        }
        __syncthreads();           // the next bag may touch the same cells (workgroup-scope fence + barrier)
    }
}

// the two launches of the window gather.  Hit lists (16 per cell) when they fit beside the taps in 60 KB of LDS and the bag is long
// enough to pay for building them; the add kernel over 32-channel slices.
static void bag_gather_launch(const float* dsample, int J, const float* centers, const int* gt_img, const float* offsets,
                              float* win_ws, int* win_org, int win, float* dmap, int N, int H, int W, int Jd, int G, int K,
                              float stride, hipStream_t stream) {
    const size_t taps = (size_t)K * 16, lists = (size_t)win * win * (4 + 16 * 2);
    const int cap = (K >= 32 && K <= 65535 && taps + lists <= 60000) ? 16 : 0;
    hipLaunchKernelGGL(bag_window_kernel, dim3(G), dim3(256), taps + (cap ? lists : 0), stream, dsample, J, centers, offsets, win_ws,
                       win_org, win, K, H, W, stride, cap);
    const int slice = J < BWA_SLICE ? J : BWA_SLICE;
    hipLaunchKernelGGL(bag_window_add_kernel, dim3(N, cdiv(J, BWA_SLICE)), dim3(win * win * slice >= 4096 ? 1024 : 256), 0, stream,
                       win_ws, win_org, win, J, gt_img, dmap, Jd, G, H, W);
}

extern "C" int cpr_loss_bwd(const float* lmap, const unsigned char* neg_mask, const float* out5, const float* bag_logits,
                            const unsigned char* valid, const int* labels, const float* gt_weight, const float* bag_ws,
                            const float* centers, const int* gt_img, const float* offsets, float* dbag_ws, float* dmap,
                            float* win_ws, int* win_org, int win,
                            int N, int H, int W, int J, int Jd, int ins_off, int G, int K, int C, float stride, float eps,
                            float w_mil, float w_gt, float w_neg, const float* upstream, hipStream_t stream) {
    CPR_CHECK_ARG(lmap && neg_mask && out5 && bag_logits && valid && labels && bag_ws && centers && gt_img && dbag_ws && dmap);
    CPR_CHECK_ARG(win == 0 || (win_ws && win_org && win >= 3 && (size_t)K * 16 <= 60000));
    CPR_CHECK_ARG(N > 0 && H > 0 && W > 0 && G > 0 && K > 0 && C > 0 && J >= ins_off + C && Jd >= J && (win == 0 || K == 1 || offsets));
    const long long NP = (long long)N * H * W;
    const int grid = (int)(cdivll(NP * Jd, 256) < 32768 ? cdivll(NP * Jd, 256) : 32768);
    hipLaunchKernelGGL(neg_loss_bwd_kernel, dim3(grid), dim3(256), 0, stream, lmap, neg_mask, out5, dmap, NP, J, Jd, C,
                       eps, w_neg, upstream);
    if (C >= 8)       // many classes: one workgroup per bag, the classes over its eight waves (bit-identical)
        hipLaunchKernelGGL(bag_loss_bwd_kernel<true>, dim3(G), dim3(512), 0, stream, bag_logits, J, ins_off, valid, labels,
                           gt_weight, bag_ws, dbag_ws, G, K, C, eps, w_mil, w_gt, upstream);
    else
        hipLaunchKernelGGL(bag_loss_bwd_kernel<false>, dim3(cdiv(G, 4)), dim3(256), 0, stream, bag_logits, J, ins_off, valid, labels,
                           gt_weight, bag_ws, dbag_ws, G, K, C, eps, w_mil, w_gt, upstream);
    if (win == 0) { CPR_LAUNCH_STATUS(); }     // the bag logits were not sampled from the map (num_cls_fcs > 0): dbag_ws is the result
    bag_gather_launch(dbag_ws, J, centers, gt_img, offsets, win_ws, win_org, win, dmap, N, H, W, Jd, G, K, stride, stream);
    CPR_LAUNCH_STATUS();
}

// The gather stage alone: dsample (G, K, J) -- the gradient wrt the bilinear samples bag_sample took from a (N, H, W, .) map --
// is ADDED onto dmap (N, H, W, Jd) through the same taps (bag_window_kernel + bag_window_add_kernel: deterministic, gt order).
// CPRHead with num_cls_fcs > 0 samples the 256-channel FEATURES (the ReLUs of the FC stack do not commute with the interpolation,
// cpr_head.py:1055-1059), so its bag gradient reaches the feature map here instead of through the logit map.
extern "C" int cpr_bag_gather_bwd(const float* dsample, int J, const float* centers, const int* gt_img, const float* offsets,
                                  float* win_ws, int* win_org, int win, float* dmap, int N, int H, int W, int Jd, int G, int K,
                                  float stride, hipStream_t stream) {
    CPR_CHECK_ARG(dsample && centers && gt_img && win_ws && win_org && dmap && win >= 3 && (size_t)K * 16 <= 60000);
    CPR_CHECK_ARG(N > 0 && H > 0 && W > 0 && G > 0 && K > 0 && J > 0 && Jd >= J && (K == 1 || offsets));
    bag_gather_launch(dsample, J, centers, gt_img, offsets, win_ws, win_org, win, dmap, N, H, W, Jd, G, K, stride, stream);
    CPR_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------ gather from a point list
// (round 5) The gather stage for the generators whose taps the window kernels above do not describe: GridCirclesPtFeatGenerator
// (cpr_head.py:296-352,405-438: a bag is the grid cells inside the circles of the gt's refine points -- exact cell values, zero
// padded -- followed by the refine points themselves) and align_corners=True sampling (:73-93,126: grid 2x/(w-1)-1, zeros
// padding, no clip).  Entries are described by the forward's own outputs: pts (E,2) and, for grid bags, code (E) = cell index
// y*W+x (exact cell value, weight 1) | -1 padding slot | <= -2 a bilinear sample at pts (NULL: every entry is a bilinear sample).
// One workgroup per (image, 32-channel slice) walks the image's gts in order; per gt the taps of its Kt entries go to LDS, the
// bounding box of the live taps is the gather window, thread (cell, j) sums the taps landing on its cell in entry order and adds
// onto dmap: one owner per output, fixed order -- deterministic like the window kernels.  wout (E), optional: the weight of each
// entry's DROPPED taps (padding slot: 1; bilinear tap outside the map under align_corners: its weight) -- what the projection's
// bias receives when the sampled map holds logits (sample_point4, csrc/cpr_points.hip: pad[c] * wout).
#define CPRG_SLICE 32
__global__ void bag_points_gather_kernel(const float* __restrict__ dsample, int J, const float* __restrict__ pts,
                                         const int* __restrict__ code, const int* __restrict__ gt_img, int Kt,
                                         float* __restrict__ dmap, int Jd, float* __restrict__ wout, int G, int H, int W,
                                         float stride, int align) {
    extern __shared__ unsigned char smem_raw[];
    int* tx0 = reinterpret_cast<int*>(smem_raw);                 // [Kt] tap cell x0
    int* ty0 = tx0 + Kt;                                          // [Kt] tap cell y0
    float* tww = reinterpret_cast<float*>(ty0 + Kt);              // [Kt] weight of x0 + 1
    float* twn = tww + Kt;                                        // [Kt] weight of y0 + 1
    int* tin = reinterpret_cast<int*>(twn + Kt);                  // [Kt] live taps: bit 0 nw, 1 ne, 2 sw, 3 se
    __shared__ int box[4], range[2];
    const int n = blockIdx.x;
    const int j0 = blockIdx.y * CPRG_SLICE, jn = min(CPRG_SLICE, J - j0);
    if (threadIdx.x == 0) {        // the gts of image n (gt_img ascends: CSR order)
        int lo = 0;
        while (lo < G && gt_img[lo] < n) ++lo;
        int hi = lo;
        while (hi < G && gt_img[hi] == n) ++hi;
        range[0] = lo;
        range[1] = hi;
    }
    __syncthreads();
    const int lo = range[0], hi = range[1];
    const float fw = (float)W, fh = (float)H;
    float* base = dmap + (size_t)n * H * W * Jd;
    for (int g = lo; g < hi; ++g) {
        if (threadIdx.x == 0) { box[0] = 0x7fffffff; box[1] = 0x7fffffff; box[2] = -1; box[3] = -1; }
        __syncthreads();
        for (int k = threadIdx.x; k < Kt; k += blockDim.x) {
            const size_t e = (size_t)g * Kt + k;
            const int c = code ? code[e] : -2;
            int x0 = 0, y0 = 0, in = 0;
            float ww = 0.f, wn_ = 0.f, dropped = 0.f;
            if (c >= 0) {                                   // a grid cell: its value, weight 1
                x0 = c % W; y0 = c / W; in = 1;
            } else if (c == -1) {                           // padding slot
                dropped = 1.f;
            } else {
                const float px = pts[e * 2], py = pts[e * 2 + 1];
                if (align) {                                // sample_point4, align != 0
                    const float gx = __fsub_rn(__fdiv_rn(2.f * __fdiv_rn(px, stride), fw - 1.f), 1.f);
                    const float gy = __fsub_rn(__fdiv_rn(2.f * __fdiv_rn(py, stride), fh - 1.f), 1.f);
                    const float ix = __fmul_rn(__fadd_rn(gx, 1.f), (fw - 1.f) * 0.5f), iy = __fmul_rn(__fadd_rn(gy, 1.f), (fh - 1.f) * 0.5f);
                    const float x0f = floorf(ix), y0f = floorf(iy);
                    ww = __fsub_rn(ix, x0f); wn_ = __fsub_rn(iy, y0f);
                    const bool xin0 = (x0f >= 0.f) && (x0f <= fw - 1.f), xin1 = (x0f + 1.f >= 0.f) && (x0f + 1.f <= fw - 1.f);
                    const bool yin0 = (y0f >= 0.f) && (y0f <= fh - 1.f), yin1 = (y0f + 1.f >= 0.f) && (y0f + 1.f <= fh - 1.f);
                    in = (xin0 && yin0 ? 1 : 0) | (xin1 && yin0 ? 2 : 0) | (xin0 && yin1 ? 4 : 0) | (xin1 && yin1 ? 8 : 0);
                    // the cell of the nw tap, also when only its neighbours are inside (x0 = -1 is a legal origin)
                    x0 = (x0f >= -1.f && x0f <= fw) ? (int)x0f : -4;
                    y0 = (y0f >= -1.f && y0f <= fh) ? (int)y0f : -4;
                    if (x0 == -4 || y0 == -4) in = 0;
                    const float we = __fsub_rn(1.f, ww), ws = __fsub_rn(1.f, wn_);
                    const float wt[4] = {__fmul_rn(ws, we), __fmul_rn(ws, ww), __fmul_rn(wn_, we), __fmul_rn(wn_, ww)};
                    for (int t = 0; t < 4; ++t) dropped += ((in >> t) & 1) ? 0.f : wt[t];
                } else {                                    // sample_point4, border clip
                    float gx = __fsub_rn(__fdiv_rn(__fadd_rn(2.f * __fdiv_rn(px, stride), 1.f), fw), 1.f);
                    float gy = __fsub_rn(__fdiv_rn(__fadd_rn(2.f * __fdiv_rn(py, stride), 1.f), fh), 1.f);
                    float ix = __fsub_rn(__fmul_rn(__fadd_rn(gx, 1.f), fw), 1.f) * 0.5f;
                    float iy = __fsub_rn(__fmul_rn(__fadd_rn(gy, 1.f), fh), 1.f) * 0.5f;
                    ix = fminf(fw - 1.f, fmaxf(ix, 0.f));
                    iy = fminf(fh - 1.f, fmaxf(iy, 0.f));
                    const float x0f = floorf(ix), y0f = floorf(iy);
                    x0 = (int)x0f; y0 = (int)y0f;
                    ww = ix - x0f; wn_ = iy - y0f;
                    const bool x1ok = x0 + 1 < W, y1ok = y0 + 1 < H;
                    in = 1 | (x1ok ? 2 : 0) | (y1ok ? 4 : 0) | (x1ok && y1ok ? 8 : 0);   // (a clipped tap carries weight 0)
                }
            }
            tx0[k] = x0; ty0[k] = y0; tww[k] = ww; twn[k] = wn_; tin[k] = in;
            if (wout && blockIdx.y == 0) wout[e] = dropped;
            if (in) {
                const int xa = (in & 5) ? x0 : x0 + 1, xb = (in & 10) ? x0 + 1 : x0;
                const int ya = (in & 3) ? y0 : y0 + 1, yb = (in & 12) ? y0 + 1 : y0;
                atomicMin(&box[0], xa); atomicMin(&box[1], ya); atomicMax(&box[2], xb); atomicMax(&box[3], yb);
            }
        }
        __syncthreads();
        const int ox = box[0], oy = box[1], bw = box[2] - box[0] + 1, bh = box[3] - box[1] + 1;
        if (box[2] >= 0) {
            const float* D = dsample + (size_t)g * Kt * J;
            const int total = bw * bh * jn;
            for (int i = threadIdx.x; i < total; i += blockDim.x) {
                const int j = j0 + i % jn, cell = i / jn;
                const int x = ox + cell % bw, y = oy + cell / bw;
                float acc = 0.f;
                for (int k = 0; k < Kt; ++k) {
                    const unsigned dx = (unsigned)(x - tx0[k]), dy = (unsigned)(y - ty0[k]);
                    if (dx > 1u || dy > 1u || !((tin[k] >> (dy * 2 + dx)) & 1)) continue;
                    const float d = D[(size_t)k * J + j];
                    if (d == 0.f) continue;
                    const float ww = tww[k], wn_ = twn[k];
                    acc += d * (dy ? wn_ : 1.f - wn_) * (dx ? ww : 1.f - ww);
                }
                if (acc != 0.f) base[((size_t)y * W + x) * Jd + j] += acc;
            }
        }
        __syncthreads();           // the next gt reuses the LDS taps and may touch the same cells
    }
}
// dbias[j] = sum_e wout[e] * dsample[e][j]: one workgroup per channel, fixed partition + fixed-order reduce (deterministic)
__global__ void pad_bias_grad_kernel(const float* __restrict__ dsample, int J, const float* __restrict__ wout, long long E,
                                     float* __restrict__ dbias) {
    __shared__ double red[4];
    const int j = blockIdx.x;
    double a = 0;
    for (long long e = threadIdx.x; e < E; e += blockDim.x) {
        const float w = wout[e];
        if (w != 0.f) a += (double)w * (double)dsample[e * J + j];
    }
    a = wave_sum_d(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) dbias[j] = (float)(red[0] + red[1] + red[2] + red[3]);
}
// dsample (G, Kt, J) ADDED onto dmap (N,H,W,Jd) through the taps of the entries (pts, code); wout_ws (G*Kt) + dbias (J): both or
// neither -- dbias <- what the projection's bias receives from the dropped taps / padding slots.
extern "C" int cpr_bag_points_gather_bwd(const float* dsample, int J, const float* pts, const int* code, const int* gt_img,
                                         float* dmap, float* wout_ws, float* dbias, int N, int H, int W, int Jd, int G, int Kt,
                                         float stride, int align_corners, hipStream_t stream) {
    CPR_CHECK_ARG(dsample && pts && gt_img && dmap && N > 0 && H > 0 && W > 0 && G > 0 && Kt > 0 && J > 0 && Jd >= J && stride > 0);
    CPR_CHECK_ARG((!wout_ws) == (!dbias) && (size_t)Kt * 20 <= 60000 && (!align_corners || (H > 1 && W > 1)));
    hipLaunchKernelGGL(bag_points_gather_kernel, dim3(N, cdiv(J, CPRG_SLICE)), dim3(256), (size_t)Kt * 20, stream, dsample, J, pts,
                       code, gt_img, Kt, dmap, Jd, wout_ws, G, H, W, stride, align_corners);
    if (dbias)
        hipLaunchKernelGGL(pad_bias_grad_kernel, dim3(J), dim3(256), 0, stream, dsample, J, wout_ws, (long long)G * Kt, dbias);
    CPR_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------ CPR loss backward, general form
// (round 5) The same gradients for the CPRHead options the kernels above do not cover -- softmax / normed_sigmoid class
// probabilities (cpr_head.py:1080-1099), MILLoss(binary_ins=True) and AllPosLoss (multi_instance_learning_loss.py:153-243), the
// merge_to_gt_bag / only_refine_bag policies and gt_loss_type='gt' (cpr_head.py:1159-1211: the bag / annotated-point geometry of
// mil_bag_kernel, csrc/cpr_points.hip), out_bg_cls (one more classifier output that is never a label), with_mil_loss=False.
// The forward kernels already take all of them; these mirror them term by term.  Probabilities p(l) of one point and the
// product of an upstream vector g = dL/dp with their Jacobian:
//   sigmoid           p_c = s(l_c)                       dL/dl_c = g_c p_c (1 - p_c)
//   softmax           p = softmax(l)                     dL/dl_c = p_c (g_c - sum_j g_j p_j)
//   normed_sigmoid    p_c = s_c / n, n = ||s||_P         dL/dl_c = (g_c / n - (sum_j g_j s_j) s_c^(P-1) n^(-1-P)) s_c (1 - s_c)
// (n >= 1e-12 always: a sum of sigmoids.)  The shipped configs keep the specialised kernels above: bit-for-bit unchanged.
#define CPRB_SIGMOID 0
#define CPRB_SOFTMAX 1
#define CPRB_NORMED 2
struct ProbNorm { float m, se, n; };     // softmax: max and sum of exp; normed_sigmoid: the norm
__device__ __forceinline__ ProbNorm prob_norm(const float* __restrict__ l, int C, int ptype, float P) {
    ProbNorm r{0.f, 1.f, 1.f};
    if (ptype == CPRB_SOFTMAX) {
        r.m = -INFINITY;
        for (int j = 0; j < C; ++j) r.m = fmaxf(r.m, l[j]);
        r.se = 0.f;
        for (int j = 0; j < C; ++j) r.se += expf(l[j] - r.m);
    } else if (ptype == CPRB_NORMED) {
        float a = 0.f;
        for (int j = 0; j < C; ++j) {
            const float sj = sigm(l[j]);
            a += (P == 1.f) ? sj : (P == 2.f ? sj * sj : powf(sj, P));
        }
        r.n = fmaxf((P == 1.f) ? a : (P == 2.f ? sqrtf(a) : powf(a, 1.f / P)), 1e-12f);
    }
    return r;
}
__device__ __forceinline__ float prob_of(const float* __restrict__ l, int c, int ptype, const ProbNorm& r) {
    if (ptype == CPRB_SOFTMAX) return expf(l[c] - r.m) / r.se;
    const float sc = sigm(l[c]);
    return ptype == CPRB_NORMED ? sc / r.n : sc;
}
// dL/dl_c from g_c = dL/dp_c and the cross term X (softmax: sum_j g_j p_j; normed_sigmoid: sum_j g_j s_j; sigmoid: unused)
__device__ __forceinline__ float prob_back(const float* __restrict__ l, int c, float gc, float X, int ptype, float P, const ProbNorm& r) {
    if (ptype == CPRB_SOFTMAX) return expf(l[c] - r.m) / r.se * (gc - X);
    const float sc = sigm(l[c]);
    if (ptype == CPRB_SIGMOID) return gc * sc * (1.f - sc);
    const float spm1 = (P == 1.f) ? 1.f : (P == 2.f ? sc : powf(sc, P - 1.f));
    return (gc / r.n - X * spm1 * powf(r.n, -1.f - P)) * sc * (1.f - sc);
}
__device__ __forceinline__ float prob_cross(int ptype, float gc, float pc, float sc) {   // one term of X
    return ptype == CPRB_SOFTMAX ? gc * pc : (ptype == CPRB_NORMED ? gc * sc : 0.f);
}

// negative-grid term, any probability type: dmap (NP, Jd) fully written (zeros beyond the class channels; all zeros when
// mask is NULL: loss_cfg with_neg=False, cpr_head.py:1219)
__global__ void neg_loss_bwd_general_kernel(const float* __restrict__ logit, const unsigned char* __restrict__ mask,
                                            const float* __restrict__ out5, const float* __restrict__ bag, int G,
                                            float* __restrict__ dmap, long long NP, int J, int Jd, int C, float eps, float w_neg,
                                            int neg_from_gt, int ptype, float P, const float* __restrict__ up) {
    __shared__ double red[4];
    if (up) w_neg *= up[3];
    double den = (double)out5[4];
    if (neg_from_gt) {      // with_mil_loss=False: averaged over the annotated-point positives (cpr_head.py:1180,1227)
        double ng = 0;
        for (int g = threadIdx.x; g < G; g += blockDim.x) ng += (double)bag[(size_t)g * 5 + 3];
        ng = wave_sum_d(ng);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ng;
        __syncthreads();
        den = fmax(red[0] + red[1] + red[2] + red[3], 1.0);
    }
    const float scale = (float)((double)w_neg / den);
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < NP; p += (long long)gridDim.x * blockDim.x) {
        const float* l = logit + p * J;
        float* d = dmap + p * Jd;
        const ProbNorm r = prob_norm(l, C, ptype, P);
        float X = 0.f;
        for (int c = 0; c < C; ++c) {
            if (!mask || !mask[p * C + c]) continue;
            const float pc = prob_of(l, c, ptype, r);
            X += prob_cross(ptype, gfocal_dp(pc, 0.f, eps) * scale, pc, sigm(l[c]));
        }
        for (int c = 0; c < C; ++c) {
            const float gc = (mask && mask[p * C + c]) ? gfocal_dp(prob_of(l, c, ptype, r), 0.f, eps) * scale : 0.f;
            d[c] = (ptype == CPRB_SIGMOID && gc == 0.f) ? 0.f : prob_back(l, c, gc, X, ptype, P, r);
        }
        for (int j = C; j < Jd; ++j) d[j] = 0.f;
    }
}

// one wave per bag (the geometry of mil_bag_kernel): dbag rows [b * bag_stride, (b + 1) * bag_stride) x J fully written
#define CPRB_MAXT 256        // class x branch terms whose statistics fit the wave's LDS slice
__global__ void bag_loss_bwd_general_kernel(const float* __restrict__ logits, int J, int ins_off,
                                            const unsigned char* __restrict__ valid, const int* __restrict__ labels,
                                            const float* __restrict__ gt_weight, const float* __restrict__ bag,
                                            float* __restrict__ dbag, int G, int bag_stride, int bag_off, int K, int ctr_off,
                                            int ctr_stride, int ctr_count, int ctr_mod, int C, float eps, int ptype, float P,
                                            int binary_ins, int allpos, float w_mil, float w_gt, const float* __restrict__ up) {
    __shared__ double red[2][4];
    __shared__ float st[4][CPRB_MAXT][5];        // per wave, per (class, branch): softmax max, sum, normaliser, P, dL/dP
    if (up) { w_gt *= up[0]; w_mil *= up[1]; }
    double ns = 0, ng = 0;
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        ns += (double)bag[(size_t)g * 5 + 2];
        ng += (double)bag[(size_t)g * 5 + 3];
    }
    ns = wave_sum_d(ns);
    ng = wave_sum_d(ng);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = ns;
        red[1][threadIdx.x >> 6] = ng;
    }
    __syncthreads();
    const double num_sample = fmax(red[0][0] + red[0][1] + red[0][2] + red[0][3], 1.0);
    const double num_pos_gt = fmax(red[1][0] + red[1][1] + red[1][2] + red[1][3], 1.0);
    const float k_mil = (float)((double)w_mil / num_sample), k_gt = (float)((double)w_gt / num_pos_gt);

    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = blockIdx.x * (blockDim.x >> 6) + wv;
    if (g >= G) return;
    const size_t full = (size_t)g * bag_stride;
    const float* L = logits + (full + bag_off) * J;
    const unsigned char* V = valid + full + bag_off;
    const int label = labels[g];
    const float wg = gt_weight ? gt_weight[g] : 1.f;
    const int nj = binary_ins ? 2 : 1;
    float (*S)[5] = st[wv];
    if (!allpos) {
        float nvalid = 0.f;
        for (int k = lane; k < K; k += 64) nvalid += V[k] ? 1.f : 0.f;
        nvalid = wave_sum(nvalid);
        const float lw = (nvalid * wg > 0.f) ? 1.f : 0.f;
        for (int c = 0; c < C; ++c) {
            for (int j = 0; j < nj; ++j) {
                const int ic = ins_off + c * nj + j;
                float m = -INFINITY;
                for (int k = lane; k < K; k += 64) m = fmaxf(m, L[(size_t)k * J + ic]);
                m = wave_max(m);
                float se = 0.f;
                for (int k = lane; k < K; k += 64) se += expf(L[(size_t)k * J + ic] - m);
                se = wave_sum(se);
                float sv = 0.f, sp = 0.f;
                for (int k = lane; k < K; k += 64) {
                    const float* lk = L + (size_t)k * J;
                    const float pi = expf(lk[ic] - m) / se * (V[k] ? wg : 0.f);
                    sv += pi;
                    sp += prob_of(lk, c, ptype, prob_norm(lk, C, ptype, P)) * pi;
                }
                sv = wave_sum(sv);
                sp = wave_sum(sp);
                const float den = fmaxf(sv, 1e-12f);
                const float Pb = sp / den;
                const float q = (j == 0 && c == label) ? 1.f : 0.f;
                if (lane == 0) {
                    float* o = S[c * nj + j];
                    o[0] = m; o[1] = se; o[2] = sv; o[3] = sp; o[4] = gfocal_dp(Pb, q, eps) * lw * k_mil;
                }
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    const bool ctr_bag = ctr_count > 0 && (g % ctr_mod) == 0;
    for (int k = lane; k < bag_stride; k += 64) {
        const size_t e = full + k;
        const float* lk = logits + e * J;
        float* D = dbag + e * J;
        for (int j = 0; j < J; ++j) D[j] = 0.f;
        const int kb = k - bag_off;
        const bool in_bag = kb >= 0 && kb < K;
        bool is_ctr = false;
        if (ctr_bag) {
            const int t = k - ctr_off;
            is_ctr = t >= 0 && t % ctr_stride == 0 && t / ctr_stride < ctr_count;
        }
        if (!in_bag && !is_ctr) continue;
        const float vk = (in_bag && V[kb]) ? wg : 0.f;               // the bag's validity weight of this entry
        const float gtv = (is_ctr && valid[e]) ? wg : 0.f;           // the annotated-point term's weight
        const ProbNorm r = prob_norm(lk, C, ptype, P);
        // dL/dp_c of this entry: bag term(s) + annotated-point term
        auto up_c = [&](int c, float pc) {
            float gc = 0.f;
            const float q = (c == label) ? 1.f : 0.f;
            if (in_bag) {
                if (allpos) gc += gfocal_dp(pc, q, eps) * vk * k_mil;
                else
                    for (int j = 0; j < nj; ++j) {
                        const float* o = S[c * nj + j];
                        const float pi = expf(lk[ins_off + c * nj + j] - o[0]) / o[1] * vk;
                        gc += o[4] * pi / fmaxf(o[2], 1e-12f);
                    }
            }
            if (is_ctr) gc += gfocal_dp(pc, q, eps) * gtv * k_gt;
            return gc;
        };
        float X = 0.f;
        if (ptype != CPRB_SIGMOID)
            for (int c = 0; c < C; ++c) {
                const float pc = prob_of(lk, c, ptype, r);
                X += prob_cross(ptype, up_c(c, pc), pc, sigm(lk[c]));
            }
        for (int c = 0; c < C; ++c) {
            const float pc = prob_of(lk, c, ptype, r);
            float dcls = prob_back(lk, c, up_c(c, pc), X, ptype, P, r);
            if (in_bag && !allpos) {
                for (int j = 0; j < nj; ++j) {      // d/d(instance logit): norm live: w (p - P); clamped: (pi p - sig sp) / den
                    const float* o = S[c * nj + j];
                    const int ic = ins_off + c * nj + j;
                    const float sig = expf(lk[ic] - o[0]) / o[1];
                    const float pi = sig * vk;
                    const float den = fmaxf(o[2], 1e-12f);
                    const float dins = (o[2] > 1e-12f) ? o[4] * (pi / den) * (pc - o[3] / den) : o[4] * (pi * pc - sig * o[3]) / den;
                    if (ins_off == 0) dcls += dins;          // ins_share_head_classifier on shared features: one logit, both roles
                    else D[ic] = dins;
                }
            }
            D[c] = dcls;
        }
    }
}

// Launcher of the general form: dmap (N,H,W,Jd) = the negative-grid term alone, dbag (num_bags * bag_stride, J) = the gradient wrt
// every bag entry's logits (NOT gathered: cpr_bag_gather_bwd adds it onto dmap when the entries were sampled from the map).
extern "C" int cpr_loss_bwd_general(const float* lmap, const unsigned char* neg_mask, const float* out5, const float* bag_logits,
                                    const unsigned char* valid, const int* labels, const float* gt_weight, const float* bag_ws,
                                    float* dbag, float* dmap, int N, int H, int W, int J, int Jd, int ins_off, int num_bags,
                                    int bag_stride, int bag_off, int bag_len, int ctr_off, int ctr_stride, int ctr_count,
                                    int ctr_mod, int C, float eps, int prob_type, float norm_p, int binary_ins, int allpos,
                                    float w_mil, float w_gt, float w_neg, int neg_from_gt, const float* upstream,
                                    hipStream_t stream) {
    CPR_CHECK_ARG(lmap && out5 && bag_logits && valid && labels && bag_ws && dbag && dmap);     // neg_mask NULL: with_neg=False
    CPR_CHECK_ARG(N > 0 && H > 0 && W > 0 && num_bags > 0 && bag_len > 0 && C > 0 && Jd >= J && bag_off >= 0 &&
                  bag_stride >= bag_off + bag_len && ctr_count >= 0 && ctr_mod >= 1 && ctr_stride >= 1);
    CPR_CHECK_ARG(J >= ins_off + C * (binary_ins ? 2 : 1) && prob_type >= 0 && prob_type <= 2 && norm_p > 0.f);
    CPR_CHECK_ARG(C * (binary_ins ? 2 : 1) <= CPRB_MAXT && !(allpos && binary_ins) && !(binary_ins && ins_off == 0));
    CPR_CHECK_ARG(ctr_count == 0 || (ctr_off >= 0 && ctr_off + (ctr_count - 1) * ctr_stride < bag_stride));
    const long long NP = (long long)N * H * W;
    const int grid = (int)(cdivll(NP, 256) < 32768 ? cdivll(NP, 256) : 32768);
    hipLaunchKernelGGL(neg_loss_bwd_general_kernel, dim3(grid), dim3(256), 0, stream, lmap, neg_mask, out5, bag_ws, num_bags, dmap,
                       NP, J, Jd, C, eps, w_neg, neg_from_gt, prob_type, norm_p, upstream);
    hipLaunchKernelGGL(bag_loss_bwd_general_kernel, dim3(cdiv(num_bags, 4)), dim3(256), 0, stream, bag_logits, J, ins_off, valid,
                       labels, gt_weight, bag_ws, dbag, num_bags, bag_stride, bag_off, bag_len, ctr_off, ctr_stride, ctr_count,
                       ctr_mod, C, eps, prob_type, norm_p, binary_ins, allpos, w_mil, w_gt, upstream);
    CPR_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------ optimizer
// global gradient norm (mmcv Fp32/OptimizerHook grad_clip -> torch clip_grad_norm_): sum of squares in double, per-buffer
// partials reduced by a second launch into norm2[0]; deterministic.
__global__ void sumsq_kernel(const float* __restrict__ g, long long n, double* __restrict__ partial) {
    __shared__ double red[4];
    double s = 0;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        s += (double)g[i] * (double)g[i];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void sumsq_final_kernel(const double* __restrict__ partial, int n, double* __restrict__ out, int accumulate) {
    __shared__ double red[4];
    double s = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += partial[i];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.0) + red[0] + red[1] + red[2] + red[3];
}
extern "C" int cpr_grad_sumsq(const float* g, long long n, double* ws_partial, double* out, int accumulate,
                              hipStream_t stream) {
    // ws_partial: 1024 doubles
    CPR_CHECK_ARG(g && ws_partial && out && n > 0);
    const int grid = (int)(cdivll(n, 1024) < 1024 ? cdivll(n, 1024) : 1024);
    hipLaunchKernelGGL(sumsq_kernel, dim3(grid), dim3(256), 0, stream, g, n, ws_partial);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, stream, ws_partial, grid, out, accumulate);
    CPR_LAUNCH_STATUS();
}
// torch.optim.SGD step on one flat fp32 buffer (momentum, dampening 0, no nesterov), with the clip coefficient read from
// the device: coef = min(1, max_norm / (sqrt(norm2) + 1e-6)) when max_norm > 0 (clip_grad_norm_), else 1.
//   g = coef*grad*grad_scale + wd*p;  buf = first ? g : mu*buf + g;  p -= lr*buf
__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ grad, float* __restrict__ buf,
                           const double* __restrict__ norm2, long long n, float lr, float mu, float wd, float max_norm,
                           float grad_scale, int first) {
    float coef = grad_scale;
    if (max_norm > 0.f) {
        const float tn = (float)sqrt(norm2[0]) * grad_scale;
        const float c = max_norm / (tn + 1e-6f);
        coef *= fminf(c, 1.f);
    }
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float g = grad[i] * coef;
        const float pv = p[i];
        if (wd != 0.f) g = g + wd * pv;
        float b = g;
        if (mu != 0.f) {
            b = first ? g : mu * buf[i] + g;
            buf[i] = b;
        }
        p[i] = pv - lr * b;
    }
}
extern "C" int cpr_sgd_step(float* p, const float* grad, float* buf, const double* norm2, long long n, float lr, float mu,
                            float wd, float max_norm, float grad_scale, int first, hipStream_t stream) {
    CPR_CHECK_ARG(p && grad && n > 0 && (mu == 0.f || buf) && (max_norm <= 0.f || norm2));
    const int grid = (int)(cdivll(n, 256) < 8192 ? cdivll(n, 256) : 8192);
    hipLaunchKernelGGL(sgd_kernel, dim3(grid), dim3(256), 0, stream, p, grad, buf, norm2, n, lr, mu, wd, max_norm,
                       grad_scale, first);
    CPR_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------ P2P loss backward
// d(sum_b loss_cls[b] + loss_pts[b]) of cpr_p2p_loss (P2PHead.loss_single, p2p_head.py:220-248) wrt the class logits and
// the regression output.  Sigmoid focal loss (py_sigmoid_focal_loss, focal_loss.py:11-56; the focal weight is NOT detached):
//   d/dx [bce * at * pt^g] = at * [pt^g (p - t) + bce * g * pt^(g-1) * (1 - 2t) * p (1 - p)]
// SmoothL1 on pred/stride/reg_norm with pred = anchor + (point_anchor + reg*gamma_p) * stride:
//   d/dreg = (|e| < beta ? e/beta : sign(e)) * gamma_p / reg_norm,  e = (pred - gt)/stride/reg_norm
// Both scaled by loss_weight / (number of positives in the batch, a device scalar).  Outputs use the padded channel counts
// of the conv gradient kernels: dcls (B*M, Cp) with the first C columns live, dreg (B*M, Rp) with the first 2 live.
__global__ void p2p_loss_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ pred,
                                    const long long* __restrict__ gt_inds, const float* __restrict__ gt_pts,
                                    const int* __restrict__ gt_labels, const int* __restrict__ gt_start,
                                    const float* __restrict__ npos, float* __restrict__ dcls, float* __restrict__ dreg,
                                    int M, int C, int Cp, int Rp, float alpha, float gamma, float beta, float pos_w,
                                    float neg_w, float reg_norm, float w_cls, float w_reg, float gamma_p,
                                    const float* __restrict__ up, int cls_mode, int reg_mode, float cls_total) {
    const int b = blockIdx.y;
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    if (up) { w_cls *= up[b * 2]; w_reg *= up[b * 2 + 1]; }     // upstream gradients of this image's (loss_cls, loss_pts)
    const size_t r = (size_t)b * M + m;
    const float inv_n = 1.f / fmaxf(npos[0], 1.f);
    const float inv_c = cls_mode == 0 ? inv_n : 1.f / cls_total;          // CrossEntropyLoss modes: averaged over all proposals
    const long long gi = gt_inds[r];
    const bool pos = gi > 0;
    const int g = pos ? gt_start[b] + (int)gi - 1 : 0;
    const int nfg = cls_mode == 2 ? C - 1 : C;
    const int label = pos ? gt_labels[g] : nfg;
    const float w = (gi < 0) ? 0.f : pos ? pos_w : (neg_w <= 0.f ? 1.f : neg_w);   // gi < 0: invalid cell, label weight 0
    float mx = -INFINITY, se = 0.f;
    if (cls_mode == 2) {
        for (int c = 0; c < C; ++c) mx = fmaxf(mx, logits[r * C + c]);
        for (int c = 0; c < C; ++c) se += expf(logits[r * C + c] - mx);
    }
    for (int c = 0; c < Cp; ++c) {
        float d = 0.f;
        if (c < C) {
            const float x = logits[r * C + c];
            if (cls_mode == 2) {
                const int lab = gi < 0 ? 0 : label;
                d = (expf(x - mx) / se - (c == lab ? 1.f : 0.f)) * w * w_cls * inv_c;
            } else {
                const float p = 1.f / (1.f + expf(-x));
                const float t = (c == label) ? 1.f : 0.f;
                if (cls_mode == 1) {
                    d = (p - t) * w * w_cls * inv_c;
                } else {
                    const float pt = (1.f - p) * t + p * (1.f - t);
                    const float at = alpha * t + (1.f - alpha) * (1.f - t);
                    const float ptg = (gamma == 2.f) ? pt * pt : powf(pt, gamma);
                    const float ptg1 = (gamma == 2.f) ? pt : powf(pt, gamma - 1.f);
                    const float bce = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
                    d = at * (ptg * (p - t) + bce * gamma * ptg1 * (1.f - 2.f * t) * p * (1.f - p)) * w * w_cls * inv_c;
                }
            }
        }
        dcls[r * Cp + c] = d;
    }
    const float s = pred[r * 3 + 2];
    for (int k = 0; k < Rp; ++k) {
        float d = 0.f;
        if (pos && k < 2) {
            const float e = pred[r * 3 + k] / s / reg_norm - gt_pts[g * 2 + k] / s / reg_norm;
            const float sg = e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f);
            const float de = reg_mode == 1 ? 2.f * e : reg_mode == 2 ? sg : (fabsf(e) < beta) ? e / beta : sg;
            d = de * gamma_p / reg_norm * w_reg * inv_n;
        }
        dreg[r * Rp + k] = d;
    }
}
extern "C" int cpr_p2p_loss_bwd(const float* logits, const float* pred, const long long* gt_inds, const float* gt_pts,
                                const int* gt_labels, const int* gt_start, const float* npos, float* dcls, float* dreg,
                                int B, int M, int C, int Cp, int Rp, float alpha, float gamma, float beta, float pos_w,
                                float neg_w, float reg_norm, float w_cls, float w_reg, float gamma_p, const float* upstream,
                                int cls_mode, int reg_mode, hipStream_t stream) {
    CPR_CHECK_ARG(B > 0 && M > 0 && C > 0 && Cp >= C && Rp >= 2 && (beta > 0 || reg_mode != 0));
    CPR_CHECK_ARG(cls_mode >= 0 && cls_mode <= 2 && reg_mode >= 0 && reg_mode <= 2 && (cls_mode != 2 || C >= 2));
    CPR_CHECK_ARG(logits && pred && gt_inds && gt_pts && gt_labels && gt_start && npos && dcls && dreg);
    hipLaunchKernelGGL(p2p_loss_bwd_kernel, dim3(cdiv(M, 256), B), dim3(256), 0, stream, logits, pred, gt_inds, gt_pts,
                       gt_labels, gt_start, npos, dcls, dreg, M, C, Cp, Rp, alpha, gamma, beta, pos_w, neg_w, reg_norm,
                       w_cls, w_reg, gamma_p, upstream, cls_mode, reg_mode, (float)((double)B * (double)M));
    CPR_LAUNCH_STATUS();
}
