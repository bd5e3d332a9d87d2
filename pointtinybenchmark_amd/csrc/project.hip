// Logit projection of the CPR head: the two classifiers cls_out / ins_out (nn.Linear(256 -> C), T/mmdet/models/point/
// dense_heads/cpr_head.py:1007-1014,1045-1078) applied to EVERY pixel of the head feature map in one streaming pass:
//   out[n][y][x][j] = bias[j] + sum_c W[j][c] * relu?(x[n][y][x][c] * a[n][c] + b[n][c])         j < J = 2C (<= 8)
// (a, b) = the GroupNorm affine of the last tower layer, applied on load, so the normalised 26 MB/img map is never
// written.  With J = 2 a 128x64 MFMA tile does 3 % useful work and the launch is bound by how the A operand streams
// through LDS; this kernel is a pure HBM stream built like the GroupNorm statistics pass (few registers, many waves):
// 16 lanes own one pixel (lane q of them holds channels [4q + 64k, +4), k < Cin/64: every load instruction of a wave is
// four contiguous 256-byte segments), so a wave covers 4 pixels per iteration; the J dot products are finished with a
// 4-step xor-shuffle inside each 16-lane group.  Algorithmic bytes: 4*Cin per pixel in, 4*J out.
#include "common.h"

template <int J, int KC>   // KC = Cin / 64 float4 loads per lane (Cin = 64, 128, 192, 256)
__global__ __launch_bounds__(256) void logit_project_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, const float* __restrict__ a,
                                                            const float* __restrict__ b, float* __restrict__ out,
                                                            long long npix, int HW, int in_relu) {
    constexpr int Cin = KC * 64;
    const int q = threadIdx.x & 15;
    const long long slot = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;   // pixel slot of this 16-lane group
    const long long nslots = ((long long)gridDim.x * blockDim.x) >> 4;
    f32x4 wj[J][KC];
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int k = 0; k < KC; ++k) wj[j][k] = *reinterpret_cast<const f32x4*>(w + (size_t)j * Cin + k * 64 + q * 4);
    const float floor_ = (in_relu && a) ? 0.f : -INFINITY;   // the ReLU belongs to the fused GroupNorm-apply
    for (long long p = slot; p < npix; p += nslots) {
        const float* xp = x + (size_t)p * Cin + q * 4;
        f32x4 xv[KC];
#pragma unroll
        for (int k = 0; k < KC; ++k) xv[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xp + k * 64));
        if (a) {
            const float* ap = a + (size_t)(p / HW) * Cin + q * 4;
            const float* bp = b + (size_t)(p / HW) * Cin + q * 4;
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(ap + k * 64), b4 = *reinterpret_cast<const f32x4*>(bp + k * 64);
                f32x4 t = xv[k] * a4 + b4;
                t.x = fmaxf(t.x, floor_); t.y = fmaxf(t.y, floor_); t.z = fmaxf(t.z, floor_); t.w = fmaxf(t.w, floor_);
                xv[k] = t;
            }
        }
        float acc[J];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < KC; ++k)
                s = fmaf(xv[k].w, wj[j][k].w, fmaf(xv[k].z, wj[j][k].z, fmaf(xv[k].y, wj[j][k].y, fmaf(xv[k].x, wj[j][k].x, s))));
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 16);
            acc[j] = s;
        }
        if (q < J) {
            float v = acc[0];
#pragma unroll
            for (int j = 1; j < J; ++j) v = (q == j) ? acc[j] : v;
            out[(size_t)p * J + q] = v + bias[q];
        }
    }
}

template <int J>
static void launch_project(const float* x, const float* w, const float* bias, const float* a, const float* b, float* out,
                           long long npix, int HW, int Cin, int in_relu, hipStream_t stream) {
    // 16 pixels per 256-thread block and iteration; enough blocks for ~7 waves per SIMD, each walking many pixels
    const long long want = (npix + 15) / 16;
    const int blocks = (int)(want < 256 * 7 ? want : 256 * 7);
#define GO(KC_) hipLaunchKernelGGL((logit_project_kernel<J, KC_>), dim3(blocks), dim3(256), 0, stream, x, w, bias, a, b, out, npix, HW, in_relu)
    switch (Cin / 64) {
        case 1: GO(1); break;
        case 2: GO(2); break;
        case 3: GO(3); break;
        default: GO(4); break;
    }
#undef GO
}

extern "C" int cpr_logit_project(const float* x, const float* w, const float* bias, const float* in_a,
                                 const float* in_b, float* out, int N, int HW, int Cin, int J, int in_relu,
                                 hipStream_t stream) {
    CPR_CHECK_ARG(x && w && bias && out && N > 0 && HW > 0 && J > 0);
    CPR_CHECK_ARG((in_a == nullptr) == (in_b == nullptr));
    if (J > 8 || Cin > 256 || Cin % 64 != 0) return CPR_ERR_UNSUPPORTED;   // caller falls back to the MFMA conv
    const long long npix = (long long)N * HW;
    switch (J) {
        case 1: launch_project<1>(x, w, bias, in_a, in_b, out, npix, HW, Cin, in_relu, stream); break;
        case 2: launch_project<2>(x, w, bias, in_a, in_b, out, npix, HW, Cin, in_relu, stream); break;
        case 3: launch_project<3>(x, w, bias, in_a, in_b, out, npix, HW, Cin, in_relu, stream); break;
        case 4: launch_project<4>(x, w, bias, in_a, in_b, out, npix, HW, Cin, in_relu, stream); break;
        case 5: launch_project<5>(x, w, bias, in_a, in_b, out, npix, HW, Cin, in_relu, stream); break;
        case 6: launch_project<6>(x, w, bias, in_a, in_b, out, npix, HW, Cin, in_relu, stream); break;
        case 7: launch_project<7>(x, w, bias, in_a, in_b, out, npix, HW, Cin, in_relu, stream); break;
        default: launch_project<8>(x, w, bias, in_a, in_b, out, npix, HW, Cin, in_relu, stream); break;
    }
    CPR_LAUNCH_STATUS();
}
