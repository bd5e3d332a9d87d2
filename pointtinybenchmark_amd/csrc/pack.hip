// Parameter re-layout kernels: run once per optimisation step per conv (the SGD kernel updates the OIHW master weights
// through raw pointers, the conv kernels read packed copies).
#include "common.h"

// OIHW fp32 master -> the implicit-GEMM layout [rows][KH][KW][cols'] (row stride Kpad floats, zero padded).
//   forward pack  (transpose = 0): rows = O, cols = I,  out[o][kh][kw][i]  = w[o][i][kh][kw]
//   dgrad pack    (transpose = 1): rows = I, cols = O,  out[i][kh][kw][o]  = w[o][i][KH-1-kh][KW-1-kw] * scale[o]
// colsp = cols padded (4 for the 3-channel stem, else cols); one thread per output element.
__global__ void pack_weights_kernel(const float* __restrict__ w, const float* __restrict__ scale, float* __restrict__ out,
                                    int O, int I, int KH, int KW, int colsp, int Kpad, int transpose) {
    const int rows = transpose ? I : O, cols = transpose ? O : I;
    const long long total = (long long)rows * Kpad;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(idx / Kpad);
        const int k = (int)(idx - (long long)r * Kpad);
        float v = 0.f;
        if (k < KH * KW * colsp) {
            const int c = k % colsp;
            const int t = k / colsp;
            const int kw = t % KW, kh = t / KW;
            if (c < cols) {
                if (transpose) v = w[(((size_t)c * I + r) * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw)] * (scale ? scale[c] : 1.f);
                else v = w[(((size_t)r * I + c) * KH + kh) * KW + kw];
            }
        }
        out[idx] = v;
    }
}
extern "C" int cpr_pack_weights(const float* w, const float* scale, float* out, int O, int I, int KH, int KW, int colsp,
                                int Kpad, int transpose, hipStream_t stream) {
    CPR_CHECK_ARG(w && out && O > 0 && I > 0 && KH > 0 && KW > 0 && Kpad >= KH * KW * colsp);
    CPR_CHECK_ARG(colsp >= (transpose ? O : I));
    const long long total = (long long)(transpose ? I : O) * Kpad;
    const int grid = (int)(cdivll(total, 256) < 4096 ? cdivll(total, 256) : 4096);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(grid), dim3(256), 0, stream, w, scale, out, O, I, KH, KW, colsp, Kpad,
                       transpose);
    CPR_LAUNCH_STATUS();
}

// eval-mode BatchNorm folded for the conv epilogue: inv_sigma = 1/sqrt(var+eps), scale = gamma*inv_sigma,
// shift = beta - mean*scale (same operation order as the torch expression it replaces: layers.folded_bn)
__global__ void bn_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ mean, const float* __restrict__ var, float eps,
                               float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ inv_sigma, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sd = __fsqrt_rn(__fadd_rn(var[c], eps));
    const float sc = __fdiv_rn(gamma[c], sd);
    scale[c] = sc;
    shift[c] = __fsub_rn(beta[c], __fmul_rn(mean[c], sc));
    if (inv_sigma) inv_sigma[c] = __fdiv_rn(1.f, sd);
}
extern "C" int cpr_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                           float* scale, float* shift, float* inv_sigma, int C, hipStream_t stream) {
    CPR_CHECK_ARG(gamma && beta && mean && var && scale && shift && C > 0);
    hipLaunchKernelGGL(bn_fold_kernel, dim3(cdiv(C, 256)), dim3(256), 0, stream, gamma, beta, mean, var, eps, scale, shift,
                       inv_sigma, C);
    CPR_LAUNCH_STATUS();
}
