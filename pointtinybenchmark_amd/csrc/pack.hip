// Parameter re-layout kernels: run once per optimisation step per conv (the SGD kernel updates the OIHW master weights
// through raw pointers, the conv kernels read packed copies).
#include "common.h"

// OIHW fp32 master -> the implicit-GEMM layout [rows][KH][KW][cols'] (row stride Kpad floats, zero padded).
//   forward pack  (transpose = 0): rows = O, cols = I,  out[o][kh][kw][i]  = w[o][i][kh][kw]
//   dgrad pack    (transpose = 1): rows = I, cols = O,  out[i][kh][kw][o]  = w[o][i][KH-1-kh][KW-1-kw] * scale[o]
// colsp = cols padded (4 for the 3-channel stem, else cols); one thread per output element.
// (the body is shared with the multi-tensor launch: `block` of `nblocks` workgroups walks the pack)
__device__ __forceinline__ void pack_weights_body(const float* __restrict__ w, const float* __restrict__ scale, float* __restrict__ out,
                                                  int O, int I, int KH, int KW, int colsp, int Kpad, int transpose, int block, int nblocks) {
    const int rows = transpose ? I : O, cols = transpose ? O : I;
    const long long total = (long long)rows * Kpad;
    for (long long idx = block * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)nblocks * blockDim.x) {
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
__global__ void pack_weights_kernel(const float* __restrict__ w, const float* __restrict__ scale, float* __restrict__ out,
                                    int O, int I, int KH, int KW, int colsp, int Kpad, int transpose) {
    pack_weights_body(w, scale, out, O, I, KH, KW, colsp, Kpad, transpose, blockIdx.x, gridDim.x);
}
// multi-tensor form (round 6, see the bf16 one below): jobs in ascending block0
struct Pack32Job {      // 64 bytes
    const float* w; const float* scale; float* out; int nblocks, pad;
    int O, I, KH, KW, colsp, Kpad, transpose, block0;
};
__global__ void pack_weights_multi_kernel(const Pack32Job* __restrict__ jobs, int n) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const Pack32Job j = jobs[lo];
    pack_weights_body(j.w, j.scale, j.out, j.O, j.I, j.KH, j.KW, j.colsp, j.Kpad, j.transpose, (int)blockIdx.x - j.block0, j.nblocks);
}
extern "C" int cpr_pack_weights_multi(const void* jobs_dev, int n, int total_blocks, hipStream_t stream) {
    CPR_CHECK_ARG(jobs_dev && n > 0 && total_blocks > 0);
    hipLaunchKernelGGL(pack_weights_multi_kernel, dim3(total_blocks), dim3(256), 0, stream, (const Pack32Job*)jobs_dev, n);
    CPR_LAUNCH_STATUS();
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

// The same two packs for the bf16 kernels (round 5): out[rows][KH][KW][cols] bf16 (no padding: cols % 64 == 0), rounded to nearest
// even from the fp32 value (scale multiplied in fp32 first) -- bit for bit what `w.permute(..).reshape(..).to(torch.bfloat16)`
// gave in three torch launches per layer -- and, when frag != NULL (rows % 64 == 0, K % 16 == 0), the fragment-order image of
// conv_bf16_dma_kernel<4, 2, 4, true> by the same thread: frag[rows / 64][K / 16][j][h][l][8] = out[64 g + 2 l + j][16 ks + 8 h ..].
// The mixed-precision step re-packs every bf16 layer after each optimizer update: ~330 tiny launches per configs[4] step before.
typedef __attribute__((ext_vector_type(2))) __bf16 pk_bf16x2_t;
// (the body is shared with the multi-tensor launch below: `block` of `nblocks` workgroups walks the pack)
__device__ __forceinline__ void pack_weights_bf16_body(const float* __restrict__ w, const float* __restrict__ scale,
                                                       unsigned short* __restrict__ out, unsigned short* __restrict__ frag, int O, int I,
                                                       int KH, int KW, int transpose, int block, int nblocks) {
    const int rows = transpose ? I : O, cols = transpose ? O : I;
    const int K = KH * KW * cols, KS = K >> 4;
    const long long total = (long long)rows * (K >> 1);          // two consecutive k per thread (cols is even)
    for (long long idx = block * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)nblocks * blockDim.x) {
        const int r = (int)(idx / (K >> 1));
        const int k = (int)(idx - (long long)r * (K >> 1)) * 2;
        float v[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int c = (k + u) % cols;
            const int t = (k + u) / cols;
            const int kw = t % KW, kh = t / KW;
            v[u] = transpose ? w[(((size_t)c * I + r) * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw)] * (scale ? scale[c] : 1.f)
                             : w[(((size_t)r * I + c) * KH + kh) * KW + kw];
        }
        const pk_bf16x2_t p = {(__bf16)v[0], (__bf16)v[1]};
        const unsigned bits = __builtin_bit_cast(unsigned, p);
        *reinterpret_cast<unsigned*>(out + (size_t)r * K + k) = bits;
        if (frag) {
            const int g = r >> 6, l = (r & 63) >> 1, j = r & 1;
            const int ks = k >> 4, h = (k >> 3) & 1, e = k & 7;
            *reinterpret_cast<unsigned*>(frag + ((((((size_t)g * KS + ks) * 2 + j) * 2 + h) * 32 + l) << 3) + e) = bits;
        }
    }
}
__global__ void pack_weights_bf16_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                         unsigned short* __restrict__ out, unsigned short* __restrict__ frag, int O, int I, int KH,
                                         int KW, int transpose) {
    pack_weights_bf16_body(w, scale, out, frag, O, I, KH, KW, transpose, blockIdx.x, gridDim.x);
}

// Multi-tensor forms (round 6): the mixed-precision step re-folds every BatchNorm and re-packs every bf16 layer after each
// optimizer update -- ~320 five-microsecond launches per configs[4] step, and with the host-side work around them 4.3 ms of a
// 37 ms step (tools/diag/stale_packs_ab.sh: the step with stale packs, profiles/round6_stale_packs_ab.txt).  One launch per kind
// over a table of jobs that lives on the device (the pointers are stable: parameters are views of the trainer's flat buffer, the
// outputs are refreshed in place).  Bit for bit the single-tensor kernels' results (the same bodies).
struct PackJob {        // 64 bytes; layers._PackCache builds the table
    const float* w; const float* scale; unsigned short* out; unsigned short* frag;
    int O, I, KH, KW, transpose, block0, nblocks, pad;
};
__global__ void pack_weights_bf16_multi_kernel(const PackJob* __restrict__ jobs, int n) {
    int lo = 0, hi = n - 1;                       // the job whose block range holds blockIdx.x (block0 ascending)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const PackJob j = jobs[lo];
    pack_weights_bf16_body(j.w, j.scale, j.out, j.frag, j.O, j.I, j.KH, j.KW, j.transpose, (int)blockIdx.x - j.block0, j.nblocks);
}
extern "C" int cpr_pack_weights_bf16_multi(const void* jobs_dev, int n, int total_blocks, hipStream_t stream) {
    CPR_CHECK_ARG(jobs_dev && n > 0 && total_blocks > 0);
    hipLaunchKernelGGL(pack_weights_bf16_multi_kernel, dim3(total_blocks), dim3(256), 0, stream, (const PackJob*)jobs_dev, n);
    CPR_LAUNCH_STATUS();
}
extern "C" int cpr_pack_weights_bf16(const float* w, const float* scale, void* out, void* frag, int O, int I, int KH, int KW,
                                     int transpose, hipStream_t stream) {
    CPR_CHECK_ARG(w && out && O > 0 && I > 0 && KH > 0 && KW > 0);
    const int rows = transpose ? I : O, cols = transpose ? O : I;
    CPR_CHECK_ARG(cols % 2 == 0 && (!frag || (rows % 64 == 0 && (KH * KW * cols) % 16 == 0)));
    const long long total = (long long)rows * (KH * KW * cols / 2);
    const int grid = (int)(cdivll(total, 256) < 4096 ? cdivll(total, 256) : 4096);
    hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3(grid), dim3(256), 0, stream, w, scale, (unsigned short*)out,
                       (unsigned short*)frag, O, I, KH, KW, transpose);
    CPR_LAUNCH_STATUS();
}

// One wave that does nothing for `ticks` of the 100 MHz wall clock: the probe training.BackwardEngine uses to find a second stream
// that sits on a different HARDWARE queue than the caller's (two of these on one queue take twice as long as on two).
__global__ void spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
extern "C" int cpr_spin(long long ticks, hipStream_t stream) {
    CPR_CHECK_ARG(ticks > 0 && ticks <= 100000000ll);        // at most a second
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, stream, ticks);
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
struct FoldJob {        // 64 bytes; scale / shift / inv may each be null
    const float *gamma, *beta, *mean, *var; float *scale, *shift, *inv; int C; float eps;
};
__global__ void bn_fold_multi_kernel(const FoldJob* __restrict__ jobs) {
    const FoldJob j = jobs[blockIdx.y];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= j.C) return;
    const float sd = __fsqrt_rn(__fadd_rn(j.var[c], j.eps));
    const float sc = __fdiv_rn(j.gamma[c], sd);
    if (j.scale) j.scale[c] = sc;
    if (j.shift) j.shift[c] = __fsub_rn(j.beta[c], __fmul_rn(j.mean[c], sc));
    if (j.inv) j.inv[c] = __fdiv_rn(1.f, sd);
}
extern "C" int cpr_bn_fold_multi(const void* jobs_dev, int n, int max_c, hipStream_t stream) {
    CPR_CHECK_ARG(jobs_dev && n > 0 && max_c > 0);
    hipLaunchKernelGGL(bn_fold_multi_kernel, dim3(cdiv(max_c, 256), n), dim3(256), 0, stream, (const FoldJob*)jobs_dev);
    CPR_LAUNCH_STATUS();
}
extern "C" int cpr_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                           float* scale, float* shift, float* inv_sigma, int C, hipStream_t stream) {
    CPR_CHECK_ARG(gamma && beta && mean && var && scale && shift && C > 0);
    hipLaunchKernelGGL(bn_fold_kernel, dim3(cdiv(C, 256)), dim3(256), 0, stream, gamma, beta, mean, var, eps, scale, shift,
                       inv_sigma, C);
    CPR_LAUNCH_STATUS();
}
