// Shared helpers for the gfx950 kernels of the CPR / P2P hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CPR_OK 0
#define CPR_ERR_ARG (-1001)   // bad shape / pointer argument
#define CPR_ERR_UNSUPPORTED (-1002)

// cpr_conv2d_fwd `flags` (mirrors include/cpr_hip.h)
#define CPR_CONV_RELU 1       // ReLU in the epilogue
#define CPR_CONV_OUT_BF16 2   // write bf16 (the fp32 stem hands a bf16 map to the bf16 layers)
#define CPR_CONV_RES_MASK 4   // `residual` is a ReLU mask source: out = residual > 0 ? v : 0 (backward of a fused ReLU)
#define CPR_CONV_COLSUM 8     // gn_part receives per-tile per-channel sums that are only reduced over the whole tensor

// cpr_conv3x3_wino_fwd `layout`: channel-blocked [N][C/8][H][W][8] tensors (mirrors include/cpr_hip.h)
#define CPR_WINO_IN_B8 1
#define CPR_WINO_OUT_B8 2

#define CPR_CHECK_ARG(cond) \
    do {                    \
        if (!(cond)) return CPR_ERR_ARG; \
    } while (0)

// every launcher ends with this: returns 0 or -hipError
#define CPR_LAUNCH_STATUS()                          \
    do {                                             \
        hipError_t e_ = hipGetLastError();           \
        return e_ == hipSuccess ? CPR_OK : -(int)e_; \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}

// torch's CPU sigmoid, bit for bit: ATen's vectorised kernel is 1 / (1 + Sleef_expf_u10(-x)); sleef_expf_u10() replays the
// Sleef routine operation for operation (0 mismatches against torch.sigmoid on 2^20 random inputs; the digest is the same
// on the Xeon build host and the EPYC GPU-box host, profiles/round2_log_probe.txt).  Used wherever a probability is
// compared against a threshold or decides a tie (Hungarian cost, PointRefiner selections).
__device__ __forceinline__ float sleef_expf_u10(float d) {
    const float q = rintf(__fmul_rn(d, 1.442695040888963407359924681001892137426645954152985934135449406931f));
    float s = __fmaf_rn(q, -0.693145751953125f, d);
    s = __fmaf_rn(q, -1.428606765330187045e-06f, s);
    float u = 0.000198527617612853646278381f;
    u = __fmaf_rn(u, s, 0.00139304355252534151077271f);
    u = __fmaf_rn(u, s, 0.00833336077630519866943359f);
    u = __fmaf_rn(u, s, 0.0416664853692054748535156f);
    u = __fmaf_rn(u, s, 0.166666671633720397949219f);
    u = __fmaf_rn(u, s, 0.5f);
    u = __fadd_rn(1.0f, __fmaf_rn(__fmul_rn(s, s), u, s));
    const int qi = (int)q, h = qi >> 1;                      // ldexp2kf: two exact power-of-two scalings
    u = __fmul_rn(u, __int_as_float((h + 127) << 23));
    u = __fmul_rn(u, __int_as_float((qi - h + 127) << 23));
    if (d < -104.f) u = 0.f;
    if (d > 104.f) u = INFINITY;
    return u;
}
__device__ __forceinline__ float sigmoid_torch_cpu(float x) { return __fdiv_rn(1.f, __fadd_rn(1.f, sleef_expf_u10(-x))); }

// torch.cdist(p=2) for > 25 rows (what every call site on this path has) is the matmul form
// [-2x, -2y, |p|^2, 1] . [cx, cy, 1, |c|^2] evaluated by MKL sgemm as a k-ordered fp32 FMA chain, then clamp_min(0), sqrt.
// d2_chain() is that chain operation for operation (the files that use it are compiled with -ffp-contract=off).
__device__ __forceinline__ float sq_norm(float x, float y) {
    // x.pow(2).sum(-1): each square rounded, then one add (cpr_head.py:277 via torch.cdist)
    return __fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y));
}
__device__ __forceinline__ float d2_chain(float px, float py, float pn, float cx, float cy, float cn) {
    float acc = __fmul_rn(-2.f * px, cx);
    acc = __fmaf_rn(-2.f * py, cy, acc);
    acc = __fadd_rn(pn, acc);
    acc = __fadd_rn(acc, cn);
    return acc;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
// The conv kernels address activations through buffer descriptors with 32-bit byte offsets (out-of-range offsets ARE the zero
// padding), so one launch takes tensors below 2 GiB; the launchers walk a larger batch in balanced chunks of whole images
// (every per-image quantity -- tile, region and GroupNorm-slot indexing -- is unchanged, so the results are bit-identical to an
// unsplit launch).  Returns the images per launch (a multiple of `align` unless everything fits in one), or 0 when a single
// image (or `align` of them) already exceeds the range.
static inline int cpr_images_per_launch(int N, long long bytes_per_image, int align = 1) {
    const long long lim = (1ll << 31) - 1;
    if (bytes_per_image <= 0 || bytes_per_image > lim) return 0;
    const long long fit = lim / bytes_per_image;
    if (fit >= N) return N;
    const long long launches = (N + fit - 1) / fit;
    long long n = (N + launches - 1) / launches;          // balanced
    n = (n + align - 1) / align * align;
    if (n > fit) n = fit / align * align;
    return (int)n;
}
static inline long long cpr_max2(long long a, long long b) { return a > b ? a : b; }
static inline long long cdivll(long long a, long long b) { return (a + b - 1) / b; }
