// Shared helpers for the gfx950 kernels of the CPR / P2P hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CPR_OK 0
#define CPR_ERR_ARG (-1001)   // bad shape / pointer argument
#define CPR_ERR_UNSUPPORTED (-1002)

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

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline long long cdivll(long long a, long long b) { return (a + b - 1) / b; }
