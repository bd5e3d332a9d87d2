// Microbenchmark (measurement tool, not product): how much non-MFMA work fits in the shadow of fp32 MFMAs on gfx950, with one
// or two waves per SIMD.  Every wave runs ITER x [8 x (v_mfma_f32_32x32x2_f32 + NV VALU fmas + NR ds_read_b32 + NW ds_write_b32
// + NB ds_read_b128 + NG global b128 loads per 8 MFMAs)], pieces pinned behind their MFMA with sched_barrier.  Prints MFMA-pipe
// cycles per MFMA per SIMD (64 = the matrix pipe is never idle).
//   hipcc --offload-arch=gfx950 -O3 -o tools/diag/mfma_shadow tools/diag/mfma_shadow.hip && tools/diag/mfma_shadow
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int WPS, int NV, int NR, int NW, int NB, int NG, int PK = 0>
__global__ __launch_bounds__(256 * WPS, 1) void shadow_kernel(float* out, const float* gsrc, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    const int tid = threadIdx.x;
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = tid * 1e-3f, b = 1.0f + tid * 1e-4f;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = tid + i;
    f32x2 v2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v2[i] = f32x2{(float)(tid + i), (float)(tid - i)};
    float rsum = 0.f;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f}, gsum = {0.f, 0.f, 0.f, 0.f};
    lds[tid] = a; lds[tid + 1024] = b;
    __syncthreads();
    const float* lp = lds + (tid & 1023);
    const f32x4* lp4 = reinterpret_cast<const f32x4*>(lds) + (tid & 1023);
    const f32x4* gp = reinterpret_cast<const f32x4*>(gsrc) + blockIdx.x * 1024 + tid;
    // every piece is an asm volatile (program order is kept, the compiler inserts no waitcnt of its own): loaded values are
    // not consumed inside the loop -- this measures ISSUE cost in the MFMA shadow, the counters' saturation is the only
    // back-pressure (vmcnt 63, lgkmcnt 15), as in a software-pipelined loop whose consumers sit a phase later
    float rr[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4 b4[2] = {bsum, bsum}, g4[4] = {gsum, gsum, gsum, gsum};
    const unsigned laddr = (unsigned)((tid & 1023) * 4), laddr4 = (unsigned)((tid & 1023) * 16);
    const float c1 = 1.0001f, c2 = 0.5f;
    const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const size_t gaddr = (size_t)gsrc;
    const i32x4 rs = {(int)(unsigned)gaddr, (int)(unsigned)(gaddr >> 32) & 0xffff, 0x7fffffff, 0x00020000};
    const int m0v = (int)(unsigned)(size_t)lds + 40960 + wave_s * 1024;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[q & 3]) : "v"(a), "v"(b));
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                if (PK == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(q * NV + k) & 7]) : "v"(c1), "v"(c2));
                else if (PK == 1) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1]" : "+v"(v2[(q * NV + k) & 7]) : "v"(v2[(q * NV + k + 3) & 7]));
                else asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[(q * NV + k) & 7]) : "v"(c2));
            }
#pragma unroll
            for (int k = 0; k < NR; ++k)
                asm volatile("ds_read_b32 %0, %1 offset:4096" : "+v"(rr[(q * NR + k) & 3]) : "v"(laddr));
#pragma unroll
            for (int k = 0; k < NW; ++k)
                asm volatile("ds_write_b32 %0, %1 offset:32768" ::"v"(laddr), "v"(v[k & 7]));
            if (NB > 0 && q < NB) asm volatile("ds_read_b128 %0, %1 offset:0" : "+v"(b4[q & 1]) : "v"(laddr4));
            if (NG > 0 && PK < 3 && q < NG) {
                const f32x4* gq = gp + ((it * 8 + q) & 63) * 4096;
                asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(g4[q & 3]) : "v"(gq));
            }
            if (NG > 0 && PK == 3 && wave_s < 4 && q < 2 * NG) {     // the same requests per SIMD, from one of its two waves
                const f32x4* gq = gp + ((it * 8 + q) & 63) * 4096;
                asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(g4[q & 3]) : "v"(gq));
            }
            if (NG > 0 && (PK == 4 || (PK == 5 && wave_s < 4)) && q < (PK == 5 ? 2 * NG : NG)) {   // global -> LDS directly
                const int soff = (((it * 8 + q) & 63) * 4096 + blockIdx.x * 1024) * 16;
                asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:0 lds"
                             :: "s"(m0v), "v"(tid * 16), "s"(rs), "s"(soff) : "memory");
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    rsum = rr[0] + rr[1] + rr[2] + rr[3];
    bsum = b4[0] + b4[1];
    gsum = g4[0] + g4[1] + g4[2] + g4[3];
    float s = rsum + bsum.x + bsum.y + bsum.z + bsum.w + gsum.x + gsum.y + gsum.z + gsum.w;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i] + v2[i].x + v2[i].y;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + tid] = s;
}

template <int WPS, int NV, int NR, int NW, int NB, int NG, int PK = 0>
static void run(float* out, const float* gsrc, double ghz) {
    const int iters = 4000, grid = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((shadow_kernel<WPS, NV, NR, NW, NB, NG, PK>), dim3(grid), dim3(256 * WPS), 0, 0, out, gsrc, 200);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((shadow_kernel<WPS, NV, NR, NW, NB, NG, PK>), dim3(grid), dim3(256 * WPS), 0, 0, out, gsrc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * 8 * WPS;
    const double cyc = ms * 1e-3 * ghz * 1e9 / mfma_per_simd;
    const double tf = (double)grid * 4 * mfma_per_simd * 4096.0 / (ms * 1e-3) / 1e12;
    printf("%s waves/SIMD %d  per MFMA: VALU %d  ds_read_b32 %d  ds_write_b32 %d  ds_read_b128 %d/8  global_b128 %d/8   %8.3f ms  %6.1f cycles/MFMA (64 ideal @%.2f GHz)  %6.1f TF\n",
           PK == 1 ? "[VALU = v_pk_add_f32]" : PK == 2 ? "[VALU = v_add_f32]" : PK == 3 ? "[loads from waves 0-3 only, 2x each]" : PK == 4 ? "[loads = LDS-DMA]" : PK == 5 ? "[LDS-DMA from waves 0-3 only, 2x each]" : "", WPS, NV, NR, NW, NB, NG, ms, cyc, ghz, tf);
}

int main() {
    float *out, *gsrc;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&gsrc, (size_t)(256 * 1024 + 64 * 4096 * 4 + 4096) * 16);
    hipMemset(gsrc, 0, (size_t)(256 * 1024 + 64 * 4096 * 4 + 4096) * 16);
    const double ghz = 2.4;
    for (int w = 0; w < 3; ++w) run<2, 0, 0, 0, 0, 0>(out, gsrc, ghz);   // clock ramp: the first lines are warm-up
    run<1, 0, 0, 0, 0, 0>(out, gsrc, ghz);
    run<2, 0, 0, 0, 0, 0>(out, gsrc, ghz);
    run<1, 2, 0, 0, 0, 0>(out, gsrc, ghz);
    run<1, 4, 0, 0, 0, 0>(out, gsrc, ghz);
    run<1, 8, 0, 0, 0, 0>(out, gsrc, ghz);
    run<2, 2, 0, 0, 0, 0>(out, gsrc, ghz);
    run<2, 4, 0, 0, 0, 0>(out, gsrc, ghz);
    run<2, 8, 0, 0, 0, 0>(out, gsrc, ghz);
    run<1, 0, 1, 0, 0, 0>(out, gsrc, ghz);
    run<1, 0, 2, 0, 0, 0>(out, gsrc, ghz);
    run<2, 0, 1, 0, 0, 0>(out, gsrc, ghz);
    run<2, 0, 2, 0, 0, 0>(out, gsrc, ghz);
    run<1, 0, 0, 1, 0, 0>(out, gsrc, ghz);
    run<2, 0, 0, 1, 0, 0>(out, gsrc, ghz);
    run<1, 0, 0, 0, 3, 0>(out, gsrc, ghz);
    run<2, 0, 0, 0, 3, 0>(out, gsrc, ghz);
    run<1, 0, 0, 0, 0, 2>(out, gsrc, ghz);
    run<2, 0, 0, 0, 0, 2>(out, gsrc, ghz);
    run<1, 0, 0, 0, 0, 4>(out, gsrc, ghz);
    // the F(4x4,3x3) one-wave-per-SIMD budget: ~2 VALU + 0.5 read + 0.5 write per MFMA, 1 b128 per 8, 2.5 global per 8
    run<1, 2, 1, 1, 1, 2>(out, gsrc, ghz);
    run<1, 3, 1, 1, 2, 3>(out, gsrc, ghz);
    run<2, 2, 1, 1, 1, 2>(out, gsrc, ghz);
    run<2, 3, 1, 1, 2, 3>(out, gsrc, ghz);
    // packed fp32 adds (two floats per lane) against scalar adds: is a packed instruction the price of one scalar instruction?
    run<2, 1, 0, 0, 0, 0, 2>(out, gsrc, ghz);
    run<2, 1, 0, 0, 0, 0, 1>(out, gsrc, ghz);
    run<2, 2, 0, 0, 0, 0, 2>(out, gsrc, ghz);
    run<2, 2, 0, 0, 0, 0, 1>(out, gsrc, ghz);
    run<2, 4, 0, 0, 0, 0, 2>(out, gsrc, ghz);
    run<2, 4, 0, 0, 0, 0, 1>(out, gsrc, ghz);
    // vector-memory issue cost: who pays for a load -- the issuing wave or the SIMD?  register loads against LDS-DMA
    run<2, 0, 0, 0, 0, 1>(out, gsrc, ghz);
    run<2, 0, 0, 0, 0, 2>(out, gsrc, ghz);
    run<2, 0, 0, 0, 0, 1, 3>(out, gsrc, ghz);
    run<2, 0, 0, 0, 0, 2, 3>(out, gsrc, ghz);
    run<2, 0, 0, 0, 0, 1, 4>(out, gsrc, ghz);
    run<2, 0, 0, 0, 0, 2, 4>(out, gsrc, ghz);
    run<2, 0, 0, 0, 0, 1, 5>(out, gsrc, ghz);
    run<2, 0, 0, 0, 0, 2, 5>(out, gsrc, ghz);
    run<2, 0, 0, 0, 0, 0>(out, gsrc, ghz);   // the bare loop again: drift of the box's clock over the run
    return 0;
}
