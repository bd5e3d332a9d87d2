// Microbenchmark: do LDS-DMA requests and bf16 MFMAs overlap on a CU?  (Round 6: every schedule of the 256 x 256 bf16 conv tile
// measured MFMA time + request time; tools/diag/dma_stream.hip shows the request stream ALONE runs at the texture path's 64 B/clk.)
// One workgroup of 8 waves per CU (two per SIMD, as the conv kernel); per mode each wave is an MFMA wave (chains of
// v_mfma_f32_32x32x16_bf16 on 8 accumulators, nothing else), a DMA wave (conv-like LDS-DMA gather, 16 requests in flight) or idle:
//   mode 0  waves 0-3 MFMA, 4-7 idle            (one MFMA wave per SIMD: the pipe's own rate)
//   mode 1  all 8 MFMA                          (two per SIMD)
//   mode 2  waves 0-3 MFMA, 4-7 DMA             (every SIMD: one MFMA wave + one DMA wave)
//   mode 3  waves 0-3 MFMA, wave 4 DMA          (one DMA wave on one SIMD)
//   mode 4  all 8 MFMA, each issuing one request per 4 MFMAs   (the conv kernels' ratio: 8 requests per 32 MFMAs)
//   mode 5  waves 0-3 MFMA, 4-7 ds_read_b128 streams (12 reads per 16 MFMAs of the partner, no DMA)
//   mode 6  waves 0-3 MFMA, 4-7 DMA + ds_read_b128
//   mode 7 / 8  as 5 / 6 with twelve 4-wide v_add per iteration in the memory wave (any vector ALU work there waits behind the partner's MFMAs)
// Each wave reports its own clock (s_memtime) for its fixed amount of work; the host prints cycles per MFMA per wave and cycles per
// request per CU.  `long_dma` = the DMA waves outlast the MFMA waves (MFMA rate under load) or the other way round (DMA rate under load).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x4v __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512, 2) void mix_kernel(const char* buf, unsigned window, int mfma_iters, int dma_iters, unsigned long long* out) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[131072];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t addr = (size_t)buf;
    const i32x4v rs = {(int)(unsigned)addr, (int)(unsigned)(addr >> 32) & 0xffff, (int)window, 0x00020000};
    const int lds0 = (int)(unsigned)(size_t)smem + wave * 1024;
    const bool is_mfma = MODE == 1 || MODE == 4 || wave < 4;
    const bool is_dma = (MODE == 2 && wave >= 4) || (MODE == 3 && wave == 4) || ((MODE == 6 || MODE == 8) && wave >= 4);
    const bool is_rd = MODE >= 5 && wave >= 4;
    unsigned voff[4];
#pragma unroll
    for (int z = 0; z < 4; ++z) voff[z] = ((blockIdx.x * 256 + wave * 32 + z * 8 + (lane >> 3)) * 512 + (lane & 7) * 16) % window;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), t1 = 0;
    if (is_mfma) {
        f32x16 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        bf16x8 a, b;
#pragma unroll
        for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (lane + e)); b[e] = (__bf16)(0.002f * (lane - e)); }
        for (int n = 0; n < mfma_iters; ++n) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                acc[q & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[q & 7], 0, 0, 0);
                if (MODE == 4 && (q & 3) == 1) {
                    asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:0 lds"
                                 :: "s"(lds0 + (q >> 2) * 8192 + (n & 1) * 65536), "v"(voff[q >> 2]), "s"(rs), "s"((n % 36) * 512) : "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (MODE == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += acc[i][0];
        t1 = __builtin_amdgcn_s_memtime();
        if (s == 12345.f) out[0] = 1;
    } else if (is_dma || is_rd) {
        f32x4 sink = {0.f, 0.f, 0.f, 0.f};
        const int rd_addr = (int)(unsigned)(size_t)smem + (lane & 31) * 128 + (((lane >> 5) ^ ((lane >> 1) & 7)) * 16);
        for (int n = 0; n < dma_iters; ++n) {
            if (is_dma) {
#pragma unroll
                for (int z = 0; z < 4; ++z)
                    asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:0 lds"
                                 :: "s"(lds0 + z * 8192 + (n & 1) * 65536), "v"(voff[z]), "s"(rs), "s"((n % 36) * 512) : "memory");
                asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            }
            if (is_rd) {
#pragma unroll
                for (int z = 0; z < 12; ++z) {
                    // NO vector ALU work in this wave: the address is loop-invariant, the data is never touched (MODE 7 / 8 add the
                    // twelve v_add the first version of this file had: they, not the reads, are what starves)
                    f32x4 v;
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(rd_addr), "n"(0) : "memory");
                    if (MODE >= 7) sink += v;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        t1 = __builtin_amdgcn_s_memtime();
        if (sink[0] == 12345.f) out[1] = 1;
    } else t1 = t0;
    if (lane == 0) out[2 + blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE> static void run(const char* buf, unsigned window, unsigned long long* out, int mfma_iters, int dma_iters, const char* what) {
    std::vector<unsigned long long> h(2 + 256 * 8);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((mix_kernel<MODE>), dim3(256), dim3(512), 0, 0, buf, window, mfma_iters, dma_iters, out);
        hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    double m = 0, d = 0; int nm = 0, nd = 0;
    for (int b = 0; b < 256; ++b)
        for (int w = 0; w < 8; ++w) {
            const bool is_mfma = MODE == 1 || MODE == 4 || w < 4;
            const bool other = (MODE == 2 && w >= 4) || (MODE == 3 && w == 4) || (MODE >= 5 && w >= 4);
            if (is_mfma) { m += (double)h[2 + b * 8 + w]; ++nm; }
            else if (other) { d += (double)h[2 + b * 8 + w]; ++nd; }
        }
    printf("mode %d %-34s: %6.1f cycles per MFMA per wave", MODE, what, m / nm / (mfma_iters * 16.0));
    if (nd) {
        const int dma_waves = MODE == 3 ? 1 : 4;
        printf(" | memory waves: %7.1f cycles per iteration per wave (4 requests and / or 12 reads) = %5.1f cycles per request per CU", d / nd / dma_iters,
               d / nd / dma_iters / 4.0 / dma_waves);
    }
    if (MODE == 4) printf(" (with 1 request per 4 MFMAs)");
    printf("\n");
}

int main() {
    const size_t bytes = 256ull << 20;
    char* buf; unsigned long long* out;
    hipMalloc(&buf, bytes); hipMalloc(&out, (2 + 256 * 8) * 8);
    hipMemset(buf, 1, bytes);
    const unsigned w = 128u << 20;
    printf("MFMA rate under load (memory waves outlast the MFMA waves)\n");
    run<0>(buf, w, out, 4000, 0, "one MFMA wave per SIMD");
    run<1>(buf, w, out, 4000, 0, "two MFMA waves per SIMD");
    run<2>(buf, w, out, 2000, 40000, "MFMA + DMA wave on every SIMD");
    run<3>(buf, w, out, 2000, 100000, "MFMA x4 + ONE DMA wave");
    run<4>(buf, w, out, 4000, 0, "8 MFMA waves issuing requests");
    run<5>(buf, w, out, 2000, 40000, "MFMA + ds_read wave on every SIMD");
    run<6>(buf, w, out, 2000, 30000, "MFMA + DMA + ds_read wave");
    printf("memory rate under load (the MFMA waves outlast the memory waves)\n");
    run<2>(buf, w, out, 40000, 4000, "MFMA + DMA wave on every SIMD");
    run<3>(buf, w, out, 40000, 8000, "MFMA x4 + ONE DMA wave");
    run<5>(buf, w, out, 40000, 4000, "MFMA + ds_read wave on every SIMD");
    run<6>(buf, w, out, 40000, 3000, "MFMA + DMA + ds_read wave");
    run<7>(buf, w, out, 40000, 1000, "MFMA + ds_read + 12 v_add wave");
    run<8>(buf, w, out, 40000, 1000, "MFMA + DMA + ds_read + 12 v_add");
    printf("(32 cycles per MFMA = the pipe's rate; 16 cycles per request per CU = the texture path's 64 B/clk; the conv tile needs 64 MFMAs per SIMD and 64 requests + 192 reads per CU per K chunk)\n");
    return 0;
}
