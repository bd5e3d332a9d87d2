// Microbenchmark: how fast can a CU pull L2-resident data into LDS with LDS-DMA (buffer_load_dwordx4 ... lds), with NO other work?
// The bf16 256 x 256 conv tile needs 64 KB per K chunk per CU against 2048 MFMA cycles (32 B/clk/CU); round 6 measured that every
// schedule of that tile ends up at MFMA time + request time.  This prices the request stream alone:
//   pattern 0  conv-like gather: a request = 8 pixel rows of 128 bytes at a 512-byte pixel stride (256-channel NHWC bf16 map, one
//              64-channel chunk), 9 taps walked per channel chunk, all workgroups inside an L2-resident window
//   pattern 1  the same bytes as contiguous 1 KB requests (what a channel-blocked [C/64][H][W][64] map would give)
//   pattern 2  every request the same KB (vector-L1 hits: the texture path + LDS write alone)
//   pattern 3  weight-like gather: 8 rows of 128 bytes at a 4608-byte row stride (K = 2304)
// and DEPTH = requests a wave keeps in flight (counted vmcnt).  8 waves per workgroup, one workgroup per CU, as the conv kernel.
// Build: hipcc --offload-arch=gfx950 -O3 -o dma_stream dma_stream.hip ; run: ./dma_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x4v __attribute__((ext_vector_type(4)));

template <int PATTERN, int DEPTH>
__global__ __launch_bounds__(512, 2) void stream_kernel(const char* buf, unsigned window, int chunks, unsigned* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[131072];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t addr = (size_t)buf;
    const i32x4v rs = {(int)(unsigned)addr, (int)(unsigned)(addr >> 32) & 0xffff, (int)window, 0x00020000};
    const int lds0 = (int)(unsigned)(size_t)smem + wave * 1024;
    const unsigned tile = blockIdx.x;
    // per-lane byte offset of the wave's request z of a chunk (before the chunk's scalar shift)
    unsigned voff[8];
#pragma unroll
    for (int z = 0; z < 8; ++z) {
        if (PATTERN == 0) voff[z] = ((tile * 256 + wave * 32 + (z & 3) * 8 + (lane >> 3)) * 512 + (lane & 7) * 16) % window;   // 4 pixel + 4 "weight" pieces
        else if (PATTERN == 1) voff[z] = ((tile * 256 + wave * 32 + (z & 3) * 8) * 128 + lane * 16) % window;
        else if (PATTERN == 2) voff[z] = lane * 16;
        else voff[z] = ((wave * 32 + (z & 3) * 8 + (lane >> 3)) * 4608 + (lane & 7) * 16) % window;
    }
    for (int n = 0; n < chunks; ++n) {
        const int tap = n % 9, cc = (n / 9) & 3;
        int soff;
        if (PATTERN == 0) soff = (((tap / 3) * 160 + tap % 3) * 512 + cc * 128);
        else if (PATTERN == 1) soff = (((tap / 3) * 160 + tap % 3) * 128 + cc * 1024 * 1024);
        else if (PATTERN == 2) soff = 0;
        else soff = (tap * 256 + cc * 64) * 2;
        const int stage = (n & 1) * 65536;
#pragma unroll
        for (int z = 0; z < 8; ++z) {
            asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:0 lds"
                         :: "s"(lds0 + stage + z * 8192), "v"(voff[z]), "s"(rs), "s"(soff) : "memory");
            if (DEPTH < 8 && ((z + 1) % (DEPTH > 0 ? DEPTH : 1)) == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DEPTH > 0 ? DEPTH - 1 : 0) : "memory");
        }
        if (DEPTH >= 8) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DEPTH - 8 < 63 ? DEPTH - 8 : 63) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) sink[blockIdx.x] = smem[0];
}

template <int P, int D> static double run(const char* buf, unsigned window, unsigned* sink, int chunks) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((stream_kernel<P, D>), dim3(256), dim3(512), 0, 0, buf, window, chunks, sink);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((stream_kernel<P, D>), dim3(256), dim3(512), 0, 0, buf, window, chunks, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / 5;
}

int main() {
    const size_t bytes = 512ull << 20;
    char* buf; unsigned* sink;
    hipMalloc(&buf, bytes); hipMalloc(&sink, 4096);
    hipMemset(buf, 1, bytes);
    const int chunks = 900;            // the head layer: 25 tiles x 36 chunks per CU
    const double kb = 256.0 * chunks * 64;      // KB moved per launch
    const char* names[] = {"conv-like gather (8 x 128 B at 512 B stride)", "contiguous 1 KB requests", "same KB (L1 hits)", "weight-like gather (4608 B stride)"};
    const unsigned windows[] = {2u << 20, 16u << 20, 128u << 20, 480u << 20};
    for (unsigned w : windows) {
        printf("window %u MB (chunks %d, 64 KB per chunk per workgroup, 256 workgroups)\n", w >> 20, chunks);
#define ROW(P) { const double t4 = run<P, 4>(buf, w, sink, chunks), t8 = run<P, 8>(buf, w, sink, chunks), t16 = run<P, 16>(buf, w, sink, chunks), t32 = run<P, 32>(buf, w, sink, chunks), t64 = run<P, 64>(buf, w, sink, chunks); \
        printf("  %-46s depth 4: %.3f ms %.2f TB/s | 8: %.3f ms %.2f | 16: %.3f ms %.2f | 32: %.3f ms %.2f | 64: %.3f ms %.2f TB/s\n", names[P], \
               t4, kb * 1024 / t4 / 1e9, t8, kb * 1024 / t8 / 1e9, t16, kb * 1024 / t16 / 1e9, t32, kb * 1024 / t32 / 1e9, t64, kb * 1024 / t64 / 1e9); }
        ROW(0) ROW(1) ROW(2) ROW(3)
    }
    printf("(the conv tile needs 64 KB per 2048 MFMA cycles per CU = 32 B/clk/CU = 16.4 TB/s at 2.0 GHz over 256 CUs)\n");
    return 0;
}
