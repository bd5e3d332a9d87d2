// Probe of ds_read_b64_tr_b16 (gfx950 LDS transpose read): what a lane receives for the addresses the lanes of its 16-lane group
// supply.  LDS holds L[i] = i (u16).  Lane i of group g supplies the address of 4 contiguous elements: row (i >> 2) of a
// [4][16] block with a row stride of `stride` elements, columns 4 (i & 3) .. + 3.  Model under test: lane i receives column i of
// the block, rows 0..3 (element j = row j).
//   hipcc --offload-arch=gfx950 -O2 tr_read_probe.hip -o tr_read_probe && ./tr_read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__global__ void probe(unsigned short* out, int stride) {
    __shared__ unsigned short L[16384];
    for (int i = threadIdx.x; i < 16384; i += 64) L[i] = (unsigned short)i;
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, i = lane & 15;
    const unsigned addr = (unsigned)(size_t)L + (unsigned)((g * 4096 + (i >> 2) * stride + (i & 3) * 4) * 2);
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[lane * 4 + 0] = (unsigned short)(v[0] & 0xffff);
    out[lane * 4 + 1] = (unsigned short)(v[0] >> 16);
    out[lane * 4 + 2] = (unsigned short)(v[1] & 0xffff);
    out[lane * 4 + 3] = (unsigned short)(v[1] >> 16);
}

int main() {
    unsigned short* d;
    unsigned short h[256];
    hipMalloc(&d, sizeof(h));
    int bad_total = 0;
    for (int stride : {16, 64, 256, 264}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, stride);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 4; ++j) {
                const int want = (lane >> 4) * 4096 + j * stride + (lane & 15);
                if (h[lane * 4 + j] != want) ++bad;
            }
        printf("stride %d elements: %d of 256 values differ from the model (lane i <- column i, element j <- row j)\n", stride, bad);
        if (bad) {
            for (int lane = 0; lane < 20; ++lane)
                printf("  lane %2d: %5d %5d %5d %5d\n", lane, h[lane * 4], h[lane * 4 + 1], h[lane * 4 + 2], h[lane * 4 + 3]);
        }
        bad_total += bad;
    }
    printf(bad_total ? "MODEL WRONG\n" : "MODEL HOLDS\n");
    return 0;
}
