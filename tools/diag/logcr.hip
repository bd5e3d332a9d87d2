// diagnostic: is (float)log((double)x) on the device the correctly rounded fp32 log?
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k(const float* x, float* a, float* b, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    a[i] = (float)log((double)x[i]);
    b[i] = logf(x[i]);
}
int main() {
    const int n = 1 << 22;
    std::vector<float> x(n), a(n), b(n);
    srand(1);
    for (int i = 0; i < n; ++i) {
        // values like p+eps and 1-p+eps: mostly in (0.5, 1] and (1e-4, 0.5)
        float u = (float)rand() / RAND_MAX;
        x[i] = (i & 1) ? 1.0f - u * u * 0.5f : u * u * 0.5f + 1e-12f;
    }
    float *dx, *da, *db;
    hipMalloc(&dx, n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, da, db, n);
    hipMemcpy(a.data(), da, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), db, n * 4, hipMemcpyDeviceToHost);
    int ma = 0, mb = 0, mab = 0;
    for (int i = 0; i < n; ++i) {
        float r = (float)std::log((double)x[i]);
        ma += (a[i] != r); mb += (b[i] != r); mab += (a[i] != b[i]);
    }
    printf("n=%d  (float)log((double)x) vs host CR: %d   logf vs host CR: %d   device double-path vs device logf: %d\n", n, ma, mb, mab);
    return 0;
}
