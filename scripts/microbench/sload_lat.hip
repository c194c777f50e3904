// sload_lat.hip -- latency of a dependent scalar load (s_load_dword, scalar cache hit) on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef const __attribute__((address_space(4))) int* cptr;
__global__ void __launch_bounds__(256) k(const int* tab, int* out, int iters) {
    cptr t = (cptr)tab;
    int idx = 0;
    for (int k = 0; k < iters; ++k) {
#pragma unroll
        for (int u = 0; u < 16; ++u) idx = t[idx];
    }
    if (threadIdx.x == 0) out[blockIdx.x] = idx;
}
int main() {
    const int n = 64;
    int h[n]; for (int i = 0; i < n; ++i) h[i] = (i * 7 + 3) % n;
    int *d, *o; CK(hipMalloc(&d, sizeof h)); CK(hipMalloc(&o, 4096 * 4)); CK(hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int nb : {1, 256, 512, 1024}) {
        hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, d, o, 10);
        CK(hipEventRecord(e0));
        const int iters = 2000;
        hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, d, o, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("blocks %4d: %.3f ms -> %.1f ns per dependent s_load (per wave)\n", nb, ms, ms * 1e6 / (iters * 16.0));
    }
    return 0;
}
