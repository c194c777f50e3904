// coissue.hip -- do two waves on one SIMD overlap VALU work of one with SALU work of the other?
// Workgroups of 256 threads (4 waves: one per SIMD); two workgroups per CU (launch 512 blocks with enough registers
// to limit residency to 2 waves / SIMD).  Even blocks run a VALU loop, odd blocks a SALU loop (or both VALU / both SALU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void __launch_bounds__(256) k(double* out, int iters, int mode_even, int mode_odd) {
    const int mode = (blockIdx.x & 256) ? mode_odd : mode_even;      // blocks 0..255 first residents, 256..511 second
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    const double c = 1.0000001, d = 1e-9;
    int s0 = blockIdx.x, s1 = 7;
    asm volatile("v_mov_b32 v200, 0" ::: "v200");                    // > 128 VGPRs: at most 2 waves per SIMD
    for (int k = 0; k < iters; ++k) {
        if (mode == 0) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
                asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(d));
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u)
                asm volatile("s_add_i32 %0, %0, %1\n s_xor_b32 %1, %1, %0\n s_add_i32 %0, %0, %1\n s_xor_b32 %1, %1, %0\n" : "+s"(s0), "+s"(s1) :: "scc");
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + s0 + s1;
}
int main() {
    double* d; CK(hipMalloc(&d, 512 * 256 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    const char* names[] = {"VALU", "SALU"};
    for (int nb : {256, 512})
        for (int me = 0; me < 2; ++me) for (int mo = 0; mo < 2; ++mo) {
            if (nb == 256 && mo != me) continue;
            hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, d, 10, me, mo);
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, d, iters, me, mo);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%d blocks (%d wave(s)/SIMD): first residents %s, second residents %s: %.3f ms (%.2f ns per instruction of one wave)\n", nb, nb / 256,
                   names[me], nb == 512 ? names[mo] : "-", ms, ms * 1e6 / (iters * 64.0));
        }
    return 0;
}
