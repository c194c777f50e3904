// power_fma.hip -- what does the chip SUSTAIN?  A stream of independent v_fma_f64 on every SIMD (WAVES waves each) for several
// seconds; prints the issue interval per wave-instruction per SIMD window by window.  Run beside scripts/clock_watch.sh, which
// samples the shader clock and the board power meanwhile: the interval in CYCLES of the sampled clock separates "the ALU takes
// 4 cycles" from "the power cap lowers the clock".   ./power_fma [seconds] [waves per SIMD] [active lanes: 64 or fewer]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k(double* out, int iters, int lanes) {
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double c = 1.0000001, d = 1e-9;
    if ((int)(threadIdx.x & 63) < lanes) {
        for (int q = 0; q < iters; ++q) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                             "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
        }
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678) out[0] = a0;
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 4.0;
    const int waves = argc > 2 ? atoi(argv[2]) : 4, lanes = argc > 3 ? atoi(argv[3]) : 64;
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    double* out; CK(hipMalloc(&out, 8));
    const int iters = 20000;                              // x 64 instructions per wave: ~ 3 ms per launch at 4 waves per SIMD
    const int blocks = cus * waves;                       // 256 threads = 4 waves = one per SIMD; `waves` blocks per CU
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const auto t0 = std::chrono::steady_clock::now();
    printf("%d CUs, %d waves per SIMD, %d active lanes, %d x 64 v_fma_f64 per wave and launch\n", cus, waves, lanes, iters);
    for (int w = 0;; ++w) {
        const int reps = 40;
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, lanes);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double ns = ms * 1e6 / ((double)reps * iters * 64 * waves);
        const double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("t=%.2f s  %.3f ns per wave-instruction per SIMD  (%.2f TFLOP/s)\n", t, ns, (double)cus * 4 * 64 * 2 / ns / 1e3);
        fflush(stdout);
        if (t > seconds) break;
    }
    return 0;
}
