// seed_acc.hip -- accuracy (in ulp) of the reciprocal / square-root refinements of evp_fast_math.h and of shorter
// variants, against correctly rounded results, for arguments over many binades.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ inline double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
__global__ void k(const double* x, double* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    // rcp: seed, 1 step, 2 steps
    double r0 = __builtin_amdgcn_rcp(v);
    double r1 = fma_(fma_(-v, r0, 1.0), r0, r0);
    double r2 = fma_(fma_(-v, r1, 1.0), r1, r1);
    // sqrt / rsqrt: seed, coupled step, +1 correction, +2 corrections
    const double y = __builtin_amdgcn_rsq(v);
    double g = v * y, h = 0.5 * y;
    const double r = fma_(-h, g, 0.5);
    g = fma_(g, r, g); h = fma_(h, r, h);
    const double g1 = g, h1 = h;
    double d = fma_(-g, g, v);
    g = fma_(d, h, g);
    const double g2 = g;
    d = fma_(-g, g, v);
    g = fma_(d, h, g);
    const double g3 = g;
    const double e = fma_(-h, g, 0.5);
    const double h3 = fma_(h, e, h);
    // variant: one correction for g (g2), reciprocal correction with g2
    const double e2 = fma_(-h1, g2, 0.5);
    const double h2 = fma_(h1, e2, h1);
    double* o = out + (size_t)i * 10;
    o[0] = r0; o[1] = r1; o[2] = r2; o[3] = y; o[4] = g1; o[5] = g2; o[6] = g3; o[7] = 2.0 * h1; o[8] = 2.0 * h2; o[9] = 2.0 * h3;
}
static double ulps(double got, long double ref) {
    if (!(got == got)) return 1e30;
    int ex; frexpl(ref, &ex);
    const long double ulp = ldexpl(1.0L, ex - 53);
    return (double)fabsl(((long double)got - ref) / ulp);
}
int main() {
    const int n = 1 << 22;
    std::vector<double> x(n);
    srand(7);
    for (int i = 0; i < n; ++i) {
        const double m = 1.0 + rand() / (double)RAND_MAX;
        const int e = (rand() % 120) - 60;
        x[i] = ldexp(m, e);
    }
    double *dx, *dout; CK(hipMalloc(&dx, n * 8)); CK(hipMalloc(&dout, (size_t)n * 80));
    CK(hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    std::vector<double> out((size_t)n * 10);
    CK(hipMemcpy(out.data(), dout, (size_t)n * 80, hipMemcpyDeviceToHost));
    const char* names[10] = {"rcp seed", "rcp 1 step", "rcp 2 steps", "rsq seed", "sqrt coupled", "sqrt +1 corr", "sqrt +2 corr (current)",
                             "rsqrt coupled", "rsqrt +1 corr (short)", "rsqrt final (current)"};
    for (int c = 0; c < 10; ++c) {
        double mx = 0, sum = 0;
        for (int i = 0; i < n; ++i) {
            const long double v = x[i];
            const long double ref = c < 3 ? 1.0L / v : (c == 3 || c >= 7 ? 1.0L / sqrtl(v) : sqrtl(v));
            const double u = ulps(out[(size_t)i * 10 + c], ref);
            mx = u > mx ? u : mx; sum += u;
        }
        printf("%-28s max %.3g ulp, mean %.3g ulp\n", names[c], mx, sum / n);
    }
    return 0;
}
