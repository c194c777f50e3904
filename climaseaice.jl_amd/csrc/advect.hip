// advect.hip -- h / aice advection tendencies and the tracer update.
//
//   k_tendencies  _compute_dynamic_tracer_tendencies!  tracer_tendency_kernel_functions.jl:27-45
//                 = - horizontal_div_Uc (sea_ice_advection.jl:51-54) with the upstream
//                 Oceananigans upwind-biased reconstructions (WENO(order = 5 | 7) in the WENO-Z
//                 form, UpwindBiased(order = 5), first-order upwind), their
//                 order reduction next to walls and immersed cells, zero flux through immersed faces.
//   k_tracer_step _dynamic_step_tracers!               sea_ice_fe_step.jl:56-82
//                 (RK3: h^n, aice^n = Psi^-, sea_ice_rk_substep.jl:140-149)
//
// This translation unit is compiled with -ffp-contract=off.  STRICT mode keeps the oracle's expression order and IEEE
// divisions: bit-identical tracer tendencies.  FAST mode (round 3) evaluates the SAME reconstructions -- same stencils,
// same smoothness indicators, same WENO-Z weights per tracer, same order reduction -- with the divisions by constants as
// multiplications, the four or five true divisions of a reconstruction as hardware reciprocal + one Newton step (fm::rcp,
// <= 11 ulp) and fused multiply-adds: a WENO7 reconstruction is ~150 instead of ~450 instructions (nine IEEE divisions of
// ~35 instructions each were two thirds of it).  Stated tolerance: |dG| <= 1e-12 max|G| on the tendencies (round 5: contracted smoothness indicators, below; 1e-13 before), 1e-13 relative on
// h, aice after a step (measured ~1e-16); masks, zero sets and the order-reduction decisions are the same code.  Each thread owns one
// face pair (west, south) of its cell: fluxes are computed once, shared with the east / north
// neighbours through LDS, so every face flux is evaluated exactly once per tile interior.
#include <cstdlib>

#include "csi_dev.h"
#include "csi_kernels.h"
#include "evp_fast_math.h"

namespace csi {
// Julia's max(a, b) for floats: NaN if either is NaN
__device__ __forceinline__ double jmax(double a, double b) { return (a != a || b != b) ? a + b : (a < b ? b : a); }

namespace adv {

#define WENO_EPS 1e-8

__device__ __forceinline__ double weno5(const double* p) {
    double q0 = (2 * p[2] + 5 * p[3] - p[4]) / 6;
    double q1 = (-p[1] + 5 * p[2] + 2 * p[3]) / 6;
    double q2 = (2 * p[0] - 7 * p[1] + 11 * p[2]) / 6;
    double b0 = (p[2] * (10 * p[2] - 31 * p[3] + 11 * p[4]) + p[3] * (25 * p[3] - 19 * p[4]) + p[4] * (4 * p[4])) / 3;
    double b1 = (p[1] * (4 * p[1] - 13 * p[2] + 5 * p[3]) + p[2] * (13 * p[2] - 13 * p[3]) + p[3] * (4 * p[3])) / 3;
    double b2 = (p[0] * (4 * p[0] - 19 * p[1] + 11 * p[2]) + p[1] * (25 * p[1] - 31 * p[2]) + p[2] * (10 * p[2])) / 3;
    double tau = fabs(b0 - b2);
    double r0 = tau / (b0 + WENO_EPS), r1 = tau / (b1 + WENO_EPS), r2 = tau / (b2 + WENO_EPS);
    double a0 = (3.0 / 10) * (1 + r0 * r0);
    double a1 = (3.0 / 5) * (1 + r1 * r1);
    double a2 = (1.0 / 10) * (1 + r2 * r2);
    double s = a0 + a1 + a2;
    return (a0 * q0 + a1 * q1 + a2 * q2) / s;
}
__device__ __forceinline__ double upwind5(const double* p) {
    return (2 * p[0] - 13 * p[1] + 47 * p[2] + 27 * p[3] - 3 * p[4]) / 60;
}
__device__ __forceinline__ double weno7(const double* p) {
    double q0 = (3 * p[3] + 13 * p[4] - 5 * p[5] + p[6]) / 12;
    double q1 = (-p[2] + 7 * p[3] + 7 * p[4] - p[5]) / 12;
    double q2 = (p[1] - 5 * p[2] + 13 * p[3] + 3 * p[4]) / 12;
    double q3 = (-3 * p[0] + 13 * p[1] - 23 * p[2] + 25 * p[3]) / 12;
    double b0 = p[3] * (2.107 * p[3] - 9.402 * p[4] + 7.042 * p[5] - 1.854 * p[6]) +
                p[4] * (11.003 * p[4] - 17.246 * p[5] + 4.642 * p[6]) + p[5] * (7.043 * p[5] - 3.882 * p[6]) + p[6] * (0.547 * p[6]);
    double b1 = p[2] * (0.547 * p[2] - 2.522 * p[3] + 1.922 * p[4] - 0.494 * p[5]) +
                p[3] * (3.443 * p[3] - 5.966 * p[4] + 1.602 * p[5]) + p[4] * (2.843 * p[4] - 1.642 * p[5]) + p[5] * (0.267 * p[5]);
    double b2 = p[1] * (0.267 * p[1] - 1.642 * p[2] + 1.602 * p[3] - 0.494 * p[4]) +
                p[2] * (2.843 * p[2] - 5.966 * p[3] + 1.922 * p[4]) + p[3] * (3.443 * p[3] - 2.522 * p[4]) + p[4] * (0.547 * p[4]);
    double b3 = p[0] * (0.547 * p[0] - 3.882 * p[1] + 4.642 * p[2] - 1.854 * p[3]) +
                p[1] * (7.043 * p[1] - 17.246 * p[2] + 7.042 * p[3]) + p[2] * (11.003 * p[2] - 9.402 * p[3]) + p[3] * (2.107 * p[3]);
    double tau = fabs(b0 + 3 * b1 - 3 * b2 - b3);
    double r0 = tau / (b0 + WENO_EPS), r1 = tau / (b1 + WENO_EPS), r2 = tau / (b2 + WENO_EPS), r3 = tau / (b3 + WENO_EPS);
    double a0 = (4.0 / 35) * (1 + r0 * r0);
    double a1 = (18.0 / 35) * (1 + r1 * r1);
    double a2 = (12.0 / 35) * (1 + r2 * r2);
    double a3 = (1.0 / 35) * (1 + r3 * r3);
    double s = a0 + a1 + a2 + a3;
    return (a0 * q0 + a1 * q1 + a2 * q2 + a3 * q3) / s;
}

__device__ __forceinline__ double weno3(const double* p) {
    double q0 = (p[1] + p[2]) / 2;
    double q1 = (-p[0] + 3 * p[1]) / 2;
    double b0 = p[1] * (p[1] - 2 * p[2]) + p[2] * p[2];
    double b1 = p[0] * (p[0] - 2 * p[1]) + p[1] * p[1];
    double tau = fabs(b0 - b1);
    double r0 = tau / (b0 + WENO_EPS), r1 = tau / (b1 + WENO_EPS);
    double a0 = (2.0 / 3) * (1 + r0 * r0);
    double a1 = (1.0 / 3) * (1 + r1 * r1);
    double s = a0 + a1;
    return (a0 * q0 + a1 * q1) / s;
}
__device__ __forceinline__ double upwind3(const double* p) { return (-p[0] + 5 * p[1] + 2 * p[2]) / 6; }

// ---- FAST-mode reconstructions (see the header).  Candidate polynomials, weights and the final combination are contracted; divisions
// are reciprocals (fm::rcp: hardware seed + one Newton step) or multiplications by constants.  The smoothness indicators and tau:
// rounds 3 - 4 evaluated them EXACTLY as in STRICT mode (no contraction) -- they are small differences of O(c^2) terms, and a
// rounding-level change of their evaluation moves the nonlinear weights by ~1e-9 relative and the tendencies by 7e-13 of max|G|
// (measured), over the 1e-13 then stated for the tendencies.  Round 5 (CSI_ADV_FAST_BETA = 1, the default): they are contracted too --
// the uncontracted quadratic forms were 92 of a WENO7 reconstruction's ~150 FP64 instructions (56 contracted) -- and the FAST
// tolerance on the TENDENCIES is stated as 1e-12 of max|G|; the tolerance on what the north star names, h and aice after an update,
// stays 1e-13 relative (dt |dG| = 120 s x 1e-12 x 1e-6 m/s against h ~ 0.3 m: four orders inside it).  STRICT is untouched.
#ifndef CSI_ADV_FAST_BETA
#define CSI_ADV_FAST_BETA 1
#endif
#if CSI_ADV_FAST_BETA
#define CSI_BETA_CONTRACT _Pragma("clang fp contract(fast)")
#else
#define CSI_BETA_CONTRACT
#endif
__device__ __forceinline__ double weno3_fast(const double* p) {
    double b0, b1;
    {
        CSI_BETA_CONTRACT
        b0 = p[1] * (p[1] - 2 * p[2]) + p[2] * p[2];
        b1 = p[0] * (p[0] - 2 * p[1]) + p[1] * p[1];
    }
    const double tau = fabs(b0 - b1);
    {
#pragma clang fp contract(fast)
        const double q0 = 0.5 * (p[1] + p[2]);
        const double q1 = 0.5 * (3 * p[1] - p[0]);
        const double r0 = tau * fm::rcp(b0 + WENO_EPS), r1 = tau * fm::rcp(b1 + WENO_EPS);
        const double a0 = (2.0 / 3) * (1 + r0 * r0);
        const double a1 = (1.0 / 3) * (1 + r1 * r1);
        return (a0 * q0 + a1 * q1) * fm::rcp(a0 + a1);
    }
}
__device__ __forceinline__ double upwind3_fast(const double* p) {
#pragma clang fp contract(fast)
    return (5 * p[1] + 2 * p[2] - p[0]) * (1.0 / 6);
}
__device__ __forceinline__ double weno5_fast(const double* p) {
    double b0, b1, b2;
    {
        CSI_BETA_CONTRACT
#if CSI_ADV_FAST_BETA
        constexpr double third = 1.0 / 3;
        b0 = (p[2] * (10 * p[2] - 31 * p[3] + 11 * p[4]) + p[3] * (25 * p[3] - 19 * p[4]) + p[4] * (4 * p[4])) * third;
        b1 = (p[1] * (4 * p[1] - 13 * p[2] + 5 * p[3]) + p[2] * (13 * p[2] - 13 * p[3]) + p[3] * (4 * p[3])) * third;
        b2 = (p[0] * (4 * p[0] - 19 * p[1] + 11 * p[2]) + p[1] * (25 * p[1] - 31 * p[2]) + p[2] * (10 * p[2])) * third;
#else
        b0 = (p[2] * (10 * p[2] - 31 * p[3] + 11 * p[4]) + p[3] * (25 * p[3] - 19 * p[4]) + p[4] * (4 * p[4])) / 3;
        b1 = (p[1] * (4 * p[1] - 13 * p[2] + 5 * p[3]) + p[2] * (13 * p[2] - 13 * p[3]) + p[3] * (4 * p[3])) / 3;
        b2 = (p[0] * (4 * p[0] - 19 * p[1] + 11 * p[2]) + p[1] * (25 * p[1] - 31 * p[2]) + p[2] * (10 * p[2])) / 3;
#endif
    }
    const double tau = fabs(b0 - b2);
    {
#pragma clang fp contract(fast)
        const double q0 = (2 * p[2] + 5 * p[3] - p[4]) * (1.0 / 6);
        const double q1 = (5 * p[2] + 2 * p[3] - p[1]) * (1.0 / 6);
        const double q2 = (2 * p[0] - 7 * p[1] + 11 * p[2]) * (1.0 / 6);
        const double r0 = tau * fm::rcp(b0 + WENO_EPS), r1 = tau * fm::rcp(b1 + WENO_EPS), r2 = tau * fm::rcp(b2 + WENO_EPS);
        const double a0 = (3.0 / 10) * (1 + r0 * r0);
        const double a1 = (3.0 / 5) * (1 + r1 * r1);
        const double a2 = (1.0 / 10) * (1 + r2 * r2);
        return (a0 * q0 + a1 * q1 + a2 * q2) * fm::rcp(a0 + a1 + a2);
    }
}
__device__ __forceinline__ double upwind5_fast(const double* p) {
#pragma clang fp contract(fast)
    return (2 * p[0] - 13 * p[1] + 47 * p[2] + 27 * p[3] - 3 * p[4]) * (1.0 / 60);
}
__device__ __forceinline__ double weno7_fast(const double* p) {
    double b0, b1, b2, b3;
    {
        CSI_BETA_CONTRACT
        b0 = p[3] * (2.107 * p[3] - 9.402 * p[4] + 7.042 * p[5] - 1.854 * p[6]) +
             p[4] * (11.003 * p[4] - 17.246 * p[5] + 4.642 * p[6]) + p[5] * (7.043 * p[5] - 3.882 * p[6]) + p[6] * (0.547 * p[6]);
        b1 = p[2] * (0.547 * p[2] - 2.522 * p[3] + 1.922 * p[4] - 0.494 * p[5]) +
             p[3] * (3.443 * p[3] - 5.966 * p[4] + 1.602 * p[5]) + p[4] * (2.843 * p[4] - 1.642 * p[5]) + p[5] * (0.267 * p[5]);
        b2 = p[1] * (0.267 * p[1] - 1.642 * p[2] + 1.602 * p[3] - 0.494 * p[4]) +
             p[2] * (2.843 * p[2] - 5.966 * p[3] + 1.922 * p[4]) + p[3] * (3.443 * p[3] - 2.522 * p[4]) + p[4] * (0.547 * p[4]);
        b3 = p[0] * (0.547 * p[0] - 3.882 * p[1] + 4.642 * p[2] - 1.854 * p[3]) +
             p[1] * (7.043 * p[1] - 17.246 * p[2] + 7.042 * p[3]) + p[2] * (11.003 * p[2] - 9.402 * p[3]) + p[3] * (2.107 * p[3]);
    }
    const double tau = fabs(b0 + 3 * b1 - 3 * b2 - b3);
    {
#pragma clang fp contract(fast)
        const double q0 = (3 * p[3] + 13 * p[4] - 5 * p[5] + p[6]) * (1.0 / 12);
        const double q1 = (7 * p[3] + 7 * p[4] - p[2] - p[5]) * (1.0 / 12);
        const double q2 = (p[1] - 5 * p[2] + 13 * p[3] + 3 * p[4]) * (1.0 / 12);
        const double q3 = (13 * p[1] - 3 * p[0] - 23 * p[2] + 25 * p[3]) * (1.0 / 12);
        const double r0 = tau * fm::rcp(b0 + WENO_EPS), r1 = tau * fm::rcp(b1 + WENO_EPS), r2 = tau * fm::rcp(b2 + WENO_EPS), r3 = tau * fm::rcp(b3 + WENO_EPS);
        const double a0 = (4.0 / 35) * (1 + r0 * r0);
        const double a1 = (18.0 / 35) * (1 + r1 * r1);
        const double a2 = (12.0 / 35) * (1 + r2 * r2);
        const double a3 = (1.0 / 35) * (1 + r3 * r3);
        return (a0 * q0 + a1 * q1 + a2 * q2 + a3 * q3) * fm::rcp(a0 + a1 + a2 + a3);
    }
}

// ---- weight_dtype f32 (csi_set_weno_weight_dtype; oracle/csi_oracle.c weno*_f32, whose statement of the ASSUMED upstream semantics
// -- newer Oceananigans versions carry a second float type FT2 = Float32 for a WENO scheme's smoothness / weight arithmetic -- this
// follows): stencil values converted to float, indicators, tau, ratios, unnormalised weights and their sum in float (same
// expressions, same order, this unit is compiled without contraction; fp32 division is correctly rounded by default), candidates in
// double, result (sum alpha_s q_s) / (sum alpha_s) in double.  STRICT: that, bit for bit.  FAST: the same float weights -- they are
// a dozen full-rate instructions, nothing to gain -- with the double part contracted and the last division a reciprocal.
#define WENO_EPS_F 1e-8f
template <bool FAST>
__device__ __forceinline__ double weno3_w32(const double* p) {
    float f[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) f[k] = (float)p[k];
    const float b0 = f[1] * (f[1] - 2 * f[2]) + f[2] * f[2];
    const float b1 = f[0] * (f[0] - 2 * f[1]) + f[1] * f[1];
    const float tau = fabsf(b0 - b1);
    const float r0 = tau / (b0 + WENO_EPS_F), r1 = tau / (b1 + WENO_EPS_F);
    const float a0 = (float)(2.0 / 3) * (1 + r0 * r0);
    const float a1 = (float)(1.0 / 3) * (1 + r1 * r1);
    const float s = a0 + a1;
    if (FAST) {
#pragma clang fp contract(fast)
        const double q0 = 0.5 * (p[1] + p[2]);
        const double q1 = 0.5 * (3 * p[1] - p[0]);
        return ((double)a0 * q0 + (double)a1 * q1) * fm::rcp((double)s);
    }
    const double q0 = (p[1] + p[2]) / 2;
    const double q1 = (-p[0] + 3 * p[1]) / 2;
    return ((double)a0 * q0 + (double)a1 * q1) / (double)s;
}
template <bool FAST>
__device__ __forceinline__ double weno5_w32(const double* p) {
    float f[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) f[k] = (float)p[k];
    const float b0 = (f[2] * (10 * f[2] - 31 * f[3] + 11 * f[4]) + f[3] * (25 * f[3] - 19 * f[4]) + f[4] * (4 * f[4])) / 3;
    const float b1 = (f[1] * (4 * f[1] - 13 * f[2] + 5 * f[3]) + f[2] * (13 * f[2] - 13 * f[3]) + f[3] * (4 * f[3])) / 3;
    const float b2 = (f[0] * (4 * f[0] - 19 * f[1] + 11 * f[2]) + f[1] * (25 * f[1] - 31 * f[2]) + f[2] * (10 * f[2])) / 3;
    const float tau = fabsf(b0 - b2);
    const float r0 = tau / (b0 + WENO_EPS_F), r1 = tau / (b1 + WENO_EPS_F), r2 = tau / (b2 + WENO_EPS_F);
    const float a0 = (float)(3.0 / 10) * (1 + r0 * r0);
    const float a1 = (float)(3.0 / 5) * (1 + r1 * r1);
    const float a2 = (float)(1.0 / 10) * (1 + r2 * r2);
    const float s = a0 + a1 + a2;
    if (FAST) {
#pragma clang fp contract(fast)
        const double q0 = (2 * p[2] + 5 * p[3] - p[4]) * (1.0 / 6);
        const double q1 = (5 * p[2] + 2 * p[3] - p[1]) * (1.0 / 6);
        const double q2 = (2 * p[0] - 7 * p[1] + 11 * p[2]) * (1.0 / 6);
        return ((double)a0 * q0 + (double)a1 * q1 + (double)a2 * q2) * fm::rcp((double)s);
    }
    const double q0 = (2 * p[2] + 5 * p[3] - p[4]) / 6;
    const double q1 = (-p[1] + 5 * p[2] + 2 * p[3]) / 6;
    const double q2 = (2 * p[0] - 7 * p[1] + 11 * p[2]) / 6;
    return ((double)a0 * q0 + (double)a1 * q1 + (double)a2 * q2) / (double)s;
}
template <bool FAST>
__device__ __forceinline__ double weno7_w32(const double* p) {
    float f[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) f[k] = (float)p[k];
    const float b0 = f[3] * (2.107f * f[3] - 9.402f * f[4] + 7.042f * f[5] - 1.854f * f[6]) +
                     f[4] * (11.003f * f[4] - 17.246f * f[5] + 4.642f * f[6]) + f[5] * (7.043f * f[5] - 3.882f * f[6]) + f[6] * (0.547f * f[6]);
    const float b1 = f[2] * (0.547f * f[2] - 2.522f * f[3] + 1.922f * f[4] - 0.494f * f[5]) +
                     f[3] * (3.443f * f[3] - 5.966f * f[4] + 1.602f * f[5]) + f[4] * (2.843f * f[4] - 1.642f * f[5]) + f[5] * (0.267f * f[5]);
    const float b2 = f[1] * (0.267f * f[1] - 1.642f * f[2] + 1.602f * f[3] - 0.494f * f[4]) +
                     f[2] * (2.843f * f[2] - 5.966f * f[3] + 1.922f * f[4]) + f[3] * (3.443f * f[3] - 2.522f * f[4]) + f[4] * (0.547f * f[4]);
    const float b3 = f[0] * (0.547f * f[0] - 3.882f * f[1] + 4.642f * f[2] - 1.854f * f[3]) +
                     f[1] * (7.043f * f[1] - 17.246f * f[2] + 7.042f * f[3]) + f[2] * (11.003f * f[2] - 9.402f * f[3]) + f[3] * (2.107f * f[3]);
    const float tau = fabsf(b0 + 3 * b1 - 3 * b2 - b3);
    const float r0 = tau / (b0 + WENO_EPS_F), r1 = tau / (b1 + WENO_EPS_F), r2 = tau / (b2 + WENO_EPS_F), r3 = tau / (b3 + WENO_EPS_F);
    const float a0 = (float)(4.0 / 35) * (1 + r0 * r0);
    const float a1 = (float)(18.0 / 35) * (1 + r1 * r1);
    const float a2 = (float)(12.0 / 35) * (1 + r2 * r2);
    const float a3 = (float)(1.0 / 35) * (1 + r3 * r3);
    const float s = a0 + a1 + a2 + a3;
    if (FAST) {
#pragma clang fp contract(fast)
        const double q0 = (3 * p[3] + 13 * p[4] - 5 * p[5] + p[6]) * (1.0 / 12);
        const double q1 = (7 * p[3] + 7 * p[4] - p[2] - p[5]) * (1.0 / 12);
        const double q2 = (p[1] - 5 * p[2] + 13 * p[3] + 3 * p[4]) * (1.0 / 12);
        const double q3 = (13 * p[1] - 3 * p[0] - 23 * p[2] + 25 * p[3]) * (1.0 / 12);
        return ((double)a0 * q0 + (double)a1 * q1 + (double)a2 * q2 + (double)a3 * q3) * fm::rcp((double)s);
    }
    const double q0 = (3 * p[3] + 13 * p[4] - 5 * p[5] + p[6]) / 12;
    const double q1 = (-p[2] + 7 * p[3] + 7 * p[4] - p[5]) / 12;
    const double q2 = (p[1] - 5 * p[2] + 13 * p[3] + 3 * p[4]) / 12;
    const double q3 = (-3 * p[0] + 13 * p[1] - 23 * p[2] + 25 * p[3]) / 12;
    return ((double)a0 * q0 + (double)a1 * q1 + (double)a2 * q2 + (double)a3 * q3) / (double)s;
}

// Boundary-order reduction next to walls (upstream topologically_conditional_interpolation, recalled -- SURVEY.md
// App. B; same rule as oracle/csi_oracle.c::reduced_buffer): the scheme with buffer B (order 2B-1) is used at face
// idx only if its biased stencil stays inside the domain, else the buffer scheme of order 2B-3, down to upwind 1.
__device__ __forceinline__ int reduced_buffer(int B, int idx, int N, bool left, bool wall_lo, bool wall_hi) {
    while (B > 1) {
        const bool lo_ok = !wall_lo || idx >= (left ? B + 1 : B);
        const bool hi_ok = !wall_hi || idx <= (left ? N + 2 - B : N + 1 - B);
        if (lo_ok && hi_ok) break;
        --B;
    }
    return B;
}

// Immersed boundaries (upstream immersed reconstruction, recalled; oracle/csi_oracle.c::reduced_buffer_immersed):
// the scheme with buffer B needs the 2B cells idx-B .. idx+B-1 around the face to be active (not immersed, not
// beyond a wall); here as distances: dl = cells below the face up to the first inactive one, dr likewise above.
__device__ __forceinline__ int reduced_buffer_immersed(const GridDev& g, int B, int i, int j, bool along_y) {
    int dl = B, dr = B;
    for (int k = B; k >= 1; --k) {
        if (along_y ? inactive_cell(g, i, j - k) : inactive_cell(g, i - k, j)) dl = k - 1;
        if (along_y ? inactive_cell(g, i, j + k - 1) : inactive_cell(g, i + k - 1, j)) dr = k - 1;
    }
    const int b = dl < dr ? dl : dr;
    return b < 1 ? 1 : b;
}

// reconstruct at a face from the line of values through `base` (cell on the high side of the
// face); st = element stride of the line; left bias (vel > 0): upwind cell is base - st.
// B: buffer of the scheme to use at this face (after the boundary-order reduction).
template <int SCHEME, bool FAST = false, bool W32 = false>
__device__ __forceinline__ double reconstruct(const double* base, long st, bool left, int B) {
    const double* up = left ? base - st : base;
    const long s = left ? st : -st;
    if (SCHEME == 1) return up[0];
    constexpr bool WENO = SCHEME > 0;
    if (B == 1) return up[0];
    if (B == 2) {
        double p[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) p[k] = up[(k - 1) * s];
        return WENO ? (W32 ? weno3_w32<FAST>(p) : (FAST ? weno3_fast(p) : weno3(p))) : (FAST ? upwind3_fast(p) : upwind3(p));
    }
    if (B == 3 || SCHEME != 7) {
        double p[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) p[k] = up[(k - 2) * s];
        return WENO ? (W32 ? weno5_w32<FAST>(p) : (FAST ? weno5_fast(p) : weno5(p))) : (FAST ? upwind5_fast(p) : upwind5(p));
    }
    double p[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) p[k] = up[(k - 3) * s];
    return W32 ? weno7_w32<FAST>(p) : (FAST ? weno7_fast(p) : weno7(p));
}
// buffer at face (i, j) of the x / y direction: immersed grid -> the immersed rule (it covers the walls: cells beyond
// them are inactive); else the topological rule next to walls; else the full scheme
template <int SCHEME>
__device__ __forceinline__ int buffer_at(const GridDev& g, int i, int j, bool along_y, bool left) {
    constexpr int B0 = SCHEME == 7 ? 4 : (SCHEME == 1 ? 1 : ((SCHEME == 3 || SCHEME == -3) ? 2 : 3));   // order 2B - 1
    if (SCHEME == 1) return 1;
    if (g.has_mask) return reduced_buffer_immersed(g, B0, i, j, along_y);
    const bool wl = (along_y ? g.ylo : g.xlo) == SIDE_WALL, wh = (along_y ? g.yhi : g.xhi) == SIDE_WALL;
    if (!(wl | wh)) return B0;
    return reduced_buffer(B0, along_y ? j : i, along_y ? g.Ny : g.Nx, left, wl, wh);
}


// measured (scripts/r03_adv_shapes.sh, us per RK3 step at 512^2 / 1024^2 / 2048^2): 64 x 4 cells per block 55 / 185 / 762, 64 x 6: 52 / 171 / 704,
// 32 x 12: 51 / 176 / 693, 40 x 10: 54 / 171 / 709, 64 x 2: 64 / 208 / 849 (the extra face row / column of a tile is redundant work)
#ifndef CSI_ADV_TY
#define CSI_ADV_TY 6
#endif
#ifndef CSI_ADV_TX
#define CSI_ADV_TX 64
#endif
constexpr int TX = CSI_ADV_TX, TY2 = CSI_ADV_TY, TY3 = 4;      // rows of a flux tile with two tracers / with three (snow): (TX + 1)(TY + 1) tracers <= 1024 threads

// One thread per cell AND TRACER of a (TX+1) x (TY+1) flux tile (threadIdx.z: h, aice [, snow thickness]): the extra column /
// row holds the east / north faces of the tile.  Fx[i] = Ax u c~ at the west face of cell i, Fy[j] at the south face.  (Round 3:
// the tracers used to share a thread -- four WENO7 reconstructions behind 28 dependent loads; at 512^2 the launch lasts as long
// as one block, so halving the chain per thread is what shortens it.)
// store_with_images for a tracer on a grid with N >= 2 H and no fold: a cell is within H of at most one side per direction, so it
// has at most one x image, one y image and their corner -- scalars instead of store_with_images' index lists (which live in
// scratch memory: inside the tendency kernel they cost more than the separate update launch they were meant to save)
__device__ __forceinline__ void store_tracer_images(const FRef& f, const GridDev& g, const ImageSpec& im, int i, int j, double val) {
    f(i, j) = val;
    int ix = 0, jy = 0;
    bool hx = false, hy = false;
    if (i <= g.Hx) {
        if (im.xhi == IMG_WRAP) { ix = i + g.Nx; hx = true; }
        if (im.xlo == IMG_MIRROR) { ix = 1 - i; hx = true; }
    } else if (i > g.Nx - g.Hx) {
        if (im.xlo == IMG_WRAP) { ix = i - g.Nx; hx = true; }
        if (im.xhi == IMG_MIRROR) { ix = 2 * g.Nx + 1 - i; hx = true; }
    }
    if (j <= g.Hy) {
        if (im.yhi == IMG_WRAP) { jy = j + g.Ny; hy = true; }
        if (im.ylo == IMG_MIRROR) { jy = 1 - j; hy = true; }
    } else if (j > g.Ny - g.Hy) {
        if (im.ylo == IMG_WRAP) { jy = j - g.Ny; hy = true; }
        if (im.yhi == IMG_MIRROR) { jy = 2 * g.Ny + 1 - j; hy = true; }
    }
    if (hx) f(ix, j) = val;
    if (hy) f(i, jy) = val;
    if (hx & hy) f(ix, jy) = val;
}

// STEP (launch_advect_stage): the launch is a whole RK stage of an advection-only model -- the h thread of a cell also does
// _dynamic_step_tracers! (k_tracer_step's arithmetic, statement for statement) with the two tendencies of its cell, into the
// stage's OUTPUT arrays.
// NT (round 5): tracers per thread.  1: one thread per cell AND tracer (threadIdx.z), the latency-optimal layout of small grids
// (round 3: at 512^2 a launch lasts as long as one block).  2: one thread per cell does h AND aice -- the face velocity and its
// sign, the order-reduction decision, the closed-face test, the face area and the stencil's address arithmetic are the same for
// both tracers and were half of a thread's instructions (ISA: ~300 of ~590 per cell and tracer were not the reconstruction's
// arithmetic); large grids, which are throughput-bound (VALU busy 0.68 at 2048^2), take this one.  Same operations per value:
// bit-identical to NT = 1.
template <int SCHEME, bool FAST, bool STEP = false, int TY = TY2, bool W32 = false, int NT = 1, int TXP = TX>
__global__ void __launch_bounds__(NT == 2 ? (TXP + 1) * (TY + 1) : 1024) k_tendencies(AdvDev A) {
    __shared__ double sFx[NT == 2 ? 2 : 3][TY + 1][TXP + 2], sFy[NT == 2 ? 2 : 3][TY + 1][TXP + 2];
    __shared__ double sG[(STEP && NT == 1) ? 2 : 1][(STEP && NT == 1) ? TY : 1][(STEP && NT == 1) ? TXP : 1];
    const GridDev& g = A.g;
    const int tx = threadIdx.x, ty = threadIdx.y, tz = NT == 2 ? 0 : threadIdx.z;          // tx in [0, TX], ty in [0, TY], tz: tracer
    const int i = 1 + blockIdx.x * TXP + tx, j = 1 + blockIdx.y * TY + ty;
    const bool in_x = i <= g.Nx + 1, in_y = j <= g.Ny + 1;
    const FRef& c = tz == 0 ? A.h : (tz == 1 ? A.a : A.hs);
    // STEP: the update's base values are independent of the fluxes: loaded (and, first stage, cached as Psi^-) up front, so that
    // behind the fluxes only a few flops and one store remain; the h thread stores h, the aice thread aice
    const bool owns = tx < TXP && ty < TY && i <= g.Nx && j <= g.Ny;
    double hn = 0.0, an = 0.0;
    if (STEP && owns) {
        hn = A.hb(i, j); an = A.ab(i, j);
        if (A.write_cache) {              // Psi^- = the state this step starts from (its halos: images, like the state's)
            if (NT == 2 || tz == 0) store_tracer_images(A.hm, A.g, A.im, i, j, hn);
            if (NT == 2 || tz != 0) store_tracer_images(A.am, A.g, A.im, i, j, an);
        }
    }
    if (in_x && in_y) {
        // x-face flux (needed for ty < TY rows), y-face flux (needed for tx < TX columns)
        if (ty < TY && j <= g.Ny) {
            const double uu = A.u(i, j);
            const bool left = uu > 0;
            const int B = buffer_at<SCHEME>(g, i, j, false, left);
            const bool closed = g.has_mask && peripheral_u(g, i, j);          // conditional_flux_fcc
            const double ax = dym(g, LOC_F, LOC_C, i, j);                    // Ax^{fcc} = dy^{fcc} * dz
            const double cc = reconstruct<SCHEME, FAST, W32>(&c(i, j), 1, left, B);
            sFx[tz][ty][tx] = closed ? 0.0 : ax * uu * cc;
            if (NT == 2) {
                const double c2 = reconstruct<SCHEME, FAST, W32>(&A.a(i, j), 1, left, B);
                sFx[1][ty][tx] = closed ? 0.0 : ax * uu * c2;
            }
        }
        if (tx < TXP && i <= g.Nx) {
            const double vv = A.v(i, j);
            const bool left = vv > 0;
            const int B = buffer_at<SCHEME>(g, i, j, true, left);
            const double dxf = dxm(g, LOC_C, LOC_F, i, j);                   // Ay^{cfc} = dx^{cfc} * dz
            const bool closed = g.has_mask && peripheral_v(g, i, j);          // conditional_flux_cfc
            const double cc = reconstruct<SCHEME, FAST, W32>(&c(i, j), c.ld, left, B);
            sFy[tz][ty][tx] = closed ? 0.0 : dxf * vv * cc;
            if (NT == 2) {
                const double c2 = reconstruct<SCHEME, FAST, W32>(&A.a(i, j), A.a.ld, left, B);
                sFy[1][ty][tx] = closed ? 0.0 : dxf * vv * c2;
            }
        }
    }
    __syncthreads();
    double G0 = 0.0, G1 = 0.0;
    if (tx < TXP && ty < TY && i <= g.Nx && j <= g.Ny) {
        const double V = azm(g, LOC_C, LOC_C, i, j);
        const double rV = FAST ? fm::rcp(V) : 1 / V;
        const double fx = sFx[tz][ty][tx + 1] - sFx[tz][ty][tx], fy = sFy[tz][ty + 1][tx] - sFy[tz][ty][tx];
        const FRef& G = tz == 0 ? A.Gh : (tz == 1 ? A.Ga : A.Ghs);         // (snow: compute_snow_advection_tendency!, tracer_tendency_kernel_functions.jl:49-52)
        const double Gv = -(rV * (fx + fy));
        G(i, j) = Gv;
        G0 = Gv;
        if (NT == 2) {
            const double fx2 = sFx[1][ty][tx + 1] - sFx[1][ty][tx], fy2 = sFy[1][ty + 1][tx] - sFy[1][ty][tx];
            G1 = -(rV * (fx2 + fy2));
            A.Ga(i, j) = G1;
        } else if (STEP) sG[tz][ty][tx] = Gv;
    }
    if (STEP) {
        if (NT == 1) __syncthreads();
        if (owns) {
            double hp = hn + A.dt * (NT == 2 ? G0 : sG[0][ty][tx]);
            double ap = an + A.dt * (NT == 2 ? G1 : sG[1][ty][tx]);
            ap = jmax(0.0, ap);
            hp = jmax(0.0, hp);
            ap = (hp == 0) ? 0.0 : ap;
            hp = (ap == 0) ? 0.0 : hp;
            const double Vp = hp * ap;
            const double a1 = (ap > 1) ? 1.0 : ap, h1 = (ap > 1) ? Vp : hp;
            if (NT == 2 || tz == 0) store_tracer_images(A.ho, A.g, A.im, i, j, h1);
            if (NT == 2 || tz != 0) store_tracer_images(A.ao, A.g, A.im, i, j, a1);
        }
    }
}

__global__ void __launch_bounds__(256) k_tracer_step(AdvDev A) {
    const int i = 1 + blockIdx.x * blockDim.x + threadIdx.x, j = 1 + blockIdx.y * blockDim.y + threadIdx.y;
    if (i > A.g.Nx || j > A.g.Ny) return;
    const double hn = A.from_cache ? A.hm(i, j) : A.h(i, j);
    const double an = A.from_cache ? A.am(i, j) : A.a(i, j);
    double hp = hn + A.dt * A.Gh(i, j);
    double ap = an + A.dt * A.Ga(i, j);
    ap = jmax(0.0, ap);
    hp = jmax(0.0, hp);
    ap = (hp == 0) ? 0.0 : ap;
    hp = (ap == 0) ? 0.0 : hp;
    const double Vp = hp * ap;
    const double a1 = (ap > 1) ? 1.0 : ap, h1 = (ap > 1) ? Vp : hp;
    if (A.fill_images) {             // update_state!'s halo fill of h, aice fused into the stores (csi_abi.hip)
        store_with_images(A.a, A.g, A.im, i, j, a1);
        store_with_images(A.h, A.g, A.im, i, j, h1);
    } else {
        A.a(i, j) = a1;
        A.h(i, j) = h1;
    }
    if (A.has_snow) {                // dynamic_step_snow!, sea_ice_fe_step.jl:86-94
        const double sn = A.from_cache ? A.hsm(i, j) : A.hs(i, j);
        double sp = sn + A.dt * A.Ghs(i, j);
        sp = jmax(0.0, sp);
        const double s1 = (a1 <= 0) ? 0.0 : sp;
        if (A.fill_images) store_with_images(A.hs, A.g, A.im, i, j, s1); else A.hs(i, j) = s1;
    }
}

}  // namespace adv

// W32: the WENO weights in single precision (AdvDev::w32; only the WENO schemes have weights)
// NT: tracers per thread (k_tendencies): 2 from CSI_ADV_NT2_CELLS cells on.  Measured round 5 (scripts/adv_bench.py, one box, WENO7, us per
// advection-only RK3 step / per tendency launch, NT = 1 -> 2): 256^2 25.7 -> 25.8 / 7.0 -> 7.4; 512^2 53.5 -> 48.5 / 14.2 -> 13.0;
// 1024^2 171 -> 142 / 47 -> 38; 2048^2 728 -> 621 / 173 -> 139 (profiles/r05_advection.md)
#ifndef CSI_ADV_NT2_CELLS
#define CSI_ADV_NT2_CELLS 200000L
#endif
static bool adv_two_tracers(const AdvDev& A) {
    if (A.has_snow) return false;
    if (A.nt > 0) return A.nt == 2;              // tuning aid / tests (CSI_ADV_NT, read when the context is created)
    return (long)A.g.Nx * (long)A.g.Ny >= CSI_ADV_NT2_CELLS;
}
// Block shapes of the two-tracers-per-thread layout by grid size (round 5, one box, us per advection-only RK3 step at 512^2 / 1024^2 /
// 2048^2; profiles/r05_advection_shapes.txt): 64 x 6 cells 48.5 / 142 / 621; 64 x 8: 46 / 161 / 678; 63 x 7 (64 x 8 threads = 8 full
// waves): 49.5 / 132 / 591; 63 x 11: 49 / 138 / 577; 63 x 5: 53.5 / 150 / 646; 63 x 9: 60 / 154 / 650; 64 x 12: 63 / 175 / 715 -- not
// monotonic in anything simple (wave quantisation of the block, blocks per CU, redundant face rows), so: the best measured shape per size
enum { SHAPE_64x8 = 0, SHAPE_63x7 = 1, SHAPE_63x11 = 2 };
static int adv_shape(const AdvDev& A) {
    const long cells = (long)A.g.Nx * (long)A.g.Ny;
    return cells < 600000L ? SHAPE_64x8 : (cells < 2500000L ? SHAPE_63x7 : SHAPE_63x11);
}
template <bool FAST, bool W32>
static void launch_tendencies_mode(const AdvDev& A, hipStream_t s) {
    const bool two = adv_two_tracers(A);
    const int shape = adv_shape(A);
    const int tx = two ? (shape == SHAPE_64x8 ? 64 : 63) : adv::TX;
    const int ty = A.has_snow ? adv::TY3 : (two ? (shape == SHAPE_64x8 ? 8 : (shape == SHAPE_63x7 ? 7 : 11)) : adv::TY2);
    dim3 b(tx + 1, ty + 1, A.has_snow ? 3 : (two ? 1 : 2));
    dim3 gr((unsigned)((A.g.Nx + tx - 1) / tx), (unsigned)((A.g.Ny + ty - 1) / ty));
#define CSI_ADV_LAUNCH(S, W) do { if (A.has_snow) hipLaunchKernelGGL((adv::k_tendencies<S, FAST, false, adv::TY3, W>), gr, b, 0, s, A); \
                                  else if (two && shape == SHAPE_64x8) hipLaunchKernelGGL((adv::k_tendencies<S, FAST, false, 8, W, 2, 64>), gr, b, 0, s, A); \
                                  else if (two && shape == SHAPE_63x7) hipLaunchKernelGGL((adv::k_tendencies<S, FAST, false, 7, W, 2, 63>), gr, b, 0, s, A); \
                                  else if (two) hipLaunchKernelGGL((adv::k_tendencies<S, FAST, false, 11, W, 2, 63>), gr, b, 0, s, A); \
                                  else hipLaunchKernelGGL((adv::k_tendencies<S, FAST, false, adv::TY2, W>), gr, b, 0, s, A); } while (0)
    switch (A.scheme) {
        case 1: CSI_ADV_LAUNCH(1, false); break;
        case 3: CSI_ADV_LAUNCH(3, W32); break;
        case -3: CSI_ADV_LAUNCH(-3, false); break;
        case 5: CSI_ADV_LAUNCH(5, W32); break;
        case -5: CSI_ADV_LAUNCH(-5, false); break;
        default: CSI_ADV_LAUNCH(7, W32); break;
    }
#undef CSI_ADV_LAUNCH
}
// mode: CSI_MODE_STRICT (0) the oracle's arithmetic, bit for bit; CSI_MODE_FAST (1) reciprocals and contraction (header)
void launch_tracer_tendencies(const AdvDev& A, int mode, hipStream_t s) {
    if (A.w32) { if (mode == 1) launch_tendencies_mode<true, true>(A, s); else launch_tendencies_mode<false, true>(A, s); }
    else { if (mode == 1) launch_tendencies_mode<true, false>(A, s); else launch_tendencies_mode<false, false>(A, s); }
}
template <bool FAST, bool W32>
static void launch_stage_mode(const AdvDev& A, hipStream_t s) {
    const bool two = adv_two_tracers(A);
    const int shape = adv_shape(A);
    const int tx = two ? (shape == SHAPE_64x8 ? 64 : 63) : adv::TX;
    const int ty = two ? (shape == SHAPE_64x8 ? 8 : (shape == SHAPE_63x7 ? 7 : 11)) : adv::TY2;
    dim3 b(tx + 1, ty + 1, two ? 1 : 2);
    dim3 gr((unsigned)((A.g.Nx + tx - 1) / tx), (unsigned)((A.g.Ny + ty - 1) / ty));
#define CSI_ADV_STAGE(S, W) do { if (two && shape == SHAPE_64x8) hipLaunchKernelGGL((adv::k_tendencies<S, FAST, true, 8, W, 2, 64>), gr, b, 0, s, A); \
                                 else if (two && shape == SHAPE_63x7) hipLaunchKernelGGL((adv::k_tendencies<S, FAST, true, 7, W, 2, 63>), gr, b, 0, s, A); \
                                 else if (two) hipLaunchKernelGGL((adv::k_tendencies<S, FAST, true, 11, W, 2, 63>), gr, b, 0, s, A); \
                                 else hipLaunchKernelGGL((adv::k_tendencies<S, FAST, true, adv::TY2, W>), gr, b, 0, s, A); } while (0)
    switch (A.scheme) {
        case 1: CSI_ADV_STAGE(1, false); break;
        case 3: CSI_ADV_STAGE(3, W32); break;
        case -3: CSI_ADV_STAGE(-3, false); break;
        case 5: CSI_ADV_STAGE(5, W32); break;
        case -5: CSI_ADV_STAGE(-5, false); break;
        default: CSI_ADV_STAGE(7, W32); break;
    }
#undef CSI_ADV_STAGE
}
// one RK stage of an advection-only model without snow: tendencies of (A.h, A.a), update A.hb + dt G -> A.ho (A.ab, A.ao)
void launch_advect_stage(const AdvDev& A, int mode, hipStream_t s) {
    if (A.w32) { if (mode == 1) launch_stage_mode<true, true>(A, s); else launch_stage_mode<false, true>(A, s); }
    else { if (mode == 1) launch_stage_mode<true, false>(A, s); else launch_stage_mode<false, false>(A, s); }
}
void launch_tracer_step(const AdvDev& A, hipStream_t s) {
    dim3 b(64, 4);
    hipLaunchKernelGGL(adv::k_tracer_step, dim3((unsigned)((A.g.Nx + 63) / 64), (unsigned)((A.g.Ny + 3) / 4)), b, 0, s, A);
}

}  // namespace csi
