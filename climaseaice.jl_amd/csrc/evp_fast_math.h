// evp_fast_math.h -- the per-cell arithmetic of the FAST-mode EVP kernels, as register-level inline
// functions shared by the three-kernel path (evp_fast.hip) and the fused sub-step kernel
// (evp_fused.hip).  Both translation units are compiled with -ffp-contract=off and every fused
// multiply-add is written explicitly, so the two paths execute the SAME floating-point operations in
// the SAME order on the same inputs: their results are bit-identical by construction (tested).
//
// Reference arithmetic restated here (paths relative to /root/reference/src):
//   strain rates            Rheologies/elasto_visco_plastic_rheology.jl:360-375
//   viscosities             :236-273        stresses / alpha   :294-354 (ice_pressure :282-289)
//   stress divergence       Rheologies/ice_stress_divergence.jl:39-51
//   velocity tendencies     SeaIceDynamics/momentum_tendencies_kernel_functions.jl:11-74
//   semi-implicit drag      SeaIceDynamics/sea_ice_external_stress.jl:176-202
//   velocity update         SeaIceDynamics/split_explicit_momentum_equations.jl:197-264
// Metric weights are folded into per-row stencil coefficients (csi_fast_coef.h); this changes rounding
// only (tolerance in DESIGN.md section 6).
#pragma once
#include "csi_dev.h"

namespace csi {
namespace fm {

#define CSI_EPS64 2.220446049250313e-16

__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
// v_max_f64; only used where neither operand can be NaN unless the inputs already are
__device__ __forceinline__ double fmax_(double a, double b) { return __builtin_fmax(a, b); }
// v_min_f64, IEEE minNum (kernels run in IEEE mode): a NaN operand yields the other one
__device__ __forceinline__ double fmin_(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ double clampd(double x, double lo, double hi) { return x > hi ? hi : (x < lo ? lo : x); }
__device__ __forceinline__ double avg2(double a, double b) { return 0.5 * (a + b); }
// y-average of x-averages, 0.5 * (0.5 * (a + b) + 0.5 * (c + d)): the halvings are exact, so one scaling at the end
// gives the same bits with two multiplications fewer (barring subnormal intermediates)
__device__ __forceinline__ double avg4(double a, double b, double c, double d) { return 0.25 * ((a + b) + (c + d)); }
__device__ __forceinline__ double sum2(double a, double b) { return a + b; }
// the same from two x-sums kept from row to row (the row pipelines of the fused kernels)
__device__ __forceinline__ double quarter(double s_a, double s_b) { return 0.25 * (s_a + s_b); }

// Reciprocal and square root without the IEEE special-case scaffolding of the library versions
// (v_div_scale / v_div_fixup, denormal rescaling): hardware seed (v_rcp_f64 / v_rsq_f64, 2^-25 relative error, measured)
// + ONE Newton / Goldschmidt step.  Measured on gfx950 over 120 binades (scripts/microbench/seed_acc.hip,
// profiles/r01_microbenchmarks.md): rcp max 11 ulp / mean 0.7 ulp; 1 / sqrt max 19.5 / mean 1.0 ulp; sqrt max 35.7 / mean
// 1.2 ulp -- i.e. <= 8e-15 relative, against a stated FAST tolerance of 1e-12 on u, v (measured whole-cycle differences to
// STRICT at 2048^2: 2.8e-15 max|u| with these, 1.9e-15 with 0.5-ulp versions).  Round 2 dropped the second refinement
// steps: 34 of 282 VALU instructions per stage-row, +11 % cell-updates/s.  The one place that keeps its residual correction
// is the square root that becomes alpha (sqrt_rsqrt's `s`: 0.5 ulp max), so that the clamp plateaus alpha- / alpha+ are the
// exact numbers the reference stores (sqrt(alpha+^2) == alpha+).
// Arguments are positive finite where the result is used; zero, infinite and NaN arguments may yield NaN, which every
// caller absorbs with the same selects the reference uses (m <= 0 ? 0 : ..., isnan(gamma^2) ? alpha+^2 : ...) or an
// explicit guard.
__device__ __forceinline__ double rcp(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    return fma_(fma_(-x, r, 1.0), r, r);
}
// 1 / sqrt(x) for x > 0: coupled Goldschmidt step on (g ~ sqrt x, h ~ 1 / (2 sqrt x)), h only
__device__ __forceinline__ double rsqrt(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double g = x * y, h = 0.5 * y;
    const double r = fma_(-h, g, 0.5);
    return fma_(y, r, y);          // = 2 (h r + h) exactly (y = 2 h): one multiplication less, the same bits
}
// s = sqrt(x) to 0.5 ulp (coupled step + one residual correction; a second correction changes nothing), rs = 1 / sqrt(x)
// from the coupled step alone
__device__ __forceinline__ void sqrt_rsqrt(double x, double& s, double& rs) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = fma_(-h, g, 0.5);
    g = fma_(g, r, g);
    h = fma_(h, r, h);
    const double d = fma_(-g, g, x);
    s = fma_(d, h, g);
    rs = 2.0 * h;
}
// sqrt(x) from the coupled step alone (the drag norm)
__device__ __forceinline__ double sqrt_fast(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double g = x * y, h = 0.5 * y;
    const double r = fma_(-h, g, 0.5);
    return fma_(g, r, g);
}

// e11 = A (u_e - u_w) + Bn v_n - Bs v_s ; e22 = Cn v_n - Cs v_s        (cell)
// UNI (uniform grid, coefficients are constants): dx does not change with j, so Bn = Bs = 0 and Cn = Cs -- the metric
// terms drop out and each strain rate is one difference times one coefficient.
template <bool UNI>
__device__ __forceinline__ void strain_cell(double A, double Bn, double Bs, double Cn, double Cs,
                                            double u_e, double u_w, double v_n, double v_s, double& e11, double& e22) {
    if (UNI) {
        e11 = A * (u_e - u_w);
        e22 = Cn * (v_n - v_s);
    } else {
        e11 = fma_(A, u_e - u_w, fma_(Bn, v_n, -(Bs * v_s)));
        e22 = fma_(Cn, v_n, -(Cs * v_s));
    }
}
// e12 = Sn u_n - Ss u_s + Sv (v_e - v_w)                                  (corner)
template <bool UNI>
__device__ __forceinline__ double strain_corner(double Sn, double Ss, double Sv, double u_n, double u_s, double v_e, double v_w) {
    if (UNI) return fma_(Sv, v_e - v_w, Sn * (u_n - u_s));          // Sn = Ss
    return fma_(Sv, v_e - v_w, fma_(Sn, u_n, -(Ss * u_s)));
}

struct StressConst {
    double em2, Dmin, Dmin2, rDmin, amin, amax, amin2, amax2, ramin, ramax;
    double hk1;      // (1 - e^-2) / 2
    int pressure_kind;
    double em2_8, Dmin2_16;      // stress_update_s only: e^-2 / 8, 16 Delta_min^2
};

struct StressOut {
    double s11, s22, s12, alpha, zc2, zf2;          // zc2, zf2: TWICE the bulk viscosities (halved where they are stored)
    double xc, rDc;                                 // Delta^2 (floored) and 1 / Delta at the cell: Delta = xc * rDc where it is stored
};

// One stress index (cell (i,j) + corner (i,j)): viscosities and sigma relaxation.
//   e11c, e22c : cell strain rates; e12f : corner strain rate
//   e11f, e22f : 4-point averages of the cell strain rates at the corner; e12c : of the corner rate at the cell
//   Pc, Pf : ice strength at the cell / averaged to the corner; mc, mf : ice mass likewise
//   hkc, hkf : HALF of c_alpha * dt / Az at the cell / corner (the stress phase works with 2 zeta = P / Delta)
//   rmc, rmf : fm::rcp(mc), fm::rcp(mf) -- depend on the ice mass only, which does not change inside a sub-cycle: the pair
//              kernel's consumer wave takes them from its producer (evp_pair_stage.h) instead of forming them again
__device__ __forceinline__ StressOut stress_update_r(const StressConst& k, double e11c, double e22c, double e12f,
                                                     double e11f, double e22f, double e12c, double Pc, double Pf,
                                                     double mc, double mf, double rmc, double rmf, double hkc, double hkf,
                                                     double s11, double s22, double s12) {
    StressOut o;
    const double dc = e11c + e22c, df = e11f + e22f;
    const double tc = e11c - e22c, tf = e11f - e22f;
    const double sc2 = fma_(tc, tc, 4.0 * (e12c * e12c));
    const double sf2 = fma_(tf, tf, 4.0 * (e12f * e12f));
    // Delta = max(sqrt(x), Dmin) and 1 / Delta (evp:265-266, 270-271): sqrt is monotone, so the argument is
    // clamped instead of the result and no select follows
    const double xc = fmax_(fma_(dc, dc, sc2 * k.em2), k.Dmin2), xf = fmax_(fma_(df, df, sf2 * k.em2), k.Dmin2);
    const double rDc = rsqrt(xc), rDf = rsqrt(xf);     // Delta itself is only stored as a diagnostic: xc * rDc
    // zeta = P / (2 Delta), eta = zeta e^-2 (evp:267-272): carried as 2 zeta, 2 eta -- the factors of two cancel
    // against sigma' = 2 eta eps + ((zeta - eta) div - P_r / 2) and fold into the constants hk1, hkc, hkf
    const double zc2 = Pc * rDc, zf2 = Pf * rDf;
    // replacement pressure P Delta / (Delta + Delta_min) (evp:282-289) = P / (1 + Delta_min / Delta): only 1 / Delta is needed
    const double Pr = (k.pressure_kind == 0) ? Pc * rcp(fma_(k.Dmin, rDc, 1.0)) : Pc;
    const double ec2 = zc2 * k.em2, ef2 = zf2 * k.em2;
    const double bulk = fma_(zc2 * k.hk1, dc, -0.5 * Pr);
    const double s11n = fma_(ec2, e11c, bulk);
    const double s22n = fma_(ec2, e22c, bulk);
    const double s12n = ef2 * e12f;
    // gamma = clamp(sqrt(gamma^2), alpha-, alpha+) with NaN -> alpha+ (evp:334-340): gamma^2 is clamped to
    // [alpha-^2, alpha+^2], so the refinement only ever sees finite positive arguments and gamma, 1 / gamma need
    // no select.  The upper side is v_min_f64 (minNum): NaN (0 * inf where there is no ice) and +inf go to alpha+^2,
    // as the reference's isnan(gamma^2) ? alpha+^2 branch does.
    const double g2c_raw = zc2 * hkc * rmc, g2f_raw = zf2 * hkf * rmf;
    const double g2c = fmax_(fmin_(g2c_raw, k.amax2), k.amin2);
    const double g2f = fmax_(fmin_(g2f_raw, k.amax2), k.amin2);
    double gc, rgc, gf, rgf;
    sqrt_rsqrt(g2c, gc, rgc);
    sqrt_rsqrt(g2f, gf, rgf);
    // sigma += (mass > 0) ? (sigma' - sigma) / gamma : 0 (evp:343-347): the mask goes onto 1 / gamma (sigma' is finite
    // where there is no ice: zeta = 0 or the ice strength is finite), and the update is one fused multiply-add
    const double wc = (mc > 0) ? rgc : 0.0, wf = (mf > 0) ? rgf : 0.0;
    o.s11 = fma_(s11n - s11, wc, s11);
    o.s22 = fma_(s22n - s22, wc, s22);
    o.s12 = fma_(s12n - s12, wf, s12);
    o.alpha = gc;
    o.zc2 = zc2; o.zf2 = zf2; o.xc = xc; o.rDc = rDc;
    return o;
}

// The same with the CORNER's inputs as SUMS (the pair kernel's row pipeline, evp_pair_stage.h): every average over four points
// reaches this function unscaled, and the corner strain rate arrives times 8 -- exact powers of two all through, so every
// result has the bits stress_update_r gives, with six multiplications fewer per stress index:
//   S11f, S22f : 4 e11f, 4 e22f (sums of the four cell strain rates around the corner)
//   E12f       : 8 e12f         (strain_corner with its coefficients times 8)
//   y2         : 2 e12c         (1/16 of the sum of the four corners' E12)
//   XP, M4     : 4 Pf, 4 mf     (sums of the four cells' ice strength / mass);  rM4 = rcp(M4) = rmf / 4
//   hkf4       : 4 hkf;   k.em2_8 = e^-2 / 8, k.Dmin2_16 = 16 Delta_min^2
// Delta_f^2 comes out times 16, 1 / Delta_f divided by 4, 2 zeta_f = XP / (4 Delta_f) exactly as before.
__device__ __forceinline__ StressOut stress_update_s(const StressConst& k, double e11c, double e22c, double E12f,
                                                     double S11f, double S22f, double y2, double Pc, double XP,
                                                     double mc, double M4, double rmc, double rM4, double hkc, double hkf4,
                                                     double s11, double s22, double s12) {
    StressOut o;
    const double dc = e11c + e22c, DF = S11f + S22f;
    const double tc = e11c - e22c, TF = S11f - S22f;
    const double sc2 = fma_(tc, tc, y2 * y2);
    const double sf2 = fma_(TF, TF, E12f * E12f);
    const double xc = fmax_(fma_(dc, dc, sc2 * k.em2), k.Dmin2), xf = fmax_(fma_(DF, DF, sf2 * k.em2), k.Dmin2_16);
    const double rDc = rsqrt(xc), rDf = rsqrt(xf);
    const double zc2 = Pc * rDc, zf2 = XP * rDf;
    const double Pr = (k.pressure_kind == 0) ? Pc * rcp(fma_(k.Dmin, rDc, 1.0)) : Pc;
    const double ec2 = zc2 * k.em2, ef2 = zf2 * k.em2_8;
    const double bulk = fma_(zc2 * k.hk1, dc, -0.5 * Pr);
    const double s11n = fma_(ec2, e11c, bulk);
    const double s22n = fma_(ec2, e22c, bulk);
    const double s12n = ef2 * E12f;
    const double g2c_raw = zc2 * hkc * rmc, g2f_raw = zf2 * hkf4 * rM4;
    const double g2c = fmax_(fmin_(g2c_raw, k.amax2), k.amin2);
    const double g2f = fmax_(fmin_(g2f_raw, k.amax2), k.amin2);
    double gc, rgc, gf, rgf;
    sqrt_rsqrt(g2c, gc, rgc);
    sqrt_rsqrt(g2f, gf, rgf);
    const double wc = (mc > 0) ? rgc : 0.0, wf = (M4 > 0) ? rgf : 0.0;
    o.s11 = fma_(s11n - s11, wc, s11);
    o.s22 = fma_(s22n - s22, wc, s22);
    o.s12 = fma_(s12n - s12, wf, s12);
    o.alpha = gc;
    o.zc2 = zc2; o.zf2 = zf2; o.xc = xc; o.rDc = rDc;
    return o;
}

__device__ __forceinline__ StressOut stress_update(const StressConst& k, double e11c, double e22c, double e12f,
                                                   double e11f, double e22f, double e12c, double Pc, double Pf,
                                                   double mc, double mf, double hkc, double hkf,
                                                   double s11, double s22, double s12) {
    return stress_update_r(k, e11c, e22c, e12f, e11f, e22f, e12c, Pc, Pf, mc, mf, rcp(mc), rcp(mf), hkc, hkf, s11, s22, s12);
}

struct VelConst {
    double dt, rdt, fcor, min_mass, min_conc;
    int has_cor;
};

// explicit part `ex` and implicit coefficient `im` of one external stress at a velocity point
//   kind 1: constant tau ; kind 2: tau from an array (passed in `tau`) ; kind 3: SemiImplicitStress with
//   own-component external velocity `we` and cross-component average `webar`
__device__ __forceinline__ void ext_stress(int kind, double tau, double rhoCd, double we, double webar, double w, double wbar,
                                           double& ex, double& im) {
    if (kind == 3) {
        const double d1 = we - w, d2 = webar - wbar;
        const double n2 = fma_(d1, d1, d2 * d2);
        // sqrt(0) through the rsq seed is NaN: the argument is floored at the smallest normal number instead of
        // selecting afterwards (a drag coefficient of 1e-154 rho C_D where the ice moves exactly with the ocean)
        const double n = sqrt_fast(fmax_(n2, 2.2250738585072014e-308));
        im = rhoCd * n;
        ex = im * we;
    } else {            // kind 1 / 2: tau; kind 0 (no stress): the caller passes tau = 0
        ex = tau; im = 0.0;
    }
}

// SemiImplicitStress against an ocean at rest (the reference's default ZeroField ocean velocities): the same bits as
// ext_stress(3, ...) with we = webar = 0 -- (0 - w)^2 = w^2 exactly, ex = im * 0 = +0 -- without the subtractions and
// the explicit part.
__device__ __forceinline__ void ext_stress_rest(double rhoCd, double w, double wbar, double& ex, double& im) {
    const double n2 = fma_(w, w, wbar * wbar);
    im = rhoCd * sqrt_fast(fmax_(n2, 2.2250738585072014e-308));
    ex = 0.0;
}

// Semi-implicit velocity update of one component.
//   w, wn   : this component now and at the start of the stage (u, u^n)
//   mi, ai, abar : ice mass, concentration, alpha averaged to the velocity point
//   div     : stress divergence ; cor : Coriolis acceleration (+f vbar for u, -f ubar for v)
//   ext/imt, exb/imb : explicit / implicit parts of the top and bottom stresses
//   peripheral : the face touches an inactive cell
__device__ __forceinline__ double vel_update_avg(const VelConst& k, double w, double wn, double mi, double ai, double abar,
                                                 double div, double cor, double ext, double imt, double exb, double imb, bool peripheral) {
    const double rm = rcp(mi);
    const double rai = rm * ai;
    const double G = fma_(wn - w, k.rdt, fma_(div, rm, fma_(exb - ext, rai, cor)));
    const double tau_i = (imb - imt) * rai;
    // The reference zeroes G and tau_i where mi <= 0 (momentum_tendencies_kernel_functions.jl:38).  With
    // minimum_mass > 0 (required in FAST mode: fast_supported) such a point is neither active nor marginal, so its
    // result is the final select's 0 (or the free-drift velocity's guard) whatever G was: the two selects are dropped.
    // (w + dtau G) / (1 + dtau tau_i) with dtau = dt / abar, as one quotient: (abar w + dt G) / (abar + dt tau_i)
    const double wD = fma_(k.dt, G, abar * w) * rcp(fma_(k.dt, tau_i, abar));
    const bool active_ice = (mi >= k.min_mass) & (ai >= k.min_conc);
    double res = active_ice ? wD : 0.0;     // free drift `nothing`: marginal ice -> 0 as well
    return peripheral ? 0.0 : res;
}
// the same with a free-drift velocity `wf` for marginal ice (split_explicit_momentum_equations.jl:219-228)
__device__ __forceinline__ double vel_update_avg_fd(const VelConst& k, double w, double wn, double mi, double ai, double abar,
                                                    double div, double cor, double ext, double imt, double exb, double imb,
                                                    bool peripheral, double wf) {
    const double wD = vel_update_avg(k, w, wn, mi, ai, abar, div, cor, ext, imt, exb, imb, false);
    const bool active_ice = (mi >= k.min_mass) & (ai >= k.min_conc);
    const bool marginal = (mi > CSI_EPS64) & (ai > CSI_EPS64);
    const double res = active_ice ? wD : (marginal ? wf : 0.0);
    return peripheral ? 0.0 : res;
}
//   m_a, m_b, a_a, a_b, al_a, al_b : ice mass, concentration, alpha at the two cells the face separates
__device__ __forceinline__ double vel_update(const VelConst& k, double w, double wn, double m_a, double m_b, double a_a, double a_b,
                                             double al_a, double al_b, double div, double cor,
                                             double ext, double imt, double exb, double imb, bool peripheral) {
    return vel_update_avg(k, w, wn, avg2(m_a, m_b), avg2(a_a, a_b), avg2(al_a, al_b), div, cor, ext, imt, exb, imb, peripheral);
}
__device__ __forceinline__ double vel_update_fd(const VelConst& k, double w, double wn, double m_a, double m_b, double a_a, double a_b,
                                                double al_a, double al_b, double div, double cor,
                                                double ext, double imt, double exb, double imb, bool peripheral, double wf) {
    return vel_update_avg_fd(k, w, wn, avg2(m_a, m_b), avg2(a_a, a_b), avg2(al_a, al_b), div, cor, ext, imt, exb, imb, peripheral, wf);
}

// The same from SUMS over the two cells the face separates (m2 = 2 mi, a2 = 2 ai, al2 = 2 abar) and a VelConst whose dt,
// min_mass, min_conc are DOUBLED (rdt is not): every intermediate is vel_update_avg's scaled by an exact power of two
// (also through rcp: seed and Newton step scale exactly), so the result has the same bits -- the three halvings of the
// averages become one doubling of 1 / m.  The row pipelines of the pair kernel (evp_pair_stage.h) use these.
// div2: TWICE the stress divergence (the row pipelines form it with doubled coefficients: div2 / (2 mi) instead of div x (1 / (2 mi) +
// 1 / (2 mi)), one addition less, the same bits)
__device__ __forceinline__ double vel_update_sum(const VelConst& k2, double w, double wn, double m2, double a2, double al2,
                                                 double div2, double cor, double ext, double imt, double exb, double imb, bool peripheral) {
    const double rm2 = rcp(m2);                             // 1 / (2 mi)
    const double rai = rm2 * a2;                            // ai / mi
    const double G = fma_(wn - w, k2.rdt, fma_(div2, rm2, fma_(exb - ext, rai, cor)));
    const double tau_i = (imb - imt) * rai;
    const double wD = fma_(k2.dt, G, al2 * w) * rcp(fma_(k2.dt, tau_i, al2));
    const bool active_ice = (m2 >= k2.min_mass) & (a2 >= k2.min_conc);
    double res = active_ice ? wD : 0.0;
    return peripheral ? 0.0 : res;
}
__device__ __forceinline__ double vel_update_sum_fd(const VelConst& k2, double w, double wn, double m2, double a2, double al2,
                                                    double div2, double cor, double ext, double imt, double exb, double imb,
                                                    bool peripheral, double wf) {
    const double wD = vel_update_sum(k2, w, wn, m2, a2, al2, div2, cor, ext, imt, exb, imb, false);
    const bool active_ice = (m2 >= k2.min_mass) & (a2 >= k2.min_conc);
    const bool marginal = (m2 > 2.0 * CSI_EPS64) & (a2 > 2.0 * CSI_EPS64);
    const double res = active_ice ? wD : (marginal ? wf : 0.0);
    return peripheral ? 0.0 : res;
}

// d_j sigma_1j = E (s11_i - s11_{i-1}) + Fn s12(j+1) - Fs s12(j)          (constant dy)
__device__ __forceinline__ double div1(double E, double Fn, double Fs, double s11_0, double s11_m, double s12_p, double s12_0) {
    return fma_(E, s11_0 - s11_m, fma_(Fn, s12_p, -(Fs * s12_0)));
}
// d_j sigma_2j = Q1n s11(j) + Q2n s22(j) - Q1s s11(j-1) - Q2s s22(j-1) + K (s12(i+1) - s12(i))
// UNI: Q1n = Q1s = 0 and Q2n = Q2s (sigma11 drops out of the v equation when dx does not change with j)
template <bool UNI>
__device__ __forceinline__ double div2(double Q1n, double Q2n, double Q1s, double Q2s, double K,
                                       double s11_0, double s22_0, double s11_m, double s22_m, double s12_p, double s12_0) {
    if (UNI) return fma_(K, s12_p - s12_0, Q2n * (s22_0 - s22_m));
    const double n = fma_(Q1n, s11_0, Q2n * s22_0);
    const double s = fma_(Q1s, s11_m, Q2s * s22_m);
    return fma_(K, s12_p - s12_0, n - s);
}

// ---- orthogonal curvilinear grids (CSI_METRIC_FULL): the operators in terms of the per-point metric planes of
// csi_fast_coef.h (C2_*).  Callers form the products of a velocity or stress with ONE plane value -- Uy = dy_fc u, Ur = u / dy_fc,
// Ux = u / dx_fc at the u points; Vx = dx_cf v, Vr = v / dx_cf, Vy = v / dy_cf at the v points; S = s11 + s22, T = dy_cc^2 (s11 - s22),
// T' = dx_cc^2 (s11 - s22) at the cells; Z = dx_ff^2 s12, Z' = dy_ff^2 s12 at the corners -- each a single operation, so every
// kernel that forms them gets the same bits; these functions fix the order of the rest.
//   e11, e22 = [ d_x(dy u) + d_y(dx v)  +/-  (dy_cc^2 d_x(u / dy) - dx_cc^2 d_y(v / dx)) ] / (2 Az)        (cell)
__device__ __forceinline__ void full_strain_cell(double Uy_e, double Uy_w, double Vx_n, double Vx_s, double Ur_e, double Ur_w,
                                                 double Vr_n, double Vr_s, double dyc2, double dxc2, double razc, double& e11, double& e22) {
    const double D1 = (Uy_e - Uy_w) + (Vx_n - Vx_s);
    const double D2 = fma_(dyc2, Ur_e - Ur_w, -(dxc2 * (Vr_n - Vr_s)));
    const double hr = 0.5 * razc;
    e11 = hr * (D1 + D2);
    e22 = hr * (D1 - D2);
}
//   e12 = [ dx_ff^2 d_y(u / dx) + dy_ff^2 d_x(v / dy) ] / (2 Az)                                              (corner)
__device__ __forceinline__ double full_strain_corner(double Ux_n, double Ux_s, double Vy_e, double Vy_w, double dxf2, double dyf2, double razf) {
    return (0.5 * razf) * fma_(dxf2, Ux_n - Ux_s, dyf2 * (Vy_e - Vy_w));
}
//   d_j sigma_1j = [ dy/2 d_x(s11 + s22) + 1/(2 dy) d_x(dy_cc^2 (s11 - s22)) + 1/dx d_y(dx_ff^2 s12) ] / Az   (u point)
__device__ __forceinline__ double full_div1(double dyu, double rdyu, double rdxu, double razu, double S_e, double S_w, double T_e, double T_w,
                                            double Z_n, double Z_s) {
    return razu * fma_(0.5 * dyu, S_e - S_w, fma_(0.5 * rdyu, T_e - T_w, rdxu * (Z_n - Z_s)));
}
//   d_j sigma_2j = [ dx/2 d_y(s11 + s22) - 1/(2 dx) d_y(dx_cc^2 (s11 - s22)) + 1/dy d_x(dy_ff^2 s12) ] / Az   (v point)
__device__ __forceinline__ double full_div2(double dxv, double rdxv, double rdyv, double razv, double S_n, double S_s, double T_n, double T_s,
                                            double Z_e, double Z_w) {
    return razv * fma_(0.5 * dxv, S_n - S_s, fma_(-0.5 * rdxv, T_n - T_s, rdyv * (Z_e - Z_w)));
}

// the pair kernel's scaled forms (evp_pair_stage.h): 8 e12 and TWICE the divergences -- the halvings inside become exact factors
__device__ __forceinline__ double full_strain_corner8(double Ux_n, double Ux_s, double Vy_e, double Vy_w, double dxf2, double dyf2, double razf) {
    return (4.0 * razf) * fma_(dxf2, Ux_n - Ux_s, dyf2 * (Vy_e - Vy_w));
}
__device__ __forceinline__ double full_div1_x2(double dyu, double rdyu, double rdxu, double razu, double S_e, double S_w, double T_e, double T_w,
                                               double Z_n, double Z_s) {
    return razu * fma_(dyu, S_e - S_w, fma_(rdyu, T_e - T_w, (rdxu + rdxu) * (Z_n - Z_s)));
}
__device__ __forceinline__ double full_div2_x2(double dxv, double rdxv, double rdyv, double razv, double S_n, double S_s, double T_n, double T_s,
                                               double Z_e, double Z_w) {
    return razv * fma_(dxv, S_n - S_s, fma_(-rdxv, T_n - T_s, (rdyv + rdyv) * (Z_e - Z_w)));
}

}  // namespace fm
}  // namespace csi
