// evp_strict.hip -- STRICT-mode EVP kernels (CSI_MODE_STRICT).
//
// One HIP kernel per reference @kernel, the reference's operation order in every expression,
// compiled with -ffp-contract=off: results are bit-for-bit those of the CPU oracle, which is
// what anchors the parity of the optimised FAST kernels (evp_fast.hip).
//
//   k_init      _initialize_evp_rhology!   Rheologies/elasto_visco_plastic_rheology.jl:211-219
//   k_visc      _compute_evp_viscosities!  :236-273 (+ strain rates :360-375)
//   k_stress    _compute_evp_stresses!     :294-354 (+ ice_pressure :282-289)
//   k_ustep     _u_velocity_step!          SeaIceDynamics/split_explicit_momentum_equations.jl:197-229
//   k_vstep     _v_velocity_step!          :231-264
// with u/v_velocity_tendency (momentum_tendencies_kernel_functions.jl:11-74), the stress
// divergence (Rheologies/ice_stress_divergence.jl:16-51) and the external stresses
// (sea_ice_external_stress.jl:8-27,176-202) inlined.  The local halo fill that follows each
// velocity kernel in the reference (:180-187) is fused into the store (store_with_images).
#include "csi_dev.h"
#include "csi_kernels.h"

namespace csi {
namespace strict {

// metrics: dxm / dym / azm(g, lx, ly, i, j) in csi_dev.h, called with the location and indices the reference's
// operators use (same calls as oracle/csi_oracle.c)
#define F_ LOC_F
#define C_ LOC_C

// Julia's max(a, b) for floats: NaN if either is NaN (fmax would drop the NaN)
__device__ __forceinline__ double jmax(double a, double b) { return (a != a || b != b) ? a + b : (a < b ? b : a); }

__device__ __forceinline__ double ice_mass(const EvpDev& P, int i, int j) {   // ClimaSeaIce.jl:42
    return P.h(i, j) * P.rho * P.a(i, j);
}

// ---- strain rates, evp:360-375 ----------------------------------------------------------------
__device__ __forceinline__ double eps_D(const EvpDev& P, int i, int j) {
    const GridDev& g = P.g;
    double a = dym(g, F_, C_, i + 1, j) * P.u(i + 1, j) - dym(g, F_, C_, i, j) * P.u(i, j);
    double b = dxm(g, C_, F_, i, j + 1) * P.v(i, j + 1) - dxm(g, C_, F_, i, j) * P.v(i, j);
    return (a + b) / azm(g, C_, C_, i, j);
}
__device__ __forceinline__ double eps_T(const EvpDev& P, int i, int j) {
    const GridDev& g = P.g;
    double dycc = dym(g, C_, C_, i, j), dxcc = dxm(g, C_, C_, i, j);
    double a = P.u(i + 1, j) / dym(g, F_, C_, i + 1, j) - P.u(i, j) / dym(g, F_, C_, i, j);
    double b = P.v(i, j + 1) / dxm(g, C_, F_, i, j + 1) - P.v(i, j) / dxm(g, C_, F_, i, j);
    return ((dycc * dycc) * a - (dxcc * dxcc) * b) / azm(g, C_, C_, i, j);
}
__device__ __forceinline__ double eps_S(const EvpDev& P, int i, int j) {
    const GridDev& g = P.g;
    double dxff = dxm(g, F_, F_, i, j), dyff = dym(g, F_, F_, i, j);
    double a = P.u(i, j) / dxm(g, F_, C_, i, j) - P.u(i, j - 1) / dxm(g, F_, C_, i, j - 1);
    double b = P.v(i, j) / dym(g, C_, F_, i, j) - P.v(i - 1, j) / dym(g, C_, F_, i - 1, j);
    return ((dxff * dxff) * a + (dyff * dyff) * b) / azm(g, F_, F_, i, j);
}
__device__ __forceinline__ double e_xx(const EvpDev& P, int i, int j) { return (eps_D(P, i, j) + eps_T(P, i, j)) / 2; }
__device__ __forceinline__ double e_yy(const EvpDev& P, int i, int j) { return (eps_D(P, i, j) - eps_T(P, i, j)) / 2; }
__device__ __forceinline__ double e_xy(const EvpDev& P, int i, int j) { return eps_S(P, i, j) / 2; }

#define AVG4_FF(fn, P, i, j) ((((fn)(P, (i) - 1, (j) - 1) + (fn)(P, (i), (j) - 1)) / 2 + ((fn)(P, (i) - 1, (j)) + (fn)(P, (i), (j))) / 2) / 2)
#define AVG4_CC(fn, P, i, j) ((((fn)(P, (i), (j)) + (fn)(P, (i) + 1, (j))) / 2 + ((fn)(P, (i), (j) + 1) + (fn)(P, (i) + 1, (j) + 1)) / 2) / 2)
__device__ __forceinline__ double P_at(const EvpDev& P, int i, int j) { return P.P(i, j); }

#define CELL_IJ(r)                                              \
    const int i = (r).i0 + (int)(blockIdx.x * blockDim.x + threadIdx.x); \
    const int j = (r).j0 + (int)(blockIdx.y * blockDim.y + threadIdx.y); \
    if (i > (r).i1 || j > (r).j1) return;

__global__ void k_init(EvpDev P, Range r, FRef un, FRef vn) {
    CELL_IJ(r)
    P.P(i, j) = P.P_star * P.h(i, j) * exp(-P.C_star * (1 - P.a(i, j)));   // ice_strength :219
    un(i, j) = P.u(i, j);
    vn(i, j) = P.v(i, j);
}

__global__ void k_visc(EvpDev P, Range r) {
    CELL_IJ(r)
    const double ie = 1.0 / P.ecc;
    const double em2 = ie * ie;
    const double Dm = P.Dmin;
    double e11c = e_xx(P, i, j);
    double e22c = e_yy(P, i, j);
    double e12f = e_xy(P, i, j);
    double e11f = AVG4_FF(e_xx, P, i, j);
    double e22f = AVG4_FF(e_yy, P, i, j);
    double e12c = AVG4_CC(e_xy, P, i, j);
    double dc = e11c + e22c;
    double df = e11f + e22f;
    double sc = sqrt((e11c - e22c) * (e11c - e22c) + 4 * (e12c * e12c));
    double sf = sqrt((e11f - e22f) * (e11f - e22f) + 4 * (e12f * e12f));
    double Dc = jmax(sqrt(dc * dc + (sc * sc) * em2), Dm);
    double Df = jmax(sqrt(df * df + (sf * sf) * em2), Dm);
    double Pc = P.P(i, j);
    double Pf = AVG4_FF(P_at, P, i, j);
    P.zf(i, j) = Pf / (2 * Df);
    P.zc(i, j) = Pc / (2 * Dc);
    P.Dl(i, j) = Dc;
}

__device__ __forceinline__ double clampd(double x, double lo, double hi) { return x > hi ? hi : (x < lo ? lo : x); }

__global__ void k_stress(EvpDev P, Range r) {
    CELL_IJ(r)
    const GridDev& g = P.g;
    const double ie = 1.0 / P.ecc;
    const double em2 = ie * ie;
    const double ap = P.amax, am = P.amin, ca = P.ca, dt = P.dt;
    double e11 = e_xx(P, i, j);
    double e22 = e_yy(P, i, j);
    double e12 = e_xy(P, i, j);
    double zc = P.zc(i, j);
    double zf = P.zf(i, j);
    double Pr;
    if (P.pressure_kind == 0) {
        double Pc = P.P(i, j), Dc = P.Dl(i, j);
        Pr = Pc * Dc / (Dc + P.Dmin);
    } else {
        Pr = P.P(i, j);
    }
    double etac = zc * em2;
    double etaf = zf * em2;
    double s11n = 2 * etac * e11 + ((zc - etac) * (e11 + e22) - Pr / 2);
    double s22n = 2 * etac * e22 + ((zc - etac) * (e11 + e22) - Pr / 2);
    double s12n = 2 * etaf * e12;
    double mc = ice_mass(P, i, j);
    double mf = AVG4_FF(ice_mass, P, i, j);
    double g2c = zc * ca * dt / mc / azm(g, C_, C_, i, j);
    g2c = isnan(g2c) ? ap * ap : g2c;
    double gc = clampd(sqrt(g2c), am, ap);
    double g2f = zf * ca * dt / mf / azm(g, F_, F_, i, j);
    g2f = isnan(g2f) ? ap * ap : g2f;
    double gf = clampd(sqrt(g2f), am, ap);
    double s11s = (s11n - P.s11(i, j)) / gc;
    double s22s = (s22n - P.s22(i, j)) / gc;
    double s12s = (s12n - P.s12(i, j)) / gf;
    P.s11(i, j) += (mc > 0) ? s11s : 0.0;
    P.s22(i, j) += (mc > 0) ? s22s : 0.0;
    P.s12(i, j) += (mf > 0) ? s12s : 0.0;
    P.al(i, j) = gc;
}

// ---- stress divergence, ice_stress_divergence.jl:16-51 ---------------------------------------
__device__ __forceinline__ double sig11(const EvpDev& P, int i, int j) { return immersed_peripheral_cc(P.g, i, j) ? 0.0 : P.s11(i, j); }
__device__ __forceinline__ double sig22(const EvpDev& P, int i, int j) { return immersed_peripheral_cc(P.g, i, j) ? 0.0 : P.s22(i, j); }
__device__ __forceinline__ double sig12(const EvpDev& P, int i, int j) { return immersed_peripheral_ff(P.g, i, j) ? 0.0 : P.s12(i, j); }
__device__ __forceinline__ double sigD(const EvpDev& P, int i, int j) { return sig11(P, i, j) + sig22(P, i, j); }
__device__ __forceinline__ double sigT(const EvpDev& P, int i, int j) { return sig11(P, i, j) - sig22(P, i, j); }

__device__ __forceinline__ double div_sigma_1(const EvpDev& P, int i, int j) {   // :39-44
    const GridDev& g = P.g;
    double dyfc = dym(g, F_, C_, i, j);
    double d = dyfc * (sigD(P, i, j) - sigD(P, i - 1, j)) / 2;
    double dyc = dym(g, C_, C_, i, j), dycm = dym(g, C_, C_, i - 1, j);
    double T = ((dyc * dyc) * sigT(P, i, j) - (dycm * dycm) * sigT(P, i - 1, j)) / dyfc / 2;
    double dxfn = dxm(g, F_, F_, i, j + 1), dxf = dxm(g, F_, F_, i, j);
    double S = ((dxfn * dxfn) * sig12(P, i, j + 1) - (dxf * dxf) * sig12(P, i, j)) / dxm(g, F_, C_, i, j);
    return (d + T + S) / azm(g, F_, C_, i, j);
}
__device__ __forceinline__ double div_sigma_2(const EvpDev& P, int i, int j) {   // :46-51
    const GridDev& g = P.g;
    double dxcf = dxm(g, C_, F_, i, j);
    double d = dxcf * (sigD(P, i, j) - sigD(P, i, j - 1)) / 2;
    double dxc = dxm(g, C_, C_, i, j), dxcm = dxm(g, C_, C_, i, j - 1);
    double T = -((dxc * dxc) * sigT(P, i, j) - (dxcm * dxcm) * sigT(P, i, j - 1)) / dxcf / 2;
    double dyfn = dym(g, F_, F_, i + 1, j), dyf = dym(g, F_, F_, i, j);
    double S = ((dyfn * dyfn) * sig12(P, i + 1, j) - (dyf * dyf) * sig12(P, i, j)) / dym(g, C_, F_, i, j);
    return (d + T + S) / azm(g, C_, F_, i, j);
}

// ---- external stresses, sea_ice_external_stress.jl:8-27,176-202 ------------------------------
__device__ __forceinline__ double ext_ue(const StressDev& s, int i, int j) {
    return s.ue_kind == 2 ? s.fu(i, j) : (s.ue_kind == 1 ? s.ue : 0.0);
}
__device__ __forceinline__ double ext_ve(const StressDev& s, int i, int j) {
    return s.ve_kind == 2 ? s.fv(i, j) : (s.ve_kind == 1 ? s.ve : 0.0);
}
#define AVG4_FC(X) (((X(i - 1, j) + X(i, j)) / 2 + (X(i - 1, j + 1) + X(i, j + 1)) / 2) / 2)
#define AVG4_CF(X) (((X(i, j - 1) + X(i + 1, j - 1)) / 2 + (X(i, j) + X(i + 1, j)) / 2) / 2)

__device__ __forceinline__ double drag_norm_u(const EvpDev& P, const StressDev& s, int i, int j) {
    double du = ext_ue(s, i, j) - P.u(i, j);
#define VE_(ii, jj) ext_ve(s, ii, jj)
#define V_(ii, jj) P.v(ii, jj)
    double dv = AVG4_FC(VE_) - AVG4_FC(V_);
#undef VE_
#undef V_
    return sqrt(du * du + dv * dv);
}
__device__ __forceinline__ double drag_norm_v(const EvpDev& P, const StressDev& s, int i, int j) {
    double dv = ext_ve(s, i, j) - P.v(i, j);
#define UE_(ii, jj) ext_ue(s, ii, jj)
#define U_(ii, jj) P.u(ii, jj)
    double du = AVG4_CF(UE_) - AVG4_CF(U_);
#undef UE_
#undef U_
    return sqrt(du * du + dv * dv);
}
__device__ __forceinline__ double explicit_tau_x(const EvpDev& P, const StressDev& s, int i, int j) {
    switch (s.kind) {
        case 1: return s.tau_u;
        case 2: return s.fu(i, j);
        case 3: return s.rho_e * s.Cd * drag_norm_u(P, s, i, j) * ext_ue(s, i, j);
        default: return 0.0;
    }
}
__device__ __forceinline__ double explicit_tau_y(const EvpDev& P, const StressDev& s, int i, int j) {
    switch (s.kind) {
        case 1: return s.tau_v;
        case 2: return s.fv(i, j);
        case 3: return s.rho_e * s.Cd * drag_norm_v(P, s, i, j) * ext_ve(s, i, j);
        default: return 0.0;
    }
}
// ---- StressBalanceFreeDrift closed forms, stress_balance_free_drift.jl:61-129 --------------------------------
// exactly one stress is a SemiImplicitStress (checked on the host); the other one gives tau.  The result depends on
// the forcing only, so it is evaluated once per sub-cycle into ufd / vfd (strict arithmetic in both modes).
__device__ __forceinline__ double free_drift_u(const EvpDev& P, int i, int j) {
    const StressDev& semi = P.bot.kind == 3 ? P.bot : P.top;
    const StressDev& expl = P.bot.kind == 3 ? P.top : P.bot;
    const double tx = explicit_tau_x(P, expl, i, j);
#define TY_(ii, jj) explicit_tau_y(P, expl, ii, jj)
    const double ty = AVG4_FC(TY_);
#undef TY_
    const double t = sqrt(tx * tx + ty * ty);
    const double C = semi.rho_e * semi.Cd;
    return ext_ue(semi, i, j) - ((t == 0) ? t : tx / sqrt(C * t));
}
__device__ __forceinline__ double free_drift_v(const EvpDev& P, int i, int j) {
    const StressDev& semi = P.bot.kind == 3 ? P.bot : P.top;
    const StressDev& expl = P.bot.kind == 3 ? P.top : P.bot;
#define TX_(ii, jj) explicit_tau_x(P, expl, ii, jj)
    const double tx = AVG4_CF(TX_);
#undef TX_
    const double ty = explicit_tau_y(P, expl, i, j);
    const double t = sqrt(tx * tx + ty * ty);
    const double C = semi.rho_e * semi.Cd;
    return ext_ve(semi, i, j) - ((t == 0) ? t : ty / sqrt(C * t));
}

__device__ __forceinline__ double implicit_tau_x(const EvpDev& P, const StressDev& s, int i, int j) {
    return s.kind == 3 ? s.rho_e * s.Cd * drag_norm_u(P, s, i, j) : 0.0;
}
__device__ __forceinline__ double implicit_tau_y(const EvpDev& P, const StressDev& s, int i, int j) {
    return s.kind == 3 ? s.rho_e * s.Cd * drag_norm_v(P, s, i, j) : 0.0;
}

#define EPS64 2.220446049250313e-16

__global__ void k_ustep(EvpDev P, Range r, ImageSpec im) {
    CELL_IJ(r)
    const double dt = P.dt;
    double mi = (ice_mass(P, i - 1, j) + ice_mass(P, i, j)) / 2;
    double ai = (P.a(i - 1, j) + P.a(i, j)) / 2;
    double abar = (P.al(i - 1, j) + P.al(i, j)) / 2;
    double dtau = dt / abar;
    // u_velocity_tendency, momentum_tendencies_kernel_functions.jl:11-41
    double cor = 0.0;
    if (P.has_cor) {
#define V_(ii, jj) P.v(ii, jj)
        cor = -fcor_at_u(P, i, j) * AVG4_FC(V_);     // FPlane / BetaPlane (f at this row's u points) / per-point f
#undef V_
    }
    const double user = P.has_forcing ? P.forcing_u(i, j) : 0.0;    // model.forcing.u as an array
    double forcing = user + (P.un(i, j) - P.u(i, j)) / dtau / abar;  // sum_of_forcing_u, evp:391-395
    double imm = immersed_div_sigma_1(P, i, j) / mi;                  // zero(grid) / FluxBoundaryCondition numbers, isd:57-85
    double G = (-cor
                - explicit_tau_x(P, P.top, i, j) / mi * ai
                + explicit_tau_x(P, P.bot, i, j) / mi * ai
                + div_sigma_1(P, i, j) / mi
                + imm
                + forcing);
    G = (mi <= 0) ? 0.0 : G;
    double tau_i = (implicit_tau_x(P, P.bot, i, j) - implicit_tau_x(P, P.top, i, j)) / mi * ai;
    tau_i = (mi <= 0) ? 0.0 : tau_i;
    double uD = (P.u(i, j) + dtau * G) / (1 + dtau * tau_i);
    double uF = P.free_drift ? P.ufd(i, j) : 0.0;                       // free_drift_u, :219
    bool marginal = (mi > EPS64) & (ai > EPS64);
    bool active_ice = (mi >= P.min_mass) & (ai >= P.min_conc);
    // `... * active` with a Julia Bool: false is a strong zero (sign kept), :228
    double sel = active_ice ? uD : (marginal ? uF : 0.0);
    double res = peripheral_u(P.g, i, j) ? copysign(0.0, sel) : sel;
    store_with_images(P.u, P.g, im, i, j, res);
}

__global__ void k_vstep(EvpDev P, Range r, ImageSpec im) {
    CELL_IJ(r)
    const double dt = P.dt;
    double mi = (ice_mass(P, i, j - 1) + ice_mass(P, i, j)) / 2;
    double ai = (P.a(i, j - 1) + P.a(i, j)) / 2;
    double abar = (P.al(i, j - 1) + P.al(i, j)) / 2;
    double dtau = dt / abar;
    double cor = 0.0;
    if (P.has_cor) {
#define U_(ii, jj) P.u(ii, jj)
        cor = fcor_at_v(P, i, j) * AVG4_CF(U_);
#undef U_
    }
    const double user = P.has_forcing ? P.forcing_v(i, j) : 0.0;
    double forcing = user + (P.vn(i, j) - P.v(i, j)) / dtau / abar;
    double imm = immersed_div_sigma_2(P, i, j) / mi;
    double G = (-cor
                - explicit_tau_y(P, P.top, i, j) / mi * ai
                + explicit_tau_y(P, P.bot, i, j) / mi * ai
                + div_sigma_2(P, i, j) / mi
                + imm
                + forcing);
    G = (mi <= 0) ? 0.0 : G;
    double tau_i = (implicit_tau_y(P, P.bot, i, j) - implicit_tau_y(P, P.top, i, j)) / mi * ai;
    tau_i = (mi <= 0) ? 0.0 : tau_i;
    double vD = (P.v(i, j) + dtau * G) / (1 + dtau * tau_i);
    double vF = P.free_drift ? P.vfd(i, j) : 0.0;
    bool marginal = (mi > EPS64) & (ai > EPS64);
    bool active_ice = (mi >= P.min_mass) & (ai >= P.min_conc);
    double sel = active_ice ? vD : (marginal ? vF : 0.0);
    double res = peripheral_v(P.g, i, j) ? copysign(0.0, sel) : sel;
    store_with_images(P.v, P.g, im, i, j, res);
}

}  // namespace strict

static inline dim3 grid_for(const Range& r, dim3 b) {
    return dim3((unsigned)((r.i1 - r.i0 + 1 + b.x - 1) / b.x), (unsigned)((r.j1 - r.j0 + 1 + b.y - 1) / b.y), 1);
}

namespace strict {
__global__ void __launch_bounds__(256) k_free_drift(EvpDev P, Range r) {
    const int i = r.i0 + (int)(blockIdx.x * blockDim.x + threadIdx.x), j = r.j0 + (int)(blockIdx.y * blockDim.y + threadIdx.y);
    if (i > r.i1 || j > r.j1) return;
    P.ufd(i, j) = free_drift_u(P, i, j);
    P.vfd(i, j) = free_drift_v(P, i, j);
}
}  // namespace strict

void launch_free_drift(const EvpDev& P, const Range& r, hipStream_t s) {
    dim3 b(64, 4);
    hipLaunchKernelGGL(strict::k_free_drift, grid_for(r, b), b, 0, s, P, r);
}
void launch_strict_init(const EvpDev& P, const Range& r, hipStream_t s) {
    dim3 b(64, 4);
    hipLaunchKernelGGL(strict::k_init, grid_for(r, b), b, 0, s, P, r, P.un, P.vn);
}
void launch_strict_visc(const EvpDev& P, const Range& r, hipStream_t s) {
    dim3 b(64, 4);
    hipLaunchKernelGGL(strict::k_visc, grid_for(r, b), b, 0, s, P, r);
}
void launch_strict_stress(const EvpDev& P, const Range& r, hipStream_t s) {
    dim3 b(64, 4);
    hipLaunchKernelGGL(strict::k_stress, grid_for(r, b), b, 0, s, P, r);
}
void launch_strict_ustep(const EvpDev& P, const Range& r, const ImageSpec& im, hipStream_t s) {
    dim3 b(64, 4);
    hipLaunchKernelGGL(strict::k_ustep, grid_for(r, b), b, 0, s, P, r, im);
}
void launch_strict_vstep(const EvpDev& P, const Range& r, const ImageSpec& im, hipStream_t s) {
    dim3 b(64, 4);
    hipLaunchKernelGGL(strict::k_vstep, grid_for(r, b), b, 0, s, P, r, im);
}

}  // namespace csi
