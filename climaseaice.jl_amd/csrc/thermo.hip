// thermo.hip -- bare-ice slab thermodynamic step (plumbing: configs 1 and 4 of BASELINE.json).
//
//   k_slab  _ice_thermodynamic_time_step!  SeaIceThermodynamics/thermodynamic_time_step.jl:75-118
//           with thermodynamic_tendency / ice_melt_freeze_tendency
//           (slab_thermodynamics_tendencies.jl:28-135, PrescribedTemperature top BC),
//           ice_volume_update (:304-324), concentration_thermodynamic_step (:358-370),
//           latent_heat (SeaIceThermodynamics.jl:161-170), slab_internal_heat_flux
//           (slab_heat_and_tracer_fluxes.jl:8-19).
//   k_layered  _layered_thermodynamic_time_step!  thermodynamic_time_step.jl:131-298 (snow on ice, resistors in
//           series: ice_snow_conductive_flux / interface_temperature, slab_heat_and_tracer_fluxes.jl:47-84;
//           snow_accumulation, snow_ice_formation :331-353).
// Top boundary conditions: PrescribedTemperature, or MeltingConstrainedFluxBalance with a numeric external flux (the
// secant solve's root in closed form, see include/csi.h).
// Per-cell, no stencil.  Compiled with -ffp-contract=off, oracle expression order.
#include "csi_dev.h"
#include "csi_kernels.h"

namespace csi {
// Julia's max(a, b) for floats: NaN if either is NaN
__device__ __forceinline__ double jmax(double a, double b) { return (a != a || b != b) ? a + b : (a < b ? b : a); }
__device__ __forceinline__ double jmin(double a, double b) { return (a != a || b != b) ? a + b : (b < a ? b : a); }


__device__ __forceinline__ double latent_heat(const SlabDev& s, double T) {
    return s.L0 + (s.rho_l * s.c_l / s.rho_pure - s.c_i) * (T - s.T0);
}

__global__ void __launch_bounds__(256) k_slab(SlabDev s, GridDev g, FRef h, FRef a, FRef mf, int has_mf, double dt) {
    const int i = 1 + blockIdx.x * blockDim.x + threadIdx.x, j = 1 + blockIdx.y * blockDim.y + threadIdx.y;
    if (i > g.Nx || j > g.Ny) return;
    const double hn = h(i, j), an = a(i, j), hc = s.hc;
    const bool consolidated = hn >= hc;
    const double Tb = s.liq_T0 - s.liq_slope * s.S;
    double Tu = s.Tu;
    if (s.top_bc_kind == 1) {      // MeltingConstrainedFluxBalance: root of Qx - Qi(T), capped at Tm(S_ice); thin slab: Tb
        const double Tm = s.liq_T0 - s.liq_slope * s.ice_salinity;
        Tu = consolidated ? jmin(Tb - s.Qu * hn / s.k, Tm) : Tb;
    }
    const double Eb = s.rho_bulk * latent_heat(s, Tb);
    const double Eu = s.rho_bulk * latent_heat(s, Tu);
    const double Qi_fun = (hn <= 0) ? 0.0 : -s.k * (Tu - Tb) / hn;
    const double Qu = (s.top_flux_kind == 1) ? Qi_fun : s.Qu;
    const double Qb = (s.bot_flux_kind == 1) ? (-(1 - an)) * s.Qb : s.Qb;
    const double Qi = consolidated ? Qi_fun : 0.0;
    const double wu = (Qu - Qi) / Eu;
    const double wb = (Qi - Qb) / Eb;
    double dtV = wu + wb;
    double V1 = hn * an + dt * dtV;
    V1 = jmax(0.0, V1);
    dtV = (V1 - hn * an) / dt;
    // `x * flag` with a Julia Bool: false is a strong zero (NaN * false == 0, sign kept)
    const bool freezing = (dtV >= 0), melting = (dtV < 0);
    const double xf = (1 - an) / hc * dtV, xm = an / (2 * hn) * dtV;
    const double daf = freezing ? xf : copysign(0.0, xf);
    const double dam = melting ? xm : copysign(0.0, xm);
    double ap = an + dt * (daf + dam);
    ap = jmax(0.0, ap);
    double hp = V1 / ap;
    hp = (ap <= 0) ? 0.0 : hp;
    ap = (dtV == 0) ? an : ap;
    hp = (dtV == 0) ? hn : hp;
    ap = (hp == 0) ? 0.0 : ap;
    hp = (ap == 0) ? 0.0 : hp;
    const double a1 = (ap > 1) ? 1.0 : ap;
    const double h1 = (ap > 1) ? hp * ap : hp;
    a(i, j) = a1;
    h(i, j) = h1;
    if (has_mf) mf(i, j) = s.rho_bulk * (h1 * a1 - hn * an) / dt;
}

// ice_volume_update, thermodynamic_time_step.jl:304-324 (+ concentration_thermodynamic_step :358-370)
__device__ __forceinline__ void ice_volume_update(double dtV, double hn, double an, double hc, double dt, double& h1, double& a1) {
    double V1 = hn * an + dt * dtV;
    V1 = jmax(0.0, V1);
    dtV = (V1 - hn * an) / dt;
    const bool freezing = (dtV >= 0), melting = (dtV < 0);
    const double xf = (1 - an) / hc * dtV, xm = an / (2 * hn) * dtV;
    const double daf = freezing ? xf : copysign(0.0, xf);
    const double dam = melting ? xm : copysign(0.0, xm);
    double ap = an + dt * (daf + dam);
    ap = jmax(0.0, ap);
    double hp = V1 / ap;
    hp = (ap <= 0) ? 0.0 : hp;
    ap = (dtV == 0) ? an : ap;
    hp = (dtV == 0) ? hn : hp;
    ap = (hp == 0) ? 0.0 : ap;
    hp = (ap == 0) ? 0.0 : hp;
    a1 = (ap > 1) ? 1.0 : ap;
    h1 = (ap > 1) ? hp * ap : hp;
}

__global__ void __launch_bounds__(256) k_layered(SlabDev s, SnowDev w, GridDev g, FRef h, FRef a, FRef hs, LayeredOut o, double dt) {
    const int i = 1 + blockIdx.x * blockDim.x + threadIdx.x, j = 1 + blockIdx.y * blockDim.y + threadIdx.y;
    if (i > g.Nx || j > g.Ny) return;
    const double hin = h(i, j), an = a(i, j), hc = s.hc;
    double hsn = hs(i, j);
    const double Vin = hin * an, Vsn = hsn * an;
    const bool consolidated = hin >= hc;
    const double Tb = s.liq_T0 - s.liq_slope * s.S;
    double Tm = s.liq_T0 - s.liq_slope * s.ice_salinity;
    const double ks = w.k, ki = s.k;
    const double Qu = s.Qu;
    Tm = (hsn > 0) ? 0.0 : Tm;
    const double R = hsn / ks + hin / ki;
    double Tus = w.Tu;
    if (w.top_bc_kind == 1) Tus = consolidated ? jmin(Tb - Qu * R, Tm) : Tb;
    const double Ri = hin / ki, Rs = hsn / ks, Rt = Rs + Ri;
    const double Tsi = (Rt <= 0) ? Tb : Tb + (Tus - Tb) * Ri / Rt;
    const double Qic = (R <= 0) ? 0.0 : (Tb - Tus) / R;
    const double Qis = consolidated ? Qic : 0.0;
    const double Qui = Qu;
    const double Qui_per_ice = (an > 0) ? Qui / an : 0.0;
    const double dQ = Qui_per_ice - Qis;
    const double melt_energy = jmax(0.0, -dQ);
    const double rs = w.rho, Ls = s.L0;
    const double cap = rs * Ls * hsn / dt;
    const double Qs = jmin(melt_energy, cap);
    const double Gsm = Qs / (rs * Ls);
    const double ri = s.rho_bulk, riL = ri * Ls;
    const double Qbi = (s.bot_flux_kind == 1) ? (-(1 - an)) * s.Qb : s.Qb;
    const double alpha = (Qui - Qbi) / riL, beta = Qs / riL;
    const double Cm = (hin > 0) ? an / (2 * hin) : 0.0;
    const double Cf = (hc > 0) ? (1 - an) / hc : 0.0;
    const double Km = dt * Cm, Kf = dt * Cf;
    const double eps = 2.220446049250313e-16;
    const double Dm = 1 - Km * beta, Df = 1 - Kf * beta;
    const double am = (fabs(Dm) > eps) ? (an + Km * alpha) / Dm : an + Km * alpha;
    const double af = (fabs(Df) > eps) ? (an + Kf * alpha) / Df : an + Kf * alpha;
    const double dtVm = alpha + beta * am;
    const bool melting = dtVm < 0;
    const double atmp = melting ? am : af;
    const double Qeff = Qui + Qs * atmp;
    const double Eb = ri * latent_heat(s, Tb), Eu = ri * latent_heat(s, Tsi);
    const double Qii_fun = (hin <= 0) ? 0.0 : -ki * (Tsi - Tb) / hin;
    const double Qii = consolidated ? Qii_fun : 0.0;
    const double wu = (Qeff - Qii) / Eu, wb = (Qii - Qbi) / Eb;
    double hi1, a1;
    ice_volume_update(wu + wb, hin, an, hc, dt, hi1, a1);
    hsn = (a1 > 0) ? hsn * an / a1 : 0.0;
    const double Gsp = (a1 > 0) ? w.snowfall / rs : 0.0;
    double hs1 = hsn + dt * (Gsp - Gsm);
    hs1 = jmax(0.0, hs1);
    {
        const double rw = s.rho_l;
        const double hf = hi1 * (1 - ri / rw) - hs1 * rs / rw;
        double dhs = (hf < 0) ? -hf * ri / rs : 0.0;
        const double hsp = jmax(0.0, hs1 - dhs);
        dhs = hs1 - hsp;
        hi1 = hi1 + dhs * rs / ri;
        hs1 = hsp;
    }
    hs1 = (a1 <= 0) ? 0.0 : hs1;
    a(i, j) = a1; h(i, j) = hi1; hs(i, j) = hs1;
    const double Pabs = rs * Gsp * a1;
    if (o.mf_ice.p) o.mf_ice(i, j) = ri * (hi1 * a1 - Vin) / dt;
    if (o.mf_snow.p) o.mf_snow(i, j) = rs * (hs1 * a1 - Vsn) / dt - Pabs;
    if (o.mf_int.p) o.mf_int(i, j) = Pabs;
    if (o.tu_ice.p) o.tu_ice(i, j) = Tsi;
    if (o.tu_snow.p) o.tu_snow(i, j) = Tus;
}

void launch_layered_step(const SlabDev& S, const SnowDev& W, const GridDev& g, const FRef& h, const FRef& a, const FRef& hs,
                         const LayeredOut& o, double dt, hipStream_t s) {
    dim3 b(64, 4);
    hipLaunchKernelGGL(k_layered, dim3((unsigned)((g.Nx + 63) / 64), (unsigned)((g.Ny + 3) / 4)), b, 0, s, S, W, g, h, a, hs, o, dt);
}

void launch_slab_step(const SlabDev& S, const GridDev& g, const FRef& h, const FRef& a, const FRef& mf, int has_mf,
                      double dt, hipStream_t s) {
    dim3 b(64, 4);
    hipLaunchKernelGGL(k_slab, dim3((unsigned)((g.Nx + 63) / 64), (unsigned)((g.Ny + 3) / 4)), b, 0, s, S, g, h, a, mf, has_mf, dt);
}

}  // namespace csi
