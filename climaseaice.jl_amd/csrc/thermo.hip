// thermo.hip -- bare-ice slab thermodynamic step (plumbing: configs 1 and 4 of BASELINE.json).
//
//   k_slab  _ice_thermodynamic_time_step!  SeaIceThermodynamics/thermodynamic_time_step.jl:75-118
//           with thermodynamic_tendency / ice_melt_freeze_tendency
//           (slab_thermodynamics_tendencies.jl:28-135, PrescribedTemperature top BC),
//           ice_volume_update (:304-324), concentration_thermodynamic_step (:358-370),
//           latent_heat (SeaIceThermodynamics.jl:161-170), slab_internal_heat_flux
//           (slab_heat_and_tracer_fluxes.jl:8-19).
// Per-cell, no stencil.  Compiled with -ffp-contract=off, oracle expression order.
#include "csi_dev.h"
#include "csi_kernels.h"

namespace csi {
// Julia's max(a, b) for floats: NaN if either is NaN
__device__ __forceinline__ double jmax(double a, double b) { return (a != a || b != b) ? a + b : (a < b ? b : a); }


__device__ __forceinline__ double latent_heat(const SlabDev& s, double T) {
    return s.L0 + (s.rho_l * s.c_l / s.rho_pure - s.c_i) * (T - s.T0);
}

__global__ void __launch_bounds__(256) k_slab(SlabDev s, GridDev g, FRef h, FRef a, FRef mf, int has_mf, double dt) {
    const int i = 1 + blockIdx.x * blockDim.x + threadIdx.x, j = 1 + blockIdx.y * blockDim.y + threadIdx.y;
    if (i > g.Nx || j > g.Ny) return;
    const double hn = h(i, j), an = a(i, j), hc = s.hc;
    const bool consolidated = hn >= hc;
    const double Tb = s.liq_T0 - s.liq_slope * s.S;
    const double Tu = s.Tu;
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

void launch_slab_step(const SlabDev& S, const GridDev& g, const FRef& h, const FRef& a, const FRef& mf, int has_mf,
                      double dt, hipStream_t s) {
    dim3 b(64, 4);
    hipLaunchKernelGGL(k_slab, dim3((unsigned)((g.Nx + 63) / 64), (unsigned)((g.Ny + 3) / 4)), b, 0, s, S, g, h, a, mf, has_mf, dt);
}

}  // namespace csi
