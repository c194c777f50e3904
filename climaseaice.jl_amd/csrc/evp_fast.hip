// evp_fast.hip -- FAST-mode EVP kernels (CSI_MODE_FAST): the benchmarked path.
//
// Three globally ordered phases per sub-step (stress -> first velocity -> second velocity):
//   k_stress  = _compute_evp_viscosities! + _compute_evp_stresses! fused
//               (Rheologies/elasto_visco_plastic_rheology.jl:236-273, 294-354): zeta, Delta never
//               round-trip through HBM (they are stored only on the last sub-step, as the
//               reference leaves them); strain rates are evaluated once per needed location.
//   k_ustep / k_vstep = _u_velocity_step! / _v_velocity_step!
//               (SeaIceDynamics/split_explicit_momentum_equations.jl:197-264) with the tendency
//               (momentum_tendencies_kernel_functions.jl:11-74), the stress divergence
//               (Rheologies/ice_stress_divergence.jl:39-51), the external stresses
//               (sea_ice_external_stress.jl:176-202) and the local halo fill (:180-187) fused.
//
// Arithmetic: metric weights are folded on the host into per-row stencil coefficients
// (FastCoef, csi_fast_coef.h); divisions by metrics become multiplications and the strain /
// divergence operators become short linear stencils.  This changes rounding only: the
// tolerance against STRICT / the oracle is stated in DESIGN.md and enforced in tests/.
//
// Algorithmic HBM bytes per cell-update (fp64, constant forcing): stress 96 B (reads u, v, P, h,
// aice, sigma x3; writes sigma x3, alpha) + 80 B per velocity step = 256 B.
#include "csi_dev.h"
#include "csi_kernels.h"
#include "csi_fast_coef.h"
#include "evp_fast_math.h"

namespace csi {
namespace fast {

template <bool UNI>
__device__ __forceinline__ double coef(const FastCoef& c, int which, int j) {
    if (UNI) return c.uni[which];
    return c.vec[(long)j * c.stride + which];
}

// Block -> tile mapping.  Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 shares an
// XCD, MI355X_MICROARCH.md "Workgroup dispatch"); each XCD has its own L2, so a tile's halo rows
// and the cache lines it shares with its x-neighbour are re-fetched through the fabric unless the
// neighbouring tile runs on the same XCD.  The remap gives every XCD one contiguous band of tile
// rows, walked in x-fastest order (speed only: any placement gives the same results).  Waves start
// at the parent row start (i = 1 - Hx), which is 64-byte aligned for even ld.
struct TileMap {
    int gx, ntiles, per_xcd;
    int ibase, jbase;
    int ty;              // rows per block (blockDim.y): TILE_Y, or 1 on short ranges (the fold band: see tile_map)
};
constexpr int TILE_X = 64, TILE_Y = 4;

// One-wave workgroups are the fold band's launches (tile_map): they run BESIDE a pair launch and are the sub-cycle's critical
// path on tripolar grids (six dependent launches per pair of sub-steps) -- their waves take the top issue priority.
#ifndef CSI_BAND_PRIO
#define CSI_BAND_PRIO 3
#endif
#define CELL_IJ(r, tm)                                                          \
    const int b_ = (int)blockIdx.x;                                             \
    const int t_ = (b_ & 7) * (tm).per_xcd + (b_ >> 3);                         \
    if ((b_ >> 3) >= (tm).per_xcd || t_ >= (tm).ntiles) return;                 \
    const int by_ = t_ / (tm).gx, bx_ = t_ - by_ * (tm).gx;                     \
    const int i = (tm).ibase + bx_ * TILE_X + (int)threadIdx.x;                 \
    const int j = __builtin_amdgcn_readfirstlane((tm).jbase + by_ * (tm).ty + (int)threadIdx.y); \
    if (i < (r).i0 || i > (r).i1 || j > (r).j1) return;                         \
    if (CSI_BAND_PRIO && (tm).ty == 1) __builtin_amdgcn_s_setprio(CSI_BAND_PRIO);

__global__ void __launch_bounds__(256) k_init(EvpDev P, Range r) {
    const int i = r.i0 + (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int j = r.j0 + (int)(blockIdx.y * blockDim.y + threadIdx.y);
    if (i > r.i1 || j > r.j1) return;
    P.P(i, j) = P.P_star * P.h(i, j) * exp(-P.C_star * (1 - P.a(i, j)));   // ice_strength, evp:219
    P.un(i, j) = P.u(i, j);
    P.vn(i, j) = P.v(i, j);
}

// ------------------------------------------------------------------------------------------------
// stress phase
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ fm::StressConst stress_const(const EvpDev& P, const FastCoef& c) {
    fm::StressConst k;
    k.em2 = c.em2; k.Dmin = P.Dmin; k.Dmin2 = c.Dmin2; k.rDmin = c.rDmin;
    k.amin = P.amin; k.amax = P.amax; k.amin2 = c.amin2; k.amax2 = c.amax2;
    k.ramin = c.ramin; k.ramax = c.ramax; k.hk1 = c.hk1;
    k.pressure_kind = P.pressure_kind;
    return k;
}
__device__ __forceinline__ fm::VelConst vel_const(const EvpDev& P, const FastCoef& c) {
    fm::VelConst k;
    k.dt = P.dt; k.rdt = c.rdt; k.fcor = P.fcor; k.min_mass = P.min_mass; k.min_conc = P.min_conc; k.has_cor = P.has_cor;
    return k;
}

template <bool UNI>
__global__ void __launch_bounds__(256) k_stress(EvpDev P, Range r, FastCoef c, TileMap tm) {
    CELL_IJ(r, tm)
    // u, v neighbourhood: u[i-1..i+1][j-1..j+1] (no (i-1, j+1)), v likewise (no (i+1, j-1))
    const double u_mm = P.u(i - 1, j - 1), u_0m = P.u(i, j - 1), u_pm = P.u(i + 1, j - 1);
    const double u_m0 = P.u(i - 1, j),     u_00 = P.u(i, j),     u_p0 = P.u(i + 1, j);
    const double                           u_0p = P.u(i, j + 1), u_pp = P.u(i + 1, j + 1);
    const double v_mm = P.v(i - 1, j - 1), v_0m = P.v(i, j - 1);
    const double v_m0 = P.v(i - 1, j),     v_00 = P.v(i, j),     v_p0 = P.v(i + 1, j);
    const double v_mp = P.v(i - 1, j + 1), v_0p = P.v(i, j + 1), v_pp = P.v(i + 1, j + 1);
    const double P_mm = P.P(i - 1, j - 1), P_0m = P.P(i, j - 1), P_m0 = P.P(i - 1, j), P_00 = P.P(i, j);
    const double h_mm = P.h(i - 1, j - 1), h_0m = P.h(i, j - 1), h_m0 = P.h(i - 1, j), h_00 = P.h(i, j);
    const double a_mm = P.a(i - 1, j - 1), a_0m = P.a(i, j - 1), a_m0 = P.a(i - 1, j), a_00 = P.a(i, j);
    const double s11 = P.s11(i, j), s22 = P.s22(i, j), s12 = P.s12(i, j);

    // centre-row coefficients for rows j-1 and j; corner-row coefficients for rows j and j+1
    const double A0 = coef<UNI>(c, FC_A, j), Bn0 = coef<UNI>(c, FC_BN, j), Bs0 = coef<UNI>(c, FC_BS, j),
                 Cn0 = coef<UNI>(c, FC_CN, j), Cs0 = coef<UNI>(c, FC_CS, j);
    const double Am = coef<UNI>(c, FC_A, j - 1), Bnm = coef<UNI>(c, FC_BN, j - 1), Bsm = coef<UNI>(c, FC_BS, j - 1),
                 Cnm = coef<UNI>(c, FC_CN, j - 1), Csm = coef<UNI>(c, FC_CS, j - 1);
    const double Sn0 = coef<UNI>(c, FC_SN, j), Ss0 = coef<UNI>(c, FC_SS, j), Sv0 = coef<UNI>(c, FC_SV, j);
    const double Snp = coef<UNI>(c, FC_SN, j + 1), Ssp = coef<UNI>(c, FC_SS, j + 1), Svp = coef<UNI>(c, FC_SV, j + 1);

    // strain rates at the four cells (i-1..i, j-1..j) and the four corners (i..i+1, j..j+1)
    double e11_00, e22_00, e11_m0, e22_m0, e11_0m, e22_0m, e11_mm, e22_mm;
    fm::strain_cell<UNI>(A0, Bn0, Bs0, Cn0, Cs0, u_p0, u_00, v_0p, v_00, e11_00, e22_00);
    fm::strain_cell<UNI>(A0, Bn0, Bs0, Cn0, Cs0, u_00, u_m0, v_mp, v_m0, e11_m0, e22_m0);
    fm::strain_cell<UNI>(Am, Bnm, Bsm, Cnm, Csm, u_pm, u_0m, v_00, v_0m, e11_0m, e22_0m);
    fm::strain_cell<UNI>(Am, Bnm, Bsm, Cnm, Csm, u_0m, u_mm, v_m0, v_mm, e11_mm, e22_mm);
    const double e12_00 = fm::strain_corner<UNI>(Sn0, Ss0, Sv0, u_00, u_0m, v_00, v_m0);
    const double e12_p0 = fm::strain_corner<UNI>(Sn0, Ss0, Sv0, u_p0, u_pm, v_p0, v_00);
    const double e12_0p = fm::strain_corner<UNI>(Snp, Ssp, Svp, u_0p, u_00, v_0p, v_mp);
    const double e12_pp = fm::strain_corner<UNI>(Snp, Ssp, Svp, u_pp, u_p0, v_pp, v_0p);

    // 4-point averages (y-average of x-averages), evp:250-252, :268, :330
    const double e11f = fm::avg4(e11_mm, e11_0m, e11_m0, e11_00);
    const double e22f = fm::avg4(e22_mm, e22_0m, e22_m0, e22_00);
    const double e12c = fm::avg4(e12_00, e12_p0, e12_0p, e12_pp);
    const double Pf = fm::avg4(P_mm, P_0m, P_m0, P_00);
    const double m_00 = h_00 * P.rho * a_00, m_m0 = h_m0 * P.rho * a_m0, m_0m = h_0m * P.rho * a_0m, m_mm = h_mm * P.rho * a_mm;
    const double mf = fm::avg4(m_mm, m_0m, m_m0, m_00);
    const double kc = UNI ? c.hkc : c.ca_dt * coef<UNI>(c, FC_RAZC, j), kf = UNI ? c.hkf : c.ca_dt * coef<UNI>(c, FC_RAZF, j);

    const fm::StressOut o = fm::stress_update(stress_const(P, c), e11_00, e22_00, e12_00, e11f, e22f, e12c, P_00, Pf,
                                              m_00, mf, kc, kf, s11, s22, s12);
    P.s11(i, j) = o.s11;
    P.s22(i, j) = o.s22;
    P.s12(i, j) = o.s12;
    P.al(i, j) = o.alpha;
    if (P.write_diag) {   // leave zeta, Delta as the reference's viscosity kernel would (evp:270-272)
        P.zf(i, j) = 0.5 * o.zf2;
        P.zc(i, j) = 0.5 * o.zc2;
        P.Dl(i, j) = o.xc * o.rDc;
    }
}

// ------------------------------------------------------------------------------------------------
// external stresses: gather the scalars fm::ext_stress needs
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void stress_x(const StressDev& s, int i, int j, double u, double vbar, double& ex, double& im) {
    double tau = 0.0, ue = 0.0, vebar = 0.0;
    if (s.kind == 1) tau = s.tau_u;
    else if (s.kind == 2) tau = s.fu(i, j);
    else if (s.kind == 3) {
        ue = s.ue_kind == 2 ? s.fu(i, j) : (s.ue_kind == 1 ? s.ue : 0.0);
        if (s.ve_kind == 2) vebar = fm::avg4(s.fv(i - 1, j), s.fv(i, j), s.fv(i - 1, j + 1), s.fv(i, j + 1));
        else vebar = (s.ve_kind == 1) ? s.ve : 0.0;
    }
    fm::ext_stress(s.kind, tau, s.rho_e * s.Cd, ue, vebar, u, vbar, ex, im);
}
__device__ __forceinline__ void stress_y(const StressDev& s, int i, int j, double v, double ubar, double& ex, double& im) {
    double tau = 0.0, ve = 0.0, uebar = 0.0;
    if (s.kind == 1) tau = s.tau_v;
    else if (s.kind == 2) tau = s.fv(i, j);
    else if (s.kind == 3) {
        ve = s.ve_kind == 2 ? s.fv(i, j) : (s.ve_kind == 1 ? s.ve : 0.0);
        if (s.ue_kind == 2) uebar = fm::avg4(s.fu(i, j - 1), s.fu(i + 1, j - 1), s.fu(i, j), s.fu(i + 1, j));
        else uebar = (s.ue_kind == 1) ? s.ue : 0.0;
    }
    fm::ext_stress(s.kind, tau, s.rho_e * s.Cd, ve, uebar, v, ubar, ex, im);
}

// ------------------------------------------------------------------------------------------------
// velocity phases
// ------------------------------------------------------------------------------------------------
template <bool UNI, bool MASK>
__global__ void __launch_bounds__(256) k_ustep(EvpDev P, Range r, ImageSpec img, FastCoef c, TileMap tm) {
    CELL_IJ(r, tm)
    const double h0 = P.h(i, j), hm = P.h(i - 1, j), a0 = P.a(i, j), am = P.a(i - 1, j);
    const double al0 = P.al(i, j), alm = P.al(i - 1, j);
    const double u = P.u(i, j), un = P.un(i, j);
    const double v_m0 = P.v(i - 1, j), v_00 = P.v(i, j), v_mp = P.v(i - 1, j + 1), v_0p = P.v(i, j + 1);
    double s11_0 = P.s11(i, j), s11_m = P.s11(i - 1, j), s12_0 = P.s12(i, j), s12_p = P.s12(i, j + 1);
    if (MASK) {
        if (immersed_peripheral_cc(P.g, i, j)) s11_0 = 0.0;
        if (immersed_peripheral_cc(P.g, i - 1, j)) s11_m = 0.0;
        if (immersed_peripheral_ff(P.g, i, j)) s12_0 = 0.0;
        if (immersed_peripheral_ff(P.g, i, j + 1)) s12_p = 0.0;
    }
    const double vbar = fm::avg4(v_m0, v_00, v_mp, v_0p);
    double div = fm::div1(coef<UNI>(c, FC_E, j), coef<UNI>(c, FC_FN, j), coef<UNI>(c, FC_FS, j), s11_0, s11_m, s12_p, s12_0);
    double ext, imt, exb, imb;
    stress_x(P.top, i, j, u, vbar, ext, imt);
    stress_x(P.bot, i, j, u, vbar, exb, imb);
    double cor = coef<UNI>(c, FC_FU, j) * vbar;               // -x_f_cross_U = +f vbar (f = 0 without Coriolis)
    if (P.extra) { if (P.has_forcing) cor += P.forcing_u(i, j); div += immersed_div_sigma_1(P, i, j); }   // model.forcing.u; immersed flux BCs
    const double mi = fm::avg2(hm * P.rho * am, h0 * P.rho * a0), ai = fm::avg2(am, a0), abar = fm::avg2(alm, al0);
    const double res = P.free_drift
        ? fm::vel_update_avg_fd(vel_const(P, c), u, un, mi, ai, abar, div, cor, ext, imt, exb, imb, peripheral_u(P.g, i, j), P.ufd(i, j))
        : fm::vel_update_avg(vel_const(P, c), u, un, mi, ai, abar, div, cor, ext, imt, exb, imb, peripheral_u(P.g, i, j));
    store_with_images(P.u, P.g, img, i, j, res);
}

template <bool UNI, bool MASK>
__global__ void __launch_bounds__(256) k_vstep(EvpDev P, Range r, ImageSpec img, FastCoef c, TileMap tm) {
    CELL_IJ(r, tm)
    const double h0 = P.h(i, j), hm = P.h(i, j - 1), a0 = P.a(i, j), am = P.a(i, j - 1);
    const double al0 = P.al(i, j), alm = P.al(i, j - 1);
    const double v = P.v(i, j), vn = P.vn(i, j);
    const double u_0m = P.u(i, j - 1), u_pm = P.u(i + 1, j - 1), u_00 = P.u(i, j), u_p0 = P.u(i + 1, j);
    double s11_0 = P.s11(i, j), s11_m = P.s11(i, j - 1), s22_0 = P.s22(i, j), s22_m = P.s22(i, j - 1);
    double s12_0 = P.s12(i, j), s12_p = P.s12(i + 1, j);
    if (MASK) {
        if (immersed_peripheral_cc(P.g, i, j)) { s11_0 = 0.0; s22_0 = 0.0; }
        if (immersed_peripheral_cc(P.g, i, j - 1)) { s11_m = 0.0; s22_m = 0.0; }
        if (immersed_peripheral_ff(P.g, i, j)) s12_0 = 0.0;
        if (immersed_peripheral_ff(P.g, i + 1, j)) s12_p = 0.0;
    }
    const double ubar = fm::avg4(u_0m, u_pm, u_00, u_p0);
    double div = fm::div2<UNI>(coef<UNI>(c, FC_Q1N, j), coef<UNI>(c, FC_Q2N, j), coef<UNI>(c, FC_Q1S, j), coef<UNI>(c, FC_Q2S, j),
                                coef<UNI>(c, FC_K, j), s11_0, s22_0, s11_m, s22_m, s12_p, s12_0);
    double ext, imt, exb, imb;
    stress_y(P.top, i, j, v, ubar, ext, imt);
    stress_y(P.bot, i, j, v, ubar, exb, imb);
    double cor = -coef<UNI>(c, FC_FV, j) * ubar;             // -y_f_cross_U = -f ubar
    if (P.extra) { if (P.has_forcing) cor += P.forcing_v(i, j); div += immersed_div_sigma_2(P, i, j); }
    const double mi = fm::avg2(hm * P.rho * am, h0 * P.rho * a0), ai = fm::avg2(am, a0), abar = fm::avg2(alm, al0);
    const double res = P.free_drift
        ? fm::vel_update_avg_fd(vel_const(P, c), v, vn, mi, ai, abar, div, cor, ext, imt, exb, imb, peripheral_v(P.g, i, j), P.vfd(i, j))
        : fm::vel_update_avg(vel_const(P, c), v, vn, mi, ai, abar, div, cor, ext, imt, exb, imb, peripheral_v(P.g, i, j));
    store_with_images(P.v, P.g, img, i, j, res);
}


// ------------------------------------------------------------------------------------------------
// Orthogonal curvilinear grids (CSI_METRIC_FULL): the same three phases with per-POINT stencil coefficients
// (csi_fast_coef.h, C2_*: fourteen metric planes; fm::full_* are the reference's operators in terms of them).
// The arithmetic after the strain rates / divergences is that of the regular-grid kernels (evp_fast_math.h).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double c2(const FastCoef& c, int w, int i, int j) { return c.c2[(long)w * c.c2_plane + i + (long)j * c.c2_ld]; }

__device__ __forceinline__ void strain_cell2(const FastCoef& c, int i, int j, double u_e, double u_w, double v_n, double v_s, double& e11, double& e22) {
    fm::full_strain_cell(c2(c, C2_DYU, i + 1, j) * u_e, c2(c, C2_DYU, i, j) * u_w, c2(c, C2_DXV, i, j + 1) * v_n, c2(c, C2_DXV, i, j) * v_s,
                         fm::rcp(c2(c, C2_DYU, i + 1, j)) * u_e, fm::rcp(c2(c, C2_DYU, i, j)) * u_w, fm::rcp(c2(c, C2_DXV, i, j + 1)) * v_n, fm::rcp(c2(c, C2_DXV, i, j)) * v_s,
                         c2(c, C2_DYC2, i, j), c2(c, C2_DXC2, i, j), c2(c, C2_RAZC, i, j), e11, e22);
}
__device__ __forceinline__ double strain_corner2(const FastCoef& c, int i, int j, double u_n, double u_s, double v_e, double v_w) {
    return fm::full_strain_corner(c2(c, C2_RDXU, i, j) * u_n, c2(c, C2_RDXU, i, j - 1) * u_s, c2(c, C2_RDYV, i, j) * v_e, c2(c, C2_RDYV, i - 1, j) * v_w,
                                  c2(c, C2_DXF2, i, j), c2(c, C2_DYF2, i, j), c2(c, C2_RAZF, i, j));
}

__global__ void __launch_bounds__(256) k_stress2(EvpDev P, Range r, FastCoef c, TileMap tm) {
    CELL_IJ(r, tm)
    const double u_mm = P.u(i - 1, j - 1), u_0m = P.u(i, j - 1), u_pm = P.u(i + 1, j - 1);
    const double u_m0 = P.u(i - 1, j),     u_00 = P.u(i, j),     u_p0 = P.u(i + 1, j);
    const double                           u_0p = P.u(i, j + 1), u_pp = P.u(i + 1, j + 1);
    const double v_mm = P.v(i - 1, j - 1), v_0m = P.v(i, j - 1);
    const double v_m0 = P.v(i - 1, j),     v_00 = P.v(i, j),     v_p0 = P.v(i + 1, j);
    const double v_mp = P.v(i - 1, j + 1), v_0p = P.v(i, j + 1), v_pp = P.v(i + 1, j + 1);
    const double P_mm = P.P(i - 1, j - 1), P_0m = P.P(i, j - 1), P_m0 = P.P(i - 1, j), P_00 = P.P(i, j);
    const double h_mm = P.h(i - 1, j - 1), h_0m = P.h(i, j - 1), h_m0 = P.h(i - 1, j), h_00 = P.h(i, j);
    const double a_mm = P.a(i - 1, j - 1), a_0m = P.a(i, j - 1), a_m0 = P.a(i - 1, j), a_00 = P.a(i, j);
    const double s11 = P.s11(i, j), s22 = P.s22(i, j), s12 = P.s12(i, j);
    double e11_00, e22_00, e11_m0, e22_m0, e11_0m, e22_0m, e11_mm, e22_mm;
    strain_cell2(c, i, j, u_p0, u_00, v_0p, v_00, e11_00, e22_00);
    strain_cell2(c, i - 1, j, u_00, u_m0, v_mp, v_m0, e11_m0, e22_m0);
    strain_cell2(c, i, j - 1, u_pm, u_0m, v_00, v_0m, e11_0m, e22_0m);
    strain_cell2(c, i - 1, j - 1, u_0m, u_mm, v_m0, v_mm, e11_mm, e22_mm);
    const double e12_00 = strain_corner2(c, i, j, u_00, u_0m, v_00, v_m0);
    const double e12_p0 = strain_corner2(c, i + 1, j, u_p0, u_pm, v_p0, v_00);
    const double e12_0p = strain_corner2(c, i, j + 1, u_0p, u_00, v_0p, v_mp);
    const double e12_pp = strain_corner2(c, i + 1, j + 1, u_pp, u_p0, v_pp, v_0p);
    const double e11f = fm::avg4(e11_mm, e11_0m, e11_m0, e11_00);
    const double e22f = fm::avg4(e22_mm, e22_0m, e22_m0, e22_00);
    const double e12c = fm::avg4(e12_00, e12_p0, e12_0p, e12_pp);
    const double Pf = fm::avg4(P_mm, P_0m, P_m0, P_00);
    const double m_00 = h_00 * P.rho * a_00, m_m0 = h_m0 * P.rho * a_m0, m_0m = h_0m * P.rho * a_0m, m_mm = h_mm * P.rho * a_mm;
    const double mf = fm::avg4(m_mm, m_0m, m_m0, m_00);
    const double kc = c.ca_dt * c2(c, C2_RAZC, i, j), kf = c.ca_dt * c2(c, C2_RAZF, i, j);
    const fm::StressOut o = fm::stress_update(stress_const(P, c), e11_00, e22_00, e12_00, e11f, e22f, e12c, P_00, Pf,
                                              m_00, mf, kc, kf, s11, s22, s12);
    P.s11(i, j) = o.s11;
    P.s22(i, j) = o.s22;
    P.s12(i, j) = o.s12;
    P.al(i, j) = o.alpha;
    if (P.write_diag) {
        P.zf(i, j) = 0.5 * o.zf2;
        P.zc(i, j) = 0.5 * o.zc2;
        P.Dl(i, j) = o.xc * o.rDc;
    }
}

template <bool MASK>
__global__ void __launch_bounds__(256) k_ustep2(EvpDev P, Range r, ImageSpec img, FastCoef c, TileMap tm) {
    CELL_IJ(r, tm)
    const double h0 = P.h(i, j), hm = P.h(i - 1, j), a0 = P.a(i, j), am = P.a(i - 1, j);
    const double al0 = P.al(i, j), alm = P.al(i - 1, j);
    const double u = P.u(i, j), un = P.un(i, j);
    const double v_m0 = P.v(i - 1, j), v_00 = P.v(i, j), v_mp = P.v(i - 1, j + 1), v_0p = P.v(i, j + 1);
    double s11_0 = P.s11(i, j), s11_m = P.s11(i - 1, j), s22_0 = P.s22(i, j), s22_m = P.s22(i - 1, j), s12_0 = P.s12(i, j), s12_p = P.s12(i, j + 1);
    if (MASK) {
        if (immersed_peripheral_cc(P.g, i, j)) { s11_0 = 0.0; s22_0 = 0.0; }
        if (immersed_peripheral_cc(P.g, i - 1, j)) { s11_m = 0.0; s22_m = 0.0; }
        if (immersed_peripheral_ff(P.g, i, j)) s12_0 = 0.0;
        if (immersed_peripheral_ff(P.g, i, j + 1)) s12_p = 0.0;
    }
    const double vbar = fm::avg4(v_m0, v_00, v_mp, v_0p);
    double div = fm::full_div1(c2(c, C2_DYU, i, j), fm::rcp(c2(c, C2_DYU, i, j)), c2(c, C2_RDXU, i, j), c2(c, C2_RAZU, i, j),
                               s11_0 + s22_0, s11_m + s22_m, c2(c, C2_DYC2, i, j) * (s11_0 - s22_0), c2(c, C2_DYC2, i - 1, j) * (s11_m - s22_m),
                               c2(c, C2_DXF2, i, j + 1) * s12_p, c2(c, C2_DXF2, i, j) * s12_0);
    double ext, imt, exb, imb;
    stress_x(P.top, i, j, u, vbar, ext, imt);
    stress_x(P.bot, i, j, u, vbar, exb, imb);
    double cor = fcor_at_u(P, i, j) * vbar;
    if (P.extra) { if (P.has_forcing) cor += P.forcing_u(i, j); div += immersed_div_sigma_1(P, i, j); }
    const double mi = fm::avg2(hm * P.rho * am, h0 * P.rho * a0), ai = fm::avg2(am, a0), abar = fm::avg2(alm, al0);
    const double res = P.free_drift
        ? fm::vel_update_avg_fd(vel_const(P, c), u, un, mi, ai, abar, div, cor, ext, imt, exb, imb, peripheral_u(P.g, i, j), P.ufd(i, j))
        : fm::vel_update_avg(vel_const(P, c), u, un, mi, ai, abar, div, cor, ext, imt, exb, imb, peripheral_u(P.g, i, j));
    store_with_images(P.u, P.g, img, i, j, res);
}

template <bool MASK>
__global__ void __launch_bounds__(256) k_vstep2(EvpDev P, Range r, ImageSpec img, FastCoef c, TileMap tm) {
    CELL_IJ(r, tm)
    const double h0 = P.h(i, j), hm = P.h(i, j - 1), a0 = P.a(i, j), am = P.a(i, j - 1);
    const double al0 = P.al(i, j), alm = P.al(i, j - 1);
    const double v = P.v(i, j), vn = P.vn(i, j);
    const double u_0m = P.u(i, j - 1), u_pm = P.u(i + 1, j - 1), u_00 = P.u(i, j), u_p0 = P.u(i + 1, j);
    double s11_0 = P.s11(i, j), s11_m = P.s11(i, j - 1), s22_0 = P.s22(i, j), s22_m = P.s22(i, j - 1);
    double s12_0 = P.s12(i, j), s12_p = P.s12(i + 1, j);
    if (MASK) {
        if (immersed_peripheral_cc(P.g, i, j)) { s11_0 = 0.0; s22_0 = 0.0; }
        if (immersed_peripheral_cc(P.g, i, j - 1)) { s11_m = 0.0; s22_m = 0.0; }
        if (immersed_peripheral_ff(P.g, i, j)) s12_0 = 0.0;
        if (immersed_peripheral_ff(P.g, i + 1, j)) s12_p = 0.0;
    }
    const double ubar = fm::avg4(u_0m, u_pm, u_00, u_p0);
    double div = fm::full_div2(c2(c, C2_DXV, i, j), fm::rcp(c2(c, C2_DXV, i, j)), c2(c, C2_RDYV, i, j), c2(c, C2_RAZV, i, j),
                               s11_0 + s22_0, s11_m + s22_m, c2(c, C2_DXC2, i, j) * (s11_0 - s22_0), c2(c, C2_DXC2, i, j - 1) * (s11_m - s22_m),
                               c2(c, C2_DYF2, i + 1, j) * s12_p, c2(c, C2_DYF2, i, j) * s12_0);
    double ext, imt, exb, imb;
    stress_y(P.top, i, j, v, ubar, ext, imt);
    stress_y(P.bot, i, j, v, ubar, exb, imb);
    double cor = -fcor_at_v(P, i, j) * ubar;
    if (P.extra) { if (P.has_forcing) cor += P.forcing_v(i, j); div += immersed_div_sigma_2(P, i, j); }
    const double mi = fm::avg2(hm * P.rho * am, h0 * P.rho * a0), ai = fm::avg2(am, a0), abar = fm::avg2(alm, al0);
    const double res = P.free_drift
        ? fm::vel_update_avg_fd(vel_const(P, c), v, vn, mi, ai, abar, div, cor, ext, imt, exb, imb, peripheral_v(P.g, i, j), P.vfd(i, j))
        : fm::vel_update_avg(vel_const(P, c), v, vn, mi, ai, abar, div, cor, ext, imt, exb, imb, peripheral_v(P.g, i, j));
    store_with_images(P.v, P.g, img, i, j, res);
}

}  // namespace fast

static inline dim3 grid_for(const Range& r, dim3 b) {
    return dim3((unsigned)((r.i1 - r.i0 + 1 + b.x - 1) / b.x), (unsigned)((r.j1 - r.j0 + 1 + b.y - 1) / b.y), 1);
}

// cross component of an array-valued external velocity averaged to the velocity points (what stress_x / stress_y
// compute per call): vbar at u points from fv, ubar at v points from fu; same avg4, so the values are identical
__global__ void __launch_bounds__(256) k_forcing_bars(FRef fu, FRef fv, FRef ubar_v, FRef vbar_u, int has_u, int has_v, Range r) {
    const int i = r.i0 + (int)(blockIdx.x * blockDim.x + threadIdx.x), j = r.j0 + (int)(blockIdx.y * blockDim.y + threadIdx.y);
    if (i > r.i1 || j > r.j1) return;
    if (has_v) vbar_u(i, j) = fm::avg4(fv(i - 1, j), fv(i, j), fv(i - 1, j + 1), fv(i, j + 1));
    if (has_u) ubar_v(i, j) = fm::avg4(fu(i, j - 1), fu(i + 1, j - 1), fu(i, j), fu(i + 1, j));
}

// the stress divergence of the immersed FluxBoundaryConditions (csi_dev.h immersed_div_sigma_1 / _2: the reference's operation
// order) at every u / v point, once per sub-cycle: the two-sub-steps kernel adds the stored numbers where the three kernels
// evaluate the functions -- the same bits
__global__ void __launch_bounds__(256) k_immersed_div(EvpDev P, FRef xd_u, FRef xd_v, Range r) {
    const int i = r.i0 + (int)(blockIdx.x * blockDim.x + threadIdx.x), j = r.j0 + (int)(blockIdx.y * blockDim.y + threadIdx.y);
    if (i > r.i1 || j > r.j1) return;
    xd_u(i, j) = immersed_div_sigma_1(P, i, j);
    xd_v(i, j) = immersed_div_sigma_2(P, i, j);
}
void launch_immersed_div(const EvpDev& P, const FRef& xd_u, const FRef& xd_v, hipStream_t s) {
    const GridDev& g = P.g;
    const Range r{2 - g.Hx, g.Nx + g.Hx - 1, 2 - g.Hy, g.Ny + g.Hy - 1};      // (one cell inside the parents: the 2 x 2 stencils)
    dim3 b(64, 4);
    hipLaunchKernelGGL(k_immersed_div, dim3((unsigned)((r.i1 - r.i0 + 64) / 64), (unsigned)((r.j1 - r.j0 + 4) / 4)), b, 0, s, P, xd_u, xd_v, r);
}

void launch_forcing_bars(const EvpDev& P, const FRef& ubar_v, const FRef& vbar_u, hipStream_t s, bool top) {
    const StressDev& b = top ? P.top : P.bot;
    if (b.kind != 3 || (b.ue_kind != 2 && b.ve_kind != 2)) return;
    // every point whose four neighbours lie inside the parent arrays
    const Range r{2 - P.g.Hx, P.g.Nx + P.g.Hx - 1, 2 - P.g.Hy, P.g.Ny + P.g.Hy - 1};
    dim3 blk(64, 4), grd((unsigned)((r.i1 - r.i0 + 64) / 64), (unsigned)((r.j1 - r.j0 + 4) / 4));
    hipLaunchKernelGGL(k_forcing_bars, grd, blk, 0, s, b.fu, b.fv, ubar_v, vbar_u, b.ue_kind == 2, b.ve_kind == 2, r);
}

bool fast_supported(const EvpDev& P) {
    // minimum_mass > 0: evp_fast_math.h drops the reference's mi <= 0 guards, which the active / marginal selection
    // makes redundant in that case (the reference's default is 1 kg m^-2)
    return P.min_mass > 0;
}

void launch_fast_init(const EvpDev& P, const Range& r, hipStream_t s) {
    dim3 b(64, 4);
    hipLaunchKernelGGL(fast::k_init, grid_for(r, b), b, 0, s, P, r);
}
static fast::TileMap tile_map(const EvpDev& P, const Range& r, dim3& grid) {
    fast::TileMap tm;
    tm.ibase = 1 - P.g.Hx;                               // parent row start (aligned), <= r.i0
    if (tm.ibase > r.i0) tm.ibase = r.i0;
    tm.jbase = r.j0;
    tm.gx = (r.i1 - tm.ibase + fast::TILE_X) / fast::TILE_X;
    // Short ranges are the fold band's: its launches run BESIDE a pair launch whose waves hold 232-240 of a SIMD's 512 registers
    // each, so a four-wave workgroup finds room only when a pair tile retires (round 4's trace of the tripolar-like case: k_stress2
    // of a 16-row band "ran" 276 us).  One wave per workgroup fits wherever ONE slot is free.
    static const int band_ty = [] { const char* e = getenv("CSI_BAND_TY"); return (e && *e) ? atoi(e) : 1; }();
    tm.ty = (r.j1 - r.j0 + 1 <= 32) ? std::min(std::max(band_ty, 1), fast::TILE_Y) : fast::TILE_Y;
    const int gy = (r.j1 - r.j0 + tm.ty) / tm.ty;
    tm.ntiles = tm.gx * gy;
    tm.per_xcd = (tm.ntiles + 7) / 8;
    grid = dim3((unsigned)(tm.per_xcd * 8), 1, 1);
    return tm;
}

void launch_fast_stress(const EvpDev& P, const Range& r, const FastCoef& c, hipStream_t s) {
    dim3 b(fast::TILE_X, fast::TILE_Y), g;
    const fast::TileMap tm = tile_map(P, r, g);
    b.y = (unsigned)tm.ty;
    if (c.full) { hipLaunchKernelGGL(fast::k_stress2, g, b, 0, s, P, r, c, tm); return; }
    if (c.uniform) hipLaunchKernelGGL(fast::k_stress<true>, g, b, 0, s, P, r, c, tm);
    else hipLaunchKernelGGL(fast::k_stress<false>, g, b, 0, s, P, r, c, tm);
}
void launch_fast_ustep(const EvpDev& P, const Range& r, const ImageSpec& im, const FastCoef& c, hipStream_t s) {
    dim3 b(fast::TILE_X, fast::TILE_Y), g;
    const fast::TileMap tm = tile_map(P, r, g);
    b.y = (unsigned)tm.ty;
    const bool m = P.g.has_mask != 0;
    if (c.full) {
        if (m) hipLaunchKernelGGL((fast::k_ustep2<true>), g, b, 0, s, P, r, im, c, tm);
        else hipLaunchKernelGGL((fast::k_ustep2<false>), g, b, 0, s, P, r, im, c, tm);
        return;
    }
    if (c.uniform) {
        if (m) hipLaunchKernelGGL((fast::k_ustep<true, true>), g, b, 0, s, P, r, im, c, tm);
        else hipLaunchKernelGGL((fast::k_ustep<true, false>), g, b, 0, s, P, r, im, c, tm);
    } else {
        if (m) hipLaunchKernelGGL((fast::k_ustep<false, true>), g, b, 0, s, P, r, im, c, tm);
        else hipLaunchKernelGGL((fast::k_ustep<false, false>), g, b, 0, s, P, r, im, c, tm);
    }
}
void launch_fast_vstep(const EvpDev& P, const Range& r, const ImageSpec& im, const FastCoef& c, hipStream_t s) {
    dim3 b(fast::TILE_X, fast::TILE_Y), g;
    const fast::TileMap tm = tile_map(P, r, g);
    b.y = (unsigned)tm.ty;
    const bool m = P.g.has_mask != 0;
    if (c.full) {
        if (m) hipLaunchKernelGGL((fast::k_vstep2<true>), g, b, 0, s, P, r, im, c, tm);
        else hipLaunchKernelGGL((fast::k_vstep2<false>), g, b, 0, s, P, r, im, c, tm);
        return;
    }
    if (c.uniform) {
        if (m) hipLaunchKernelGGL((fast::k_vstep<true, true>), g, b, 0, s, P, r, im, c, tm);
        else hipLaunchKernelGGL((fast::k_vstep<true, false>), g, b, 0, s, P, r, im, c, tm);
    } else {
        if (m) hipLaunchKernelGGL((fast::k_vstep<false, true>), g, b, 0, s, P, r, im, c, tm);
        else hipLaunchKernelGGL((fast::k_vstep<false, false>), g, b, 0, s, P, r, im, c, tm);
    }
}

}  // namespace csi
