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

namespace csi {
namespace fast {

#define EPS64 2.220446049250313e-16

template <bool UNI>
__device__ __forceinline__ double coef(const FastCoef& c, int which, int j) {
    if (UNI) return c.uni[which];
    return c.vec[(long)which * c.stride + j];
}

__device__ __forceinline__ double clampd(double x, double lo, double hi) { return x > hi ? hi : (x < lo ? lo : x); }

// Block -> tile mapping.  Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 shares an
// XCD, MI355X_MICROARCH.md "Workgroup dispatch"); each XCD has its own L2, so a tile's halo rows
// and the cache lines it shares with its x-neighbour are re-fetched through the fabric unless the
// neighbouring tile runs on the same XCD.  The remap gives every XCD one contiguous band of tile
// rows, walked in x-fastest order (speed only: any placement gives the same results).  Waves start
// at the parent row start (i = 1 - Hx), which is 64-byte aligned for even ld.
struct TileMap {
    int gx, ntiles, per_xcd;
    int ibase, jbase;
};
constexpr int TILE_X = 64, TILE_Y = 4;

#define CELL_IJ(r, tm)                                                          \
    const int b_ = (int)blockIdx.x;                                             \
    const int t_ = (b_ & 7) * (tm).per_xcd + (b_ >> 3);                         \
    if ((b_ >> 3) >= (tm).per_xcd || t_ >= (tm).ntiles) return;                 \
    const int by_ = t_ / (tm).gx, bx_ = t_ - by_ * (tm).gx;                     \
    const int i = (tm).ibase + bx_ * TILE_X + (int)threadIdx.x;                 \
    const int j = __builtin_amdgcn_readfirstlane((tm).jbase + by_ * TILE_Y + (int)threadIdx.y); \
    if (i < (r).i0 || i > (r).i1 || j > (r).j1) return;

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

    // strain rates at the four cells (i-1..i, j-1..j): e11 = A du + Bn vn - Bs vs ; e22 = Cn vn - Cs vs
    const double e11_00 = A0 * (u_p0 - u_00) + (Bn0 * v_0p - Bs0 * v_00);
    const double e22_00 = Cn0 * v_0p - Cs0 * v_00;
    const double e11_m0 = A0 * (u_00 - u_m0) + (Bn0 * v_mp - Bs0 * v_m0);
    const double e22_m0 = Cn0 * v_mp - Cs0 * v_m0;
    const double e11_0m = Am * (u_pm - u_0m) + (Bnm * v_00 - Bsm * v_0m);
    const double e22_0m = Cnm * v_00 - Csm * v_0m;
    const double e11_mm = Am * (u_0m - u_mm) + (Bnm * v_m0 - Bsm * v_mm);
    const double e22_mm = Cnm * v_m0 - Csm * v_mm;
    // e12 at the four corners (i..i+1, j..j+1): e12 = Sn u(jj) - Ss u(jj-1) + Sv (v(ii) - v(ii-1))
    const double e12_00 = (Sn0 * u_00 - Ss0 * u_0m) + Sv0 * (v_00 - v_m0);
    const double e12_p0 = (Sn0 * u_p0 - Ss0 * u_pm) + Sv0 * (v_p0 - v_00);
    const double e12_0p = (Snp * u_0p - Ssp * u_00) + Svp * (v_0p - v_mp);
    const double e12_pp = (Snp * u_pp - Ssp * u_p0) + Svp * (v_pp - v_0p);

    // 4-point averages (y-average of x-averages), evp:250-252
    const double e11f = 0.5 * (0.5 * (e11_mm + e11_0m) + 0.5 * (e11_m0 + e11_00));
    const double e22f = 0.5 * (0.5 * (e22_mm + e22_0m) + 0.5 * (e22_m0 + e22_00));
    const double e12c = 0.5 * (0.5 * (e12_00 + e12_p0) + 0.5 * (e12_0p + e12_pp));

    const double em2 = c.em2;
    // evp:255-272
    const double dc = e11_00 + e22_00, df = e11f + e22f;
    const double tc = e11_00 - e22_00, tf = e11f - e22f;
    const double sc2 = tc * tc + 4.0 * (e12c * e12c);
    const double sf2 = tf * tf + 4.0 * (e12_00 * e12_00);
    const double Dc = fmax(sqrt(dc * dc + sc2 * em2), P.Dmin);
    const double Df = fmax(sqrt(df * df + sf2 * em2), P.Dmin);
    const double Pf = 0.5 * (0.5 * (P_mm + P_0m) + 0.5 * (P_m0 + P_00));
    const double zc = P_00 * (0.5 / Dc);
    const double zf = Pf * (0.5 / Df);

    // evp:318-327
    const double Pr = (P.pressure_kind == 0) ? P_00 * Dc / (Dc + P.Dmin) : P_00;
    const double etac = zc * em2, etaf = zf * em2;
    const double bulk = (zc - etac) * dc - 0.5 * Pr;
    const double s11n = 2.0 * etac * e11_00 + bulk;
    const double s22n = 2.0 * etac * e22_00 + bulk;
    const double s12n = 2.0 * etaf * e12_00;

    // evp:329-340
    const double m_00 = h_00 * P.rho * a_00, m_m0 = h_m0 * P.rho * a_m0, m_0m = h_0m * P.rho * a_0m, m_mm = h_mm * P.rho * a_mm;
    const double mf = 0.5 * (0.5 * (m_mm + m_0m) + 0.5 * (m_m0 + m_00));
    const double kc = c.ca_dt * coef<UNI>(c, FC_RAZC, j), kf = c.ca_dt * coef<UNI>(c, FC_RAZF, j);
    double g2c = zc * kc / m_00;
    g2c = isnan(g2c) ? P.amax * P.amax : g2c;
    const double gc = clampd(sqrt(g2c), P.amin, P.amax);
    double g2f = zf * kf / mf;
    g2f = isnan(g2f) ? P.amax * P.amax : g2f;
    const double gf = clampd(sqrt(g2f), P.amin, P.amax);
    const double rgc = 1.0 / gc, rgf = 1.0 / gf;

    // evp:345-352
    P.s11(i, j) = s11 + ((m_00 > 0) ? (s11n - s11) * rgc : 0.0);
    P.s22(i, j) = s22 + ((m_00 > 0) ? (s22n - s22) * rgc : 0.0);
    P.s12(i, j) = s12 + ((mf > 0) ? (s12n - s12) * rgf : 0.0);
    P.al(i, j) = gc;
    if (P.write_diag) {   // leave zeta, Delta as the reference's viscosity kernel would (evp:270-272)
        P.zf(i, j) = zf;
        P.zc(i, j) = zc;
        P.Dl(i, j) = Dc;
    }
}

// ------------------------------------------------------------------------------------------------
// external stresses
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double ext_ue(const StressDev& s, int i, int j) {
    return s.ue_kind == 2 ? s.fu(i, j) : (s.ue_kind == 1 ? s.ue : 0.0);
}
__device__ __forceinline__ double ext_ve(const StressDev& s, int i, int j) {
    return s.ve_kind == 2 ? s.fv(i, j) : (s.ve_kind == 1 ? s.ve : 0.0);
}
// explicit part and implicit coefficient of one stress at the u point; vbar = Ixy^{fc}(v)
__device__ __forceinline__ void stress_x(const StressDev& s, int i, int j, double u, double vbar, double& ex, double& im) {
    ex = 0.0; im = 0.0;
    if (s.kind == 1) ex = s.tau_u;
    else if (s.kind == 2) ex = s.fu(i, j);
    else if (s.kind == 3) {
        const double ue = ext_ue(s, i, j);
        double vebar;
        if (s.ve_kind == 2) vebar = 0.5 * (0.5 * (s.fv(i - 1, j) + s.fv(i, j)) + 0.5 * (s.fv(i - 1, j + 1) + s.fv(i, j + 1)));
        else vebar = (s.ve_kind == 1) ? s.ve : 0.0;
        const double du = ue - u, dv = vebar - vbar;
        im = s.rho_e * s.Cd * sqrt(du * du + dv * dv);
        ex = im * ue;
    }
}
__device__ __forceinline__ void stress_y(const StressDev& s, int i, int j, double v, double ubar, double& ex, double& im) {
    ex = 0.0; im = 0.0;
    if (s.kind == 1) ex = s.tau_v;
    else if (s.kind == 2) ex = s.fv(i, j);
    else if (s.kind == 3) {
        const double ve = ext_ve(s, i, j);
        double uebar;
        if (s.ue_kind == 2) uebar = 0.5 * (0.5 * (s.fu(i, j - 1) + s.fu(i + 1, j - 1)) + 0.5 * (s.fu(i, j) + s.fu(i + 1, j)));
        else uebar = (s.ue_kind == 1) ? s.ue : 0.0;
        const double dv = ve - v, du = uebar - ubar;
        im = s.rho_e * s.Cd * sqrt(du * du + dv * dv);
        ex = im * ve;
    }
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
    const double mi = 0.5 * (hm * P.rho * am + h0 * P.rho * a0);
    const double ai = 0.5 * (am + a0);
    const double abar = 0.5 * (alm + al0);
    const double dtau = P.dt / abar;
    const double rm = 1.0 / mi;
    const double vbar = 0.5 * (0.5 * (v_m0 + v_00) + 0.5 * (v_mp + v_0p));
    // d_j sigma_1j with constant dy: E (s11_i - s11_{i-1}) + Fn s12(j+1) - Fs s12(j)   (isd:39-44)
    const double div = coef<UNI>(c, FC_E, j) * (s11_0 - s11_m) + (coef<UNI>(c, FC_FN, j) * s12_p - coef<UNI>(c, FC_FS, j) * s12_0);
    double ext, imt, exb, imb;
    stress_x(P.top, i, j, u, vbar, ext, imt);
    stress_x(P.bot, i, j, u, vbar, exb, imb);
    const double cor = P.has_cor ? P.fcor * vbar : 0.0;           // -x_f_cross_U = +f vbar
    const double rai = rm * ai;
    double G = cor + (exb - ext) * rai + div * rm + (un - u) * c.rdt;
    double tau_i = (imb - imt) * rai;
    G = (mi <= 0) ? 0.0 : G;
    tau_i = (mi <= 0) ? 0.0 : tau_i;
    const double uD = (u + dtau * G) / (1.0 + dtau * tau_i);
    const bool marginal = (mi > EPS64) & (ai > EPS64);
    const bool active_ice = (mi >= P.min_mass) & (ai >= P.min_conc);
    (void)marginal;                                                // free drift `nothing` -> 0 either way
    double res = active_ice ? uD : 0.0;
    if (peripheral_u(P.g, i, j)) res = 0.0;                        // NaN * 0 cannot occur: uD finite wherever mi > 0
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
    const double mi = 0.5 * (hm * P.rho * am + h0 * P.rho * a0);
    const double ai = 0.5 * (am + a0);
    const double abar = 0.5 * (alm + al0);
    const double dtau = P.dt / abar;
    const double rm = 1.0 / mi;
    const double ubar = 0.5 * (0.5 * (u_0m + u_pm) + 0.5 * (u_00 + u_p0));
    // d_j sigma_2j (isd:46-51): Q1n s11(j) + Q2n s22(j) - Q1s s11(j-1) - Q2s s22(j-1) + K (s12(i+1) - s12(i))
    const double div = (coef<UNI>(c, FC_Q1N, j) * s11_0 + coef<UNI>(c, FC_Q2N, j) * s22_0)
                     - (coef<UNI>(c, FC_Q1S, j) * s11_m + coef<UNI>(c, FC_Q2S, j) * s22_m)
                     + coef<UNI>(c, FC_K, j) * (s12_p - s12_0);
    double ext, imt, exb, imb;
    stress_y(P.top, i, j, v, ubar, ext, imt);
    stress_y(P.bot, i, j, v, ubar, exb, imb);
    const double cor = P.has_cor ? -P.fcor * ubar : 0.0;          // -y_f_cross_U = -f ubar
    const double rai = rm * ai;
    double G = cor + (exb - ext) * rai + div * rm + (vn - v) * c.rdt;
    double tau_i = (imb - imt) * rai;
    G = (mi <= 0) ? 0.0 : G;
    tau_i = (mi <= 0) ? 0.0 : tau_i;
    const double vD = (v + dtau * G) / (1.0 + dtau * tau_i);
    const bool active_ice = (mi >= P.min_mass) & (ai >= P.min_conc);
    double res = active_ice ? vD : 0.0;
    if (peripheral_v(P.g, i, j)) res = 0.0;
    store_with_images(P.v, P.g, img, i, j, res);
}

}  // namespace fast

static inline dim3 grid_for(const Range& r, dim3 b) {
    return dim3((unsigned)((r.i1 - r.i0 + 1 + b.x - 1) / b.x), (unsigned)((r.j1 - r.j0 + 1 + b.y - 1) / b.y), 1);
}

bool fast_supported(const EvpDev& P) {
    // free-drift closed forms and field-valued forcing besides the bound stress slots are "next"
    (void)P;
    return true;
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
    const int gy = (r.j1 - r.j0 + fast::TILE_Y) / fast::TILE_Y;
    tm.ntiles = tm.gx * gy;
    tm.per_xcd = (tm.ntiles + 7) / 8;
    grid = dim3((unsigned)(tm.per_xcd * 8), 1, 1);
    return tm;
}

void launch_fast_stress(const EvpDev& P, const Range& r, const FastCoef& c, hipStream_t s) {
    dim3 b(fast::TILE_X, fast::TILE_Y), g;
    const fast::TileMap tm = tile_map(P, r, g);
    if (c.uniform) hipLaunchKernelGGL(fast::k_stress<true>, g, b, 0, s, P, r, c, tm);
    else hipLaunchKernelGGL(fast::k_stress<false>, g, b, 0, s, P, r, c, tm);
}
void launch_fast_ustep(const EvpDev& P, const Range& r, const ImageSpec& im, const FastCoef& c, hipStream_t s) {
    dim3 b(fast::TILE_X, fast::TILE_Y), g;
    const fast::TileMap tm = tile_map(P, r, g);
    const bool m = P.g.has_mask != 0;
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
    const bool m = P.g.has_mask != 0;
    if (c.uniform) {
        if (m) hipLaunchKernelGGL((fast::k_vstep<true, true>), g, b, 0, s, P, r, im, c, tm);
        else hipLaunchKernelGGL((fast::k_vstep<true, false>), g, b, 0, s, P, r, im, c, tm);
    } else {
        if (m) hipLaunchKernelGGL((fast::k_vstep<false, true>), g, b, 0, s, P, r, im, c, tm);
        else hipLaunchKernelGGL((fast::k_vstep<false, false>), g, b, 0, s, P, r, im, c, tm);
    }
}

}  // namespace csi
