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

// (the value functions below are the kernels' bodies: the three kernels store what they return, the fold band's fused launches
//  -- k_band_stress, k_band_uv further down -- evaluate the same functions, so that both give the same bits)
template <bool UNI>
__device__ __forceinline__ fm::StressOut stress_value(const EvpDev& P, const FastCoef& c, int i, int j) {
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

    return fm::stress_update(stress_const(P, c), e11_00, e22_00, e12_00, e11f, e22f, e12c, P_00, Pf,
                             m_00, mf, kc, kf, s11, s22, s12);
}
__device__ __forceinline__ void store_stress(const EvpDev& P, int i, int j, const fm::StressOut& o) {
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
template <bool UNI>
__global__ void __launch_bounds__(256) k_stress(EvpDev P, Range r, FastCoef c, TileMap tm) {
    CELL_IJ(r, tm)
    store_stress(P, i, j, stress_value<UNI>(P, c, i, j));
}

// ------------------------------------------------------------------------------------------------
// external stresses: gather the scalars fm::ext_stress needs.  Two phases -- the loads (ext_gather_*), then the arithmetic
// (stress_*_from) -- so that a velocity function can issue every load of its point before it uses any (see ustep_value).
// ------------------------------------------------------------------------------------------------
struct ExtIn { double own, c0, c1, c2, c3; };      // array values: the own component's (kind 2: tau) and the cross component's four
__device__ __forceinline__ ExtIn ext_gather_x(const StressDev& s, int i, int j) {
    ExtIn e{0.0, 0.0, 0.0, 0.0, 0.0};
    if ((s.kind == 2) | ((s.kind == 3) & (s.ue_kind == 2))) e.own = s.fu(i, j);
    if ((s.kind == 3) & (s.ve_kind == 2)) { e.c0 = s.fv(i - 1, j); e.c1 = s.fv(i, j); e.c2 = s.fv(i - 1, j + 1); e.c3 = s.fv(i, j + 1); }
    return e;
}
__device__ __forceinline__ ExtIn ext_gather_y(const StressDev& s, int i, int j) {
    ExtIn e{0.0, 0.0, 0.0, 0.0, 0.0};
    if ((s.kind == 2) | ((s.kind == 3) & (s.ve_kind == 2))) e.own = s.fv(i, j);
    if ((s.kind == 3) & (s.ue_kind == 2)) { e.c0 = s.fu(i, j - 1); e.c1 = s.fu(i + 1, j - 1); e.c2 = s.fu(i, j); e.c3 = s.fu(i + 1, j); }
    return e;
}
__device__ __forceinline__ void stress_x_from(const StressDev& s, const ExtIn& e, double u, double vbar, double& ex, double& im) {
    double tau = 0.0, ue = 0.0, vebar = 0.0;
    if (s.kind == 1) tau = s.tau_u;
    else if (s.kind == 2) tau = e.own;
    else if (s.kind == 3) {
        ue = s.ue_kind == 2 ? e.own : (s.ue_kind == 1 ? s.ue : 0.0);
        if (s.ve_kind == 2) vebar = fm::avg4(e.c0, e.c1, e.c2, e.c3);
        else vebar = (s.ve_kind == 1) ? s.ve : 0.0;
    }
    fm::ext_stress(s.kind, tau, s.rho_e * s.Cd, ue, vebar, u, vbar, ex, im);
}
__device__ __forceinline__ void stress_y_from(const StressDev& s, const ExtIn& e, double v, double ubar, double& ex, double& im) {
    double tau = 0.0, ve = 0.0, uebar = 0.0;
    if (s.kind == 1) tau = s.tau_v;
    else if (s.kind == 2) tau = e.own;
    else if (s.kind == 3) {
        ve = s.ve_kind == 2 ? e.own : (s.ve_kind == 1 ? s.ve : 0.0);
        if (s.ue_kind == 2) uebar = fm::avg4(e.c0, e.c1, e.c2, e.c3);
        else uebar = (s.ue_kind == 1) ? s.ue : 0.0;
    }
    fm::ext_stress(s.kind, tau, s.rho_e * s.Cd, ve, uebar, v, ubar, ex, im);
}

// the mask bytes of the six cells a velocity point's tests look at, loaded before any is used (csi_dev.h: inactive_cell and the
// tests made of it -- the same truth tables, from the gathered bytes)
struct MaskIn { unsigned b0, b1, b2, b3, b4, b5; };      // (a register each: packing bytes would be a use of the loaded values)
__device__ __forceinline__ unsigned mask_byte(const GridDev& g, int i, int j) {      // (only where the grid has a mask: the kernels' MASK = has_mask)
    const int ic = min(max(i, 1 - g.Hx), g.Nx + g.Hx), jc = min(max(j, 1 - g.Hy), g.Ny + g.Hy);
    return g.mask[ic + (long)jc * g.mask_ld];
}
__device__ __forceinline__ bool inactive_from(const GridDev& g, int i, int j, unsigned byte) {
    bool out = inactive_cell_underlying(g, i, j);
    if (g.has_mask) out |= (i < 1 - g.Hx) | (i > g.Nx + g.Hx) | (j < 1 - g.Hy) | (j > g.Ny + g.Hy) | (byte == 0);
    return out;
}
// everything fm::vel_update_* needs that is not arithmetic on gathered values: the loads of one velocity point
struct VelIn {
    double h0, hm, a0, am, al0, alm, w, wn, fd, forcing;
    ExtIn top, bot;
    MaskIn m;
};
// the stresses a u point reads with the immersed cells' zeroed (MASK), its peripheral flag; cells: 0 (i, j), 1 (i-1, j), 2 (i, j-1), 3 (i-1, j-1), 4 (i, j+1), 5 (i-1, j+1)
template <bool MASK>
__device__ __forceinline__ VelIn gather_u(const EvpDev& P, int i, int j) {
    VelIn q;
    q.h0 = P.h(i, j); q.hm = P.h(i - 1, j); q.a0 = P.a(i, j); q.am = P.a(i - 1, j);
    q.al0 = P.al(i, j); q.alm = P.al(i - 1, j);
    q.w = P.u(i, j); q.wn = P.un(i, j);
    q.fd = P.free_drift ? P.ufd(i, j) : 0.0;
    q.forcing = (P.extra && P.has_forcing) ? P.forcing_u(i, j) : 0.0;
    q.top = ext_gather_x(P.top, i, j);
    q.bot = ext_gather_x(P.bot, i, j);
    q.m.b0 = q.m.b1 = q.m.b2 = q.m.b3 = q.m.b4 = q.m.b5 = 1u;
    if (MASK) { q.m.b0 = mask_byte(P.g, i, j); q.m.b1 = mask_byte(P.g, i - 1, j); q.m.b2 = mask_byte(P.g, i, j - 1); q.m.b3 = mask_byte(P.g, i - 1, j - 1); q.m.b4 = mask_byte(P.g, i, j + 1); q.m.b5 = mask_byte(P.g, i - 1, j + 1); }
    return q;
}
// v point; cells: 0 (i, j), 1 (i, j-1), 2 (i-1, j), 3 (i-1, j-1), 4 (i+1, j), 5 (i+1, j-1)
template <bool MASK>
__device__ __forceinline__ VelIn gather_v(const EvpDev& P, int i, int j) {
    VelIn q;
    q.h0 = P.h(i, j); q.hm = P.h(i, j - 1); q.a0 = P.a(i, j); q.am = P.a(i, j - 1);
    q.al0 = P.al(i, j); q.alm = P.al(i, j - 1);
    q.w = P.v(i, j); q.wn = P.vn(i, j);
    q.fd = P.free_drift ? P.vfd(i, j) : 0.0;
    q.forcing = (P.extra && P.has_forcing) ? P.forcing_v(i, j) : 0.0;
    q.top = ext_gather_y(P.top, i, j);
    q.bot = ext_gather_y(P.bot, i, j);
    q.m.b0 = q.m.b1 = q.m.b2 = q.m.b3 = q.m.b4 = q.m.b5 = 1u;
    if (MASK) { q.m.b0 = mask_byte(P.g, i, j); q.m.b1 = mask_byte(P.g, i, j - 1); q.m.b2 = mask_byte(P.g, i - 1, j); q.m.b3 = mask_byte(P.g, i - 1, j - 1); q.m.b4 = mask_byte(P.g, i + 1, j); q.m.b5 = mask_byte(P.g, i + 1, j - 1); }
    return q;
}
// immersed_peripheral_cc / _ff (csi_dev.h) from gathered bytes: n = inactive, w = inactive without the mask
struct UMask { bool cc0, ccm, ff0, ffp, peripheral; };
template <bool MASK>
__device__ __forceinline__ UMask umask_from(const GridDev& g, int i, int j, const MaskIn& m) {
    UMask r{false, false, false, false, false};
    const bool n0 = inactive_from(g, i, j, m.b0), n1 = inactive_from(g, i - 1, j, m.b1);
    r.peripheral = n0 | n1;
    if (MASK) {
        const bool n2 = inactive_from(g, i, j - 1, m.b2), n3 = inactive_from(g, i - 1, j - 1, m.b3), n4 = inactive_from(g, i, j + 1, m.b4), n5 = inactive_from(g, i - 1, j + 1, m.b5);
        const bool w0 = inactive_cell_underlying(g, i, j), w1 = inactive_cell_underlying(g, i - 1, j), w2 = inactive_cell_underlying(g, i, j - 1),
                   w3 = inactive_cell_underlying(g, i - 1, j - 1), w4 = inactive_cell_underlying(g, i, j + 1), w5 = inactive_cell_underlying(g, i - 1, j + 1);
        r.cc0 = n0 & !w0; r.ccm = n1 & !w1;                                  // cells (i, j), (i - 1, j)
        r.ff0 = (n0 | n1 | n2 | n3) & !(w0 | w1 | w2 | w3);                  // corner (i, j): cells (i, j), (i-1, j), (i, j-1), (i-1, j-1)
        r.ffp = (n4 | n5 | n0 | n1) & !(w4 | w5 | w0 | w1);                  // corner (i, j + 1)
    }
    return r;
}
template <bool MASK>
__device__ __forceinline__ UMask vmask_from(const GridDev& g, int i, int j, const MaskIn& m) {
    UMask r{false, false, false, false, false};
    const bool n0 = inactive_from(g, i, j, m.b0), n1 = inactive_from(g, i, j - 1, m.b1);
    r.peripheral = n0 | n1;
    if (MASK) {
        const bool n2 = inactive_from(g, i - 1, j, m.b2), n3 = inactive_from(g, i - 1, j - 1, m.b3), n4 = inactive_from(g, i + 1, j, m.b4), n5 = inactive_from(g, i + 1, j - 1, m.b5);
        const bool w0 = inactive_cell_underlying(g, i, j), w1 = inactive_cell_underlying(g, i, j - 1), w2 = inactive_cell_underlying(g, i - 1, j),
                   w3 = inactive_cell_underlying(g, i - 1, j - 1), w4 = inactive_cell_underlying(g, i + 1, j), w5 = inactive_cell_underlying(g, i + 1, j - 1);
        r.cc0 = n0 & !w0; r.ccm = n1 & !w1;                                  // cells (i, j), (i, j - 1)
        r.ff0 = (n0 | n2 | n1 | n3) & !(w0 | w2 | w1 | w3);                  // corner (i, j)
        r.ffp = (n4 | n0 | n5 | n1) & !(w4 | w0 | w5 | w1);                  // corner (i + 1, j): cells (i+1, j), (i, j), (i+1, j-1), (i, j-1)
    }
    return r;
}

// ------------------------------------------------------------------------------------------------
// velocity phases
// ------------------------------------------------------------------------------------------------
// u after the sub-step at (i, j), given v at the four points around it: (i - 1, j), (i, j), (i - 1, j + 1), (i, j + 1)
struct UInRow { VelIn q; double s11_0, s11_m, s12_0, s12_p, cE, cFN, cFS, cFU; };
template <bool UNI, bool MASK>
__device__ __forceinline__ UInRow ustep_gather(const EvpDev& P, const FastCoef& c, int i, int j) {
    // every load of this point (nothing between them waits for a loaded value: one memory round trip instead of a dozen)
    UInRow g;
    g.q = gather_u<MASK>(P, i, j);
    g.s11_0 = P.s11(i, j); g.s11_m = P.s11(i - 1, j); g.s12_0 = P.s12(i, j); g.s12_p = P.s12(i, j + 1);
    g.cE = coef<UNI>(c, FC_E, j); g.cFN = coef<UNI>(c, FC_FN, j); g.cFS = coef<UNI>(c, FC_FS, j); g.cFU = coef<UNI>(c, FC_FU, j);
    return g;
}
template <bool UNI, bool MASK>
__device__ __forceinline__ double ustep_compute(const EvpDev& P, const FastCoef& c, int i, int j, const UInRow& g, double v_m0, double v_00, double v_mp, double v_0p) {
    const VelIn& q = g.q;
    double s11_0 = g.s11_0, s11_m = g.s11_m, s12_0 = g.s12_0, s12_p = g.s12_p;
    const double cE = g.cE, cFN = g.cFN, cFS = g.cFS, cFU = g.cFU;
    const UMask k = umask_from<MASK>(P.g, i, j, q.m);
    if (MASK) {
        if (k.cc0) s11_0 = 0.0;
        if (k.ccm) s11_m = 0.0;
        if (k.ff0) s12_0 = 0.0;
        if (k.ffp) s12_p = 0.0;
    }
    const double u = q.w, un = q.wn;
    const double vbar = fm::avg4(v_m0, v_00, v_mp, v_0p);
    double div = fm::div1(cE, cFN, cFS, s11_0, s11_m, s12_p, s12_0);
    double ext, imt, exb, imb;
    stress_x_from(P.top, q.top, u, vbar, ext, imt);
    stress_x_from(P.bot, q.bot, u, vbar, exb, imb);
    double cor = cFU * vbar;               // -x_f_cross_U = +f vbar (f = 0 without Coriolis)
    if (P.extra) { if (P.has_forcing) cor += q.forcing; div += immersed_div_sigma_1(P, i, j); }   // model.forcing.u; immersed flux BCs
    const double mi = fm::avg2(q.hm * P.rho * q.am, q.h0 * P.rho * q.a0), ai = fm::avg2(q.am, q.a0), abar = fm::avg2(q.alm, q.al0);
    return P.free_drift
        ? fm::vel_update_avg_fd(vel_const(P, c), u, un, mi, ai, abar, div, cor, ext, imt, exb, imb, k.peripheral, q.fd)
        : fm::vel_update_avg(vel_const(P, c), u, un, mi, ai, abar, div, cor, ext, imt, exb, imb, k.peripheral);
}
template <bool UNI, bool MASK>
__device__ __forceinline__ double ustep_value(const EvpDev& P, const FastCoef& c, int i, int j, double v_m0, double v_00, double v_mp, double v_0p) {
    const UInRow g = ustep_gather<UNI, MASK>(P, c, i, j);
    __builtin_amdgcn_sched_barrier(0);
    return ustep_compute<UNI, MASK>(P, c, i, j, g, v_m0, v_00, v_mp, v_0p);
}
template <bool UNI, bool MASK>
__global__ void __launch_bounds__(256) k_ustep(EvpDev P, Range r, ImageSpec img, FastCoef c, TileMap tm) {
    CELL_IJ(r, tm)
    store_with_images(P.u, P.g, img, i, j, ustep_value<UNI, MASK>(P, c, i, j, P.v(i - 1, j), P.v(i, j), P.v(i - 1, j + 1), P.v(i, j + 1)));
}

// v after the sub-step at (i, j), given u at (i, j - 1), (i + 1, j - 1), (i, j), (i + 1, j)
struct VInRow { VelIn q; double s11_0, s11_m, s22_0, s22_m, s12_0, s12_p, cQ1N, cQ2N, cQ1S, cQ2S, cK, cFV; };
template <bool UNI, bool MASK>
__device__ __forceinline__ VInRow vstep_gather(const EvpDev& P, const FastCoef& c, int i, int j) {
    VInRow g;
    g.q = gather_v<MASK>(P, i, j);
    g.s11_0 = P.s11(i, j); g.s11_m = P.s11(i, j - 1); g.s22_0 = P.s22(i, j); g.s22_m = P.s22(i, j - 1);
    g.s12_0 = P.s12(i, j); g.s12_p = P.s12(i + 1, j);
    g.cQ1N = coef<UNI>(c, FC_Q1N, j); g.cQ2N = coef<UNI>(c, FC_Q2N, j); g.cQ1S = coef<UNI>(c, FC_Q1S, j); g.cQ2S = coef<UNI>(c, FC_Q2S, j);
    g.cK = coef<UNI>(c, FC_K, j); g.cFV = coef<UNI>(c, FC_FV, j);
    return g;
}
template <bool UNI, bool MASK>
__device__ __forceinline__ double vstep_compute(const EvpDev& P, const FastCoef& c, int i, int j, const VInRow& g, double u_0m, double u_pm, double u_00, double u_p0) {
    const VelIn& q = g.q;
    double s11_0 = g.s11_0, s11_m = g.s11_m, s22_0 = g.s22_0, s22_m = g.s22_m, s12_0 = g.s12_0, s12_p = g.s12_p;
    const double cQ1N = g.cQ1N, cQ2N = g.cQ2N, cQ1S = g.cQ1S, cQ2S = g.cQ2S, cK = g.cK, cFV = g.cFV;
    const UMask k = vmask_from<MASK>(P.g, i, j, q.m);
    if (MASK) {
        if (k.cc0) { s11_0 = 0.0; s22_0 = 0.0; }
        if (k.ccm) { s11_m = 0.0; s22_m = 0.0; }
        if (k.ff0) s12_0 = 0.0;
        if (k.ffp) s12_p = 0.0;
    }
    const double v = q.w, vn = q.wn;
    const double ubar = fm::avg4(u_0m, u_pm, u_00, u_p0);
    double div = fm::div2<UNI>(cQ1N, cQ2N, cQ1S, cQ2S, cK, s11_0, s22_0, s11_m, s22_m, s12_p, s12_0);
    double ext, imt, exb, imb;
    stress_y_from(P.top, q.top, v, ubar, ext, imt);
    stress_y_from(P.bot, q.bot, v, ubar, exb, imb);
    double cor = -cFV * ubar;             // -y_f_cross_U = -f ubar
    if (P.extra) { if (P.has_forcing) cor += q.forcing; div += immersed_div_sigma_2(P, i, j); }
    const double mi = fm::avg2(q.hm * P.rho * q.am, q.h0 * P.rho * q.a0), ai = fm::avg2(q.am, q.a0), abar = fm::avg2(q.alm, q.al0);
    return P.free_drift
        ? fm::vel_update_avg_fd(vel_const(P, c), v, vn, mi, ai, abar, div, cor, ext, imt, exb, imb, k.peripheral, q.fd)
        : fm::vel_update_avg(vel_const(P, c), v, vn, mi, ai, abar, div, cor, ext, imt, exb, imb, k.peripheral);
}
template <bool UNI, bool MASK>
__device__ __forceinline__ double vstep_value(const EvpDev& P, const FastCoef& c, int i, int j, double u_0m, double u_pm, double u_00, double u_p0) {
    const VInRow g = vstep_gather<UNI, MASK>(P, c, i, j);
    __builtin_amdgcn_sched_barrier(0);
    return vstep_compute<UNI, MASK>(P, c, i, j, g, u_0m, u_pm, u_00, u_p0);
}
template <bool UNI, bool MASK>
__global__ void __launch_bounds__(256) k_vstep(EvpDev P, Range r, ImageSpec img, FastCoef c, TileMap tm) {
    CELL_IJ(r, tm)
    store_with_images(P.v, P.g, img, i, j, vstep_value<UNI, MASK>(P, c, i, j, P.u(i, j - 1), P.u(i + 1, j - 1), P.u(i, j), P.u(i + 1, j)));
}


// ------------------------------------------------------------------------------------------------
// Orthogonal curvilinear grids (CSI_METRIC_FULL): the same three phases with per-POINT stencil coefficients
// (csi_fast_coef.h, C2_*: fourteen metric planes; fm::full_* are the reference's operators in terms of them).
// The arithmetic after the strain rates / divergences is that of the regular-grid kernels (evp_fast_math.h).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double c2(const FastCoef& c, int w, int i, int j) { return c.c2[(long)w * c.c2_plane + i + (long)j * c.c2_ld]; }

// the plane values either straight from memory (where they are used) or from a window around (i, j) loaded beforehand (the fold band's
// one-wave launches: every load of the point in flight at once, stress_value2_hoisted)
struct C2Direct {
    const FastCoef& c;
    static constexpr bool hoisted = false;
    __device__ __forceinline__ double operator()(int w, int i, int j) const { return c2(c, w, i, j); }
};
struct C2Window {
    double v[C2_COUNT][3][3];
    int i0, j0;                       // v[w][a][b] = plane w at (i0 + a, j0 + b)
    static constexpr bool hoisted = true;
    __device__ __forceinline__ double operator()(int w, int i, int j) const { return v[w][i - i0][j - j0]; }
};
template <class A>
__device__ __forceinline__ void strain_cell2(const A& m, int i, int j, double u_e, double u_w, double v_n, double v_s, double& e11, double& e22) {
    fm::full_strain_cell(m(C2_DYU, i + 1, j) * u_e, m(C2_DYU, i, j) * u_w, m(C2_DXV, i, j + 1) * v_n, m(C2_DXV, i, j) * v_s,
                         fm::rcp(m(C2_DYU, i + 1, j)) * u_e, fm::rcp(m(C2_DYU, i, j)) * u_w, fm::rcp(m(C2_DXV, i, j + 1)) * v_n, fm::rcp(m(C2_DXV, i, j)) * v_s,
                         m(C2_DYC2, i, j), m(C2_DXC2, i, j), m(C2_RAZC, i, j), e11, e22);
}
template <class A>
__device__ __forceinline__ double strain_corner2(const A& m, int i, int j, double u_n, double u_s, double v_e, double v_w) {
    return fm::full_strain_corner(m(C2_RDXU, i, j) * u_n, m(C2_RDXU, i, j - 1) * u_s, m(C2_RDYV, i, j) * v_e, m(C2_RDYV, i - 1, j) * v_w,
                                  m(C2_DXF2, i, j), m(C2_DYF2, i, j), m(C2_RAZF, i, j));
}

template <class A>
__device__ __forceinline__ fm::StressOut stress_core2(const EvpDev& P, const FastCoef& c, const A& m, int i, int j) {
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
    if (A::hoisted) __builtin_amdgcn_sched_barrier(0);
    double e11_00, e22_00, e11_m0, e22_m0, e11_0m, e22_0m, e11_mm, e22_mm;
    strain_cell2(m, i, j, u_p0, u_00, v_0p, v_00, e11_00, e22_00);
    strain_cell2(m, i - 1, j, u_00, u_m0, v_mp, v_m0, e11_m0, e22_m0);
    strain_cell2(m, i, j - 1, u_pm, u_0m, v_00, v_0m, e11_0m, e22_0m);
    strain_cell2(m, i - 1, j - 1, u_0m, u_mm, v_m0, v_mm, e11_mm, e22_mm);
    const double e12_00 = strain_corner2(m, i, j, u_00, u_0m, v_00, v_m0);
    const double e12_p0 = strain_corner2(m, i + 1, j, u_p0, u_pm, v_p0, v_00);
    const double e12_0p = strain_corner2(m, i, j + 1, u_0p, u_00, v_0p, v_mp);
    const double e12_pp = strain_corner2(m, i + 1, j + 1, u_pp, u_p0, v_pp, v_0p);
    const double e11f = fm::avg4(e11_mm, e11_0m, e11_m0, e11_00);
    const double e22f = fm::avg4(e22_mm, e22_0m, e22_m0, e22_00);
    const double e12c = fm::avg4(e12_00, e12_p0, e12_0p, e12_pp);
    const double Pf = fm::avg4(P_mm, P_0m, P_m0, P_00);
    const double m_00 = h_00 * P.rho * a_00, m_m0 = h_m0 * P.rho * a_m0, m_0m = h_0m * P.rho * a_0m, m_mm = h_mm * P.rho * a_mm;
    const double mf = fm::avg4(m_mm, m_0m, m_m0, m_00);
    const double kc = c.ca_dt * m(C2_RAZC, i, j), kf = c.ca_dt * m(C2_RAZF, i, j);
    return fm::stress_update(stress_const(P, c), e11_00, e22_00, e12_00, e11f, e22f, e12c, P_00, Pf,
                             m_00, mf, kc, kf, s11, s22, s12);
}
__device__ __forceinline__ fm::StressOut stress_value2(const EvpDev& P, const FastCoef& c, int i, int j) {
    return stress_core2(P, c, C2Direct{c}, i, j);
}
// the same with the 48 plane values of the point loaded first (160 vector registers more: one-wave workgroups only)
__device__ __forceinline__ fm::StressOut stress_value2_hoisted(const EvpDev& P, const FastCoef& c, int i, int j) {
    C2Window w;
    w.i0 = i - 1; w.j0 = j - 1;
#define CSI_W(pl, a, b) w.v[pl][(a) + 1][(b) + 1] = c2(c, pl, i + (a), j + (b))
#pragma unroll
    for (int a = -1; a <= 1; ++a)
#pragma unroll
        for (int b = -1; b <= 0; ++b) CSI_W(C2_DYU, a, b);           // cells (i-1 .. i, j-1 .. j): their east and west faces
#pragma unroll
    for (int a = -1; a <= 0; ++a)
#pragma unroll
        for (int b = -1; b <= 1; ++b) CSI_W(C2_DXV, a, b);           // ... north and south faces
#pragma unroll
    for (int a = -1; a <= 0; ++a)
#pragma unroll
        for (int b = -1; b <= 0; ++b) { CSI_W(C2_DYC2, a, b); CSI_W(C2_DXC2, a, b); CSI_W(C2_RAZC, a, b); }
#pragma unroll
    for (int a = 0; a <= 1; ++a)
#pragma unroll
        for (int b = -1; b <= 1; ++b) CSI_W(C2_RDXU, a, b);          // corners (i .. i+1, j .. j+1): the u points north and south of them
#pragma unroll
    for (int a = -1; a <= 1; ++a)
#pragma unroll
        for (int b = 0; b <= 1; ++b) CSI_W(C2_RDYV, a, b);           // ... the v points east and west
#pragma unroll
    for (int a = 0; a <= 1; ++a)
#pragma unroll
        for (int b = 0; b <= 1; ++b) { CSI_W(C2_DXF2, a, b); CSI_W(C2_DYF2, a, b); CSI_W(C2_RAZF, a, b); }
#undef CSI_W
    return stress_core2(P, c, w, i, j);
}
__global__ void __launch_bounds__(256) k_stress2(EvpDev P, Range r, FastCoef c, TileMap tm) {
    CELL_IJ(r, tm)
    store_stress(P, i, j, stress_value2(P, c, i, j));
}

struct UInFull { VelIn q; double s11_0, s11_m, s22_0, s22_m, s12_0, s12_p, dyu, rdxu, razu, dyc2_0, dyc2_m, dxf2_p, dxf2_0, f; };
template <bool MASK>
__device__ __forceinline__ UInFull ustep_gather2(const EvpDev& P, const FastCoef& c, int i, int j) {
    UInFull g;
    g.q = gather_u<MASK>(P, i, j);
    g.s11_0 = P.s11(i, j); g.s11_m = P.s11(i - 1, j); g.s22_0 = P.s22(i, j); g.s22_m = P.s22(i - 1, j); g.s12_0 = P.s12(i, j); g.s12_p = P.s12(i, j + 1);
    g.dyu = c2(c, C2_DYU, i, j); g.rdxu = c2(c, C2_RDXU, i, j); g.razu = c2(c, C2_RAZU, i, j); g.dyc2_0 = c2(c, C2_DYC2, i, j); g.dyc2_m = c2(c, C2_DYC2, i - 1, j);
    g.dxf2_p = c2(c, C2_DXF2, i, j + 1); g.dxf2_0 = c2(c, C2_DXF2, i, j); g.f = fcor_at_u(P, i, j);
    return g;
}
template <bool MASK>
__device__ __forceinline__ double ustep_compute2(const EvpDev& P, const FastCoef& c, int i, int j, const UInFull& g, double v_m0, double v_00, double v_mp, double v_0p) {
    const VelIn& q = g.q;
    double s11_0 = g.s11_0, s11_m = g.s11_m, s22_0 = g.s22_0, s22_m = g.s22_m, s12_0 = g.s12_0, s12_p = g.s12_p;
    const double dyu = g.dyu, rdxu = g.rdxu, razu = g.razu, dyc2_0 = g.dyc2_0, dyc2_m = g.dyc2_m, dxf2_p = g.dxf2_p, dxf2_0 = g.dxf2_0, f = g.f;
    const UMask k = umask_from<MASK>(P.g, i, j, q.m);
    if (MASK) {
        if (k.cc0) { s11_0 = 0.0; s22_0 = 0.0; }
        if (k.ccm) { s11_m = 0.0; s22_m = 0.0; }
        if (k.ff0) s12_0 = 0.0;
        if (k.ffp) s12_p = 0.0;
    }
    const double u = q.w, un = q.wn;
    const double vbar = fm::avg4(v_m0, v_00, v_mp, v_0p);
    double div = fm::full_div1(dyu, fm::rcp(dyu), rdxu, razu,
                               s11_0 + s22_0, s11_m + s22_m, dyc2_0 * (s11_0 - s22_0), dyc2_m * (s11_m - s22_m),
                               dxf2_p * s12_p, dxf2_0 * s12_0);
    double ext, imt, exb, imb;
    stress_x_from(P.top, q.top, u, vbar, ext, imt);
    stress_x_from(P.bot, q.bot, u, vbar, exb, imb);
    double cor = f * vbar;
    if (P.extra) { if (P.has_forcing) cor += q.forcing; div += immersed_div_sigma_1(P, i, j); }
    const double mi = fm::avg2(q.hm * P.rho * q.am, q.h0 * P.rho * q.a0), ai = fm::avg2(q.am, q.a0), abar = fm::avg2(q.alm, q.al0);
    return P.free_drift
        ? fm::vel_update_avg_fd(vel_const(P, c), u, un, mi, ai, abar, div, cor, ext, imt, exb, imb, k.peripheral, q.fd)
        : fm::vel_update_avg(vel_const(P, c), u, un, mi, ai, abar, div, cor, ext, imt, exb, imb, k.peripheral);
}
template <bool MASK>
__device__ __forceinline__ double ustep_value2(const EvpDev& P, const FastCoef& c, int i, int j, double v_m0, double v_00, double v_mp, double v_0p) {
    const UInFull g = ustep_gather2<MASK>(P, c, i, j);
    __builtin_amdgcn_sched_barrier(0);
    return ustep_compute2<MASK>(P, c, i, j, g, v_m0, v_00, v_mp, v_0p);
}
template <bool MASK>
__global__ void __launch_bounds__(256) k_ustep2(EvpDev P, Range r, ImageSpec img, FastCoef c, TileMap tm) {
    CELL_IJ(r, tm)
    store_with_images(P.u, P.g, img, i, j, ustep_value2<MASK>(P, c, i, j, P.v(i - 1, j), P.v(i, j), P.v(i - 1, j + 1), P.v(i, j + 1)));
}

struct VInFull { VelIn q; double s11_0, s11_m, s22_0, s22_m, s12_0, s12_p, dxv, rdyv, razv, dxc2_0, dxc2_m, dyf2_p, dyf2_0, f; };
template <bool MASK>
__device__ __forceinline__ VInFull vstep_gather2(const EvpDev& P, const FastCoef& c, int i, int j) {
    VInFull g;
    g.q = gather_v<MASK>(P, i, j);
    g.s11_0 = P.s11(i, j); g.s11_m = P.s11(i, j - 1); g.s22_0 = P.s22(i, j); g.s22_m = P.s22(i, j - 1);
    g.s12_0 = P.s12(i, j); g.s12_p = P.s12(i + 1, j);
    g.dxv = c2(c, C2_DXV, i, j); g.rdyv = c2(c, C2_RDYV, i, j); g.razv = c2(c, C2_RAZV, i, j); g.dxc2_0 = c2(c, C2_DXC2, i, j); g.dxc2_m = c2(c, C2_DXC2, i, j - 1);
    g.dyf2_p = c2(c, C2_DYF2, i + 1, j); g.dyf2_0 = c2(c, C2_DYF2, i, j); g.f = fcor_at_v(P, i, j);
    return g;
}
template <bool MASK>
__device__ __forceinline__ double vstep_compute2(const EvpDev& P, const FastCoef& c, int i, int j, const VInFull& g, double u_0m, double u_pm, double u_00, double u_p0) {
    const VelIn& q = g.q;
    double s11_0 = g.s11_0, s11_m = g.s11_m, s22_0 = g.s22_0, s22_m = g.s22_m, s12_0 = g.s12_0, s12_p = g.s12_p;
    const double dxv = g.dxv, rdyv = g.rdyv, razv = g.razv, dxc2_0 = g.dxc2_0, dxc2_m = g.dxc2_m, dyf2_p = g.dyf2_p, dyf2_0 = g.dyf2_0, f = g.f;
    const UMask k = vmask_from<MASK>(P.g, i, j, q.m);
    if (MASK) {
        if (k.cc0) { s11_0 = 0.0; s22_0 = 0.0; }
        if (k.ccm) { s11_m = 0.0; s22_m = 0.0; }
        if (k.ff0) s12_0 = 0.0;
        if (k.ffp) s12_p = 0.0;
    }
    const double v = q.w, vn = q.wn;
    const double ubar = fm::avg4(u_0m, u_pm, u_00, u_p0);
    double div = fm::full_div2(dxv, fm::rcp(dxv), rdyv, razv,
                               s11_0 + s22_0, s11_m + s22_m, dxc2_0 * (s11_0 - s22_0), dxc2_m * (s11_m - s22_m),
                               dyf2_p * s12_p, dyf2_0 * s12_0);
    double ext, imt, exb, imb;
    stress_y_from(P.top, q.top, v, ubar, ext, imt);
    stress_y_from(P.bot, q.bot, v, ubar, exb, imb);
    double cor = -f * ubar;
    if (P.extra) { if (P.has_forcing) cor += q.forcing; div += immersed_div_sigma_2(P, i, j); }
    const double mi = fm::avg2(q.hm * P.rho * q.am, q.h0 * P.rho * q.a0), ai = fm::avg2(q.am, q.a0), abar = fm::avg2(q.alm, q.al0);
    return P.free_drift
        ? fm::vel_update_avg_fd(vel_const(P, c), v, vn, mi, ai, abar, div, cor, ext, imt, exb, imb, k.peripheral, q.fd)
        : fm::vel_update_avg(vel_const(P, c), v, vn, mi, ai, abar, div, cor, ext, imt, exb, imb, k.peripheral);
}
template <bool MASK>
__device__ __forceinline__ double vstep_value2(const EvpDev& P, const FastCoef& c, int i, int j, double u_0m, double u_pm, double u_00, double u_p0) {
    const VInFull g = vstep_gather2<MASK>(P, c, i, j);
    __builtin_amdgcn_sched_barrier(0);
    return vstep_compute2<MASK>(P, c, i, j, g, u_0m, u_pm, u_00, u_p0);
}
template <bool MASK>
__global__ void __launch_bounds__(256) k_vstep2(EvpDev P, Range r, ImageSpec img, FastCoef c, TileMap tm) {
    CELL_IJ(r, tm)
    store_with_images(P.v, P.g, img, i, j, vstep_value2<MASK>(P, c, i, j, P.u(i, j - 1), P.u(i + 1, j - 1), P.u(i, j), P.u(i + 1, j)));
}

// ------------------------------------------------------------------------------------------------
// The fold band's launches (csi_fold.hip, round 6b).  Beside a pair launch the band of rows next to a north fold is the critical path
// of a tripolar sub-cycle: a chain of dependent, latency-bound launches (alone on the chip 96 us per pair of sub-steps -- eight
// launches of 5-10 us with 3-4 us between them -- against the pair launch's 97; profiles/r06_band.md).  What shortens the chain:
//   * every load of a point in flight before any is used (the value functions above; the stress launch with its 48 plane values
//     loaded first, stress_value2_hoisted: a one-wave workgroup has the registers);
//   * no copy kernels: inputs and outputs are separate arrays -- the first sub-step reads the current buffer and writes the band's
//     copies, the second works on those in place and stores rows >= M + 1 (BandOut::j0) into the other buffer as well.
// (Measured and NOT kept: both velocity components in one launch, the first evaluated again at the four points the second reads --
//  five evaluations in one thread take longer than two launches of one, and a 256-register wave finds no slot beside a pair launch.)
// KIND: 0 uniform coefficients, 1 per row, 2 per point (CSI_METRIC_FULL).
template <int KIND>
__device__ __forceinline__ fm::StressOut band_stress_value(const EvpDev& P, const FastCoef& c, int i, int j) {
    if constexpr (KIND == 2) return stress_value2_hoisted(P, c, i, j);
    else return stress_value<KIND == 0>(P, c, i, j);
}
template <int KIND>
__global__ void __launch_bounds__(64) k_band_stress(EvpDev P, Range r, FastCoef c, TileMap tm, BandOut o) {
    CELL_IJ(r, tm)
    const fm::StressOut s = band_stress_value<KIND>(P, c, i, j);
    o.a(i, j) = s.s11; o.b(i, j) = s.s22; o.c(i, j) = s.s12;
    P.al(i, j) = s.alpha;
    if (o.da.p && j >= o.j0) { o.da(i, j) = s.s11; o.db(i, j) = s.s22; o.dc(i, j) = s.s12; }
    if (P.write_diag) {
        P.zf(i, j) = 0.5 * s.zf2;
        P.zc(i, j) = 0.5 * s.zc2;
        P.Dl(i, j) = s.xc * s.rDc;
    }
}

// one velocity component (U: u, else v) of the band: inputs from P, the result (+ its halo images) into o.a and, rows >= o.j0, into o.da
template <int KIND, bool MASK, bool U>
__global__ void __launch_bounds__(64) k_band_vel(EvpDev P, Range r, ImageSpec img, FastCoef c, TileMap tm, BandOut o) {
    CELL_IJ(r, tm)
    double res;
    if constexpr (U) {
        const double v_m0 = P.v(i - 1, j), v_00 = P.v(i, j), v_mp = P.v(i - 1, j + 1), v_0p = P.v(i, j + 1);
        if constexpr (KIND == 2) res = ustep_value2<MASK>(P, c, i, j, v_m0, v_00, v_mp, v_0p);
        else res = ustep_value<KIND == 0, MASK>(P, c, i, j, v_m0, v_00, v_mp, v_0p);
    } else {
        const double u_0m = P.u(i, j - 1), u_pm = P.u(i + 1, j - 1), u_00 = P.u(i, j), u_p0 = P.u(i + 1, j);
        if constexpr (KIND == 2) res = vstep_value2<MASK>(P, c, i, j, u_0m, u_pm, u_00, u_p0);
        else res = vstep_value<KIND == 0, MASK>(P, c, i, j, u_0m, u_pm, u_00, u_p0);
    }
    store_with_images(o.a, P.g, img, i, j, res);
    if (o.da.p && j >= o.j0) store_with_images(o.da, P.g, img, i, j, res);
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

// the fold band's fused launches (fast::k_band_stress, fast::k_band_uv)
static fast::TileMap tile_map_band(const EvpDev& P, const Range& r, dim3& grid) {      // one-wave workgroups whatever the number of rows
    fast::TileMap tm;
    tm.ibase = 1 - P.g.Hx;
    if (tm.ibase > r.i0) tm.ibase = r.i0;
    tm.jbase = r.j0;
    tm.gx = (r.i1 - tm.ibase + fast::TILE_X) / fast::TILE_X;
    tm.ty = 1;
    tm.ntiles = tm.gx * (r.j1 - r.j0 + 1);
    tm.per_xcd = (tm.ntiles + 7) / 8;
    grid = dim3((unsigned)(tm.per_xcd * 8), 1, 1);
    return tm;
}
void launch_band_stress(const EvpDev& P, const Range& r, const FastCoef& c, const BandOut& o, hipStream_t s) {
    dim3 b(fast::TILE_X, 1), g;
    const fast::TileMap tm = tile_map_band(P, r, g);
    if (c.full) hipLaunchKernelGGL((fast::k_band_stress<2>), g, b, 0, s, P, r, c, tm, o);
    else if (c.uniform) hipLaunchKernelGGL((fast::k_band_stress<0>), g, b, 0, s, P, r, c, tm, o);
    else hipLaunchKernelGGL((fast::k_band_stress<1>), g, b, 0, s, P, r, c, tm, o);
}
template <int KIND>
static void launch_band_vel_kind(const EvpDev& P, const Range& r, const ImageSpec& im, const FastCoef& c, bool u, const BandOut& o, hipStream_t s) {
    dim3 b(fast::TILE_X, 1), g;
    const fast::TileMap tm = tile_map_band(P, r, g);
    const bool m = P.g.has_mask != 0;
    if (m) {
        if (u) hipLaunchKernelGGL((fast::k_band_vel<KIND, true, true>), g, b, 0, s, P, r, im, c, tm, o);
        else hipLaunchKernelGGL((fast::k_band_vel<KIND, true, false>), g, b, 0, s, P, r, im, c, tm, o);
    } else {
        if (u) hipLaunchKernelGGL((fast::k_band_vel<KIND, false, true>), g, b, 0, s, P, r, im, c, tm, o);
        else hipLaunchKernelGGL((fast::k_band_vel<KIND, false, false>), g, b, 0, s, P, r, im, c, tm, o);
    }
}
void launch_band_vel(const EvpDev& P, const Range& r, const ImageSpec& im, const FastCoef& c, bool u, const BandOut& o, hipStream_t s) {
    if (c.full) launch_band_vel_kind<2>(P, r, im, c, u, o, s);
    else if (c.uniform) launch_band_vel_kind<0>(P, r, im, c, u, o, s);
    else launch_band_vel_kind<1>(P, r, im, c, u, o, s);
}

}  // namespace csi
