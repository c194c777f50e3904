// csi_fast_coef.h -- per-row stencil coefficients of the FAST kernels.
//
// The reference evaluates the strain rates and the stress divergence with metric weights inside
// the differences (Rheologies/elasto_visco_plastic_rheology.jl:360-375,
// Rheologies/ice_stress_divergence.jl:39-51).  For the grids this library supports
// (regular rectilinear, regular latitude-longitude: dy constant, dx and Az functions of j only)
// those operators are short linear stencils whose weights depend on the row j alone.  The
// weights are computed once on the host in fp64 (build_fast_coef) and read by the kernels as
// wave-uniform scalars; on a uniform grid they are kernel-argument constants.
#pragma once
#include <vector>

namespace csi {

enum : int {
    // Order = order of use, so that the constants of one phase sit next to each other (the scalar loads of adjacent
    // table entries merge into wide ones).  Centre-row (j) and face-row coefficients are marked (c) / (f).
    FC_A = 0,   // (c) e11: A (u[i+1] - u[i])               = dy / Az^cc
    FC_CN,      // (c) e22: + CN v[j+1]
    FC_BN,      // (c) e11: + BN v[j+1]                      = (dxf[j+1]/Az - dxc^2/(dxf[j+1] Az)) / 2
    FC_BS,      // (c) e11: - BS v[j]
    FC_CS,      // (c) e22: - CS v[j]
    FC_RAZC,    // (c) 1 / Az^cc
    FC_RAZF,    // (f) 1 / Az^ff
    FC_SN,      // (f) e12: + SN u[j]                        = dxf[j]^2 / (2 dxc[j] Az^ff)
    FC_SV,      // (f) e12: + SV (v[i] - v[i-1])             = dy / (2 Az^ff)
    FC_SS,      // (f) e12: - SS u[j-1]
    FC_E,       // (c) div1: E (s11[i] - s11[i-1])           = dy / Az^fc
    FC_FN,      // (c) div1: + FN s12[j+1]                   = dxf[j+1]^2 / (dxc[j] Az^fc)
    FC_FS,      // (c) div1: - FS s12[j]
    FC_FU,      // (c) Coriolis parameter at the u points of the row (FPlane: the same in every row; BetaPlane: csi_coriolis_rows_set)
    FC_Q2N,     // (f) div2: + Q1N s11[j] + Q2N s22[j] - Q1S s11[j-1] - Q2S s22[j-1] + K (s12[i+1] - s12[i])
    FC_K,
    FC_FV,      // (f) Coriolis parameter at the v points of the row
    FC_Q1N,
    FC_Q1S,
    FC_Q2S,
    FC_COUNT
};

// Per-point metric planes for grids whose metrics vary in both directions (CSI_METRIC_FULL): the reference's operators with
// the metric at the location and index each one names (elasto_visco_plastic_rheology.jl:360-375,
// ice_stress_divergence.jl:39-51), written in terms of TWELVE planes -- three at the u points, three at the v points,
// three at the cells, three at the corners -- instead of one folded coefficient per stencil entry (26 planes, round 2 a):
// the kernels on such grids are bound by the planes' traffic and load count, not by arithmetic (fm::full_* in
// evp_fast_math.h spell the stencils out).
enum : int {
    C2_DYU = 0, C2_RDXU, C2_RAZU,               // u points (Face, Center): dy, 1 / dx, 1 / Az     (1 / dy: fm::rcp in the kernels,
    C2_DXV, C2_RDYV, C2_RAZV,                   // v points (Center, Face): dx, 1 / dy, 1 / Az      1 / dx likewise: two loads fewer)
    C2_DYC2, C2_DXC2, C2_RAZC,                  // cells: dy^2, dx^2, 1 / Az
    C2_DXF2, C2_DYF2, C2_RAZF,                  // corners (Face, Face): dx^2, dy^2, 1 / Az
    C2_COUNT
};

// The two-sub-steps kernel's row pipelines (evp_pair_stage.h) read their own copy of the coefficients, scaled by exact powers of
// two: the corner strain rate times 8 and the stress divergences times 2 (fm::stress_update_s, fm::vel_update_sum)
inline double pair_coef_scale(int which) {
    switch (which) {
        case FC_SN: case FC_SS: case FC_SV: return 8.0;
        case FC_E: case FC_FN: case FC_FS: case FC_Q1N: case FC_Q2N: case FC_Q1S: case FC_Q2S: case FC_K: return 2.0;
        default: return 1.0;
    }
}

struct FastCoef {
    double uni[FC_COUNT];   // uniform grid: the (row-independent) values
    const double* vec;      // per-j: device array [rows][FC_COUNT] (row-major), pre-offset so vec[j*stride + w] is row j
    const double* vec_pair; // ... and the copy scaled by pair_coef_scale
    int stride;
    int uniform;
    int jmin, jmax;         // valid row range of `vec`
    double em2;             // e^-2
    double ca_dt;           // HALF of c_alpha * dt (stage step): see fm::stress_update
    double hkc, hkf;        // uniform grid: ca_dt / Az at cells / corners
    double hk1;             // (1 - e^-2) / 2
    // orthogonal curvilinear grids (CSI_METRIC_FULL): C2_COUNT planes of per-POINT coefficients, element (i, j) of
    // plane w at c2[w * c2_plane + i + j * c2_ld] (pointer pre-offset); full != 0 selects the k_*2 kernels
    const double* c2;
    long c2_plane;
    int c2_ld;
    int full;
    // ... rows whose plane values are the same in every column (csi_core.hip ensure_row_constant): C2_COUNT + 2 vectors of c2row_n
    // doubles (the planes, then the per-point Coriolis planes at u / v points), entry [parent row]; rcsum: prefix sums of the marks
    const double* c2row;
    long c2row_n;
    const int* rcsum;
    double rdt;             // 1 / dt
    double Dmin2, rDmin, amin2, amax2, ramin, ramax;   // Delta_min^2, 1/Delta_min, alpha-^2, alpha+^2, 1/alpha-, 1/alpha+
};

// Host: fill `uni` (uniform) or `out` ([FC_COUNT][n], n = Ny + 2Hy + 1, entry for row j at [j + Hy - 1]).
inline void build_fast_coef_uniform(double dx, double dy, double* uni) {
    const double raz = 1.0 / (dx * dy);
    uni[FC_A] = dy * raz;
    const double d = dx * raz, t = dx * dx / dx * raz;
    uni[FC_BN] = 0.5 * (d - t); uni[FC_BS] = 0.5 * (d - t);
    uni[FC_CN] = 0.5 * (d + t); uni[FC_CS] = 0.5 * (d + t);
    uni[FC_E] = dy * raz;
    uni[FC_FN] = dx * dx / dx * raz; uni[FC_FS] = uni[FC_FN];
    uni[FC_RAZC] = raz;
    uni[FC_SN] = 0.5 * dx * dx / dx * raz; uni[FC_SS] = uni[FC_SN];
    uni[FC_SV] = 0.5 * dy * raz;
    const double G = 0.5 * dx * raz, H = 0.5 * dx * dx / dx * raz;
    uni[FC_Q1N] = G - H; uni[FC_Q2N] = G + H; uni[FC_Q1S] = G - H; uni[FC_Q2S] = G + H;
    uni[FC_K] = dy * raz;
    uni[FC_RAZF] = raz;
}

inline void build_fast_coef_per_j(int n, double dy, const double* dxc, const double* dxf, const double* azc,
                                  const double* azf, std::vector<double>& out) {
    out.assign((size_t)FC_COUNT * n, 0.0);
    auto at = [&](int w, int t) -> double& { return out[(size_t)w * n + t]; };
    for (int t = 0; t < n; ++t) {
        const int tp = t + 1 < n ? t + 1 : t, tm = t > 0 ? t - 1 : t;   // clamped neighbours (edge rows are never used)
        const double razc = 1.0 / azc[t], razf = 1.0 / azf[t];
        at(FC_A, t) = dy * razc;
        const double dn = dxf[tp] * razc, ds = dxf[t] * razc;
        const double tn = dxc[t] * dxc[t] / dxf[tp] * razc, ts = dxc[t] * dxc[t] / dxf[t] * razc;
        at(FC_BN, t) = 0.5 * (dn - tn); at(FC_BS, t) = 0.5 * (ds - ts);
        at(FC_CN, t) = 0.5 * (dn + tn); at(FC_CS, t) = 0.5 * (ds + ts);
        at(FC_E, t) = dy * razc;
        at(FC_FN, t) = dxf[tp] * dxf[tp] / dxc[t] * razc;
        at(FC_FS, t) = dxf[t] * dxf[t] / dxc[t] * razc;
        at(FC_RAZC, t) = razc;
        at(FC_SN, t) = 0.5 * dxf[t] * dxf[t] / dxc[t] * razf;
        at(FC_SS, t) = 0.5 * dxf[t] * dxf[t] / dxc[tm] * razf;
        at(FC_SV, t) = 0.5 * dy * razf;
        const double G = 0.5 * dxf[t] * razf;
        const double Hn = 0.5 * dxc[t] * dxc[t] / dxf[t] * razf, Hs = 0.5 * dxc[tm] * dxc[tm] / dxf[t] * razf;
        at(FC_Q1N, t) = G - Hn; at(FC_Q2N, t) = G + Hn; at(FC_Q1S, t) = G - Hs; at(FC_Q2S, t) = G + Hs;
        at(FC_K, t) = dy * razf;
        at(FC_RAZF, t) = razf;
    }
}

// m[k]: the twelve metric planes of csi_metrics.full (dx, dy, Az at (c,c), (f,c), (c,f), (f,f)), each nj rows of ni
// (dense); out: C2_COUNT planes of the same shape.
inline void build_fast_coef_full(int ni, int nj, const double* const* m, std::vector<double>& out) {
    out.assign((size_t)C2_COUNT * ni * nj, 0.0);
    enum { CC = 0, FC = 1, CF = 2, FF = 3 };
    auto M = [&](int which, int loc, size_t t) -> double { return m[4 * which + loc][t]; };     // which: 0 dx, 1 dy, 2 Az
    auto O = [&](int w, size_t t) -> double& { return out[(size_t)w * nj * ni + t]; };
    for (size_t t = 0; t < (size_t)ni * nj; ++t) {
        O(C2_DYU, t) = M(1, FC, t); O(C2_RDXU, t) = 1.0 / M(0, FC, t); O(C2_RAZU, t) = 1.0 / M(2, FC, t);
        O(C2_DXV, t) = M(0, CF, t); O(C2_RDYV, t) = 1.0 / M(1, CF, t); O(C2_RAZV, t) = 1.0 / M(2, CF, t);
        O(C2_DYC2, t) = M(1, CC, t) * M(1, CC, t); O(C2_DXC2, t) = M(0, CC, t) * M(0, CC, t); O(C2_RAZC, t) = 1.0 / M(2, CC, t);
        O(C2_DXF2, t) = M(0, FF, t) * M(0, FF, t); O(C2_DYF2, t) = M(1, FF, t) * M(1, FF, t); O(C2_RAZF, t) = 1.0 / M(2, FF, t);
    }
}

}  // namespace csi
