// csi_dev.h -- device-side problem description shared by all kernels of libcsi_hip.so.
// gfx950 (MI355X) only.  Indices (i, j) are the reference's 1-based indices; a field
// reference points at the (virtual) element (0, 0) so that element (i, j) = p[i + j * ld].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace csi {

enum : int { SIDE_PERIODIC = 0, SIDE_WALL = 1, SIDE_CONNECTED = 2, SIDE_FOLD = 3 };   // SIDE_FOLD: north fold of a TripolarGrid (yhi only)
enum : int { LOC_C = 0, LOC_F = 1 };
enum : int { IMG_NONE = 0, IMG_WRAP = 1, IMG_MIRROR = 2, IMG_VALUE = 3, IMG_FOLD = 4 };

struct FRef {
    double* p;   // element (0, 0) in reference indexing
    int ld;
    __device__ __forceinline__ double& operator()(int i, int j) const { return p[i + (long)j * ld]; }
    __device__ __forceinline__ double ld_(int i, int j) const { return p[i + (long)j * ld]; }
};

struct GridDev {
    int Nx, Ny, Hx, Hy;
    int xlo, xhi, ylo, yhi;   // SIDE_* of each edge of this tile
    int metric_kind;          // 0 uniform, 1 per-j, 2 full 2-D arrays (orthogonal curvilinear grids; STRICT kernels)
    int has_mask;
    double dx, dy;
    // per-j vectors, indexable directly with the reference j (pointer pre-offset): j in [1-Hy, Ny+Hy+1]
    const double *dxc, *dxf, *azc, *azf;
    const double *rdxc, *rdxf, *razc, *razf;   // reciprocals (FAST mode)
    const uint8_t* mask;      // element (0,0)-offset like FRef
    int mask_ld;
    // metric_kind 2: twelve planes dx, dy, Az at (c,c), (f,c), (c,f), (f,f) -- plane 4 * {dx 0, dy 1, Az 2} + (x Face) +
    // 2 * (y Face) -- element (i, j) of plane k at m2[k * m2_plane + i + j * m2_ld] (pointer pre-offset)
    const double* m2;
    long m2_plane;
    int m2_ld;
};

// ---- metrics at a location (Oceananigans.Operators dx / dy / Az at (lx, ly)) --------------------
__device__ __forceinline__ double metric2(const GridDev& g, int which, int lx, int ly, int i, int j) {
    return g.m2[(4 * which + (lx == LOC_F ? 1 : 0) + (ly == LOC_F ? 2 : 0)) * g.m2_plane + i + (long)j * g.m2_ld];
}
__device__ __forceinline__ double dxm(const GridDev& g, int lx, int ly, int i, int j) {
    if (g.metric_kind == 0) return g.dx;
    if (g.metric_kind == 1) return ly == LOC_C ? g.dxc[j] : g.dxf[j];
    return metric2(g, 0, lx, ly, i, j);
}
__device__ __forceinline__ double dym(const GridDev& g, int lx, int ly, int i, int j) {
    if (g.metric_kind != 2) return g.dy;
    return metric2(g, 1, lx, ly, i, j);
}
__device__ __forceinline__ double azm(const GridDev& g, int lx, int ly, int i, int j) {
    if (g.metric_kind == 0) return g.dx * g.dy;
    if (g.metric_kind == 1) return ly == LOC_C ? g.azc[j] : g.azf[j];
    return metric2(g, 2, lx, ly, i, j);
}

struct StressDev {
    int kind, ue_kind, ve_kind, pad;
    double tau_u, tau_v, ue, ve, rho_e, Cd;
    FRef fu, fv;
};

struct EvpDev {
    GridDev g;
    FRef u, v, h, a, s11, s22, s12, zc, zf, Dl, al, P, un, vn;
    StressDev top, bot;
    double P_star, C_star, ecc, Dmin, amin, amax, ca;
    double min_mass, min_conc, rho, fcor;
    const double *fcor_u, *fcor_v;   // BetaPlane: f per row at the u / v points (ptr[j] = row j); NULL: fcor
    const double *fcor2_u, *fcor2_v; // per-point f on CSI_METRIC_FULL grids (ptr[i + j * fcor2_ld]); NULL: rows / fcor
    long fcor2_ld;
    int pressure_kind, has_cor;
    int free_drift;           // 1: StressBalanceFreeDrift, velocities of marginal ice from ufd / vfd
    FRef ufd, vfd;            // free-drift velocities at u / v points (library scratch, once per sub-cycle)
    double dt;
    int write_diag;   // FAST: also store zeta_c, zeta_f, Delta (last sub-step only)
    // rarely used terms of the velocity tendencies (three-kernel paths only; `extra` = any of them present):
    int extra;
    int has_forcing;          // model.forcing.u / .v as arrays (user_forcing of sum_of_forcing_u / _v, evp:391-401)
    FRef forcing_u, forcing_v;
    double ibc_u[4], ibc_v[4];   // immersed FluxBoundaryCondition numbers of u / v: west, east, south, north (isd:65-123)
};

struct Range { int i0, i1, j0, j1; };

// Coriolis parameter at the u / v point (i, j): per point, per row or a number
// (one conditional load from a selected address: nested loads under nested tests were two memory round trips)
__device__ __forceinline__ double fcor_at_u(const EvpDev& P, int i, int j) {
    const double* p = P.fcor2_u ? P.fcor2_u + (i + (long)j * P.fcor2_ld) : (P.fcor_u ? P.fcor_u + j : nullptr);
    return p ? *p : P.fcor;
}
__device__ __forceinline__ double fcor_at_v(const EvpDev& P, int i, int j) {
    const double* p = P.fcor2_v ? P.fcor2_v + (i + (long)j * P.fcor2_ld) : (P.fcor_v ? P.fcor_v + j : nullptr);
    return p ? *p : P.fcor;
}

// ---- activity / peripheral nodes (upstream Grids.inactive_cell / peripheral_node) ------------
// (No branch between the tests, and the mask byte is loaded from a clamped index whatever they say: a velocity kernel looks at a dozen
//  cells, and with a branch around every load each of them was a memory round trip of its own -- forty in k_ustep2 with a mask, the
//  whole duration of the fold band's latency-bound launches; round 6, profiles/r06_band.md.)
__device__ __forceinline__ bool inactive_cell_underlying(const GridDev& g, int i, int j) {
    return ((g.xlo == SIDE_WALL) & (i < 1)) | ((g.xhi == SIDE_WALL) & (i > g.Nx)) | ((g.ylo == SIDE_WALL) & (j < 1)) | ((g.yhi == SIDE_WALL) & (j > g.Ny));
}
__device__ __forceinline__ bool inactive_cell(const GridDev& g, int i, int j) {
    bool out = inactive_cell_underlying(g, i, j);
    if (g.has_mask) {
        const bool beyond = (i < 1 - g.Hx) | (i > g.Nx + g.Hx) | (j < 1 - g.Hy) | (j > g.Ny + g.Hy);
        const int ic = min(max(i, 1 - g.Hx), g.Nx + g.Hx), jc = min(max(j, 1 - g.Hy), g.Ny + g.Hy);
        out |= beyond | (g.mask[ic + (long)jc * g.mask_ld] == 0);
    }
    return out;
}
__device__ __forceinline__ bool peripheral_u(const GridDev& g, int i, int j) {
    return inactive_cell(g, i, j) | inactive_cell(g, i - 1, j);
}
__device__ __forceinline__ bool peripheral_v(const GridDev& g, int i, int j) {
    return inactive_cell(g, i, j) | inactive_cell(g, i, j - 1);
}
__device__ __forceinline__ bool immersed_peripheral_cc(const GridDev& g, int i, int j) {
    if (!g.has_mask) return false;
    return inactive_cell(g, i, j) && !inactive_cell_underlying(g, i, j);
}
__device__ __forceinline__ bool immersed_peripheral_ff(const GridDev& g, int i, int j) {
    if (!g.has_mask) return false;
    bool p = inactive_cell(g, i, j) | inactive_cell(g, i - 1, j) | inactive_cell(g, i, j - 1) | inactive_cell(g, i - 1, j - 1);
    bool pu = inactive_cell_underlying(g, i, j) | inactive_cell_underlying(g, i - 1, j) |
              inactive_cell_underlying(g, i, j - 1) | inactive_cell_underlying(g, i - 1, j - 1);
    return p && !pu;
}

// ---- immersed_dj_sigma_1j / _2j with FluxBoundaryCondition numbers (ice_stress_divergence.jl:65-123): the stress on an
// immersed face is -flux (west, south) / +flux (east, north); conditional_flux_* picks it where the node is an immersed
// peripheral node; index_left / index_right: Face -> (i - 1, i), Center -> (i, i + 1).  Single-layer grid, dz = 1:
// Ax = dy, Ay = dx, V = Az.  Reference operation order (used by STRICT and FAST alike: the term is rare).
__device__ __forceinline__ double immersed_div_sigma_1(const EvpDev& P, int i, int j) {
    const GridDev& g = P.g;
    if (!g.has_mask) return 0.0;
    const double qtW = -P.ibc_u[0], qtE = P.ibc_u[1], qtS = -P.ibc_u[2], qtN = P.ibc_u[3];
    const int iW = i - 1, iE = i, jS = j, jN = j + 1;
    const double qW = (immersed_peripheral_cc(g, iW, j) ? qtW : 0.0) * dym(g, LOC_C, LOC_C, iW, j);
    const double qE = (immersed_peripheral_cc(g, iE, j) ? qtE : 0.0) * dym(g, LOC_C, LOC_C, iE, j);
    const double qS = (immersed_peripheral_ff(g, i, jS) ? qtS : 0.0) * dxm(g, LOC_F, LOC_F, i, jS);
    const double qN = (immersed_peripheral_ff(g, i, jN) ? qtN : 0.0) * dxm(g, LOC_F, LOC_F, i, jN);
    return (qE - qW + qN - qS) / azm(g, LOC_F, LOC_C, i, j);
}
__device__ __forceinline__ double immersed_div_sigma_2(const EvpDev& P, int i, int j) {
    const GridDev& g = P.g;
    if (!g.has_mask) return 0.0;
    const double qtW = -P.ibc_v[0], qtE = P.ibc_v[1], qtS = -P.ibc_v[2], qtN = P.ibc_v[3];
    const int iW = i, iE = i + 1, jS = j - 1, jN = j;
    const double qW = (immersed_peripheral_ff(g, iW, j) ? qtW : 0.0) * dym(g, LOC_F, LOC_F, iW, j);
    const double qE = (immersed_peripheral_ff(g, iE, j) ? qtE : 0.0) * dym(g, LOC_F, LOC_F, iE, j);
    const double qS = (immersed_peripheral_cc(g, i, jS) ? qtS : 0.0) * dxm(g, LOC_C, LOC_C, i, jS);
    const double qN = (immersed_peripheral_cc(g, i, jN) ? qtN : 0.0) * dxm(g, LOC_C, LOC_C, i, jN);
    return (qE - qW + qN - qS) / azm(g, LOC_C, LOC_F, i, j);
}

// ---- fused local halo fill: the thread that owns interior element (i, j) also writes the halo
// images that fill_halo_regions!(...; only_local_halos = true) would copy from it
// (split_explicit_momentum_equations.jl:170-187; upstream BC semantics).
// imgx / imgy: IMG_WRAP (Periodic), IMG_MIRROR (no-flux, Center location on a wall), IMG_NONE.
// A mirrored low side and a connected high side (tile edges) are encoded per side.
struct ImageSpec {
    int xlo, xhi, ylo, yhi;   // IMG_* per side
    int ex, ey;               // 1: the field has an extra column / row of points on a high wall (Face location): the
                              // wall faces have images in the OTHER direction like any interior point
    int fold_sign;            // yhi == IMG_FOLD (Zipper): +1 / -1 (velocity components)
    int fold_fx, fold_fy;     // ... and the field's location (1 = Face) in x / y
    double vxlo, vxhi, vylo, vyhi;   // IMG_VALUE sides (ValueBoundaryCondition on a tangential velocity): the first halo
                                     // cell holds 2 * value - c[first interior cell]; deeper halo cells are left alone
};
__device__ __forceinline__ int image_lo(int mode, int i, int N, int H, bool& has) {
    // halo index on the LOW side that copies from interior index i
    if (mode == IMG_WRAP) { has = (i > N - H); return i - N; }
    if (mode == IMG_MIRROR) { has = (i <= H); return 1 - i; }
    if (mode == IMG_VALUE) { has = (i == 1); return 0; }
    has = false; return 0;
}
__device__ __forceinline__ int image_hi(int mode, int i, int N, int H, bool& has) {
    if (mode == IMG_WRAP) { has = (i <= H); return i + N; }
    if (mode == IMG_MIRROR) { has = (i > N - H); return 2 * N + 1 - i; }
    if (mode == IMG_VALUE) { has = (i == N); return N + 1; }
    has = false; return 0;
}
__device__ __forceinline__ void store_with_images(const FRef& f, const GridDev& g, const ImageSpec& im, int i, int j, double val) {
    f(i, j) = val;
    // Elements within H of an edge have halo images; the test is cheap and almost always false.
    // x-images exist for interior columns (any row, including the ring rows a tile recomputes for
    // its neighbours), y-images for interior rows; corners are the product of the two.
    const bool in_x = (i >= 1) & (i <= g.Nx), in_y = (j >= 1) & (j <= g.Ny);
    const bool near_x = in_x & ((i <= g.Hx) | (i > g.Nx - g.Hx));
    const bool fold = im.yhi == IMG_FOLD;
    const bool near_y = in_y & ((j <= g.Hy) | (j > g.Ny - g.Hy - (fold ? 1 : 0)));
    if (!(near_x | near_y)) return;
    if (fold & in_x & in_y) {
        // North fold (Zipper; recalled upstream semantics, oracle/csi_oracle.c fold_north): the owner of (i, j) writes
        //   c[it, 2 Ny - j] (Center in y, rows Ny-Hy .. Ny-1) / c[it, 2 Ny + 1 - j] (Face in y, rows Ny-Hy+1 .. Ny)
        // with it = Nx - i + 1 (Center in x) / Nx - i + 2 (Face in x; column 1 folds onto itself without the sign change),
        // and the periodic x images of that target column.
        const int jt = im.fold_fy ? 2 * g.Ny + 1 - j : 2 * g.Ny - j;
        if ((jt > g.Ny) & (jt <= g.Ny + g.Hy)) {
            int it = im.fold_fx ? g.Nx - i + 2 : g.Nx - i + 1;
            double w = (double)im.fold_sign * val;
            if (it > g.Nx) { it -= g.Nx; w = fabs((double)im.fold_sign) * val; }
            f(it, jt) = w;
            if (it <= g.Hx) f(it + g.Nx, jt) = w;
            if (it > g.Nx - g.Hx) f(it - g.Nx, jt) = w;
        }
    }
    int xi[3], yj[3];
    double cx[3], cy[3];          // image value = c - val (c = 2 * bc value on an IMG_VALUE side), or val itself (c = NaN marker unused)
    bool fx[3], fy[3];            // the image is a ValueBoundaryCondition reflection
    int nx = 0, ny = 0;
    xi[nx] = i; fx[nx] = false; cx[nx++] = 0.0;
    yj[ny] = j; fy[ny] = false; cy[ny++] = 0.0;
    bool has;
    int t;
    if (in_x) {
        t = image_lo(im.xlo, i, g.Nx, g.Hx, has); if (has) { xi[nx] = t; fx[nx] = im.xlo == IMG_VALUE; cx[nx++] = 2 * im.vxlo; }
        t = image_hi(im.xhi, i, g.Nx, g.Hx, has); if (has) { xi[nx] = t; fx[nx] = im.xhi == IMG_VALUE; cx[nx++] = 2 * im.vxhi; }
    }
    if (in_y) {
        t = image_lo(im.ylo, j, g.Ny, g.Hy, has); if (has) { yj[ny] = t; fy[ny] = im.ylo == IMG_VALUE; cy[ny++] = 2 * im.vylo; }
        t = image_hi(im.yhi, j, g.Ny, g.Hy, has); if (has) { yj[ny] = t; fy[ny] = im.yhi == IMG_VALUE; cy[ny++] = 2 * im.vyhi; }
    }
    for (int b = 0; b < ny; ++b)
        for (int a = 0; a < nx; ++a)
            if (a | b) {
                double w = val;
                if (fx[a]) w = cx[a] - w;
                if (fy[b]) w = cy[b] - w;
                f(xi[a], yj[b]) = w;
            }
}

}  // namespace csi
