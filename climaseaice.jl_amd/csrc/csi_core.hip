// csi_core.hip -- context helpers, index ranges, launch geometries of the fused kernels  (split out of csi_abi.hip in round 4; see csi_ctx.h)
#include "csi_ctx.h"

namespace csi_host {

std::string g_create_error;


int32_t fail(csi_context* c, int32_t code, const std::string& msg) {
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}

int side_lo(int topo) {
    switch (topo) {
        case CSI_PERIODIC: return SIDE_PERIODIC;
        case CSI_BOUNDED: return SIDE_WALL;
        case CSI_FULLY_CONNECTED: return SIDE_CONNECTED;
        case CSI_LEFT_CONNECTED: return SIDE_CONNECTED;
        case CSI_LEFT_CONNECTED_RIGHT_FOLDED: return SIDE_CONNECTED;
        default: return SIDE_WALL;   // RIGHT_CONNECTED, RIGHT_FOLDED: low side is the wall
    }
}
int side_hi(int topo) {
    switch (topo) {
        case CSI_PERIODIC: return SIDE_PERIODIC;
        case CSI_BOUNDED: return SIDE_WALL;
        case CSI_FULLY_CONNECTED: return SIDE_CONNECTED;
        case CSI_LEFT_CONNECTED: return SIDE_WALL;
        case CSI_RIGHT_FOLDED: return SIDE_FOLD;
        case CSI_LEFT_CONNECTED_RIGHT_FOLDED: return SIDE_FOLD;
        default: return SIDE_CONNECTED;
    }
}
int img_of(int side, int loc) {
    if (side == SIDE_PERIODIC) return IMG_WRAP;
    if (side == SIDE_WALL) return loc == LOC_C ? IMG_MIRROR : IMG_NONE;
    if (side == SIDE_FOLD) return IMG_FOLD;
    return IMG_NONE;
}
ImageSpec image_spec(const csi_context* c, int fid) {
    ImageSpec im;
    im.xlo = img_of(c->g.xlo, kLoc[fid][0]);
    im.xhi = img_of(c->g.xhi, kLoc[fid][0]);
    im.ylo = img_of(c->g.ylo, kLoc[fid][1]);
    im.yhi = img_of(c->g.yhi, kLoc[fid][1]);
    im.ex = (kLoc[fid][0] == LOC_F && c->g.xhi == SIDE_WALL) ? 1 : 0;
    im.ey = (kLoc[fid][1] == LOC_F && c->g.yhi == SIDE_WALL) ? 1 : 0;
    im.vxlo = im.vxhi = im.vylo = im.vyhi = 0.0;
    // Zipper (north fold): vector components change sign (sea_ice_model.jl:57-64 for u, v; the stress / ocean-velocity /
    // forcing arrays at the velocity points are built with the same boundary conditions, test/distributed_tests_utils.jl:196-197)
    im.fold_fx = kLoc[fid][0] == LOC_F; im.fold_fy = kLoc[fid][1] == LOC_F;
    im.fold_sign = (kLoc[fid][0] != kLoc[fid][1]) ? -1 : 1;       // (f,c) and (c,f) fields are all velocity-like here
    // ValueBoundaryCondition on the tangential velocity at a wall replaces the no-flux mirror (one halo cell)
    if (fid == CSI_F_U) {
        if (im.ylo == IMG_MIRROR && c->vel_bc_on[0][0]) { im.ylo = IMG_VALUE; im.vylo = c->vel_bc_value[0][0]; }
        if (im.yhi == IMG_MIRROR && c->vel_bc_on[0][1]) { im.yhi = IMG_VALUE; im.vyhi = c->vel_bc_value[0][1]; }
    } else if (fid == CSI_F_V) {
        if (im.xlo == IMG_MIRROR && c->vel_bc_on[1][0]) { im.xlo = IMG_VALUE; im.vxlo = c->vel_bc_value[1][0]; }
        if (im.xhi == IMG_MIRROR && c->vel_bc_on[1][1]) { im.xhi = IMG_VALUE; im.vxhi = c->vel_bc_value[1][1]; }
    }
    return im;
}
// a Face-located field has one extra point where the HIGH side of that direction is a wall
int extra_x(const csi_context* c, int fid) { return (kLoc[fid][0] == LOC_F && c->g.xhi == SIDE_WALL) ? 1 : 0; }
int extra_y(const csi_context* c, int fid) { return (kLoc[fid][1] == LOC_F && c->g.yhi == SIDE_WALL) ? 1 : 0; }

FRef ref_of(const csi_context* c, int fid) {
    FRef r;
    const Bound& b = c->f[fid];
    r.p = b.p ? b.p + (c->Hx - 1) + (int64_t)(c->Hy - 1) * b.ld : nullptr;
    r.ld = (int)b.ld;
    return r;
}
int32_t need(csi_context* c, std::initializer_list<int> ids) {
    if (!c->grid_set) return fail(c, CSI_ERR_NOT_BOUND, "csi_grid_set has not been called");
    for (int id : ids)
        if (!c->f[id].p) return fail(c, CSI_ERR_NOT_BOUND, std::string("field not bound: ") + kName[id]);
    return CSI_OK;
}

StressDev stress_dev(const csi_context* c, int side) {
    const csi_stress& s = c->stress[side];
    StressDev d{};
    d.kind = s.kind; d.ue_kind = s.ue_kind; d.ve_kind = s.ve_kind;
    d.tau_u = s.tau_u; d.tau_v = s.tau_v; d.ue = s.ue; d.ve = s.ve; d.rho_e = s.rho_e; d.Cd = s.Cd;
    d.fu = ref_of(c, side == CSI_STRESS_TOP ? CSI_F_TOP_U : CSI_F_BOT_U);
    d.fv = ref_of(c, side == CSI_STRESS_TOP ? CSI_F_TOP_V : CSI_F_BOT_V);
    return d;
}
int32_t check_stress_fields(csi_context* c, int side) {
    const csi_stress& s = c->stress[side];
    int fu = side == CSI_STRESS_TOP ? CSI_F_TOP_U : CSI_F_BOT_U, fv = side == CSI_STRESS_TOP ? CSI_F_TOP_V : CSI_F_BOT_V;
    bool need_u = s.kind == CSI_STRESS_FIELD || (s.kind == CSI_STRESS_SEMI_IMPLICIT && s.ue_kind == CSI_VEL_FIELD);
    bool need_v = s.kind == CSI_STRESS_FIELD || (s.kind == CSI_STRESS_SEMI_IMPLICIT && s.ve_kind == CSI_VEL_FIELD);
    if (need_u && !c->f[fu].p) return fail(c, CSI_ERR_NOT_BOUND, std::string("stress field not bound: ") + kName[fu]);
    if (need_v && !c->f[fv].p) return fail(c, CSI_ERR_NOT_BOUND, std::string("stress field not bound: ") + kName[fv]);
    return CSI_OK;
}

// Coriolis parameter of the FAST kernels: two columns of the per-row coefficient table (uniform metrics + FPlane:
// two of the table's constants).  Rebuilt when the FPlane value, the BetaPlane rows or the grid changed; a
// BetaPlane on uniform metrics switches the kernels to their per-row-coefficient instantiation.
int32_t sync_coriolis(csi_context* c) {
    const csi_evp_params& e = c->evp;
    const double f0 = e.has_coriolis ? e.coriolis_f : 0.0;
    if (!c->cor_dirty && f0 == c->cor_synced) return CSI_OK;
    if (c->metric_kind == CSI_METRIC_FULL) { c->cor_dirty = false; c->cor_synced = f0; return CSI_OK; }   // no FAST table
    const bool rows = e.has_coriolis && !c->fcor_rows[0].empty();
    const bool metrics_uniform = c->metric_kind == CSI_METRIC_UNIFORM;
    c->coef.uni[FC_FU] = f0; c->coef.uni[FC_FV] = f0;
    c->coef.uniform = metrics_uniform && !rows;
    if (!c->coef.uniform) {
        const int n = c->Ny + 2 * c->Hy + 1;
        // device layout: ROW-major, the FC_COUNT coefficients of one row contiguous (a kernel reads a row's
        // coefficients with a few wide scalar loads from one base address)
        // (rows n .. 2n - 1: the pair kernel's copy, scaled by exact powers of two: pair_coef_scale)
        std::vector<double> host((size_t)FC_COUNT * n * 2);
        for (int w = 0; w < FC_COUNT; ++w)
            for (int t = 0; t < n; ++t)
                host[(size_t)t * FC_COUNT + w] = metrics_uniform ? c->coef.uni[w] : c->coef_host[(size_t)w * n + t];
        for (int t = 0; t < n; ++t) {
            host[(size_t)t * FC_COUNT + FC_FU] = rows ? c->fcor_rows[0][t] : f0;
            host[(size_t)t * FC_COUNT + FC_FV] = rows ? c->fcor_rows[1][t] : f0;
        }
        for (int t = 0; t < n; ++t)
            for (int w = 0; w < FC_COUNT; ++w)
                host[(size_t)(n + t) * FC_COUNT + w] = pair_coef_scale(w) * host[(size_t)t * FC_COUNT + w];
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (!c->dev_coef) HIP_TRY(c, hipMalloc((void**)&c->dev_coef, sizeof(double) * host.size()));
        HIP_TRY(c, hipMemcpy(c->dev_coef, host.data(), sizeof(double) * host.size(), hipMemcpyHostToDevice));
        c->coef.vec = c->dev_coef + (size_t)(c->Hy - 1) * FC_COUNT;      // so that vec[j * stride + which] is row j
        c->coef.vec_pair = c->coef.vec + (size_t)n * FC_COUNT;
        c->coef.stride = FC_COUNT;
        c->coef.jmin = 1 - c->Hy;
        c->coef.jmax = c->Ny + c->Hy + 1;
    }
    c->cor_dirty = false;
    c->cor_synced = f0;
    return CSI_OK;
}

EvpDev evp_dev(const csi_context* c, double dt) {
    EvpDev P{};
    P.g = c->g;
    P.u = ref_of(c, CSI_F_U); P.v = ref_of(c, CSI_F_V); P.h = ref_of(c, CSI_F_H); P.a = ref_of(c, CSI_F_A);
    P.s11 = ref_of(c, CSI_F_S11); P.s22 = ref_of(c, CSI_F_S22); P.s12 = ref_of(c, CSI_F_S12);
    P.zc = ref_of(c, CSI_F_ZETA_C); P.zf = ref_of(c, CSI_F_ZETA_F); P.Dl = ref_of(c, CSI_F_DELTA);
    P.al = ref_of(c, CSI_F_ALPHA); P.P = ref_of(c, CSI_F_P); P.un = ref_of(c, CSI_F_UN); P.vn = ref_of(c, CSI_F_VN);
    P.top = stress_dev(c, CSI_STRESS_TOP);
    P.bot = stress_dev(c, CSI_STRESS_BOTTOM);
    const csi_evp_params& e = c->evp;
    P.P_star = e.ice_compressive_strength; P.C_star = e.ice_compaction_hardening; P.ecc = e.yield_curve_eccentricity;
    P.Dmin = e.minimum_plastic_stress; P.amin = e.min_relaxation_parameter; P.amax = e.max_relaxation_parameter;
    P.ca = e.relaxation_strength; P.min_mass = e.minimum_mass; P.min_conc = e.minimum_concentration;
    P.rho = e.sea_ice_density; P.fcor = e.has_coriolis ? e.coriolis_f : 0.0; P.has_cor = e.has_coriolis;   // FAST kernels multiply by fcor unconditionally
    if (c->dev_fcor && e.has_coriolis) {
        const size_t n = (size_t)c->Ny + 2 * (size_t)c->Hy + 1;
        P.fcor_u = c->dev_fcor + (c->Hy - 1); P.fcor_v = c->dev_fcor + n + (c->Hy - 1);   // ptr[j] is row j
    }
    if (c->dev_fcor2 && e.has_coriolis) {
        const long off = (c->Hx - 1) + (long)(c->Hy - 1) * c->fcor2_ld;
        P.fcor2_u = c->dev_fcor2 + off; P.fcor2_v = c->dev_fcor2 + c->fcor2_plane + off; P.fcor2_ld = c->fcor2_ld;
    }
    P.pressure_kind = e.pressure_formulation;
    P.dt = dt;
    P.write_diag = 0;
    P.has_forcing = (c->f[CSI_F_FORCING_U].p && c->f[CSI_F_FORCING_V].p) ? 1 : 0;
    if (P.has_forcing) { P.forcing_u = ref_of(c, CSI_F_FORCING_U); P.forcing_v = ref_of(c, CSI_F_FORCING_V); }
    bool any_ibc = false;
    for (int k = 0; k < 4; ++k) { P.ibc_u[k] = c->ibc[0][k]; P.ibc_v[k] = c->ibc[1][k]; any_ibc |= (c->ibc[0][k] != 0.0) | (c->ibc[1][k] != 0.0); }
    P.extra = (P.has_forcing || (any_ibc && c->g.has_mask)) ? 1 : 0;
    P.free_drift = c->free_drift;
    if (c->free_drift && c->fd[0] && c->fd[1]) {
        P.ufd.p = c->fd[0] + (c->Hx - 1) + (int64_t)(c->Hy - 1) * c->f[CSI_F_U].ld; P.ufd.ld = (int)c->f[CSI_F_U].ld;
        P.vfd.p = c->fd[1] + (c->Hx - 1) + (int64_t)(c->Hy - 1) * c->f[CSI_F_V].ld; P.vfd.ld = (int)c->f[CSI_F_V].ld;
    }
    return P;
}

// Index ranges (1-based, inclusive).  Stress kernels: Auxiliaries kernel parameters
// -H+2 : N+H-1 (elasto_visco_plastic_rheology.jl:145); velocity kernels: :xy on a serial grid,
// split_explicit_kernel_size on connected (tile) sides (split_explicit_momentum_equations.jl:40-46).
// Connected (tile) sides.  Between two halo exchanges only V layers of u, v beyond the owned cells are
// valid (V = exchange width at the start of a batch, shrinking by 2 per sub-step):
//   stress                       [2-V, N+V-1]
//   first velocity  (u first)    x [3-V, N+V-1], y [2-V, N+V-2]      (v first: x and y swapped)
//   second velocity              [3-V, N+V-2]
// recomputed redundantly on the ring so that sigma, alpha never need exchanging inside the sub-cycle.
// V = 2 (exchange every sub-step) gives stress [0, N+1], first velocity [1, N+1] x [0, N], second [1, N].
// Sides with a local boundary condition keep the reference's ranges (-H+2 : N+H-1 and 1 : N).
Range stress_range(const csi_context* c, int V) {
    const GridDev& g = c->g;
    return Range{g.xlo == SIDE_CONNECTED ? 2 - V : -c->Hx + 2, g.xhi == SIDE_CONNECTED ? c->Nx + V - 1 : c->Nx + c->Hx - 1,
                 g.ylo == SIDE_CONNECTED ? 2 - V : -c->Hy + 2, g.yhi == SIDE_CONNECTED ? c->Ny + V - 1 : c->Ny + c->Hy - 1};
}
Range first_u_range(const csi_context* c, int V) {
    const GridDev& g = c->g;
    return Range{g.xlo == SIDE_CONNECTED ? 3 - V : 1, g.xhi == SIDE_CONNECTED ? c->Nx + V - 1 : c->Nx,
                 g.ylo == SIDE_CONNECTED ? 2 - V : 1, g.yhi == SIDE_CONNECTED ? c->Ny + V - 2 : c->Ny};
}
Range first_v_range(const csi_context* c, int V) {
    const GridDev& g = c->g;
    return Range{g.xlo == SIDE_CONNECTED ? 2 - V : 1, g.xhi == SIDE_CONNECTED ? c->Nx + V - 2 : c->Nx,
                 g.ylo == SIDE_CONNECTED ? 3 - V : 1, g.yhi == SIDE_CONNECTED ? c->Ny + V - 1 : c->Ny};
}
Range second_range(const csi_context* c, int V) {
    const GridDev& g = c->g;
    return Range{g.xlo == SIDE_CONNECTED ? 3 - V : 1, g.xhi == SIDE_CONNECTED ? c->Nx + V - 2 : c->Nx,
                 g.ylo == SIDE_CONNECTED ? 3 - V : 1, g.yhi == SIDE_CONNECTED ? c->Ny + V - 2 : c->Ny};
}
bool is_tiled(const csi_context* c) {
    const GridDev& g = c->g;
    return g.xlo == SIDE_CONNECTED || g.xhi == SIDE_CONNECTED || g.ylo == SIDE_CONNECTED || g.yhi == SIDE_CONNECTED;
}
Range interior_range(const csi_context* c) { return Range{1, c->Nx, 1, c->Ny}; }
Range parent_range(const csi_context* c) { return Range{1 - c->Hx, c->Nx + c->Hx, 1 - c->Hy, c->Ny + c->Hy}; }


int32_t fill_halo(csi_context* c, int fid) {
    launch_fill_halo(ref_of(c, fid), c->g, image_spec(c, fid), c->stream);
    HIP_TRY(c, hipGetLastError());
    return CSI_OK;
}

int32_t copy_parent(csi_context* c, int dst, int src) {
    const Bound &d = c->f[dst], &s = c->f[src];
    if (d.ld != s.ld || d.nj != s.nj) return fail(c, CSI_ERR_INVALID_ARGUMENT, std::string("parent shape mismatch: ") + kName[dst] + " vs " + kName[src]);
    HIP_TRY(c, hipMemcpyAsync(d.p, s.p, sizeof(double) * (size_t)d.ld * (size_t)d.nj, hipMemcpyDeviceToDevice, c->stream));
    return CSI_OK;
}

int32_t do_initialize(csi_context* c) {
    EvpDev P = evp_dev(c, 0.0);
    if (c->mode == CSI_MODE_FAST) launch_fast_init(P, parent_range(c), c->stream);
    else launch_strict_init(P, parent_range(c), c->stream);
    HIP_TRY(c, hipGetLastError());
    return CSI_OK;
}


// ---- fused sub-step path (evp_fused.hip) -----------------------------------------------------------------

FRef alt_ref(const csi_context* c, int k) {
    FRef r;
    const Bound& b = c->f[kPing[k]];
    r.p = c->alt[k] + (c->Hx - 1) + (int64_t)(c->Hy - 1) * b.ld;
    r.ld = (int)b.ld;
    return r;
}


FusedGeom fused_geom(const csi_context* c, int V) {
    FusedGeom G;
    G.rs = stress_range(c, V);
    // decomposition of the stress range into (60-column strip) x (rows) wave tiles.  Measured (round 2, MI355X): 2048^2
    // rows 12 -> 32.2, 24 -> 28.8, 48 -> 20.7 G cell-updates/s; 1024 x 512: rows 2 -> 21.0, 4 -> 19.0, 12 -> 12.1: many short
    // tiles beat few tall ones (the kernel is bandwidth-bound, its 3 ring rows are re-read from L2)
    const int width = G.rs.i1 - G.rs.i0 + 1, height = G.rs.j1 - G.rs.j0 + 1;
    G.nstrips = (width + 59) / 60;
    long strip_rows = (long)G.nstrips * height;
    int rows = (int)(strip_rows / 6000);
    if (rows < 3) rows = 3;
    if (rows > 12) rows = 12;
    if (c->tune.fused_rows >= 0) rows = c->tune.fused_rows;         // tuning aid (CSI_FUSED_ROWS)
    if (rows > height) rows = height;
    if (rows < 1) rows = 1;
    G.rows = rows;
    G.nchunks = (height + rows - 1) / rows;
    return G;
}

void velocity_ranges(const csi_context* c, bool ufirst, int V, Range& r1, Range& r1c, Range& r2) {
    const GridDev& g = c->g;
    r1 = ufirst ? first_u_range(c, V) : first_v_range(c, V);
    r2 = second_range(c, V);
    r1c = r1;
    // a periodic side keeps halo images of the first velocity; the second velocity next to that edge reads
    // them, so the ring recomputation there extends one cell into the halo (inputs in the halo are images too)
    if (ufirst) {
        if (g.ylo == SIDE_PERIODIC) r1c.j0 -= 1;
        if (g.xhi == SIDE_PERIODIC) r1c.i1 += 1;
    } else {
        if (g.xlo == SIDE_PERIODIC) r1c.i0 -= 1;
        if (g.yhi == SIDE_PERIODIC) r1c.j1 += 1;
    }
}

int32_t ensure_alt(csi_context* c) {
    for (int k = 0; k < 5; ++k) {
        const Bound& b = c->f[kPing[k]];
        const size_t n = (size_t)b.ld * (size_t)b.nj;
        if (c->alt_elems[k] != n) {
            if (c->alt[k]) { HIP_TRY(c, hipStreamSynchronize(c->stream)); hipFree(c->alt[k]); c->alt[k] = nullptr; }
            HIP_TRY(c, hipMalloc((void**)&c->alt[k], n * sizeof(double)));
            c->alt_elems[k] = n;
        }
    }
    return CSI_OK;
}

int exchange_interval(const csi_context* c) {
    if (!is_tiled(c)) return 1;
    const int hmin = c->Hx < c->Hy ? c->Hx : c->Hy, nmin = c->Nx < c->Ny ? c->Nx : c->Ny;
    int k = c->exch_k > 0 ? c->exch_k : (hmin / 2 < 16 ? hmin / 2 : 16);   // automatic: as rare as the halo allows (<= 16)
    if (k > kMaxExchangeInterval) k = kMaxExchangeInterval;                 // the fused path's table has that many batch positions
    while (k > 1 && (2 * k > hmin || 2 * k > nmin)) --k;
    return k < 1 ? 1 : k;
}

// ---- two sub-steps per launch (evp_fused2.hip) -----------------------------------------------------------
// Valid halo width per side at the start of a sub-step: connected sides follow the exchange batch (W - 2m),
// periodic and wall sides are refreshed by the owner's halo images after every pair (4 for the first sub-step of a
// pair, 2 for the second; beyond a wall the "valid" cells are mirror images or never-written zeros, exactly what
// the reference's kernels read there).
SideV pair_side_v(const csi_context* c, int v_connected, int v_periodic) {
    const GridDev& g = c->g;
    auto v = [&](int side) { return side == SIDE_CONNECTED ? v_connected : v_periodic; };
    return SideV{v(g.xlo), v(g.xhi), v(g.ylo), v(g.yhi)};
}
Range v_stress_range(const csi_context* c, const SideV& v) { return Range{2 - v.xlo, c->Nx + v.xhi - 1, 2 - v.ylo, c->Ny + v.yhi - 1}; }
Range v_first_range(const csi_context* c, const SideV& v, bool ufirst) {
    return ufirst ? Range{3 - v.xlo, c->Nx + v.xhi - 1, 2 - v.ylo, c->Ny + v.yhi - 2}
                  : Range{2 - v.xlo, c->Nx + v.xhi - 2, 3 - v.ylo, c->Ny + v.yhi - 1};
}
Range v_second_range(const csi_context* c, const SideV& v) { return Range{3 - v.xlo, c->Nx + v.xhi - 2, 3 - v.ylo, c->Ny + v.yhi - 2}; }
// periodic sides: the owner stores interior cells only, the halo copies are written as images of that store;
// wall sides: velocities on 1 : N (split_explicit_momentum_equations.jl:40-46, the wall face N + 1 is never
// written), stresses on 1 : N + 1 (sigma12 lives on the wall corners)
Range clip_store(const csi_context* c, Range r, bool sigma) {
    const GridDev& g = c->g;
    const int ex = sigma ? 1 : 0;
    if (g.xlo != SIDE_CONNECTED && r.i0 < 1) r.i0 = 1;
    if (g.xhi == SIDE_PERIODIC && r.i1 > c->Nx) r.i1 = c->Nx;
    if (g.xhi == SIDE_WALL && r.i1 > c->Nx + ex) r.i1 = c->Nx + ex;
    if (g.ylo != SIDE_CONNECTED && r.j0 < 1) r.j0 = 1;
    if (g.yhi == SIDE_PERIODIC && r.j1 > c->Ny) r.j1 = c->Ny;
    if (g.yhi == SIDE_WALL && r.j1 > c->Ny + ex) r.j1 = c->Ny + ex;
    return r;
}
bool has_walls(const csi_context* c) {
    const GridDev& g = c->g;
    return g.xlo == SIDE_WALL || g.xhi == SIDE_WALL || g.ylo == SIDE_WALL || g.yhi == SIDE_WALL;
}
// The fused kernels address every field with 32-bit unsigned BYTE offsets from the parent's first element (one SGPR
// base + one VGPR offset per access): a parent of 4 GiB or more (about 23k x 23k cells; it fits the 288 GB of HBM) would
// wrap silently, so such grids run the three-kernel path, whose FRef indexes with 64-bit integers.
bool offsets_fit_32bit(int Nx, int Ny, int Hx, int Hy, int64_t max_ld) {
    const int64_t ld = max_ld > 0 ? max_ld : (int64_t)Nx + 2 * Hx + 1, nj = (int64_t)Ny + 2 * Hy + 1;
    return ld * nj * 8 < ((int64_t)1 << 32);
}
int64_t max_bound_ld(const csi_context* c) {
    int64_t m = 0;
    for (int k = 0; k < CSI_F_COUNT; ++k) if (c->f[k].p && c->f[k].ld > m) m = c->f[k].ld;
    if (c->g.has_mask && c->g.mask_ld > m) m = c->g.mask_ld;
    return m;
}
bool pair_supported(const csi_context* c) {
    if (!c->pairing) return false;
    if (!offsets_fit_32bit(c->Nx, c->Ny, c->Hx, c->Hy, max_bound_ld(c))) return false;
    const GridDev& g = c->g;
    // per-point coefficients (CSI_METRIC_FULL): the pair kernel streams the 14 metric planes; a periodic y side
    // would need the ring rows beyond the seam to see their owners' coefficients -- the planes' halo entries are images
    // of the interior (csi.h), so that holds; a north fold does not pair
    if (c->metric_kind == CSI_METRIC_FULL && !c->dev_coef2) return false;
    auto ok = [](int s) { return s == SIDE_PERIODIC || s == SIDE_CONNECTED || s == SIDE_WALL; };
    // per-row metrics with a periodic y side: the ring rows recomputed beyond the seam would use other metrics than
    // their owners (an unphysical grid anyway) -- three kernels
    if (c->metric_kind == CSI_METRIC_PER_J && (g.ylo == SIDE_PERIODIC || g.yhi == SIDE_PERIODIC)) return false;   // (BetaPlane rows wrap: csi.h)
    return ok(g.xlo) && ok(g.xhi) && ok(g.ylo) && ok(g.yhi) && c->Hx >= 4 && c->Hy >= 4 && c->Nx >= 2 * c->Hx && c->Ny >= 2 * c->Hy;
}
// CSI_METRIC_FULL: which rows of the coefficient planes hold ONE value per row (a tripolar grid south of its cap, any grid handed over
// as twelve arrays that is a latitude-longitude or rectilinear grid in disguise).  All Nx + 2Hx + 1 columns of all twelve planes --
// and of the per-point Coriolis planes, where set -- are compared bit for bit (rc_rtol > 0: within that relative distance of the first
// interior column, whose value then stands for the row: the caller's choice, changes results at that level).  The pair kernel reads
// the marked rows' values from per-row vectors (evp_fused2.hip `rcd`).
int32_t ensure_row_constant(csi_context* c) {
    if (!c->rc_dirty) return CSI_OK;
    c->rc_dirty = false;
    c->rc_rows = 0;
    c->coef.c2row = nullptr; c->coef.rcsum = nullptr; c->coef.c2row_n = 0;
    if (c->metric_kind != CSI_METRIC_FULL || !c->rc_enabled || c->coef2_host.empty()) return CSI_OK;
    const long ni = c->Nx + 2 * c->Hx + 1, nj = c->Ny + 2 * c->Hy + 1;
    const bool hasf = !c->fcor2_host.empty();
    const int nv = C2_COUNT + 2;
    std::vector<double> rowv((size_t)nv * nj, 0.0);
    std::vector<int> sum((size_t)nj + 1, 0);
    const long ref = c->Hx;                               // parent column of i = 1
    for (long t = 0; t < nj; ++t) {
        bool same = true;
        for (int k = 0; k < nv && same; ++k) {
            const double* row;
            if (k < C2_COUNT) row = c->coef2_host.data() + ((size_t)k * nj + t) * ni;
            else if (hasf) row = c->fcor2_host.data() + ((size_t)(k - C2_COUNT) * nj + t) * ni;
            else continue;
            const double v = row[ref];
            rowv[(size_t)k * nj + t] = v;
            if (c->rc_rtol > 0.0) {
                const double tol = c->rc_rtol * std::fabs(v);
                for (long a = 0; a < ni && same; ++a) same = std::fabs(row[a] - v) <= tol;
            } else {
                for (long a = 0; a < ni && same; ++a) same = std::memcmp(&row[a], &v, sizeof(double)) == 0;
            }
        }
        sum[t + 1] = sum[t] + (same ? 1 : 0);
    }
    c->rc_rows = sum[nj];
    if (c->rc_rows == 0) return CSI_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->dev_c2row) { hipFree(c->dev_c2row); c->dev_c2row = nullptr; }
    if (c->dev_rcsum) { hipFree(c->dev_rcsum); c->dev_rcsum = nullptr; }
    HIP_TRY(c, hipMalloc((void**)&c->dev_c2row, sizeof(double) * rowv.size()));
    HIP_TRY(c, hipMalloc((void**)&c->dev_rcsum, sizeof(int) * sum.size()));
    HIP_TRY(c, hipMemcpy(c->dev_c2row, rowv.data(), sizeof(double) * rowv.size(), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->dev_rcsum, sum.data(), sizeof(int) * sum.size(), hipMemcpyHostToDevice));
    c->coef.c2row = c->dev_c2row; c->coef.c2row_n = nj; c->coef.rcsum = c->dev_rcsum;
    return CSI_OK;
}

constexpr long kWriteThroughCells = 1152L * 1024L;      // (scripts/wt_sweep.py, profiles/r04_wt_sweep.txt: 512^2 +4 %, 1024 x 512 +3.4 %, 2048 x 256 +3.7 %, 2048 x 512 +1.8 %, 1024^2 +0.6 %; 1536^2 -1 %, 2048^2 -9 %)
// tile_scale > 1: so many times the tiles of the one-round geometry (run_fused: launches of the LIVE tiles only -- the fraction of
// live tiles times the number of tiles is what has to fit one round)
FusedGeom pair_geom(const csi_context* c, const Range& dec, double tile_scale) {
    FusedGeom G;
    G.rs = dec;
    // (56-column strip) x (rows) tiles, one workgroup of two waves (producer: first sub-step, consumer: second) per tile.
    // Exactly ONE round of tiles, as tall as possible: every SIMD keeps its waves from start to end and each tile pays its ring
    // rows once.  How many tiles a round holds follows the family's register budget (evp_fused2.hip, k_pair's launch bounds):
    // 3 waves per SIMD (<= 168 VGPRs) = 256 CUs x 6 workgroups = 1536; 2 waves per SIMD = 1024.
    const int width = dec.i1 - dec.i0 + 1, height = dec.j1 - dec.j0 + 1;
    G.nstrips = (width + 55) / 56;
    // (per-point coefficients: the kernel is compiled for 2 waves per SIMD -> 1024 resident tiles; measured at 2048^2:
    // 1024 tiles 22.9, 1536 tiles 18.9, 768 tiles 20.8 G cell-updates/s)
    // (array forcing on uniform / per-row metrics, round 4: the forcing values travel through the LDS ring, 36 KB per workgroup ->
    //  four workgroups per CU, two waves per SIMD, like the per-point-metric instantiations)
    const bool force_ring = c->metric_kind != CSI_METRIC_FULL && evp_ring_forcing(evp_dev(c, 0.0));
    int target = (c->metric_kind == CSI_METRIC_FULL || force_ring) ? 1024 : 1536;
    // Uniform coefficients (plain / walls / mask families on uniform metrics with an FPlane): two waves per SIMD with taller tiles
    // beat three at every size measured (round 5, profiles/r05_tile_count.txt: 1536^2 +4.5 %, 2048^2 +1.2 ... 3 %, 3072^2 +7.5 %,
    // 4096^2 +11 %, config 5's masked 4096^2 64.8 -> 71.8 G): a launch at 2048^2 is 83 iterations of 1.42 us instead of 57 of 2.09.
    // Per-row coefficients (lat-lon, BetaPlane: 65 vs 59 G at 2048^2) and the immersed flux conditions (55 vs 49) keep three:
    // their scalar loads per row want the third wave.
    {
        bool any_ibc = false;
        for (int k = 0; k < 4; ++k) any_ibc |= (c->ibc[0][k] != 0.0) | (c->ibc[1][k] != 0.0);
        if (target == 1536 && (c->coef.uniform != 0 || c->tune.row_target_1024 != 0) && !(any_ibc && c->g.has_mask)) target = 1024;
    }
    if (c->tune.pair_target >= 0) target = c->tune.pair_target;      // tuning aid (CSI_PAIR_TARGET: the rules below still apply)
    // Beside a fold band (its own stream: eight small launches per pair of sub-steps) the pair launch leaves a third of the wave
    // slots free, so that the band runs DURING the launch instead of in its tail -- a launch that fills every slot lets only the
    // band's first kernel in (round 3: 123 + 31 us per pair of sub-steps at 2048^2).  Measured at 2048^2, round 4: fold on uniform
    // metrics 1536 tiles 53.1, 1280 52.6, 1024 58.7, 896 56.1 G; tripolar-like (per-point metrics) 1024 tiles 20.4, 896 21.9, 768 20.5
    if (c->geom_band) target = c->metric_kind == CSI_METRIC_FULL ? 896 : 1024;
    if (tile_scale > 1.0) target = (int)std::lround(target * tile_scale);
    bool forced = false;
    if (c->tune.pair_tiles >= 0) { target = c->tune.pair_tiles; forced = true; }   // tuning aid (CSI_PAIR_TILES)
    int max_chunks = target / G.nstrips;
    if (max_chunks < 1) max_chunks = 1;
    int rows = (height + max_chunks - 1) / max_chunks;
    if (!forced && rows < 16) {
        // small grids (tiles of a multi-GPU decomposition): the 6 ring rows dominate short tiles; two waves per SIMD
        // (1024 tiles) with taller tiles beat three (measured: 1024 x 512 tile 35.3 vs 32.4, 1024 x 1024 47.0 vs 45.6 G cell-updates/s)
        max_chunks = 1024 / G.nstrips;
        if (max_chunks < 1) max_chunks = 1;
        rows = (height + max_chunks - 1) / max_chunks;
    }
    int min_rows = 6;                  // small grids: parallelism beats the 6 ring rows
    if (c->tune.pair_minrows >= 0) min_rows = c->tune.pair_minrows;
    if (rows < min_rows) rows = min_rows;
    if (c->tune.pair_rows >= 0) rows = c->tune.pair_rows;          // tuning aid (CSI_PAIR_ROWS)
    if (rows > height) rows = height;
    if (rows < 1) rows = 1;
    G.rows = rows;
    G.nchunks = (height + rows - 1) / rows;
    // write-through result stores on small grids (evp_fused2.hip, FI_WT): measured round 4 (profiles/r04_tile_1024x512.md)
    G.wt = c->tune.write_through >= 0 ? (c->tune.write_through != 0) : ((long)width * height <= kWriteThroughCells);
    // Peer transport: the chunk next to a connected y side waits for its neighbour's flags, stores the halo images (write-through,
    // into the neighbour's memory) and publishes -- about four row iterations' worth -- and a launch is as long as its slowest
    // tile: that chunk gets four rows less (not fewer than the halo: the next chunk must stay out of the side's tile set).
    // Measured on a 2048 x 256 tile connected to itself in y (round 4): profiles/r04_tile_1024x512.md.
    if (c->geom_peer && !forced && c->tune.pair_rows < 0 && c->tune.peer_edge != 0) {
        // (the NEXT chunk must stay out of the side's tile set: it may neither read halo rows nor own rows whose images go to the
        //  neighbour -- rows 1 .. max(Hy, 4) / the last max(Hy, 4) rows; dec starts at row 0 / ends at row Ny + 1)
        // The short chunks must not push the tile count over the resident set (one tile too many puts a third wave on some SIMDs
        // for the whole launch: 2048 x 1024 connected in y, 28 x 37 tiles: 52.6 G; 27 x 37: 68): taller rows until they fit.
        for (;; ++rows) {
            const int reach = std::max(c->Hy, 4), want = rows - (c->tune.peer_edge > 0 ? c->tune.peer_edge : 4);
            const int e_lo = std::max(reach + 1 - dec.j0, want), e_hi = std::max(reach + (dec.j1 - c->Ny), want);
            const bool lo = c->peer.sync_rank[2] >= 0 && e_lo < rows, hi = c->peer.sync_rank[3] >= 0 && e_hi < rows;
            G.rows = rows; G.elo = G.ehi = 0;
            G.nchunks = (height + rows - 1) / rows;
            if (height >= 4 * rows && (lo || hi)) {
                G.elo = lo ? e_lo : rows;
                G.ehi = hi ? e_hi : 0;
                const int mid = height - G.elo - G.ehi;
                G.nchunks = 1 + (mid + rows - 1) / rows + (hi ? 1 : 0);
                if (!hi) {
                    // (no short chunk at the high side: the formula's last chunk is simply what is left behind chunk 0)
                    G.nchunks = 1 + (height - G.elo + rows - 1) / rows;
                }
            }
            if (G.nchunks <= max_chunks || rows >= height) break;
        }
    }
    return G;
}


}  // namespace csi_host
