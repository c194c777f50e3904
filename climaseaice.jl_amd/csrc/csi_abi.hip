// csi_abi.hip -- the C ABI of libcsi_hip.so (include/csi.h): argument checks, binding, the entry points of the time steppers.
//
// The host code behind it (round 4: split out of this file, see csi_ctx.h) replaces the host side of the reference's
//   time_step_momentum!   SeaIceDynamics/split_explicit_momentum_equations.jl:103-195      csi_launch.hip
//   rk_substep! / cache_current_fields! / dynamic_time_step!   sea_ice_rk_substep.jl:29-152   csi_launch.hip, this file
//   time_step!(::FESeaIceModel)   sea_ice_fe_step.jl:13-34                                  this file
// All work is ordered on the context's stream; nothing here synchronises with the host
// except csi_sync / csi_context_destroy.
#include "csi_ctx.h"


static SlabDev slab_dev(const csi_slab_params* p) {
    SlabDev S{};
    S.k = p->conductivity; S.rho_bulk = p->sea_ice_density; S.rho_pure = p->density; S.rho_l = p->liquid_density;
    S.c_l = p->liquid_heat_capacity; S.c_i = p->heat_capacity; S.L0 = p->reference_latent_heat; S.T0 = p->reference_temperature;
    S.liq_slope = p->liquidus_slope; S.liq_T0 = p->freshwater_melting_temperature; S.S = p->bottom_salinity;
    S.hc = p->ice_consolidation_thickness; S.Tu = p->top_temperature; S.Qu = p->top_heat_flux; S.Qb = p->bottom_heat_flux;
    S.top_flux_kind = p->top_flux_kind; S.bot_flux_kind = p->bottom_flux_kind;
    S.top_bc_kind = p->top_bc_kind; S.ice_salinity = p->ice_salinity;
    return S;
}
static SnowDev snow_dev(const csi_snow_params* p) {
    SnowDev W{};
    W.k = p->conductivity; W.rho = p->snow_density; W.snowfall = p->snowfall; W.Tu = p->top_temperature; W.top_bc_kind = p->top_bc_kind;
    return W;
}
static int32_t do_layered(csi_context* c, const SlabDev& S, const SnowDev& W, double dt) {
    int32_t rc = need(c, {CSI_F_H, CSI_F_A, CSI_F_HS});
    if (rc) return rc;
    if (S.top_flux_kind != 0) return fail(c, CSI_ERR_UNSUPPORTED, "the layered (snow) step takes a numeric top heat flux (top_flux_kind 0)");
    LayeredOut o{};
    o.mf_ice = ref_of(c, CSI_F_MASS_FLUX); o.mf_snow = ref_of(c, CSI_F_MASS_FLUX_SNOW); o.mf_int = ref_of(c, CSI_F_SNOWFALL_INTERCEPTED);
    o.tu_ice = ref_of(c, CSI_F_TU); o.tu_snow = ref_of(c, CSI_F_TUS);
    launch_layered_step(S, W, c->g, ref_of(c, CSI_F_H), ref_of(c, CSI_F_A), ref_of(c, CSI_F_HS), o, dt, c->stream);
    HIP_TRY(c, hipGetLastError());
    return CSI_OK;
}
// thermodynamic_time_step!(model, ice_thermodynamics, snow_thermodynamics, dt): dispatch on the snow layer
static int32_t do_thermo(csi_context* c, double dt);
static int32_t do_slab(csi_context* c, const SlabDev& S, double dt) {
    const bool has_mf = c->f[CSI_F_MASS_FLUX].p != nullptr;
    if (S.top_bc_kind == 1 && S.top_flux_kind != 0)
        return fail(c, CSI_ERR_UNSUPPORTED, "MeltingConstrainedFluxBalance takes a numeric top heat flux (top_flux_kind 0)");
    launch_slab_step(S, c->g, ref_of(c, CSI_F_H), ref_of(c, CSI_F_A), ref_of(c, CSI_F_MASS_FLUX), has_mf, dt, c->stream);
    HIP_TRY(c, hipGetLastError());
    return CSI_OK;
}
static int32_t do_thermo(csi_context* c, double dt) {
    if (!c->slab_set) return CSI_OK;                       // thermodynamic_time_step!(model, ::Nothing, ...) = nothing
    return c->snow_set ? do_layered(c, c->slab, c->snow, dt) : do_slab(c, c->slab, dt);
}


// ============================================================================================
extern "C" {

int32_t csi_version(void) { return CSI_VERSION; }

const char* csi_last_error(const csi_context* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int32_t csi_context_create(int32_t device_id, void* hip_stream, csi_context** out) {
    if (!out) return fail(nullptr, CSI_ERR_INVALID_ARGUMENT, "out == NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return fail(nullptr, CSI_ERR_NO_DEVICE, "no HIP device visible (libcsi_hip has no CPU fallback)");
    if (device_id < 0 || device_id >= n) return fail(nullptr, CSI_ERR_INVALID_ARGUMENT, "device_id out of range");
    HIP_TRY(nullptr, hipSetDevice(device_id));
    csi_context* c = new csi_context();
    c->device = device_id;
    if (hip_stream) {
        c->stream = (hipStream_t)hip_stream;
    } else {
        e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete c; return fail(nullptr, CSI_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
        c->own_stream = true;
    }
    hipEventCreate(&c->ev0);
    hipEventCreate(&c->ev1);
    c->stress[0].kind = CSI_STRESS_NONE;
    c->stress[1].kind = CSI_STRESS_NONE;
    {
        auto env_int = [](const char* name) { const char* e = getenv(name); return (e && *e) ? atoi(e) : -1; };
        c->tune.fused_rows = env_int("CSI_FUSED_ROWS"); c->tune.pair_tiles = env_int("CSI_PAIR_TILES");
        c->tune.pair_minrows = env_int("CSI_PAIR_MINROWS"); c->tune.pair_rows = env_int("CSI_PAIR_ROWS");
        c->tune.pair_common = env_int("CSI_PAIR_COMMON"); c->tune.no_geom_sig = env_int("CSI_DEBUG_NO_GEOM_SIG");
        c->tune.peer_edge = env_int("CSI_PEER_EDGE");          // rows the chunks next to a peer-connected y side are shorter by (default 4; 0: uniform chunks)
        c->tune.write_through = env_int("CSI_WRITE_THROUGH");  // 0 / 1: never / always store the pair kernel's results write-through (default: by grid size)
        c->tune.adv_nt = env_int("CSI_ADV_NT"); c->tune.pair_target = env_int("CSI_PAIR_TARGET");
        { const char* e = getenv("CSI_ADV_STAGE_MAX_CELLS"); if (e && *e) c->tune.adv_stage_max_cells = atol(e); }      // A/B: largest grid that takes one launch per RK stage
        { const int v = env_int("CSI_ROW_TARGET_1024"); if (v >= 0) c->tune.row_target_1024 = v; }      // 0: per-row coefficients keep 1536 tiles (rounds 3-5a)
        { const int v = env_int("CSI_TILE_SKIPPING"); if (v >= 0) c->act.enabled = v != 0; }      // A/B: the defaults of csi_set_tile_skipping / csi_set_row_constant
        { const int v = env_int("CSI_ROW_CONSTANT"); if (v >= 0) c->rc_enabled = v != 0; }
        c->tune.band_fused = env_int("CSI_BAND_FUSED"); c->tune.band_event_flags = env_int("CSI_BAND_EVENT_FLAGS");
        c->tune.band_cus = env_int("CSI_BAND_CUS"); c->tune.band_cus_share = env_int("CSI_BAND_CUS_SHARE");
        c->tune.exp_band_only = env_int("CSI_EXP_BAND_ONLY");
        c->tune.exp_overlap = env_int("CSI_EXP_OVERLAP");    // timing experiment, tiles connected to themselves only (scripts/tile_overlap_dependent.py)
        c->tune.peer_kernel = env_int("CSI_PEER_KERNEL");      // 1: untiled grids run the PEER instantiation of the pair kernel (no neighbour, no waits): what the instantiation itself costs
    }
    *out = c;
    return CSI_OK;
}

int32_t csi_context_destroy(csi_context* c) {
    if (!c) return CSI_OK;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    if (c->dev_metrics) hipFree(c->dev_metrics);
    if (c->dev_fcor) hipFree(c->dev_fcor);
    if (c->dev_fcor2) hipFree(c->dev_fcor2);
    peer_release(c);
    if (c->peer.slots) hipFree(c->peer.slots);
    if (c->peer.err) hipFree(c->peer.err);
    if (c->peer.err_host) hipHostFree(c->peer.err_host);
    if (c->peer.xbuf) hipFree(c->peer.xbuf);
    if (c->dev_coef2) hipFree(c->dev_coef2);
    if (c->dev_c2row) hipFree(c->dev_c2row);
    if (c->dev_rcsum) hipFree(c->dev_rcsum);
    if (c->act.flags) hipFree(c->act.flags);
    if (c->act.list) hipFree(c->act.list);
    if (c->act.list0) hipFree(c->act.list0);
    if (c->act.host) hipHostFree(c->act.host);
    if (c->host_ring) hipHostFree(c->host_ring);
    for (auto& e : c->ring_ev) if (e) hipEventDestroy(e);
    if (c->dev_coef) hipFree(c->dev_coef);
    for (int k = 0; k < 5; ++k) if (c->alt[k]) hipFree(c->alt[k]);
    for (int k = 0; k < 4; ++k) if (c->adv_buf[k]) hipFree(c->adv_buf[k]);
    for (int k = 0; k < 9; ++k) if (c->band[k]) hipFree(c->band[k]);
    if (c->band_ev_pair) hipEventDestroy(c->band_ev_pair);
    if (c->band_ev_band) hipEventDestroy(c->band_ev_band);
    if (c->band_stream) hipStreamDestroy(c->band_stream);
    if (c->pair_stream) hipStreamDestroy(c->pair_stream);
    for (hipEvent_t e : c->exp_ev) if (e) hipEventDestroy(e);
    for (int k = 0; k < 2; ++k) if (c->fbar[k]) hipFree(c->fbar[k]);
    for (int k = 0; k < 2; ++k) if (c->fbar_top[k]) hipFree(c->fbar_top[k]);
    for (int k = 0; k < 2; ++k) if (c->fd[k]) hipFree(c->fd[k]);
    for (int k = 0; k < 2; ++k) if (c->xd[k]) hipFree(c->xd[k]);
    if (c->dev_tables) hipFree(c->dev_tables);
    if (c->sendbuf) hipFree(c->sendbuf);
    if (c->recvbuf) hipFree(c->recvbuf);
    if (c->comm) ncclCommDestroy(c->comm);
    if (c->hostg) hostgroup_leave(c->hostg);
    for (hipEvent_t e : c->stats.ev) hipEventDestroy(e);
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    if (c->own_stream) hipStreamDestroy(c->stream);
    delete c;
    return CSI_OK;
}

// A wait of the peer halo transport has given up (this rank's error word, copied to pinned memory behind every sub-cycle): the
// sub-cycle that saw it is invalid on this rank and -- through the abort words -- on its neighbours.  Reported by EVERY entry
// point that advances the model (at its start and at its end) and by csi_sync, whichever comes first.
// The error is STICKY (round 5): the flag protocol cannot recover by itself -- aborted edge tiles never published
// their launch number while the host kept counting, so every later launch would wait three seconds for flags that never come, and a
// neighbour still inside its aborted sub-cycle can raise the abort words again behind any local clean-up.  From here on every entry
// point that advances the model returns CSI_ERR_COMM until the CALLER re-arms the transport on ALL ranks: csi_set_halo_transport
// (either kind; CSI_TRANSPORT_PEER makes the next sub-cycle run the collective set-up again, which clears flags, abort words and
// launch numbers once every rank has arrived) or a new csi_comm_init*.  Ranks that are not neighbours of the rank that gave up
// learn of it when their own waits time out in turn (each within 3 s), or at once through csi_validate_all.
static int32_t peer_check(csi_context* c);
extern "C++" { namespace csi_host { int32_t peer_check_entry(csi_context* c) { return peer_check(c); } } }
static int32_t peer_check(csi_context* c) {
    if (c->peer.err_host && *c->peer.err_host) c->peer.aborted = true;
    if (c->peer.aborted)
        return fail(c, CSI_ERR_COMM, "peer halo transport: a tile waited 3 s for its neighbour's flags and gave up (or a neighbouring rank did) -- the results of "
                                     "that sub-cycle are invalid (a rank that fell behind or died) and the transport stays refused until EVERY rank has called "
                                     "csi_set_halo_transport again (CSI_TRANSPORT_PEER: a new collective set-up at the next sub-cycle; CSI_TRANSPORT_RCCL: the "
                                     "message exchange)");
    return CSI_OK;
}
int32_t csi_sync(csi_context* c) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return peer_check(c);
}
// testing aid: what a wait of the flag protocol that gave up leaves behind on the host side (the pinned error word), without the
// three seconds -- tests/test_gpu_evp.py::test_peer_abort_is_sticky_until_rearmed drives the recovery path with it
int32_t csi_debug_peer_abort(csi_context* c) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    if (!c->peer.err_host) return fail(c, CSI_ERR_NOT_BOUND, "the peer transport has not been set up");
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *c->peer.err_host = 1u;
    return CSI_OK;
}
// csi_sync on every rank + the transport's status reduced over ALL ranks (collective: every rank of the communicator calls it
// between the same two steps).  What output writers / checkpointers call before they read a tiled model's fields: a peer-transport
// abort reaches only the direct neighbours' abort words, this makes every rank see it -- and take the same decision.
int32_t csi_validate_all(csi_context* c) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    int bad = (c->peer.aborted || (c->peer.err_host && *c->peer.err_host)) ? 1 : 0;
    int32_t rc;
    if ((rc = comm_allreduce_max(c, &bad))) return rc;
    if (bad) c->peer.aborted = true;
    return peer_check(c);
}

int32_t csi_set_mode(csi_context* c, int32_t mode) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    if (mode != CSI_MODE_STRICT && mode != CSI_MODE_FAST) return fail(c, CSI_ERR_INVALID_ARGUMENT, "unknown mode");
    c->mode = mode;
    return CSI_OK;
}

int32_t csi_set_tile_skipping(csi_context* c, int32_t on) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    c->act.enabled = on ? 1 : 0;
    c->act.scale = 1.0; c->act.last_live = -1; c->act.last_tiles = 0; c->act.last_used = 0; c->act.since_probe = 0;
    return CSI_OK;
}

int32_t csi_tile_activity(csi_context* c, int32_t* tiles, int32_t* live, int32_t* used) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    // the newest sample the device has written (no synchronisation: call csi_sync first for the last sub-cycle's)
    activity_sample(c);
    if (tiles) *tiles = c->act.last_tiles;
    if (live) *live = c->act.last_live;
    if (used) *used = c->act.last_used;
    return CSI_OK;
}

int32_t csi_set_row_constant(csi_context* c, int32_t on, double rtol) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    if (!(rtol >= 0.0) || rtol > 1e-6) return fail(c, CSI_ERR_INVALID_ARGUMENT, "row-constant tolerance: 0 <= rtol <= 1e-6");
    c->rc_enabled = on ? 1 : 0;
    c->rc_rtol = rtol;
    c->rc_dirty = true;
    return CSI_OK;
}

int32_t csi_row_constant_rows(csi_context* c, int32_t* rows) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    int32_t rc = ensure_row_constant(c);
    if (rc) return rc;
    if (rows) *rows = c->rc_rows;
    return CSI_OK;
}

int32_t csi_grid_set(csi_context* c, int32_t Nx, int32_t Ny, int32_t Hx, int32_t Hy, int32_t topo_x, int32_t topo_y,
                     int32_t metric_kind, const csi_metrics* m) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    if (Nx < 1 || Ny < 1 || Hx < 1 || Hy < 1) return fail(c, CSI_ERR_INVALID_ARGUMENT, "grid sizes must be >= 1");
    if ((topo_y == CSI_RIGHT_FOLDED || topo_y == CSI_LEFT_CONNECTED_RIGHT_FOLDED) && topo_x != CSI_PERIODIC)
        return fail(c, CSI_ERR_UNSUPPORTED, "a north fold needs a Periodic, unpartitioned x direction (Partition(1, Ry))");
    if (topo_x < CSI_PERIODIC || topo_x > CSI_RIGHT_CONNECTED || topo_y < CSI_PERIODIC || topo_y > CSI_LEFT_CONNECTED_RIGHT_FOLDED)
        return fail(c, CSI_ERR_INVALID_ARGUMENT, "unknown topology");
    if (!m) return fail(c, CSI_ERR_INVALID_ARGUMENT, "metrics == NULL");
    if (metric_kind != CSI_METRIC_UNIFORM && metric_kind != CSI_METRIC_PER_J && metric_kind != CSI_METRIC_FULL)
        return fail(c, CSI_ERR_INVALID_ARGUMENT, "unknown metric kind");
    HIP_TRY(c, hipSetDevice(c->device));
    c->Nx = Nx; c->Ny = Ny; c->Hx = Hx; c->Hy = Hy; c->topo_x = topo_x; c->topo_y = topo_y; c->metric_kind = metric_kind;
    GridDev& g = c->g;
    g = GridDev{};   // a new grid also drops any mask
    g.Nx = Nx; g.Ny = Ny; g.Hx = Hx; g.Hy = Hy;
    g.xlo = side_lo(topo_x); g.xhi = side_hi(topo_x); g.ylo = side_lo(topo_y); g.yhi = side_hi(topo_y);
    g.metric_kind = metric_kind;
    g.dx = m->dx; g.dy = m->dy;
    HIP_TRY(c, hipStreamSynchronize(c->stream));      // kernels still in flight may read the tables freed below
    if (c->dev_metrics) { hipFree(c->dev_metrics); c->dev_metrics = nullptr; }
    if (metric_kind == CSI_METRIC_PER_J) {
        if (!m->dxc || !m->dxf || !m->azc || !m->azf) return fail(c, CSI_ERR_INVALID_ARGUMENT, "PER_J metrics need dxc, dxf, azc, azf");
        const size_t n = (size_t)Ny + 2 * (size_t)Hy + 1;
        std::vector<double> host(8 * n);
        const double* src[4] = {m->dxc, m->dxf, m->azc, m->azf};
        for (int k = 0; k < 4; ++k)
            for (size_t t = 0; t < n; ++t) {
                host[k * n + t] = src[k][t];
                host[(4 + k) * n + t] = 1.0 / src[k][t];
            }
        HIP_TRY(c, hipMalloc((void**)&c->dev_metrics, sizeof(double) * 8 * n));
        HIP_TRY(c, hipMemcpy(c->dev_metrics, host.data(), sizeof(double) * 8 * n, hipMemcpyHostToDevice));
        const double* base = c->dev_metrics + (Hy - 1);   // so that ptr[j] is row j
        g.dxc = base; g.dxf = base + n; g.azc = base + 2 * n; g.azf = base + 3 * n;
        g.rdxc = base + 4 * n; g.rdxf = base + 5 * n; g.razc = base + 6 * n; g.razf = base + 7 * n;
    }
    if (metric_kind == CSI_METRIC_FULL) {
        const int64_t ni = (int64_t)Nx + 2 * Hx + 1, nj = (int64_t)Ny + 2 * Hy + 1;
        if (m->full_ld < ni) return fail(c, CSI_ERR_INVALID_ARGUMENT, "FULL metrics: full_ld < Nx + 2Hx + 1");
        std::vector<double> host((size_t)(12 * ni * nj));
        for (int k = 0; k < 12; ++k) {
            if (!m->full[k]) return fail(c, CSI_ERR_INVALID_ARGUMENT, "FULL metrics need all twelve arrays");
            for (int64_t r = 0; r < nj; ++r)
                for (int64_t q = 0; q < ni; ++q) host[(size_t)((k * nj + r) * ni + q)] = m->full[k][r * m->full_ld + q];
        }
        HIP_TRY(c, hipMalloc((void**)&c->dev_metrics, sizeof(double) * host.size()));
        HIP_TRY(c, hipMemcpy(c->dev_metrics, host.data(), sizeof(double) * host.size(), hipMemcpyHostToDevice));
        g.m2 = c->dev_metrics + (Hx - 1) + (int64_t)(Hy - 1) * ni;      // so that m2[k * plane + i + j * ld] is (i, j)
        g.m2_plane = (long)(ni * nj);
        g.m2_ld = (int)ni;
        // FAST mode: per-point stencil coefficients (three-kernel path)
        std::vector<double> coef2;
        const double* planes[12];
        for (int k = 0; k < 12; ++k) planes[k] = host.data() + (size_t)k * ni * nj;
        build_fast_coef_full((int)ni, (int)nj, planes, coef2);
        if (c->dev_coef2) { hipFree(c->dev_coef2); c->dev_coef2 = nullptr; }
        HIP_TRY(c, hipMalloc((void**)&c->dev_coef2, sizeof(double) * coef2.size()));
        HIP_TRY(c, hipMemcpy(c->dev_coef2, coef2.data(), sizeof(double) * coef2.size(), hipMemcpyHostToDevice));
        c->coef2_host.swap(coef2);      // (row-constant marks: ensure_row_constant)
    } else {
        c->coef2_host.clear();
    }
    c->fcor2_host.clear();
    c->rc_dirty = true;
    // FAST-mode stencil coefficients
    if (c->dev_coef) { hipFree(c->dev_coef); c->dev_coef = nullptr; }
    if (c->dev_fcor) { hipFree(c->dev_fcor); c->dev_fcor = nullptr; }
    if (c->dev_fcor2) { hipFree(c->dev_fcor2); c->dev_fcor2 = nullptr; }
    c->coef_host.clear(); c->fcor_rows[0].clear(); c->fcor_rows[1].clear();
    c->cor_dirty = true;
    c->coef = FastCoef{};
    c->coef.uniform = metric_kind == CSI_METRIC_UNIFORM;
    if (metric_kind == CSI_METRIC_FULL) {
        const int64_t ni = (int64_t)Nx + 2 * Hx + 1, nj = (int64_t)Ny + 2 * Hy + 1;
        c->coef.full = 1;
        c->coef.c2 = c->dev_coef2 + (Hx - 1) + (int64_t)(Hy - 1) * ni;
        c->coef.c2_plane = (long)(ni * nj);
        c->coef.c2_ld = (int)ni;
    }
    if (metric_kind == CSI_METRIC_UNIFORM) {
        build_fast_coef_uniform(m->dx, m->dy, c->coef.uni);
    } else if (metric_kind == CSI_METRIC_PER_J) {
        const int n = Ny + 2 * Hy + 1;
        std::vector<double> host;
        build_fast_coef_per_j(n, m->dy, m->dxc, m->dxf, m->azc, m->azf, host);
        c->coef_host = host;          // uploaded, with the Coriolis columns, by sync_coriolis
    }
    for (auto& b : c->f) b = Bound{};   // bindings refer to the previous grid
    c->act.scale = 1.0; c->act.last_live = -1; c->act.last_tiles = 0; c->act.since_probe = 0;      // (tile activity: what was learnt belongs to the previous grid)
    c->grid_set = true;
    return CSI_OK;
}

int32_t csi_mask_set(csi_context* c, const uint8_t* dev_mask, int64_t ld) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    if (!c->grid_set) return fail(c, CSI_ERR_NOT_BOUND, "csi_grid_set has not been called");
    if (!dev_mask) { c->g.mask = nullptr; c->g.has_mask = 0; c->g.mask_ld = 0; return CSI_OK; }
    if (ld < c->Nx + 2 * c->Hx) return fail(c, CSI_ERR_INVALID_ARGUMENT, "mask ld too small");
    c->g.mask = dev_mask + (c->Hx - 1) + (int64_t)(c->Hy - 1) * ld;
    c->g.mask_ld = (int)ld;
    c->g.has_mask = 1;
    return CSI_OK;
}

int32_t csi_field_bind(csi_context* c, int32_t fid, void* dev_ptr, int64_t ld, int32_t ni, int32_t nj) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    if (!c->grid_set) return fail(c, CSI_ERR_NOT_BOUND, "csi_grid_set has not been called");
    if (fid < 0 || fid >= CSI_F_COUNT) return fail(c, CSI_ERR_INVALID_ARGUMENT, "unknown field id");
    if (!dev_ptr) { c->f[fid] = Bound{}; return CSI_OK; }
    const int eni = c->Nx + 2 * c->Hx + extra_x(c, fid), enj = c->Ny + 2 * c->Hy + extra_y(c, fid);
    if (ni != eni || nj != enj || ld < ni) {
        char buf[256];
        snprintf(buf, sizeof buf, "field %s: parent extents (%d, %d, ld %lld) do not match the grid (expected %d x %d)",
                 kName[fid], ni, nj, (long long)ld, eni, enj);
        return fail(c, CSI_ERR_INVALID_ARGUMENT, buf);
    }
    if (ld > 0x7fffffff) return fail(c, CSI_ERR_INVALID_ARGUMENT, "ld too large");
    if (((uintptr_t)dev_ptr) & 7) return fail(c, CSI_ERR_INVALID_ARGUMENT, "field pointer must be 8-byte aligned");
    const bool peer_field = fid == CSI_F_U || fid == CSI_F_V || fid == CSI_F_S11 || fid == CSI_F_S22 || fid == CSI_F_S12 || fid == CSI_F_ALPHA ||
                            fid == CSI_F_ZETA_C || fid == CSI_F_ZETA_F || fid == CSI_F_DELTA;
    if (peer_field && c->f[fid].p != (double*)dev_ptr && c->peer.ready) {
        // the neighbours hold mappings of the OLD array: the next sub-cycle sets the peer transport up again.  That set-up is
        // collective -- on a tiled model, re-binding one of these nine arrays is something every rank has to do between the same
        // two steps (include/csi.h)
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        peer_release(c);
    }
    c->f[fid].p = (double*)dev_ptr; c->f[fid].ld = ld; c->f[fid].ni = ni; c->f[fid].nj = nj;
    return CSI_OK;
}

int32_t csi_immersed_flux_bc_set(csi_context* c, int32_t fid, double west, double east, double south, double north) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    if (fid != CSI_F_U && fid != CSI_F_V) return fail(c, CSI_ERR_INVALID_ARGUMENT, "immersed flux boundary conditions: CSI_F_U or CSI_F_V");
    double* q = c->ibc[fid == CSI_F_U ? 0 : 1];
    q[0] = west; q[1] = east; q[2] = south; q[3] = north;
    return CSI_OK;
}

int32_t csi_evp_params_set(csi_context* c, const csi_evp_params* p) {
    if (!c || !p) return CSI_ERR_INVALID_ARGUMENT;
    if (p->pressure_formulation != CSI_PRESSURE_REPLACEMENT && p->pressure_formulation != CSI_PRESSURE_ICE_STRENGTH)
        return fail(c, CSI_ERR_INVALID_ARGUMENT, "unknown pressure formulation");
    c->evp = *p;
    c->evp_set = true;
    return CSI_OK;
}

int32_t csi_coriolis_rows_set(csi_context* c, const double* f_u, const double* f_v, int32_t n) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    if (!c->grid_set) return fail(c, CSI_ERR_NOT_BOUND, "csi_grid_set has not been called");
    if ((f_u == nullptr) != (f_v == nullptr)) return fail(c, CSI_ERR_INVALID_ARGUMENT, "f_u and f_v: both or neither");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->cor_dirty = true;
    if (c->dev_fcor) { hipFree(c->dev_fcor); c->dev_fcor = nullptr; }
    c->fcor_rows[0].clear(); c->fcor_rows[1].clear();
    if (!f_u) return CSI_OK;
    const int need = c->Ny + 2 * c->Hy + 1;
    if (n != need) return fail(c, CSI_ERR_INVALID_ARGUMENT, "coriolis rows: n must be Ny + 2Hy + 1");
    c->fcor_rows[0].assign(f_u, f_u + n); c->fcor_rows[1].assign(f_v, f_v + n);
    std::vector<double> host(2 * (size_t)n);
    for (int t = 0; t < n; ++t) { host[t] = f_u[t]; host[(size_t)n + t] = f_v[t]; }
    HIP_TRY(c, hipMalloc((void**)&c->dev_fcor, sizeof(double) * host.size()));
    HIP_TRY(c, hipMemcpy(c->dev_fcor, host.data(), sizeof(double) * host.size(), hipMemcpyHostToDevice));
    return CSI_OK;
}

int32_t csi_coriolis_points_set(csi_context* c, const double* f_u, const double* f_v, int64_t ld) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    if (!c->grid_set) return fail(c, CSI_ERR_NOT_BOUND, "csi_grid_set has not been called");
    if ((f_u == nullptr) != (f_v == nullptr)) return fail(c, CSI_ERR_INVALID_ARGUMENT, "f_u and f_v: both or neither");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->dev_fcor2) { hipFree(c->dev_fcor2); c->dev_fcor2 = nullptr; }
    c->fcor2_host.clear();
    c->rc_dirty = true;
    if (!f_u) return CSI_OK;
    if (c->metric_kind != CSI_METRIC_FULL)
        return fail(c, CSI_ERR_UNSUPPORTED, "per-point Coriolis parameter: CSI_METRIC_FULL grids only (use csi_coriolis_rows_set on per-row grids)");
    const long ni = c->Nx + 2 * c->Hx + 1, nj = c->Ny + 2 * c->Hy + 1;
    if (ld < ni) return fail(c, CSI_ERR_INVALID_ARGUMENT, "coriolis points: ld must be >= Nx + 2Hx + 1");
    std::vector<double> host(2 * (size_t)ni * nj);
    for (long b = 0; b < nj; ++b)
        for (long a = 0; a < ni; ++a) { host[a + b * ni] = f_u[a + b * ld]; host[(size_t)ni * nj + a + b * ni] = f_v[a + b * ld]; }
    HIP_TRY(c, hipMalloc((void**)&c->dev_fcor2, sizeof(double) * host.size()));
    HIP_TRY(c, hipMemcpy(c->dev_fcor2, host.data(), sizeof(double) * host.size(), hipMemcpyHostToDevice));
    c->fcor2_ld = ni; c->fcor2_plane = ni * nj;
    c->fcor2_host.swap(host);
    return CSI_OK;
}

int32_t csi_velocity_bc_set(csi_context* c, int32_t field_id, int32_t side, int32_t kind, double value) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    if (field_id != CSI_F_U && field_id != CSI_F_V) return fail(c, CSI_ERR_INVALID_ARGUMENT, "velocity boundary conditions: CSI_F_U or CSI_F_V");
    if (side != 0 && side != 1) return fail(c, CSI_ERR_INVALID_ARGUMENT, "side: 0 (south / west) or 1 (north / east)");
    if (kind != 0 && kind != 1) return fail(c, CSI_ERR_INVALID_ARGUMENT, "kind: 0 (default no-flux) or 1 (ValueBoundaryCondition)");
    const int q = field_id == CSI_F_U ? 0 : 1;
    c->vel_bc_on[q][side] = kind;
    c->vel_bc_value[q][side] = kind ? value : 0.0;
    return CSI_OK;
}

int32_t csi_stress_set(csi_context* c, int32_t side, const csi_stress* s) {
    if (!c || !s) return CSI_ERR_INVALID_ARGUMENT;
    if (side != CSI_STRESS_TOP && side != CSI_STRESS_BOTTOM) return fail(c, CSI_ERR_INVALID_ARGUMENT, "unknown stress side");
    if (s->kind < CSI_STRESS_NONE || s->kind > CSI_STRESS_SEMI_IMPLICIT) return fail(c, CSI_ERR_INVALID_ARGUMENT, "unknown stress kind");
    c->stress[side] = *s;
    return CSI_OK;
}

int32_t csi_evp_initialize(csi_context* c) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    int32_t rc = need_evp(c);
    if (rc) return rc;
    return do_initialize(c);
}

int32_t csi_evp_subcycle(csi_context* c, double dt, int32_t substeps, int32_t first_substep) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    int32_t rc = need_evp(c);
    if (rc) return rc;
    if (substeps < 0 || first_substep < 1) return fail(c, CSI_ERR_INVALID_ARGUMENT, "substeps >= 0 and first_substep >= 1 required");
    rc = do_subcycle(c, dt, substeps, first_substep);
    return rc ? rc : peer_check(c);
}

int32_t csi_evp_finalize(csi_context* c) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    int32_t rc = need(c, {CSI_F_S11, CSI_F_S22, CSI_F_S12});
    if (rc) return rc;
    return do_finalize(c);
}

int32_t csi_time_step_momentum(csi_context* c, double dt, int32_t substeps, int32_t rk_reset) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    int32_t rc = need_evp(c);
    if (rc) return rc;
    if (substeps < 0) return fail(c, CSI_ERR_INVALID_ARGUMENT, "substeps >= 0 required");
    rc = do_time_step_momentum(c, dt, substeps, rk_reset);
    return rc ? rc : peer_check(c);
}

int32_t csi_compute_tracer_tendencies(csi_context* c, int32_t scheme) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    return do_tendencies(c, scheme);
}

int32_t csi_dynamic_step_tracers(csi_context* c, double dt, int32_t from_cache) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    return do_tracer_step(c, dt, from_cache);
}

int32_t csi_cache_current_fields(csi_context* c) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    int32_t rc = need(c, {CSI_F_H, CSI_F_A, CSI_F_HM, CSI_F_AM});
    if (rc) return rc;
    // Psi^- = the prognostic fields, whole parents, in one launch
    CopyBatch B{};
    B.aligned16 = 1;
    auto add = [&](int dst, int src) -> int32_t {
        const Bound &d = c->f[dst], &q = c->f[src];
        if (d.ld != q.ld || d.nj != q.nj) return fail(c, CSI_ERR_INVALID_ARGUMENT, std::string("parent shape mismatch: ") + kName[dst] + " vs " + kName[src]);
        B.src[B.count] = q.p; B.dst[B.count] = d.p; B.n[B.count] = (long)d.ld * d.nj; ++B.count;
        if ((((uintptr_t)q.p) | ((uintptr_t)d.p)) & 15) B.aligned16 = 0;
        return CSI_OK;
    };
    if ((rc = add(CSI_F_HM, CSI_F_H))) return rc;
    if ((rc = add(CSI_F_AM, CSI_F_A))) return rc;
    if (c->f[CSI_F_HS].p && c->f[CSI_F_HSM].p && (rc = add(CSI_F_HSM, CSI_F_HS))) return rc;
    if (c->f[CSI_F_U].p && c->f[CSI_F_UM].p) {
        if ((rc = need(c, {CSI_F_V, CSI_F_VM}))) return rc;
        if ((rc = add(CSI_F_UM, CSI_F_U))) return rc;
        if ((rc = add(CSI_F_VM, CSI_F_V))) return rc;
    }
    launch_copy_batch(B, c->stream);
    HIP_TRY(c, hipGetLastError());
    return CSI_OK;
}

int32_t csi_update_state(csi_context* c) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    return do_update_state(c);
}

int32_t csi_fill_halo_local(csi_context* c, int32_t fid) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    if (fid < 0 || fid >= CSI_F_COUNT) return fail(c, CSI_ERR_INVALID_ARGUMENT, "unknown field id");
    int32_t rc = need(c, {fid});
    if (rc) return rc;
    return fill_halo(c, fid);
}

int32_t csi_time_step_fe(csi_context* c, double dt, int32_t substeps, int32_t scheme, int32_t first_iteration) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    // dynamics = nothing (csi_evp_params_set never called): prescribed velocities, time_step_momentum! is a no-op
    // (SeaIceDynamics.jl:40) -- the advection-only models of examples/ice_advected_by_anticyclone.jl's family
    const bool dynamics = c->evp_set;
    int32_t rc = dynamics ? need_evp(c) : need(c, {CSI_F_H, CSI_F_A});
    if (rc) return rc;
    if (first_iteration && (rc = do_update_state(c))) return rc;          // sea_ice_fe_step.jl:16
    if ((rc = do_tendencies_or_zero(c, scheme))) return rc;               // :19
    if (dynamics && (rc = do_time_step_momentum(c, dt, substeps, 0))) return rc;      // :22
    // without a mask and without a thermodynamic step the tracer update's stores write the halo images themselves
    const bool fused_fill = !c->g.has_mask && !c->slab_set;
    if ((rc = do_tracer_step(c, dt, 0, fused_fill))) return rc;           // :25
    if ((rc = do_thermo(c, dt))) return rc;                               // :28 thermodynamic_time_step!
    if ((rc = do_update_state(c, true, fused_fill))) return rc;           // :31
    return peer_check(c);
}

// An RK3 step of an advection-only model (prescribed velocities: examples/ice_advected_by_anticyclone.jl's family, BASELINE
// config 2) with ONE launch per stage: nothing happens between a stage's tendencies and its tracer update, so the kernel that
// computes G also applies it -- into another copy of (h, a), because its neighbours still read the stage's input: the state
// rotates bound arrays -> copy 1 -> copy 2 -> bound arrays.  The first stage also writes Psi^- (cache_current_fields!).  Same
// arithmetic as the separate kernels, statement for statement: bit-identical (tests/test_gpu_steps.py).
bool advect_stage_supported(const csi_context* c, int scheme) {
    const ImageSpec im = image_spec(c, CSI_F_H);
    for (int side : {im.xlo, im.xhi, im.ylo, im.yhi}) if (side != IMG_WRAP && side != IMG_MIRROR) return false;      // (periodic / no-flux walls)
    if (c->Nx < 2 * c->Hx || c->Ny < 2 * c->Hy) return false;
    // Round 3 cut this path off above 3 M cells (2048^2: 768 -> 794 us per RK3 step with it).  Re-measured in round 6 with two tracers
    // per thread and the round-5 block shapes (scripts/ab_adv_stage.sh, profiles/r06_advection_stage_ab.txt; WENO7, us per RK3 step,
    // separate launches -> one per stage): 1536^2 314 -> 264, 2048^2 538 -> 476, 3072^2 1205 -> 988, 4096^2 2110 -> 1711 -- the
    // separate update streams another 7 arrays per stage, which the fused stage never touches: no cut any more (A/B knob:
    // CSI_ADV_STAGE_MAX_CELLS)
    if ((long)c->Nx * c->Ny > c->tune.adv_stage_max_cells) return false;
    return !c->evp_set && scheme != 0 && c->fusion && !c->slab_set && !c->g.has_mask && !is_tiled(c) &&
           c->f[CSI_F_HS].p == nullptr && c->f[CSI_F_H].ld == c->f[CSI_F_HM].ld && c->f[CSI_F_A].ld == c->f[CSI_F_AM].ld;
}
int32_t rk3_advection_only(csi_context* c, double dt, int scheme) {
    int32_t rc;
    if ((rc = need(c, {CSI_F_U, CSI_F_V, CSI_F_H, CSI_F_A, CSI_F_GH, CSI_F_GA, CSI_F_HM, CSI_F_AM}))) return rc;
    const int src[4] = {CSI_F_H, CSI_F_A, CSI_F_H, CSI_F_A};
    for (int q = 0; q < 4; ++q) {
        const Bound& b = c->f[src[q]];
        const size_t n = (size_t)b.ld * (size_t)b.nj;
        if (c->adv_elems[q] != n) {
            if (c->adv_buf[q]) { HIP_TRY(c, hipStreamSynchronize(c->stream)); hipFree(c->adv_buf[q]); c->adv_buf[q] = nullptr; }
            HIP_TRY(c, hipMalloc((void**)&c->adv_buf[q], n * sizeof(double)));
            // beyond walls the halo holds mirror images the stores rewrite; cells nobody writes (wall corners) start as the state's
            HIP_TRY(c, hipMemcpyAsync(c->adv_buf[q], b.p, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
            c->adv_elems[q] = n;
        }
    }
    auto buf = [&](int q) { FRef r; const Bound& b = c->f[src[q]]; r.p = c->adv_buf[q] + (c->Hx - 1) + (int64_t)(c->Hy - 1) * b.ld; r.ld = (int)b.ld; return r; };
    const FRef H0 = ref_of(c, CSI_F_H), A0 = ref_of(c, CSI_F_A), H1 = buf(0), A1 = buf(1), H2 = buf(2), A2 = buf(3);
    const FRef hin[3] = {H0, H1, H2}, ain[3] = {A0, A1, A2}, hout[3] = {H1, H2, H0}, aout[3] = {A1, A2, A0};
    int stage = 0;
    for (int beta = 3; beta >= 1; --beta, ++stage) {
        AdvDev A = adv_dev(c, scheme, dt / beta, 1);
        A.h = hin[stage]; A.a = ain[stage];
        A.hb = stage == 0 ? H0 : A.hm; A.ab = stage == 0 ? A0 : A.am;
        A.ho = hout[stage]; A.ao = aout[stage];
        A.write_cache = stage == 0;
        A.fill_images = 1;
        launch_advect_stage(A, c->mode, c->stream);
        HIP_TRY(c, hipGetLastError());
    }
    return CSI_OK;
}

int32_t csi_time_step_rk3(csi_context* c, double dt, int32_t substeps, int32_t scheme) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    const bool dynamics = c->evp_set;                                     // see csi_time_step_fe
    int32_t rc = dynamics ? need_evp(c) : need(c, {CSI_F_H, CSI_F_A});
    if (rc) return rc;
    if ((rc = dynamics ? need(c, {CSI_F_HM, CSI_F_AM, CSI_F_UM, CSI_F_VM}) : need(c, {CSI_F_HM, CSI_F_AM}))) return rc;
    if (advect_stage_supported(c, scheme)) {
        const bool third = scheme == CSI_ADVECT_WENO3 || scheme == CSI_ADVECT_UPWIND3;
        const int need_h = scheme == CSI_ADVECT_WENO7 ? 4 : (scheme == CSI_ADVECT_UPWIND1 ? 1 : (third ? 2 : 3));
        if (c->Hx >= need_h && c->Hy >= need_h &&
            (scheme == CSI_ADVECT_UPWIND1 || scheme == CSI_ADVECT_WENO5 || scheme == CSI_ADVECT_WENO7 || scheme == CSI_ADVECT_UPWIND5 || third))
            return rk3_advection_only(c, dt, scheme);
    }
    if ((rc = csi_cache_current_fields(c))) return rc;                    // sea_ice_rk_substep.jl:29-42
    for (int beta = 3; beta >= 1; --beta) {                               // upstream stage loop
        const double dtau = dt / beta;
        if ((rc = do_tendencies_or_zero(c, scheme))) return rc;           // :84
        if (dynamics && (rc = do_time_step_momentum(c, dtau, substeps, 1))) return rc;   // :87
        const bool fused_fill = !c->g.has_mask && !c->slab_set;
        if ((rc = do_tracer_step(c, dtau, 1, fused_fill))) return rc;     // :89
        if ((rc = do_thermo(c, dtau))) return rc;                         // :91 thermodynamic_time_step!
        if ((rc = do_update_state(c, true, fused_fill))) return rc;
    }
    return peer_check(c);
}

int32_t csi_slab_thermo_step(csi_context* c, const csi_slab_params* p, double dt) {
    if (!c || !p) return CSI_ERR_INVALID_ARGUMENT;
    int32_t rc = need(c, {CSI_F_H, CSI_F_A});
    if (rc) return rc;
    return do_slab(c, slab_dev(p), dt);
}

int32_t csi_layered_thermo_step(csi_context* c, const csi_slab_params* p, const csi_snow_params* w, double dt) {
    if (!c || !p || !w) return CSI_ERR_INVALID_ARGUMENT;
    return do_layered(c, slab_dev(p), snow_dev(w), dt);
}

int32_t csi_snow_params_set(csi_context* c, const csi_snow_params* p) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    c->snow_set = p != nullptr;
    if (p) c->snow = snow_dev(p);
    return CSI_OK;
}

int32_t csi_slab_params_set(csi_context* c, const csi_slab_params* p) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    c->slab_set = p != nullptr;
    if (p) c->slab = slab_dev(p);
    return CSI_OK;
}

int32_t csi_tile_set(csi_context* c, int32_t rx, int32_t ry, int32_t Rx, int32_t Ry, int32_t periodic_x, int32_t periodic_y) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    if (Rx < 1 || Ry < 1 || rx < 0 || rx >= Rx || ry < 0 || ry >= Ry) return fail(c, CSI_ERR_INVALID_ARGUMENT, "tile coordinates out of range");
    c->tile.rx = rx; c->tile.ry = ry; c->tile.Rx = Rx; c->tile.Ry = Ry;
    c->tile.periodic_x = periodic_x != 0; c->tile.periodic_y = periodic_y != 0;
    c->tile.set = true;
    peer_release(c); c->peer.failed = false; c->peer.aborted = false; if (c->peer.err_host) *c->peer.err_host = 0;           // (the neighbours may be other ranks now)
    return CSI_OK;
}

int32_t csi_comm_unique_id(uint8_t* id128) {
    if (!id128) return fail(nullptr, CSI_ERR_INVALID_ARGUMENT, "id128 == NULL");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, CSI_ERR_COMM, std::string("ncclGetUniqueId: ") + ncclGetErrorString(r));
    memcpy(id128, &id, 128);
    return CSI_OK;
}

int32_t csi_comm_init(csi_context* c, int32_t world_size, int32_t rank, const uint8_t* id128) {
    if (!c || !id128) return CSI_ERR_INVALID_ARGUMENT;
    if (world_size < 1 || rank < 0 || rank >= world_size) return fail(c, CSI_ERR_INVALID_ARGUMENT, "rank / world_size out of range");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->comm) { ncclCommDestroy(c->comm); c->comm = nullptr; }
    if (c->hostg) { hostgroup_leave(c->hostg); c->hostg = nullptr; }
    c->local = nullptr;
    peer_release(c); c->peer.failed = false; c->peer.aborted = false; if (c->peer.err_host) *c->peer.err_host = 0;           // (mappings of the previous communicator's neighbours)
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    NCCL_TRY(c, ncclCommInitRank(&c->comm, world_size, id, rank));
    c->world = world_size; c->rank = rank;
    return CSI_OK;
}

int32_t csi_local_group_create(int32_t world_size, csi_local_group** out) {
    if (!out || world_size < 1) return CSI_ERR_INVALID_ARGUMENT;
    csi_local_group* G = new csi_local_group;
    G->world = world_size;
    G->box.resize((size_t)world_size * world_size);
    G->posted.assign(world_size, 0); G->consumed.assign(world_size, 0);
    G->payload.resize(world_size);
    *out = G;
    return CSI_OK;
}

void csi_local_group_destroy(csi_local_group* G) { delete G; }

int32_t csi_comm_init_local(csi_context* c, csi_local_group* G, int32_t rank) {
    if (!c || !G) return CSI_ERR_INVALID_ARGUMENT;
    if (rank < 0 || rank >= G->world) return fail(c, CSI_ERR_INVALID_ARGUMENT, "rank out of range for this group");
    if (c->comm) { ncclCommDestroy(c->comm); c->comm = nullptr; }
    if (c->hostg) { hostgroup_leave(c->hostg); c->hostg = nullptr; }
    peer_release(c); c->peer.failed = false; c->peer.aborted = false; if (c->peer.err_host) *c->peer.err_host = 0;
    c->local = G;
    c->world = G->world; c->rank = rank;
    // The peer transport's kernels wait for flags the OTHER tiles' kernels of this process publish.  HIP maps streams onto
    // GPU_MAX_HW_QUEUES hardware queues (default 4) and two streams that share one run in submission order: with fewer queues
    // than tiles a waiting kernel can sit in front of the one it waits for until its 3 s time-out.  The variable has to be set
    // before the runtime initialises, so the library can only check it: without enough queues the group runs the message
    // exchange (device copies), and asking for the peer transport fails with a clear error instead of timing out at run time.
    {
        const char* q = getenv("GPU_MAX_HW_QUEUES");
        const int queues = (q && *q) ? atoi(q) : 4;
        c->peer.local_queues_ok = queues > G->world;
        if (!c->peer.local_queues_ok) c->peer.want = 0;
    }
    std::unique_lock<std::mutex> lk(G->mu);
    ++G->joined;
    if ((int)G->device.size() != G->world) G->device.assign((size_t)G->world, -1);
    G->device[(size_t)rank] = c->device;
    return CSI_OK;
}

int32_t csi_comm_init_host(csi_context* c, const char* shm_name, int32_t world_size, int32_t rank) {
    if (!c || !shm_name) return CSI_ERR_INVALID_ARGUMENT;
    if (world_size < 2 || rank < 0 || rank >= world_size) return fail(c, CSI_ERR_INVALID_ARGUMENT, "rank / world_size out of range (a host-channel group has at least two ranks)");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->comm) { ncclCommDestroy(c->comm); c->comm = nullptr; }
    if (c->hostg) { hostgroup_leave(c->hostg); c->hostg = nullptr; }
    c->local = nullptr;
    peer_release(c); c->peer.failed = false; c->peer.aborted = false; if (c->peer.err_host) *c->peer.err_host = 0;
    std::string e;
    c->hostg = hostgroup_join(shm_name, world_size, rank, &e);
    if (!c->hostg) return fail(c, CSI_ERR_COMM, e);
    c->world = world_size; c->rank = rank;
    if (c->sendbuf && !hostgroup_set_sendbuf(c->hostg, c->sendbuf, c->buf_cap * sizeof(double), &c->err)) return CSI_ERR_COMM;
    return CSI_OK;
}
int32_t csi_comm_count(csi_context* c, int32_t* ranks) {
    if (!c || !ranks) return CSI_ERR_INVALID_ARGUMENT;
    *ranks = 0;
    if (c->local) { *ranks = c->local->world; return CSI_OK; }
    if (c->hostg) { *ranks = hostgroup_world(c->hostg); return CSI_OK; }
    if (!c->comm) return CSI_OK;
    int n = 0;
    NCCL_TRY(c, ncclCommCount(c->comm, &n));
    *ranks = n;
    return CSI_OK;
}

int32_t csi_halo_exchange(csi_context* c, const int32_t* field_ids, int32_t nfields, int32_t width) {
    if (!c || !field_ids) return CSI_ERR_INVALID_ARGUMENT;
    if (!c->grid_set) return fail(c, CSI_ERR_NOT_BOUND, "csi_grid_set has not been called");
    for (int k = 0; k < nfields; ++k)
        if (field_ids[k] < 0 || field_ids[k] >= CSI_F_COUNT) return fail(c, CSI_ERR_INVALID_ARGUMENT, "unknown field id");
    return exchange(c, field_ids, nfields, width);
}

int32_t csi_plan_exchange(int32_t Nx, int32_t Ny, int32_t Hx, int32_t Hy, int32_t topo_x, int32_t topo_y,
                          int32_t rx, int32_t ry, int32_t Rx, int32_t Ry, int32_t periodic_x, int32_t periodic_y,
                          int32_t width, int32_t halo, int32_t* out40) {
    if (!out40 || Nx < 1 || Ny < 1 || Rx < 1 || Ry < 1) return CSI_ERR_INVALID_ARGUMENT;
    GridDev g{};
    g.Nx = Nx; g.Ny = Ny; g.Hx = Hx; g.Hy = Hy;
    g.xlo = side_lo(topo_x); g.xhi = side_hi(topo_x); g.ylo = side_lo(topo_y); g.yhi = side_hi(topo_y);
    TileInfo t;
    t.rx = rx; t.ry = ry; t.Rx = Rx; t.Ry = Ry; t.periodic_x = periodic_x != 0; t.periodic_y = periodic_y != 0; t.set = true;
    FRef dummy{nullptr, 0};
    ExPlan pl;
    long off[8], cnt[8];
    int peer[8];
    build_plan(g, t, &dummy, 1, width, halo, pl, off, cnt, peer);
    int s = 0;
    for (int k = 0; k < 8; ++k) {
        out40[5 * k] = peer[k];
        if (peer[k] >= 0) {
            const ExSeg& e = pl.seg[s++];
            out40[5 * k + 1] = e.i0; out40[5 * k + 2] = e.j0; out40[5 * k + 3] = e.ni; out40[5 * k + 4] = e.nj;
        } else {
            out40[5 * k + 1] = out40[5 * k + 2] = out40[5 * k + 3] = out40[5 * k + 4] = 0;
        }
    }
    return CSI_OK;
}

int32_t csi_free_drift_set(csi_context* c, int32_t kind) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    if (kind != 0 && kind != 1) return fail(c, CSI_ERR_INVALID_ARGUMENT, "free drift kind: 0 (nothing) or 1 (StressBalanceFreeDrift)");
    c->free_drift = kind;
    return CSI_OK;
}

int32_t csi_set_weno_weight_dtype(csi_context* c, int32_t dtype) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    if (dtype != CSI_WEIGHTS_F64 && dtype != CSI_WEIGHTS_F32) return fail(c, CSI_ERR_INVALID_ARGUMENT, "WENO weight dtype: CSI_WEIGHTS_F64 or CSI_WEIGHTS_F32");
    c->weno_w32 = dtype == CSI_WEIGHTS_F32;
    return CSI_OK;
}
int32_t csi_weno_weight_dtype(csi_context* c, int32_t* dtype) {
    if (!c || !dtype) return CSI_ERR_INVALID_ARGUMENT;
    *dtype = c->weno_w32 ? CSI_WEIGHTS_F32 : CSI_WEIGHTS_F64;
    return CSI_OK;
}

int32_t csi_set_fusion(csi_context* c, int32_t on) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    c->fusion = on != 0;
    if (on < 0 || on > 2) return fail(c, CSI_ERR_INVALID_ARGUMENT, "fusion level must be 0, 1 or 2 (level 3, three sub-steps per launch, was withdrawn in round 4: slower than level 2 at every size but one)");
    c->pairing = on != 1;         // 1: one sub-step per launch only; 2: pairs where supported
    return CSI_OK;
}

int32_t csi_set_exchange_interval(csi_context* c, int32_t k) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    if (k < 0 || k > 64) return fail(c, CSI_ERR_INVALID_ARGUMENT, "0 <= k <= 64");
    c->exch_k = k;
    return CSI_OK;
}

int32_t csi_set_halo_transport(csi_context* c, int32_t kind) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    if (kind != CSI_TRANSPORT_RCCL && kind != CSI_TRANSPORT_PEER) return fail(c, CSI_ERR_INVALID_ARGUMENT, "unknown halo transport");
    if (kind == CSI_TRANSPORT_PEER && c->local && !c->peer.local_queues_ok)
        return fail(c, CSI_ERR_UNSUPPORTED, "peer halo transport on an in-process tile group needs GPU_MAX_HW_QUEUES > number of tiles, set BEFORE the HIP "
                                            "runtime initialises (the tiles' kernels wait for each other; with fewer hardware queues a waiting kernel can "
                                            "block the one it waits for): export GPU_MAX_HW_QUEUES=16, or keep the message exchange (CSI_TRANSPORT_RCCL)");
    c->peer.want = kind == CSI_TRANSPORT_PEER;
    if (c->peer.want) c->peer.failed = false;            // (asking again retries the set-up)
    if (c->peer.aborted) {
        // re-arming after a wait that gave up (peer_check): the caller does this on EVERY rank.  The stream is drained (this rank's
        // aborted launches are over), the mappings go, and the next sub-cycle on the peer transport runs the collective set-up.
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        peer_release(c);
        if (c->peer.err_host) *c->peer.err_host = 0;
        if (c->peer.err) HIP_TRY(c, hipMemset(c->peer.err, 0, sizeof(unsigned)));
        c->peer.aborted = false;
        c->peer.last = 0;
    }
    return CSI_OK;
}
int32_t csi_halo_transport(csi_context* c, int32_t* kind) {
    if (!c || !kind) return CSI_ERR_INVALID_ARGUMENT;
    *kind = c->peer.last ? CSI_TRANSPORT_PEER : CSI_TRANSPORT_RCCL;
    return CSI_OK;
}
int32_t csi_set_peer_tier(csi_context* c, int32_t tier) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    if (tier < -1 || tier > 2) return fail(c, CSI_ERR_INVALID_ARGUMENT, "peer protocol tier must be -1 (automatic), 0, 1 or 2");
    c->peer.tier = tier;
    return CSI_OK;
}
int32_t csi_peer_tier(csi_context* c, int32_t* tier) {
    if (!c || !tier) return CSI_ERR_INVALID_ARGUMENT;
    *tier = peer_effective_tier(c);      // (what the kernels run: automatic resolves to 1 across processes / devices, 0 otherwise)
    return CSI_OK;
}

int32_t csi_plan_ranges(int32_t Nx, int32_t Ny, int32_t Hx, int32_t Hy, int32_t topo_x, int32_t topo_y, int32_t V, int32_t* out16) {
    if (!out16 || Nx < 1 || Ny < 1 || V < 2) return CSI_ERR_INVALID_ARGUMENT;
    csi_context tmp;
    tmp.Nx = Nx; tmp.Ny = Ny; tmp.Hx = Hx; tmp.Hy = Hy;
    tmp.g.xlo = side_lo(topo_x); tmp.g.xhi = side_hi(topo_x); tmp.g.ylo = side_lo(topo_y); tmp.g.yhi = side_hi(topo_y);
    const Range r[4] = {stress_range(&tmp, V), first_u_range(&tmp, V), first_v_range(&tmp, V), second_range(&tmp, V)};
    for (int k = 0; k < 4; ++k) { out16[4 * k] = r[k].i0; out16[4 * k + 1] = r[k].i1; out16[4 * k + 2] = r[k].j0; out16[4 * k + 3] = r[k].j1; }
    return CSI_OK;
}

int32_t csi_plan_pair(int32_t Nx, int32_t Ny, int32_t Hx, int32_t Hy, int32_t topo_x, int32_t topo_y, int32_t k, int32_t m,
                      int32_t* out32) {
    if (!out32 || Nx < 1 || Ny < 1 || k < 1 || m < 0 || m + 1 >= 2 * ((k + 1) / 2) + (k == 1 ? 1 : 0)) return CSI_ERR_INVALID_ARGUMENT;
    csi_context tmp;
    tmp.Nx = Nx; tmp.Ny = Ny; tmp.Hx = Hx; tmp.Hy = Hy;
    tmp.g.xlo = side_lo(topo_x); tmp.g.xhi = side_hi(topo_x); tmp.g.ylo = side_lo(topo_y); tmp.g.yhi = side_hi(topo_y);
    tmp.coef.uniform = 1; tmp.metric_kind = CSI_METRIC_UNIFORM;   // (per-row metrics with a periodic y side are the one grid kind that never pairs)
    const bool tiled = is_tiled(&tmp);
    memset(out32, 0, 32 * sizeof(int32_t));
    out32[0] = (pair_supported(&tmp) && (!tiled || k % 2 == 0)) ? 1 : 0;
    if (!out32[0]) return CSI_OK;
    const int W = 2 * k;
    const SideV va = pair_side_v(&tmp, W - 2 * m, 4), vb = pair_side_v(&tmp, W - 2 * m - 2, 2);
    const Range ra = v_stress_range(&tmp, va), dec = v_stress_range(&tmp, vb);
    const FusedGeom G = pair_geom(&tmp, dec);
    out32[1] = G.nstrips; out32[2] = G.nchunks; out32[3] = G.rows;
    const Range r[6] = {ra, dec, clip_store(&tmp, dec, true), clip_store(&tmp, v_first_range(&tmp, vb, true), false),
                        clip_store(&tmp, v_first_range(&tmp, vb, false), false), clip_store(&tmp, v_second_range(&tmp, vb), false)};
    for (int q = 0; q < 6; ++q) { out32[4 + 4 * q] = r[q].i0; out32[5 + 4 * q] = r[q].i1; out32[6 + 4 * q] = r[q].j0; out32[7 + 4 * q] = r[q].j1; }
    out32[28] = has_walls(&tmp) ? 1 : 0;
    return CSI_OK;
}

int32_t csi_plan_peer_chunks(int32_t Nx, int32_t Ny, int32_t Hx, int32_t Hy, int32_t peer_south, int32_t peer_north, int32_t cus,
                             int32_t* out8, int32_t* rows_out, int32_t max_chunks) {
    if (!out8 || !rows_out || Nx < 1 || Ny < 1 || max_chunks < 1) return CSI_ERR_INVALID_ARGUMENT;
    csi_context tmp;
    tmp.Nx = Nx; tmp.Ny = Ny; tmp.Hx = Hx; tmp.Hy = Hy;
    // the launch loop of the peer transport sees connected sides as periodic ones (PeerView)
    tmp.g.xlo = tmp.g.xhi = tmp.g.ylo = tmp.g.yhi = SIDE_PERIODIC;
    tmp.coef.uniform = 1; tmp.metric_kind = CSI_METRIC_UNIFORM;
    tmp.geom_peer = 1;
    for (int d = 0; d < 8; ++d) tmp.peer.sync_rank[d] = -1;
    if (peer_south) tmp.peer.sync_rank[2] = 0;
    if (peer_north) tmp.peer.sync_rank[3] = 0;
    (void)cus;
    memset(out8, 0, 8 * sizeof(int32_t));
    if (!pair_supported(&tmp)) return CSI_OK;
    const Range dec = v_stress_range(&tmp, pair_side_v(&tmp, 2, 2));
    const FusedGeom G = pair_geom(&tmp, dec);
    const PeerSets ps = peer_wait_counts(&tmp, G);
    out8[0] = 1; out8[1] = G.nstrips; out8[2] = G.nchunks; out8[3] = G.rows; out8[4] = G.elo; out8[5] = G.ehi; out8[6] = ps.nS; out8[7] = ps.nN;
    for (int q = 0; q < G.nchunks && q < max_chunks; ++q) { int ja, jb; chunk_rows(G, q, &ja, &jb); rows_out[2 * q] = ja; rows_out[2 * q + 1] = jb; }
    return CSI_OK;
}

int32_t csi_profile_substeps(csi_context* c, double dt, int32_t substeps, double* out_ms4) {
    if (!c || !out_ms4) return CSI_ERR_INVALID_ARGUMENT;
    int32_t rc = need_evp(c);
    if (rc) return rc;
    if (substeps < 2 || substeps > 64) return fail(c, CSI_ERR_INVALID_ARGUMENT, "2 <= substeps <= 64");
    EvpDev P = evp_dev(c, dt);
    FastCoef fc = c->coef;
    { const double ie = 1.0 / P.ecc; fc.em2 = ie * ie; fc.ca_dt = 0.5 * (P.ca * dt); fc.hkc = fc.ca_dt * fc.uni[FC_RAZC]; fc.hkf = fc.ca_dt * fc.uni[FC_RAZF]; fc.hk1 = 0.5 * (1.0 - ie * ie); fc.rdt = 1.0 / dt;
      fc.Dmin2 = P.Dmin * P.Dmin; fc.rDmin = 1.0 / P.Dmin;
      fc.amin2 = P.amin * P.amin; fc.amax2 = P.amax * P.amax; fc.ramin = 1.0 / P.amin; fc.ramax = 1.0 / P.amax; }
    const bool fast = c->mode == CSI_MODE_FAST, tiled = is_tiled(c);
    const Range rs = stress_range(c), rv = interior_range(c), ru1 = first_u_range(c), rv1 = first_v_range(c);
    const ImageSpec imu = image_spec(c, CSI_F_U), imv = image_spec(c, CSI_F_V);
    const int uv[2] = {CSI_F_U, CSI_F_V};
    if (fast && c->fusion && fused_supported(P)) {
        // the fused path: one launch per sub-step or per pair (csi_last_launches); bracket the whole run with two events
        if (substeps & 1) ++substeps;                      // even count: the state ends in the caller's arrays
        bool peer = false;
        if ((rc = peer_decide(c, P, substeps, &peer))) return rc;
        if ((rc = peer ? run_fused_peer(c, dt, fc, substeps, 1) : run_fused(c, P, fc, substeps, 1))) return rc;
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        float t = 0.f;
        HIP_TRY(c, hipEventElapsedTime(&t, c->ev0, c->ev1));
        out_ms4[0] = t / (c->last_launches > 0 ? c->last_launches : substeps); out_ms4[1] = out_ms4[2] = out_ms4[3] = 0.0;
        return CSI_OK;
    }
    std::vector<hipEvent_t> ev((size_t)substeps * 4 + 1);
    for (auto& e : ev) HIP_TRY(c, hipEventCreate(&e));
    if (tiled && (rc = exchange(c, uv, 2, 2))) return rc;    // sizes the buffers outside the timed part
    size_t k = 0;
    HIP_TRY(c, hipEventRecord(ev[k++], c->stream));
    for (int s = 1; s <= substeps; ++s) {
        if (fast) launch_fast_stress(P, rs, fc, c->stream);
        else { launch_strict_visc(P, rs, c->stream); launch_strict_stress(P, rs, c->stream); }
        HIP_TRY(c, hipEventRecord(ev[k++], c->stream));
        // u then v on every sub-step here (the order only permutes which kernel has the ring range)
        if (fast) launch_fast_ustep(P, ru1, imu, fc, c->stream); else launch_strict_ustep(P, ru1, imu, c->stream);
        HIP_TRY(c, hipEventRecord(ev[k++], c->stream));
        if (fast) launch_fast_vstep(P, rv, imv, fc, c->stream); else launch_strict_vstep(P, rv, imv, c->stream);
        HIP_TRY(c, hipEventRecord(ev[k++], c->stream));
        if (tiled && (rc = exchange(c, uv, 2, 2))) return rc;
        HIP_TRY(c, hipEventRecord(ev[k++], c->stream));
        (void)rv1;
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    double acc[4] = {0, 0, 0, 0};
    for (int s = 0; s < substeps; ++s)
        for (int q = 0; q < 4; ++q) {
            float t = 0.f;
            HIP_TRY(c, hipEventElapsedTime(&t, ev[(size_t)s * 4 + q], ev[(size_t)s * 4 + q + 1]));
            acc[q] += t;
        }
    for (int q = 0; q < 4; ++q) out_ms4[q] = acc[q] / substeps;
    for (auto& e : ev) hipEventDestroy(e);
    return CSI_OK;
}

int32_t csi_last_subcycle_ms(csi_context* c, double* ms) {
    if (!c || !ms) return CSI_ERR_INVALID_ARGUMENT;
    if (!c->timed) return fail(c, CSI_ERR_NOT_BOUND, "no sub-cycle has been timed yet");
    float t = 0.f;
    HIP_TRY(c, hipEventSynchronize(c->ev1));
    HIP_TRY(c, hipEventElapsedTime(&t, c->ev0, c->ev1));
    *ms = (double)t;
    return CSI_OK;
}

int32_t csi_subcycle_stats_begin(csi_context* c) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    for (hipEvent_t e : c->stats.ev) hipEventDestroy(e);
    c->stats.ev.clear(); c->stats.launches.clear();
    c->stats.on = true;
    return CSI_OK;
}
int32_t csi_subcycle_stats_end(csi_context* c, double* total_ms, int32_t* cycles, int32_t* launches) {
    if (!c || !total_ms || !cycles || !launches) return CSI_ERR_INVALID_ARGUMENT;
    c->stats.on = false;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    double sum = 0.0; int n = 0, nl = 0;
    for (size_t k = 0; k + 1 < c->stats.ev.size() && k / 2 < c->stats.launches.size(); k += 2) {
        float t = 0.f;
        HIP_TRY(c, hipEventElapsedTime(&t, c->stats.ev[k], c->stats.ev[k + 1]));
        sum += t; ++n; nl += c->stats.launches[k / 2];
    }
    for (hipEvent_t e : c->stats.ev) hipEventDestroy(e);
    c->stats.ev.clear(); c->stats.launches.clear();
    *total_ms = sum; *cycles = n; *launches = nl;
    return CSI_OK;
}

int32_t csi_last_path(csi_context* c, int32_t* fused, int32_t* exchange_interval, int32_t* exchanges) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    if (fused) *fused = c->last_fused;
    if (exchange_interval) *exchange_interval = c->last_k;
    if (exchanges) *exchanges = c->last_exchanges;
    return CSI_OK;
}

int32_t csi_last_launches(csi_context* c, int32_t* launches, int32_t* substeps) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    if (launches) *launches = c->last_launches;
    if (substeps) *substeps = c->last_substeps;
    return CSI_OK;
}

int32_t csi_launches_per_substep(csi_context* c, int32_t* n) {
    if (!c || !n) return CSI_ERR_INVALID_ARGUMENT;
    *n = c->launches_per_substep;
    return CSI_OK;
}

}  // extern "C"
