// csi_abi.hip -- the C ABI of libcsi_hip.so (include/csi.h) and the HIP launch loops.
//
// The launch loop here replaces the host side of the reference's
//   time_step_momentum!   SeaIceDynamics/split_explicit_momentum_equations.jl:103-195
//   rk_substep! / cache_current_fields! / dynamic_time_step!   sea_ice_rk_substep.jl:29-152
//   time_step!(::FESeaIceModel)   sea_ice_fe_step.jl:13-34
// All work is ordered on the context's stream; nothing here synchronises with the host
// except csi_sync / csi_context_destroy.
#include "../../include/csi.h"
#include "csi_dev.h"
#include "csi_kernels.h"
#include "csi_hostgroup.h"
#include "csi_comm.h"

#include <rccl/rccl.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <vector>

using namespace csi;

namespace {

struct Bound {
    double* p = nullptr;
    int64_t ld = 0;
    int ni = 0, nj = 0;
};

// (x, y) location of every field slot
const int kLoc[CSI_F_COUNT][2] = {
    {LOC_F, LOC_C}, {LOC_C, LOC_F}, {LOC_C, LOC_C}, {LOC_C, LOC_C},   // U V H A
    {LOC_C, LOC_C}, {LOC_C, LOC_C}, {LOC_F, LOC_F},                   // S11 S22 S12
    {LOC_F, LOC_C}, {LOC_C, LOC_F}, {LOC_C, LOC_C}, {LOC_C, LOC_C}, {LOC_C, LOC_C},  // UN VN P ALPHA DELTA
    {LOC_F, LOC_F}, {LOC_C, LOC_C},                                   // ZETA_F ZETA_C
    {LOC_C, LOC_C}, {LOC_C, LOC_C}, {LOC_C, LOC_C}, {LOC_C, LOC_C},   // GH GA HM AM
    {LOC_F, LOC_C}, {LOC_C, LOC_F},                                   // UM VM
    {LOC_F, LOC_C}, {LOC_C, LOC_F}, {LOC_F, LOC_C}, {LOC_C, LOC_F},   // TOP_U TOP_V BOT_U BOT_V
    {LOC_C, LOC_C},                                                   // MASS_FLUX
    {LOC_C, LOC_C}, {LOC_C, LOC_C}, {LOC_C, LOC_C},                   // HS GHS HSM
    {LOC_C, LOC_C}, {LOC_C, LOC_C}, {LOC_C, LOC_C}, {LOC_C, LOC_C},   // MASS_FLUX_SNOW SNOWFALL_INTERCEPTED TU TUS
    {LOC_F, LOC_C}, {LOC_C, LOC_F}};                                  // FORCING_U FORCING_V
const char* kName[CSI_F_COUNT] = {"u", "v", "h", "aice", "sigma11", "sigma22", "sigma12", "un", "vn", "P", "alpha",
                                  "Delta", "zeta_f", "zeta_c", "Gh", "Gaice", "h-", "aice-", "u-", "v-",
                                  "top_u", "top_v", "bottom_u", "bottom_v", "mass_flux",
                                  "hs", "Ghs", "hs-", "mass_flux_snow", "intercepted_snowfall", "Tu", "Tu_snow", "forcing_u", "forcing_v"};

std::string g_create_error;

}  // namespace

// ---- in-process tile group (csi_local_group_create / csi_comm_init_local) --------------------------------------------------
// Several contexts of ONE process, one host thread each, exchange halos through device-to-device copies: what RCCL's grouped
// ncclSend / ncclRecv do between processes, with the same matching rule (messages between a pair of ranks match in the order
// they were posted).  Host-synchronous -- a sender waits for its pack kernel before it posts, a receiver for its copies before
// it acknowledges -- because it exists for correctness runs of real decompositions on one GPU (RCCL refuses two ranks on one
// device), not for speed.  The peer halo transport on such a group addresses the neighbours' arrays directly.
struct csi_local_group {
    struct Msg { const double* ptr; size_t count; };
    int world = 0;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::deque<Msg>> box;              // [src * world + dst]
    std::vector<long> posted, consumed;            // per sender: messages posted / copied out of its send buffer
    // collectives (all ranks call them in the same order)
    std::vector<std::vector<uint8_t>> payload;
    long arrived = 0, generation = 0;
    int joined = 0;
};

struct csi_context {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    bool grid_set = false, evp_set = false;
    int Nx = 0, Ny = 0, Hx = 0, Hy = 0, topo_x = 0, topo_y = 0, metric_kind = 0;
    GridDev g{};
    double* dev_metrics = nullptr;   // 8 vectors of length Ny + 2Hy + 1 (PER_J) or 12 planes (FULL)
    double* dev_coef = nullptr;      // FAST per-row stencil coefficients [Ny + 2Hy + 1][FC_COUNT]
    double* dev_coef2 = nullptr;     // FAST per-point stencil coefficients of a CSI_METRIC_FULL grid, C2_COUNT planes
    FastCoef coef{};
    std::vector<double> coef_host;       // host copy of the per-row table built from PER_J metrics (empty: uniform metrics)
    std::vector<double> fcor_rows[2];    // csi_coriolis_rows_set: f per row at u / v points (empty: FPlane scalar)
    double* dev_fcor = nullptr;          // the same on the device (STRICT kernels), 2 x (Ny + 2Hy + 1)
    double* dev_fcor2 = nullptr;         // csi_coriolis_points_set: two planes (u points, v points) of ni x nj
    long fcor2_ld = 0, fcor2_plane = 0;
    bool cor_dirty = true;               // Coriolis columns of the FAST table need (re)building
    double cor_synced = 0.0;             // FPlane value they were built with
    Bound f[CSI_F_COUNT];
    csi_evp_params evp{};
    csi_stress stress[2]{};
    int mode = CSI_MODE_STRICT;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    int launches_per_substep = 0;
    // multi-GPU tiles
    TileInfo tile;
    ncclComm_t comm = nullptr;
    csi_local_group* local = nullptr;      // in-process tile group instead of an RCCL communicator (csi_comm_init_local)
    HostGroup* hostg = nullptr;            // host-channel group of PROCESSES (shared memory + HIP IPC, csi_comm_init_host): RCCL-free runs of several ranks on one GPU
    int world = 1, rank = 0;
    double *sendbuf = nullptr, *recvbuf = nullptr;
    size_t buf_cap = 0;   // elements per buffer
    int last_exchanges = 0, last_k = 1;
    // Halo transport of the two-sub-steps kernel on tiles.  "peer" (default where it can be set up): the neighbouring tiles'
    // arrays are mapped into this process (HIP IPC; xGMI peer access) and a connected side behaves like a periodic one whose halo
    // lives on another GPU -- the owner's stores write the halo images straight into the neighbour's arrays, and flags in
    // device memory order the launches of neighbouring ranks (evp_fused2.hip): no pack, no RCCL kernel, no unpack, no widened
    // halo.  "rccl": ncclSend / ncclRecv of width-2k strips every k sub-steps (the fallback, and what every other path uses).
    struct Peer {
        static constexpr int NARR = 14;      // u, v, sigma11, sigma22, sigma12 (caller's), the same five (library's ping-pong copies), alpha, zeta_c, zeta_f, Delta
        static constexpr int SLOTS = kPeerSlots;   // flag slots per direction (the last one is the block's abort word)
        int want = 1;                        // csi_set_halo_transport: 1 peer where possible, 0 RCCL only
        int dld[8][2] = {};                  // per direction x {Center, Face in x}: the neighbour's row stride minus this tile's, bytes
        int nbr_wait[8] = {};                // flags to wait for per direction: the size of the NEIGHBOUR's opposite set (its own geometry)
        int ny_below = 0;                    // rows of the tile below (all tiles of a decomposition have this tile's UNCUT height)
        bool ready = false, failed = false;  // set up (collectively) / cannot be set up (stays on RCCL)
        const void* sig[NARR] = {};          // the local arrays the set-up was made for
        int img_rank[8], sync_rank[8];       // per direction: the rank whose arrays receive this tile's images there; the neighbour to wait for (-1: none)
        void* arr[8][NARR] = {};             // that rank's arrays as this process addresses them
        unsigned long long* nbr_slots[8] = {};   // its flag array
        unsigned long long* slots = nullptr; // this rank's flag array: 8 directions x SLOTS
        unsigned* err = nullptr;             // device word set by a wait that timed out
        unsigned* err_host = nullptr;        // pinned copy, refreshed after every sub-cycle
        unsigned long long seq = 0;          // launches of the flag protocol so far (the same number on every rank)
        std::vector<void*> opened;           // IPC mappings
        uint8_t* xbuf = nullptr;             // device staging of the set-up's all-gather
        int last = 0;                        // the last sub-cycle used the peer transport
        int tier = 0;                        // protocol tier (csi_set_peer_tier; FI_PTIER of the kernel tables)
        bool local_queues_ok = true;         // in-process tile group: GPU_MAX_HW_QUEUES > tiles (csi_comm_init_local)
        size_t xbuf_cap = 0;                 // bytes of xbuf
    } peer;
    ExPlan pending_rp;                   // the receive plan of an exchange that has been begun
    // fused sub-step kernel: ping-pong copies of u, v, sigma11, sigma22, sigma12
    FusedTable* dev_tables = nullptr;   // uniform-input tables of the fused kernel
    // pinned staging ring for their upload: the host never waits for the stream (a slot is reused after its own copy
    // has completed, four sub-cycles later)
    static constexpr int kRing = 4;
    FusedTable* host_ring = nullptr;
    hipEvent_t ring_ev[kRing] = {nullptr, nullptr, nullptr, nullptr};
    bool ring_used[kRing] = {false, false, false, false};
    unsigned ring_pos = 0;
    double* alt[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    double* adv_buf[4] = {nullptr, nullptr, nullptr, nullptr};      // RK stages of an advection-only model in one launch each: (h, a) x 2 rotating copies
    size_t adv_elems[4] = {0, 0, 0, 0};
    // north fold (FoldBand): the band's own copies of u, v, sigma and of the four diagnostics, its stream and the two events
    // that order it against the pair launches
    double* band[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t band_elems[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    hipStream_t band_stream = nullptr;
    hipEvent_t band_ev_pair = nullptr, band_ev_band = nullptr;
    double* fbar[2] = {nullptr, nullptr};   // ocean ubar at v points, vbar at u points (array-valued bottom drag)
    double* fbar_top[2] = {nullptr, nullptr};   // the same of the air velocities (array-valued wind drag)
    size_t fbar_top_elems[2] = {0, 0};
    double* fd[2] = {nullptr, nullptr};     // free-drift velocities at u / v points (StressBalanceFreeDrift)
    double* xd[2] = {nullptr, nullptr};     // stress divergence of the immersed flux boundary conditions at u / v points (two-sub-steps kernel)
    size_t xd_elems[2] = {0, 0};
    size_t fd_elems[2] = {0, 0};
    int free_drift = 0;                     // csi_free_drift_set
    size_t fbar_elems[2] = {0, 0};
    size_t alt_elems[5] = {0, 0, 0, 0, 0};
    bool slab_set = false;   // thermodynamic step inside csi_time_step_fe / _rk3
    SlabDev slab{};
    int vel_bc_on[2][2] = {{0, 0}, {0, 0}};          // csi_velocity_bc_set: [u | v][low | high] ValueBoundaryCondition
    double vel_bc_value[2][2] = {{0, 0}, {0, 0}};
    bool snow_set = false;   // layered (snow + ice) step instead of the bare-ice one
    SnowDev snow{};
    int trios = 0;        // three sub-steps per launch where the configuration allows it: csi_set_fusion level 3 (not the default: measured
                          // 5 % slower than pairs at 2048^2)
    int last_trios = 0;   // launches of the last sub-cycle that did three sub-steps
    int fusion = 1;       // 1: use the fused sub-step kernel when the configuration allows it
    int pairing = 1;      // 1: two sub-steps per launch where supported (csi_set_fusion level 2)
    int last_launches = 0, last_substeps = 0, last_used_pairs = 0;   // kernel launches / sub-steps of the last fused sub-cycle
    int last_fused = 0;
    double ibc[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};   // csi_immersed_flux_bc_set: [u | v][west, east, south, north]
    int exch_k = 0;       // sub-steps per halo exchange (0 = auto: the largest k with 2k <= halo, at most 4)
    int geom_band = 0;    // the pair launches being laid out run beside a fold band (FoldCut / PeerView of a fold tile): see pair_geom
    // tuning aids (A/B runs), read from the environment ONCE, when the context is created; -1 = not set
    struct Tuning { int fused_rows = -1, pair_tiles = -1, pair_minrows = -1, pair_rows = -1, pair_common = -1, trio_tiles = -1, peer_kernel = -1; } tune;
};

namespace {

int32_t fail(csi_context* c, int32_t code, const std::string& msg) {
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}
#define HIP_TRY(c, expr)                                                                        \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(c, CSI_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));     \
    } while (0)

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
        std::vector<double> host((size_t)FC_COUNT * n);
        for (int w = 0; w < FC_COUNT; ++w)
            for (int t = 0; t < n; ++t)
                host[(size_t)t * FC_COUNT + w] = metrics_uniform ? c->coef.uni[w] : c->coef_host[(size_t)w * n + t];
        for (int t = 0; t < n; ++t) {
            host[(size_t)t * FC_COUNT + FC_FU] = rows ? c->fcor_rows[0][t] : f0;
            host[(size_t)t * FC_COUNT + FC_FV] = rows ? c->fcor_rows[1][t] : f0;
        }
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (!c->dev_coef) HIP_TRY(c, hipMalloc((void**)&c->dev_coef, sizeof(double) * host.size()));
        HIP_TRY(c, hipMemcpy(c->dev_coef, host.data(), sizeof(double) * host.size(), hipMemcpyHostToDevice));
        c->coef.vec = c->dev_coef + (size_t)(c->Hy - 1) * FC_COUNT;      // so that vec[j * stride + which] is row j
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
Range stress_range(const csi_context* c, int V = 2) {
    const GridDev& g = c->g;
    return Range{g.xlo == SIDE_CONNECTED ? 2 - V : -c->Hx + 2, g.xhi == SIDE_CONNECTED ? c->Nx + V - 1 : c->Nx + c->Hx - 1,
                 g.ylo == SIDE_CONNECTED ? 2 - V : -c->Hy + 2, g.yhi == SIDE_CONNECTED ? c->Ny + V - 1 : c->Ny + c->Hy - 1};
}
Range first_u_range(const csi_context* c, int V = 2) {
    const GridDev& g = c->g;
    return Range{g.xlo == SIDE_CONNECTED ? 3 - V : 1, g.xhi == SIDE_CONNECTED ? c->Nx + V - 1 : c->Nx,
                 g.ylo == SIDE_CONNECTED ? 2 - V : 1, g.yhi == SIDE_CONNECTED ? c->Ny + V - 2 : c->Ny};
}
Range first_v_range(const csi_context* c, int V = 2) {
    const GridDev& g = c->g;
    return Range{g.xlo == SIDE_CONNECTED ? 2 - V : 1, g.xhi == SIDE_CONNECTED ? c->Nx + V - 2 : c->Nx,
                 g.ylo == SIDE_CONNECTED ? 3 - V : 1, g.yhi == SIDE_CONNECTED ? c->Ny + V - 1 : c->Ny};
}
Range second_range(const csi_context* c, int V = 2) {
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

#define NCCL_TRY(c, expr)                                                                       \
    do {                                                                                        \
        ncclResult_t r_ = (expr);                                                               \
        if (r_ != ncclSuccess)                                                                  \
            return fail(c, CSI_ERR_COMM, std::string(#expr) + ": " + ncclGetErrorString(r_));   \
    } while (0)

// Exchange `W` halo layers of the given fields with the neighbouring tiles (no-op on an untiled grid).
int32_t exchange_refs(csi_context* c, const FRef* fr, int nf, int W);

int32_t exchange(csi_context* c, const int* fids, int nf, int W) {
    if (!is_tiled(c)) return CSI_OK;
    if (nf > MAX_EX_FIELDS) return fail(c, CSI_ERR_INVALID_ARGUMENT, "too many fields in one exchange");
    FRef fr[MAX_EX_FIELDS];
    for (int k = 0; k < nf; ++k) {
        if (!c->f[fids[k]].p) return fail(c, CSI_ERR_NOT_BOUND, std::string("field not bound: ") + kName[fids[k]]);
        fr[k] = ref_of(c, fids[k]);
    }
    return exchange_refs(c, fr, nf, W);
}

bool has_comm(const csi_context* c) { return c->comm != nullptr || c->local != nullptr || c->hostg != nullptr; }

constexpr int kLocalTimeoutSeconds = 120;
// all ranks of the group: rank r's `nb` bytes end up in out[r * nb ...] everywhere
int32_t local_allgather(csi_context* c, const void* mine, size_t nb, std::vector<uint8_t>& out) {
    csi_local_group* G = c->local;
    std::unique_lock<std::mutex> lk(G->mu);
    const long gen = G->generation;
    G->payload[c->rank].assign((const uint8_t*)mine, (const uint8_t*)mine + nb);
    if (++G->arrived == G->world) {
        // the last one in publishes: the payloads stay untouched until everybody of the NEXT collective has arrived
        G->arrived = 0; ++G->generation;
        G->cv.notify_all();
    } else if (!G->cv.wait_for(lk, std::chrono::seconds(kLocalTimeoutSeconds), [&] { return G->generation != gen; })) {
        return fail(c, CSI_ERR_COMM, "in-process tile group: a collective timed out (a rank did not arrive)");
    }
    out.resize(nb * (size_t)G->world);
    for (int r = 0; r < G->world; ++r) {
        if (G->payload[r].size() != nb) return fail(c, CSI_ERR_COMM, "in-process tile group: payload sizes differ");
        memcpy(out.data() + (size_t)r * nb, G->payload[r].data(), nb);
    }
    // second phase: nobody overwrites its payload before all have read
    const long gen2 = G->generation;
    if (++G->arrived == G->world) { G->arrived = 0; ++G->generation; G->cv.notify_all(); }
    else if (!G->cv.wait_for(lk, std::chrono::seconds(kLocalTimeoutSeconds), [&] { return G->generation != gen2; }))
        return fail(c, CSI_ERR_COMM, "in-process tile group: a collective timed out (a rank did not arrive)");
    return CSI_OK;
}
int32_t local_allreduce_min(csi_context* c, int* v) {
    std::vector<uint8_t> all;
    int32_t rc;
    if ((rc = local_allgather(c, v, sizeof(int), all))) return rc;
    for (int r = 0; r < c->world; ++r) { int x; memcpy(&x, all.data() + (size_t)r * sizeof(int), sizeof(int)); if (x < *v) *v = x; }
    return CSI_OK;
}
// before the send buffer is packed again: every message posted from it has been copied out
int32_t local_wait_consumed(csi_context* c) {
    csi_local_group* G = c->local;
    std::unique_lock<std::mutex> lk(G->mu);
    if (!G->cv.wait_for(lk, std::chrono::seconds(kLocalTimeoutSeconds), [&] { return G->consumed[c->rank] == G->posted[c->rank]; }))
        return fail(c, CSI_ERR_COMM, "in-process tile group: a neighbour never received this rank's previous halo message");
    return CSI_OK;
}
// the grouped send / receive of exchange_refs
int32_t local_sendrecv(csi_context* c, const long* soff, const long* scnt, const int* speer, const long* roff, const long* rcnt, const int* rpeer) {
    csi_local_group* G = c->local;
    HIP_TRY(c, hipStreamSynchronize(c->stream));                         // the pack kernel has filled the send buffer
    int from[8], nfrom = 0;
    {
        std::unique_lock<std::mutex> lk(G->mu);
        for (int k = 0; k < 8; ++k)
            if (speer[k] >= 0 && scnt[k] > 0) {
                G->box[(size_t)c->rank * G->world + speer[k]].push_back(csi_local_group::Msg{c->sendbuf + soff[k], (size_t)scnt[k]});
                ++G->posted[c->rank];
            }
        G->cv.notify_all();
        for (int k = 0; k < 8; ++k)
            if (rpeer[k] >= 0 && rcnt[k] > 0) {
                std::deque<csi_local_group::Msg>& q = G->box[(size_t)rpeer[k] * G->world + c->rank];
                if (!G->cv.wait_for(lk, std::chrono::seconds(kLocalTimeoutSeconds), [&] { return !q.empty(); }))
                    return fail(c, CSI_ERR_COMM, "in-process tile group: a halo message never arrived (a rank fell behind or died)");
                const csi_local_group::Msg m = q.front();
                q.pop_front();
                if (m.count != (size_t)rcnt[k]) return fail(c, CSI_ERR_COMM, "in-process tile group: halo message of unexpected size (send / receive plans do not match)");
                HIP_TRY(c, hipMemcpyAsync(c->recvbuf + roff[k], m.ptr, m.count * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
                from[nfrom++] = rpeer[k];
            }
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));                         // the copies are done: the senders may repack
    {
        std::unique_lock<std::mutex> lk(G->mu);
        for (int q = 0; q < nfrom; ++q) ++G->consumed[from[q]];
        G->cv.notify_all();
    }
    return CSI_OK;
}

// the same on explicit array references (the fused path exchanges whichever ping-pong buffer is current): pack, one grouped
// send / receive, unpack, all on the context stream
int32_t exchange_refs(csi_context* c, const FRef* fr, int nf, int W) {
    if (!is_tiled(c)) return CSI_OK;
    if (!c->tile.set) return fail(c, CSI_ERR_NOT_BOUND, "connected topology but csi_tile_set has not been called");
    if (!has_comm(c)) return fail(c, CSI_ERR_NOT_BOUND, "connected topology but csi_comm_init has not been called");
    if (nf > MAX_EX_FIELDS) return fail(c, CSI_ERR_INVALID_ARGUMENT, "too many fields in one exchange");
    if (W < 1 || W > c->Hx || W > c->Hy || W > c->Nx || W > c->Ny) return fail(c, CSI_ERR_INVALID_ARGUMENT, "exchange width out of range");
    ExPlan sp;
    long soff[8], scnt[8], roff[8], rcnt[8];
    int speer[8], rpeer[8];
    build_plan(c->g, c->tile, fr, nf, W, 0, sp, soff, scnt, speer);
    build_plan(c->g, c->tile, fr, nf, W, 1, c->pending_rp, roff, rcnt, rpeer);
    const size_t need_elems = (size_t)(sp.total > c->pending_rp.total ? sp.total : c->pending_rp.total);
    int32_t lrc;
    if (c->local && (lrc = local_wait_consumed(c))) return lrc;      // (before the send buffer is repacked -- or freed)
    if (c->hostg && !hostgroup_wait_consumed(c->hostg, &c->err)) return CSI_ERR_COMM;
    if (need_elems > c->buf_cap) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (c->sendbuf) hipFree(c->sendbuf);
        if (c->recvbuf) hipFree(c->recvbuf);
        c->sendbuf = c->recvbuf = nullptr;
        const size_t cap = need_elems * 2;
        HIP_TRY(c, hipMalloc((void**)&c->sendbuf, cap * sizeof(double)));
        HIP_TRY(c, hipMalloc((void**)&c->recvbuf, cap * sizeof(double)));
        c->buf_cap = cap;
        if (c->hostg && !hostgroup_set_sendbuf(c->hostg, c->sendbuf, cap * sizeof(double), &c->err)) return CSI_ERR_COMM;
    }
    launch_pack(sp, c->sendbuf, 0, c->stream);
    if (c->local) {
        if ((lrc = local_sendrecv(c, soff, scnt, speer, roff, rcnt, rpeer))) return lrc;
        launch_pack(c->pending_rp, c->recvbuf, 1, c->stream);
        HIP_TRY(c, hipGetLastError());
        return CSI_OK;
    }
    if (c->hostg) {
        if (!hostgroup_sendrecv(c->hostg, c->stream, c->recvbuf, soff, scnt, speer, roff, rcnt, rpeer, &c->err)) return CSI_ERR_COMM;
        launch_pack(c->pending_rp, c->recvbuf, 1, c->stream);
        HIP_TRY(c, hipGetLastError());
        return CSI_OK;
    }
    NCCL_TRY(c, ncclGroupStart());
    for (int k = 0; k < 8; ++k)
        if (speer[k] >= 0 && scnt[k] > 0) NCCL_TRY(c, ncclSend(c->sendbuf + soff[k], (size_t)scnt[k], ncclDouble, speer[k], c->comm, c->stream));
    for (int k = 0; k < 8; ++k)
        if (rpeer[k] >= 0 && rcnt[k] > 0) NCCL_TRY(c, ncclRecv(c->recvbuf + roff[k], (size_t)rcnt[k], ncclDouble, rpeer[k], c->comm, c->stream));
    NCCL_TRY(c, ncclGroupEnd());
    launch_pack(c->pending_rp, c->recvbuf, 1, c->stream);
    HIP_TRY(c, hipGetLastError());
    return CSI_OK;
}

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
const int kPing[5] = {CSI_F_U, CSI_F_V, CSI_F_S11, CSI_F_S22, CSI_F_S12};

FRef alt_ref(const csi_context* c, int k) {
    FRef r;
    const Bound& b = c->f[kPing[k]];
    r.p = c->alt[k] + (c->Hx - 1) + (int64_t)(c->Hy - 1) * b.ld;
    r.ld = (int)b.ld;
    return r;
}

struct FusedGeom { Range rs; int nstrips, nchunks, rows; };

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

constexpr int kMaxExchangeInterval = 16;
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
struct SideV { int xlo, xhi, ylo, yhi; };
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
FusedGeom pair_geom(const csi_context* c, const Range& dec) {
    FusedGeom G;
    G.rs = dec;
    // (56-column strip) x (rows) tiles, one workgroup of two waves (producer: first sub-step, consumer: second) per tile.
    // The kernel is compiled for 3 waves per SIMD (<= 168 VGPRs): 256 CUs x 6 workgroups = 1536 resident tiles.  Exactly
    // one round of tiles, as tall as possible: every SIMD keeps its waves from start to end and each tile pays its 6 ring
    // rows once.
    const int width = dec.i1 - dec.i0 + 1, height = dec.j1 - dec.j0 + 1;
    G.nstrips = (width + 55) / 56;
    // (per-point coefficients: the kernel is compiled for 2 waves per SIMD -> 1024 resident tiles; measured at 2048^2:
    // 1024 tiles 22.9, 1536 tiles 18.9, 768 tiles 20.8 G cell-updates/s)
    int target = c->metric_kind == CSI_METRIC_FULL ? 1024 : 1536;
    // Beside a fold band (its own stream: eight small launches per pair of sub-steps) the pair launch leaves a third of the wave
    // slots free, so that the band runs DURING the launch instead of in its tail -- a launch that fills every slot lets only the
    // band's first kernel in (round 3: 123 + 31 us per pair of sub-steps at 2048^2).  Measured at 2048^2, round 4: fold on uniform
    // metrics 1536 tiles 53.1, 1280 52.6, 1024 58.7, 896 56.1 G; tripolar-like (per-point metrics) 1024 tiles 20.4, 896 21.9, 768 20.5
    if (c->geom_band) target = c->metric_kind == CSI_METRIC_FULL ? 896 : 1024;
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
    return G;
}

// ---- peer halo transport (csi_context::Peer) -------------------------------------------------------------------------------
// Directions: 0 W, 1 E, 2 S, 3 N, 4 SW, 5 SE, 6 NW, 7 NE (the order evp_fused2.hip's D_* and the table's FP_IMG0 rows use).
const int kPeerDx[8] = {-1, 1, 0, 0, -1, 1, -1, 1}, kPeerDy[8] = {0, 0, -1, 1, -1, -1, 1, 1};
const int kPeerOpp[8] = {1, 0, 3, 2, 7, 6, 5, 4};

// Which tiles of a pair launch touch the halo beyond each side -- read it, store images of their own cells into the neighbour's,
// or share a 128-byte line with it -- and therefore wait for / signal that neighbour: the first nW / last nE strips, the first
// nS / last nN chunks.  size[d]: tiles in this rank's set of direction d; n[d]: slots to wait for from the neighbour in direction
// d = the size of ITS set towards this rank (tiles of one decomposition have the same shape, hence the same sets).
struct PeerSets { int nW, nE, nS, nN, size[8], n[8]; };
PeerSets peer_wait_counts(const csi_context* c, const FusedGeom& G) {
    PeerSets ps{};
    constexpr int P_LO = 4, P_W = 56;                       // evp_pair_stage.h: a strip is 64 lanes wide and owns lanes 4 .. 59
    for (int st = 0; st < G.nstrips; ++st) {
        const int i0s = G.rs.i0 - P_LO + st * P_W;
        if (i0s <= c->Hx + 16) ++ps.nW;
        if (i0s + 63 + 16 > c->Nx - c->Hx) ++ps.nE;
    }
    for (int q = 0; q < G.nchunks; ++q) {
        const int ja = G.rs.j0 + q * G.rows, jb = std::min(ja + G.rows - 1, G.rs.j1);
        if (ja <= c->Hy + 4) ++ps.nS;
        if (jb + 4 > c->Ny - c->Hy) ++ps.nN;
    }
    const int sz[8] = {ps.nW * G.nchunks, ps.nE * G.nchunks, ps.nS * G.nstrips, ps.nN * G.nstrips,
                       ps.nW * ps.nS, ps.nE * ps.nS, ps.nW * ps.nN, ps.nE * ps.nN};
    for (int d = 0; d < 8; ++d) ps.size[d] = sz[d];
    for (int d = 0; d < 8; ++d) ps.n[d] = sz[kPeerOpp[d]];
    return ps;
}

// the 14 local arrays a neighbour stores images into, in Peer::arr order
void peer_local_arrays(const csi_context* c, const void* out[csi_context::Peer::NARR]) {
    for (int q = 0; q < 5; ++q) { out[q] = c->f[kPing[q]].p; out[5 + q] = c->alt[q]; }
    out[10] = c->f[CSI_F_ALPHA].p; out[11] = c->f[CSI_F_ZETA_C].p; out[12] = c->f[CSI_F_ZETA_F].p; out[13] = c->f[CSI_F_DELTA].p;
}

void peer_release(csi_context* c) {
    for (void* m : c->peer.opened) hipIpcCloseMemHandle(m);
    c->peer.opened.clear();
    c->peer.ready = false;
}

struct PeerRec {                 // what a rank tells the others about one of its buffers
    hipIpcMemHandle_t handle;    // of the allocation that holds it
    uint64_t offset;             // of the buffer inside that allocation
    int64_t ld;                  // leading dimension (images use the sender's strides: they must agree)
    int32_t ok, pad;
    uint64_t local_ptr;          // in-process tile group: the buffer itself (same address space)
    int32_t set_size[8];         // (record 0) tiles of this rank's launches in each direction's set: what the neighbour waits for
};
constexpr int kPeerRecs = csi_context::Peer::NARR + 1;      // + the flag array

// The grid descriptor the launches of the peer transport see: connected sides count as periodic ones; a fold tile is cut below
// its three-kernel band (FoldBand), whose side then counts as "connected" (halo rows = interior rows of the same arrays).
struct PeerView {
    csi_context* c; GridDev g; int Ny;
    explicit PeerView(csi_context* cc) : c(cc), g(cc->g), Ny(cc->Ny) {
        for (int* side : {&c->g.xlo, &c->g.xhi, &c->g.ylo, &c->g.yhi}) if (*side == SIDE_CONNECTED) *side = SIDE_PERIODIC;
        if (c->g.yhi == SIDE_FOLD) { const int M = c->Ny - c->Hy - 4; c->Ny = M; c->g.Ny = M; c->g.yhi = SIDE_CONNECTED; band = c->geom_band; c->geom_band = 1; }
    }
    int band = -1;
    ~PeerView() { c->g = g; c->Ny = Ny; if (band >= 0) c->geom_band = band; }
};
bool fold_cut_possible(const csi_context* c) {
    const GridDev& g = c->g;
    return g.yhi == SIDE_FOLD && g.xlo == SIDE_PERIODIC && g.xhi == SIDE_PERIODIC && c->Hy >= 4 && c->Ny - c->Hy - 4 >= 2 * c->Hy + 8;
}
// does the two-sub-steps kernel take this tile on the peer transport?  (P: the tile as it is)
bool peer_tile_supported(csi_context* c, const EvpDev& Pfull) {
    if (c->g.yhi != SIDE_FOLD) return pair_supported(c) && pair_forcing_kind(Pfull) >= 0;      // (the tile as it is: connected sides)
    if (!fold_cut_possible(c)) return false;
    const GridDev g = c->g;
    const int Ny = c->Ny, M = c->Ny - c->Hy - 4;
    c->Ny = M; c->g.Ny = M; c->g.yhi = SIDE_CONNECTED;                                        // cut below the band
    EvpDev P = Pfull;
    P.g = c->g;
    const bool ok = pair_supported(c) && pair_forcing_kind(P) >= 0;
    c->g = g; c->Ny = Ny;
    return ok;
}
PeerSets peer_my_sets(csi_context* c) {
    PeerView view(c);
    const Range dec = v_stress_range(c, pair_side_v(c, 2, 2));
    return peer_wait_counts(c, pair_geom(c, dec));
}

// Collective over the context's communicator: every rank publishes IPC handles of its arrays and flags, maps its neighbours'.
// Failure anywhere (no IPC, strides that differ across a side, sets larger than the flag array) makes EVERY rank stay on RCCL.
int32_t peer_setup(csi_context* c, bool local_ok) {
    csi_context::Peer& pr = c->peer;
    HIP_TRY(c, hipSetDevice(c->device));                   // (allocations and IPC mappings below belong to the context's device)
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    peer_release(c);
    const int me = c->tile.ry * c->tile.Rx + c->tile.rx;
    if (!pr.slots) {
        // fine-grained (uncached) device memory where the runtime offers it: the flags are polled while remote ranks write them
        if (hipExtMallocWithFlags((void**)&pr.slots, sizeof(unsigned long long) * 8 * csi_context::Peer::SLOTS, hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            HIP_TRY(c, hipMalloc((void**)&pr.slots, sizeof(unsigned long long) * 8 * csi_context::Peer::SLOTS));
        }
        HIP_TRY(c, hipMalloc((void**)&pr.err, sizeof(unsigned)));
        HIP_TRY(c, hipHostMalloc((void**)&pr.err_host, sizeof(unsigned), hipHostMallocDefault));
    }
    {
        const size_t need_x = sizeof(PeerRec) * kPeerRecs * (size_t)(c->world + 1) + 64;      // (a later csi_comm_init may have a larger world)
        if (need_x > pr.xbuf_cap) {
            if (pr.xbuf) hipFree(pr.xbuf);
            pr.xbuf = nullptr; pr.xbuf_cap = 0;
            HIP_TRY(c, hipMalloc((void**)&pr.xbuf, need_x));
            pr.xbuf_cap = need_x;
        }
    }
    HIP_TRY(c, hipMemset(pr.slots, 0, sizeof(unsigned long long) * 8 * csi_context::Peer::SLOTS));
    HIP_TRY(c, hipMemset(pr.err, 0, sizeof(unsigned)));
    *pr.err_host = 0;
    pr.seq = 0;
    pr.ny_below = c->Ny;
    // neighbours: where this tile's images go (a periodic or wall component keeps the coordinate: wraps / mirrors are local
    // in that direction) and whom to wait for (connected components only)
    for (int d = 0; d < 8; ++d) {
        pr.sync_rank[d] = tile_neighbor(c->tile, kPeerDx[d], kPeerDy[d], c->g.xlo, c->g.xhi, c->g.ylo, c->g.yhi);
        int rx = c->tile.rx, ry = c->tile.ry;
        if (kPeerDx[d] < 0 && c->g.xlo == SIDE_CONNECTED) rx = (rx - 1 + c->tile.Rx) % c->tile.Rx;
        if (kPeerDx[d] > 0 && c->g.xhi == SIDE_CONNECTED) rx = (rx + 1) % c->tile.Rx;
        if (kPeerDy[d] < 0 && c->g.ylo == SIDE_CONNECTED) ry = (ry - 1 + c->tile.Ry) % c->tile.Ry;
        if (kPeerDy[d] > 0 && c->g.yhi == SIDE_CONNECTED) ry = (ry + 1) % c->tile.Ry;
        pr.img_rank[d] = ry * c->tile.Rx + rx;
    }
    const void* local[csi_context::Peer::NARR];
    peer_local_arrays(c, local);
    const int64_t lds[csi_context::Peer::NARR] = {c->f[CSI_F_U].ld, c->f[CSI_F_V].ld, c->f[CSI_F_S11].ld, c->f[CSI_F_S22].ld, c->f[CSI_F_S12].ld,
                                                  c->f[CSI_F_U].ld, c->f[CSI_F_V].ld, c->f[CSI_F_S11].ld, c->f[CSI_F_S22].ld, c->f[CSI_F_S12].ld,
                                                  c->f[CSI_F_ALPHA].ld, c->f[CSI_F_ZETA_C].ld, c->f[CSI_F_ZETA_F].ld, c->f[CSI_F_DELTA].ld};
    std::vector<PeerRec> mine(kPeerRecs), all((size_t)kPeerRecs * c->world);
    int ok = local_ok ? 1 : 0;          // (a rank whose own configuration rules the transport out still takes part: every rank or none)
    for (int q = 0; q < kPeerRecs; ++q) {
        const void* ptr = q < csi_context::Peer::NARR ? local[q] : (const void*)pr.slots;
        PeerRec& r = mine[q];
        memset(&r, 0, sizeof r);
        r.ld = q < csi_context::Peer::NARR ? lds[q] : 0;
        r.local_ptr = (uint64_t)ptr;
        if (q == 0 && local_ok) {
            const PeerSets ps = peer_my_sets(c);
            for (int d = 0; d < 8; ++d) { r.set_size[d] = ps.size[d]; if (ps.size[d] >= csi_context::Peer::SLOTS) ok = 0; }
        }
        if (ok && c->world > 1 && !c->local) {               // (a single rank / an in-process group addresses the arrays directly)
            hipDeviceptr_t base = nullptr; size_t size = 0;
            if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)ptr) != hipSuccess || hipIpcGetMemHandle(&r.handle, base) != hipSuccess) {
                (void)hipGetLastError();
                ok = 0;
            } else {
                r.offset = (uint64_t)((const char*)ptr - (const char*)base);
            }
        }
        r.ok = ok;
    }
    if (c->local) {
        std::vector<uint8_t> bytes;
        int32_t lrc;
        if ((lrc = local_allgather(c, mine.data(), sizeof(PeerRec) * kPeerRecs, bytes))) return lrc;
        memcpy(all.data(), bytes.data(), bytes.size());
    } else if (c->hostg) {
        std::vector<uint8_t> bytes;
        if (!hostgroup_allgather(c->hostg, mine.data(), sizeof(PeerRec) * kPeerRecs, bytes, &c->err)) return CSI_ERR_COMM;
        memcpy(all.data(), bytes.data(), bytes.size());
    } else if (c->world > 1) {
        const size_t nb = sizeof(PeerRec) * kPeerRecs;
        HIP_TRY(c, hipMemcpy(pr.xbuf, mine.data(), nb, hipMemcpyHostToDevice));
        NCCL_TRY(c, ncclAllGather(pr.xbuf, pr.xbuf + nb, nb, ncclUint8, c->comm, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipMemcpy(all.data(), pr.xbuf + nb, nb * c->world, hipMemcpyDeviceToHost));
    } else {
        all = mine;
    }
    // map the neighbours' buffers (one mapping per distinct allocation)
    struct Mapped { int rank; hipIpcMemHandle_t h; void* p; };
    std::vector<Mapped> cache;
    auto resolve = [&](int rank, int q, void** out) -> bool {
        const PeerRec& r = all[(size_t)rank * kPeerRecs + q];
        if (!r.ok) return false;
        if (c->local) { *out = (void*)r.local_ptr; return true; }
        for (const Mapped& m : cache)
            if (m.rank == rank && memcmp(&m.h, &r.handle, sizeof r.handle) == 0) { *out = (char*)m.p + r.offset; return true; }
        void* mp = nullptr;
        if (hipIpcOpenMemHandle(&mp, r.handle, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); return false; }
        cache.push_back(Mapped{rank, r.handle, mp});
        pr.opened.push_back(mp);
        *out = (char*)mp + r.offset;
        return true;
    };
    for (int d = 0; d < 8 && ok; ++d) {
        const int r = pr.img_rank[d];
        for (int q = 0; q < csi_context::Peer::NARR; ++q) {
            if (r == me) { pr.arr[d][q] = const_cast<void*>(local[q]); continue; }
            if (!resolve(r, q, &pr.arr[d][q])) { ok = 0; break; }
        }
        // the neighbour's row strides may differ from this tile's (a Bounded x direction partitioned in x gives the easternmost
        // tile one more column of Face points): one difference per direction and location in x, which the image stores add per
        // parent row -- provided the neighbour's arrays of one location share a stride, as this tile's do
        pr.dld[d][0] = pr.dld[d][1] = 0;
        if (ok && r != me) {
            static const int cls[csi_context::Peer::NARR] = {1, 0, 0, 0, 1, 1, 0, 0, 0, 1, 0, 0, 1, 0};      // Face in x: u, sigma12, zeta_f
            for (int q = 0; q < csi_context::Peer::NARR; ++q) {
                const int64_t diff = all[(size_t)r * kPeerRecs + q].ld - lds[q];
                if (diff != all[(size_t)r * kPeerRecs + (cls[q] ? 0 : 1)].ld - lds[cls[q] ? 0 : 1] || diff < -64 || diff > 64) { ok = 0; break; }
                pr.dld[d][cls[q]] = (int)diff * 8;
            }
        }
        pr.nbr_slots[d] = nullptr;
        pr.nbr_wait[d] = (ok && pr.sync_rank[d] >= 0) ? all[(size_t)pr.sync_rank[d] * kPeerRecs].set_size[kPeerOpp[d]] : 0;
        if (ok && pr.sync_rank[d] >= 0) {
            void* sp = pr.slots;
            if (pr.sync_rank[d] != me && !resolve(pr.sync_rank[d], csi_context::Peer::NARR, &sp)) ok = 0;
            pr.nbr_slots[d] = (unsigned long long*)sp;
        }
    }
    if (c->local) {
        int32_t lrc;
        if ((lrc = local_allreduce_min(c, &ok))) return lrc;
    } else if (c->hostg) {
        std::vector<uint8_t> bytes;
        if (!hostgroup_allgather(c->hostg, &ok, sizeof(int), bytes, &c->err)) return CSI_ERR_COMM;
        for (int r = 0; r < c->world; ++r) { int x; memcpy(&x, bytes.data() + (size_t)r * sizeof(int), sizeof(int)); if (x < ok) ok = x; }
    } else if (c->world > 1) {                               // every rank or none
        int* flag = (int*)pr.xbuf;
        HIP_TRY(c, hipMemcpy(flag, &ok, sizeof(int), hipMemcpyHostToDevice));
        NCCL_TRY(c, ncclAllReduce(flag, flag, 1, ncclInt32, ncclMin, c->comm, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipMemcpy(&ok, flag, sizeof(int), hipMemcpyDeviceToHost));
    }
    if (!ok) { peer_release(c); pr.failed = true; return CSI_OK; }
    for (int q = 0; q < csi_context::Peer::NARR; ++q) pr.sig[q] = local[q];
    pr.ready = true;
    return CSI_OK;
}

// Does this sub-cycle run on the peer transport?  Every condition is the same on all ranks of a decomposition (they share the
// configuration and the tile shape), so the ranks agree without talking; the set-up itself is collective.
int32_t peer_decide(csi_context* c, const EvpDev& P, int substeps, bool* use) {
    *use = false;
    csi_context::Peer& pr = c->peer;
    if (!is_tiled(c) || !pr.want || pr.failed || !has_comm(c) || !c->tile.set) return CSI_OK;
    if (c->exch_k > 0) return CSI_OK;                        // an explicit exchange interval asks for the RCCL exchange
    if (c->mode != CSI_MODE_FAST || !c->fusion || !c->pairing || substeps < 1) return CSI_OK;      // (an odd count ends with one single-mode launch)
    // Everything above is the same on every rank; what follows may differ from tile to tile (the fold lives on the northernmost
    // tile of a y partition only, a Bounded x partition has tiles of unequal strides): those conditions go INTO the collective
    // set-up, which answers with the minimum over the ranks -- every rank or none.
    const bool local_ok = peer_tile_supported(c, P) &&
                          c->f[CSI_F_U].ld == c->f[CSI_F_S12].ld && c->f[CSI_F_V].ld == c->f[CSI_F_S11].ld &&
                          c->Nx >= 128;                      // (a wave's x images all go to ONE neighbour: evp_fused2.hip)
    int32_t rc;
    if ((rc = ensure_alt(c))) return rc;
    const void* local[csi_context::Peer::NARR];
    peer_local_arrays(c, local);
    bool same = pr.ready;
    for (int q = 0; q < csi_context::Peer::NARR && same; ++q) same = pr.sig[q] == local[q];
    if (!same && (rc = peer_setup(c, local_ok))) return rc;
    *use = pr.ready && local_ok;
    return CSI_OK;
}

// redirect the halo images of a pair table to the neighbours and describe the flag protocol (G: the launch geometry)
int32_t peer_fill_table(csi_context* c, const FusedGeom& G, bool out_is_alt, FusedTable* t) {
    const csi_context::Peer& pr = c->peer;
    const PeerSets ps = peer_wait_counts(c, G);
    for (int d = 0; d < 8; ++d)
        if (ps.size[d] >= csi_context::Peer::SLOTS) return fail(c, CSI_ERR_UNSUPPORTED, "peer halo transport: more edge tiles than flag slots");
    static const int karr[9] = {2, 3, 4, 0, 1, 10, 11, 12, 13};      // kernel order (sigma11, sigma22, sigma12, u, v, alpha, zeta_c, zeta_f, Delta) -> Peer::arr
    for (int k = 0; k < 9; ++k)
        for (int d = 0; d < 8; ++d) {
            const int q = karr[k] < 5 ? karr[k] + (out_is_alt ? 5 : 0) : karr[k];
            t->P[FP_IMG0 + d * 9 + k] = (unsigned long)pr.arr[d][q];
        }
    int mask = 0;
    for (int d = 0; d < 8; ++d) {
        t->P[FP_SLOT_IN + d] = (unsigned long)(pr.slots + (size_t)d * csi_context::Peer::SLOTS);
        t->P[FP_SLOT_OUT + d] = pr.nbr_slots[d] ? (unsigned long)(pr.nbr_slots[d] + (size_t)kPeerOpp[d] * csi_context::Peer::SLOTS) : 0ul;
        t->I[FI_PWAIT + d] = pr.nbr_wait[d];      // (the neighbour's own set: a fold tile's launches have another geometry)
        if (pr.sync_rank[d] >= 0) mask |= 1 << d;
    }
    int any = 0;
    for (int d = 0; d < 8; ++d)
        for (int q = 0; q < 2; ++q) { t->I[FI_PDLD + d * 2 + q] = pr.dld[d][q]; any |= pr.dld[d][q] != 0; }
    t->I[FI_PHASDLD] = any;
    t->I[FI_NYLO] = c->peer.ny_below > 0 ? c->peer.ny_below : c->Ny;
    t->P[FP_PERR] = (unsigned long)pr.err;
    t->I[FI_PEER] = 1; t->I[FI_PMASK] = mask;
    t->I[FI_PTIER] = pr.tier;
    t->I[FI_PSET] = ps.nW; t->I[FI_PSET + 1] = ps.nE; t->I[FI_PSET + 2] = ps.nS; t->I[FI_PSET + 3] = ps.nN;
    return CSI_OK;
}

// Three sub-steps per launch (evp_fused3.hip): what the plain instantiation of the pair kernel takes, on a fully periodic,
// untiled grid whose halo holds the 6-cell dependency radius of three sub-steps.
bool trio_supported(const csi_context* c, const EvpDev& P) {
    const GridDev& g = c->g;
    if (!c->trios || !pair_supported(c) || is_tiled(c)) return false;
    if (g.xlo != SIDE_PERIODIC || g.xhi != SIDE_PERIODIC || g.ylo != SIDE_PERIODIC || g.yhi != SIDE_PERIODIC) return false;
    if (g.has_mask || P.free_drift || c->metric_kind == CSI_METRIC_FULL || pair_forcing_kind(P) != 0) return false;
    return c->Hx >= 6 && c->Hy >= 6 && c->Nx >= 2 * c->Hx && c->Ny >= 2 * c->Hy;
}
FusedGeom trio_geom(const csi_context* c, const Range& dec) {
    FusedGeom G;
    G.rs = dec;
    // (52-column strip) x (rows) tiles, one workgroup of three waves per tile, 40 KB of LDS: 256 CUs x 4 workgroups = 1024 tiles
    const int width = dec.i1 - dec.i0 + 1, height = dec.j1 - dec.j0 + 1;
    G.nstrips = (width + 51) / 52;
    int target = 1024;
    if (c->tune.trio_tiles >= 0) target = c->tune.trio_tiles;         // tuning aid (CSI_TRIO_TILES)
    int max_chunks = target / G.nstrips;
    if (max_chunks < 1) max_chunks = 1;
    int rows = (height + max_chunks - 1) / max_chunks;
    if (rows < 8) rows = 8;
    if (rows > height) rows = height;
    G.rows = rows;
    G.nchunks = (height + rows - 1) / rows;
    return G;
}

// ---- north fold: a band of rows next to the fold on the three kernels, everything below on the two-sub-steps kernel -------
// The two-sub-steps kernel cannot reproduce the reference next to a fold (it would have to recompute halo rows the reference
// READS as stored images).  But only the rows within reach of the fold need that: rows 1 .. M
// (M = Ny - Hy - 4) run through the pair kernel as a tile whose north side is "connected" -- its halo rows M + 1 .. M + 4 are
// interior rows of the same arrays --, rows above M through the three kernels, which store and read fold images exactly like
// the reference's.  Per pair of sub-steps the band, on its own stream and in its own copies of the arrays: copy rows >= M - 7
// of u, v, sigma from the current buffer, advance them by two three-kernel sub-steps on shrinking row ranges (valid from row M
// on after the second), copy rows >= M + 1 into the other buffer -- while the pair launch reads the current buffer and stores
// rows <= M of the other one.  Two events: a pair launch waits for the previous band (its halo rows), a band for the previous
// pair launch (rows M - 7 .. M of its input).
struct FoldBand {
    int M;
    bool tiled;                     // the fold tile of a y partition: its south side is connected
    int k;                          // exchange interval (tiled; 2 otherwise: one pair launch per band step)
    GridDev g_full, g_cut;          // the tile as it is / with the band cut off and the north side "connected"
    int Ny_full;
    EvpDev P;                       // the whole grid (fold geometry)
    ImageSpec imu, imv;
    Range rs, ru1, rv1, r2;         // the three kernels' ranges on the whole grid
};
const int kBandDiag[4] = {CSI_F_ALPHA, CSI_F_ZETA_C, CSI_F_ZETA_F, CSI_F_DELTA};
const Bound& band_bound(const csi_context* c, int q) { return c->f[q < 5 ? kPing[q] : kBandDiag[q - 5]]; }
FRef band_ref(const csi_context* c, int q) {
    const Bound& b = band_bound(c, q);
    FRef r;
    r.p = c->band[q] + (c->Hx - 1) + (int64_t)(c->Hy - 1) * b.ld;
    r.ld = (int)b.ld;
    return r;
}
int32_t ensure_band(csi_context* c) {
    for (int q = 0; q < 9; ++q) {
        const Bound& b = band_bound(c, q);
        const size_t n = (size_t)b.ld * (size_t)b.nj;
        if (c->band_elems[q] != n) {
            if (c->band[q]) { HIP_TRY(c, hipDeviceSynchronize()); hipFree(c->band[q]); c->band[q] = nullptr; }
            HIP_TRY(c, hipMalloc((void**)&c->band[q], n * sizeof(double)));
            HIP_TRY(c, hipMemsetAsync(c->band[q], 0, n * sizeof(double), c->stream));
            c->band_elems[q] = n;
        }
    }
    if (!c->band_stream) {
        HIP_TRY(c, hipStreamCreateWithFlags(&c->band_stream, hipStreamNonBlocking));
        HIP_TRY(c, hipEventCreateWithFlags(&c->band_ev_pair, hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&c->band_ev_band, hipEventDisableTiming));
    }
    return CSI_OK;
}
// one three-kernel sub-step on rows >= jlo, in place: u, v, sigma in b[0..4], diagnostics (last sub-step) in d[0..3] (alpha,
// zeta_c, zeta_f, Delta; nullptr: the caller's arrays); jlo hugely negative: the whole grid
int32_t band_substep(csi_context* c, const FoldBand& bd, const FastCoef& fc, const FRef* b, const FRef* d, bool ufirst, int jlo, bool last, hipStream_t st) {
    EvpDev Q = bd.P;
    Q.u = b[0]; Q.v = b[1]; Q.s11 = b[2]; Q.s22 = b[3]; Q.s12 = b[4];
    if (d) { Q.al = d[0]; Q.zc = d[1]; Q.zf = d[2]; Q.Dl = d[3]; }
    Q.write_diag = last;
    auto from = [&](Range r, int j0) { if (j0 > r.j0) r.j0 = j0; return r; };
    launch_fast_stress(Q, from(bd.rs, jlo), fc, st);
    if (ufirst) { launch_fast_ustep(Q, from(bd.ru1, jlo + 1), bd.imu, fc, st); launch_fast_vstep(Q, from(bd.r2, jlo + 1), bd.imv, fc, st); }
    else { launch_fast_vstep(Q, from(bd.rv1, jlo + 1), bd.imv, fc, st); launch_fast_ustep(Q, from(bd.r2, jlo + 1), bd.imu, fc, st); }
    return CSI_OK;
}
// two sub-steps (or the trailing single one) of the band: buffer `cur` (0: the caller's arrays) -> the other one, on the band's stream
int32_t band_substeps(csi_context* c, const FoldBand& bd, const FastCoef& fc, int cur, int s, int n, bool last) {
    hipStream_t st = c->band_stream;
    HIP_TRY(c, hipStreamWaitEvent(st, c->band_ev_pair, 0));
    auto rows_from = [&](int q, int j0, const double* src, double* dst, CopyBatch& B) {
        const Bound& b = band_bound(c, q);
        const size_t row = (size_t)(j0 - 1 + c->Hy), off = row * (size_t)b.ld;
        B.src[B.count] = src + off; B.dst[B.count] = dst + off; B.n[B.count] = (long)(((size_t)b.nj - row) * (size_t)b.ld);
        ++B.count;
    };
    CopyBatch in{}, out{}, diag{};
    for (int q = 0; q < 5; ++q) {
        const Bound& b = band_bound(c, q);
        rows_from(q, bd.M - 7, cur == 0 ? b.p : c->alt[q], c->band[q], in);
        rows_from(q, bd.M + 1, c->band[q], cur == 0 ? c->alt[q] : b.p, out);
    }
    launch_copy_batch(in, st);
    FRef b[5], d[4];
    for (int q = 0; q < 5; ++q) b[q] = band_ref(c, q);
    for (int q = 0; q < 4; ++q) d[q] = band_ref(c, 5 + q);
    // validity after the first sub-step: sigma from row M - 5, velocities from M - 3; after the second: sigma M - 2, velocities M
    int32_t rc;
    if (n == 2) {
        if ((rc = band_substep(c, bd, fc, b, d, (s % 2) == 0, bd.M - 5, false, st))) return rc;
        if ((rc = band_substep(c, bd, fc, b, d, ((s + 1) % 2) == 0, bd.M - 2, last, st))) return rc;
    } else if ((rc = band_substep(c, bd, fc, b, d, (s % 2) == 0, bd.M - 3, last, st))) return rc;
    launch_copy_batch(out, st);
    if (last) {
        for (int q = 5; q < 9; ++q) rows_from(q, bd.M + 1, c->band[q], band_bound(c, q).p, diag);
        launch_copy_batch(diag, st);
    }
    HIP_TRY(c, hipEventRecord(c->band_ev_band, st));
    return CSI_OK;
}

static inline int nxf_of(int k) { return k > 1 ? 5 : 2; }     // sigma travels with u, v when k > 1 (see do_subcycle)

// peer: the caller (run_fused_peer) has turned the connected sides of c->g / P.g into periodic ones: the launch loop is that of an
// untiled periodic grid, the halo images of those sides go to the neighbouring tiles' arrays and every pair launch carries a
// number of the flag protocol.  band: the caller (run_fused_fold) has cut the rows next to a north fold off c->g / P.g.
int32_t run_fused(csi_context* c, const EvpDev& P, const FastCoef& fc, int substeps, int first, bool peer = false, const FoldBand* band = nullptr) {
    int32_t rc;
    if ((rc = ensure_alt(c))) return rc;
    if (band && (rc = ensure_band(c))) return rc;
    const bool tiled = band ? band->tiled : is_tiled(c);
    const int k = band ? band->k : exchange_interval(c), W = 2 * k;
    // (band: the halo exchange of the fold tile is that of the tile as it is -- its north side has no neighbour)
    auto exchange_tile = [&](const FRef* fr) -> int32_t {
        if (band) { c->g = band->g_full; c->Ny = band->Ny_full; }
        const int32_t r = exchange_refs(c, fr, nxf_of(k), W);
        if (band) { c->g = band->g_cut; c->Ny = band->M; }
        return r;
    };
    const bool masked = P.g.has_mask != 0;
    const bool force = pair_forcing_kind(P) == 1;           // array-valued forcing: two-sub-steps kernel only
    const bool pairs = peer || (pair_supported(c) && (!tiled || k % 2 == 0));
    // number-valued top stress (or none) and a bottom SemiImplicitStress with number-valued ocean velocities: the kernels'
    // compile-time forcing kinds
    auto ocean_at_rest = [](int kind, double value) { return kind == 0 || (kind == 1 && value == 0.0 && !std::signbit(value)); };   // (-0.0 would flip signed zeros)
    int common_forcing = !force && P.top.kind <= 1 && P.bot.kind == 3 && P.bot.ue_kind != 2 && P.bot.ve_kind != 2 &&
                         P.pressure_kind == 0;            // ... and the default ReplacementPressure
    if (common_forcing && ocean_at_rest(P.bot.ue_kind, P.bot.ue) && ocean_at_rest(P.bot.ve_kind, P.bot.ve))
        common_forcing = 2;                               // ZeroField ocean velocities (the reference's default)
    if (c->tune.pair_common >= 0 && common_forcing > c->tune.pair_common) common_forcing = c->tune.pair_common;   // A/B knob (CSI_PAIR_COMMON)
    FRef ubar_v{nullptr, 0}, vbar_u{nullptr, 0};
    if (force && P.bot.kind == 3 && (P.bot.ue_kind == 2 || P.bot.ve_kind == 2)) {
        // cross components of the ocean velocity averaged to the velocity points, once per sub-cycle
        const Bound* src[2] = {&c->f[CSI_F_V], &c->f[CSI_F_U]};       // shapes: ubar lives at v points, vbar at u points
        for (int q = 0; q < 2; ++q) {
            const size_t n = (size_t)src[q]->ld * (size_t)src[q]->nj;
            if (c->fbar_elems[q] != n) {
                if (c->fbar[q]) { HIP_TRY(c, hipStreamSynchronize(c->stream)); hipFree(c->fbar[q]); c->fbar[q] = nullptr; }
                HIP_TRY(c, hipMalloc((void**)&c->fbar[q], n * sizeof(double)));
                HIP_TRY(c, hipMemsetAsync(c->fbar[q], 0, n * sizeof(double), c->stream));
                c->fbar_elems[q] = n;
            }
        }
        ubar_v.p = c->fbar[0] + (c->Hx - 1) + (int64_t)(c->Hy - 1) * src[0]->ld; ubar_v.ld = (int)src[0]->ld;
        vbar_u.p = c->fbar[1] + (c->Hx - 1) + (int64_t)(c->Hy - 1) * src[1]->ld; vbar_u.ld = (int)src[1]->ld;
        launch_forcing_bars(P, ubar_v, vbar_u, c->stream);
    }
    FRef tbar_v{nullptr, 0}, tbar_u{nullptr, 0};              // wind drag: the air velocities' cross averages
    const bool wind = force && P.top.kind == 3 && (P.top.ue_kind == 2 || P.top.ve_kind == 2);
    const int extra_kind = P.extra ? 1 : ((wind || (force && P.bot.kind == 2)) ? 2 : 0);      // which family of array-forcing instantiations
    if (wind) {
        const Bound* src[2] = {&c->f[CSI_F_V], &c->f[CSI_F_U]};
        for (int q = 0; q < 2; ++q) {
            const size_t n = (size_t)src[q]->ld * (size_t)src[q]->nj;
            if (c->fbar_top_elems[q] != n) {
                if (c->fbar_top[q]) { HIP_TRY(c, hipStreamSynchronize(c->stream)); hipFree(c->fbar_top[q]); c->fbar_top[q] = nullptr; }
                HIP_TRY(c, hipMalloc((void**)&c->fbar_top[q], n * sizeof(double)));
                HIP_TRY(c, hipMemsetAsync(c->fbar_top[q], 0, n * sizeof(double), c->stream));
                c->fbar_top_elems[q] = n;
            }
        }
        tbar_v.p = c->fbar_top[0] + (c->Hx - 1) + (int64_t)(c->Hy - 1) * src[0]->ld; tbar_v.ld = (int)src[0]->ld;
        tbar_u.p = c->fbar_top[1] + (c->Hx - 1) + (int64_t)(c->Hy - 1) * src[1]->ld; tbar_u.ld = (int)src[1]->ld;
        launch_forcing_bars(P, tbar_v, tbar_u, c->stream, true);
    }
    // model.forcing arrays / immersed flux boundary conditions (the EXTRA instantiations of the pair kernel): the divergence of
    // the immersed fluxes is a function of the mask and the metrics only -- once per sub-cycle into two arrays
    const bool extra = P.extra != 0;
    FRef xd_u{nullptr, 0}, xd_v{nullptr, 0};
    if (extra && P.g.has_mask) {
        bool any_ibc = false;
        for (int q = 0; q < 4; ++q) any_ibc |= (P.ibc_u[q] != 0.0) | (P.ibc_v[q] != 0.0);
        if (any_ibc) {
            const Bound* src[2] = {&c->f[CSI_F_U], &c->f[CSI_F_V]};
            for (int q = 0; q < 2; ++q) {
                const size_t n = (size_t)src[q]->ld * (size_t)src[q]->nj;
                if (c->xd_elems[q] != n) {
                    if (c->xd[q]) { HIP_TRY(c, hipStreamSynchronize(c->stream)); hipFree(c->xd[q]); c->xd[q] = nullptr; }
                    HIP_TRY(c, hipMalloc((void**)&c->xd[q], n * sizeof(double)));
                    HIP_TRY(c, hipMemsetAsync(c->xd[q], 0, n * sizeof(double), c->stream));
                    c->xd_elems[q] = n;
                }
            }
            xd_u.p = c->xd[0] + (c->Hx - 1) + (int64_t)(c->Hy - 1) * src[0]->ld; xd_u.ld = (int)src[0]->ld;
            xd_v.p = c->xd[1] + (c->Hx - 1) + (int64_t)(c->Hy - 1) * src[1]->ld; xd_v.ld = (int)src[1]->ld;
            launch_immersed_div(P, xd_u, xd_v, c->stream);
        }
    }
    const int kb = tiled ? k : (pairs ? 2 : 1);             // batch length: positions 0 .. kb-1
    FRef orig[5], alt[5];
    for (int q = 0; q < 5; ++q) { orig[q] = ref_of(c, kPing[q]); alt[q] = alt_ref(c, q); }
    if (tiled && (rc = exchange_tile(orig))) return rc;
    // both buffers start identical, so cells no sub-step ever writes (wall halos, the outermost halo layer of sigma
    // under the one-sub-step kernel) agree in both.  A fully periodic, untiled grid advanced by pair launches only
    // rewrites every cell of the five parents -- interior and all halo images -- at every launch: no copy needed.
    const bool every_cell_written = pairs && !tiled && !has_walls(c) && (substeps % 2 == 0 || (trio_supported(c, P) && substeps >= 2));
    if (!every_cell_written && !peer)                      // (peer: run_fused_peer has made the copy, BEFORE its exchange)
        for (int q = 0; q < 5; ++q) {
            const Bound& b = c->f[kPing[q]];
            HIP_TRY(c, hipMemcpyAsync(c->alt[q], b.p, c->alt_elems[q] * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        }
    const ImageSpec imu = image_spec(c, CSI_F_U), imv = image_spec(c, CSI_F_V);
    // tables: singles (position in the exchange batch) x (which buffer is current) x (u first / v first), then
    // pairs (pair position) x (buffer) x (first sub-step u first / v first)
    constexpr int KMAX = kMaxExchangeInterval, NSINGLE = KMAX * 4, NPAIR = (KMAX / 2) * 4 + 4;      // (+ 4: three sub-steps per launch)
    constexpr int TRIO0 = NSINGLE + (KMAX / 2) * 4;
    const bool trios = pairs && !peer && trio_supported(c, P);
    FusedGeom GT{};
    if (k > KMAX) return fail(c, CSI_ERR_UNSUPPORTED, "exchange interval too large for the fused path");
    if (!c->dev_tables) HIP_TRY(c, hipMalloc((void**)&c->dev_tables, (NSINGLE + NPAIR) * sizeof(FusedTable)));
    FusedGeom G[KMAX], GP[KMAX / 2];
    // configurations only the two-sub-steps kernel takes (masks, array forcing, per-point metrics): a single sub-step (the odd
    // trailing one) runs through that kernel too, its consumer wave storing stage A's results (evp_fused2.hip, `single`)
    const bool single_by_pair = pairs && (masked || force || c->metric_kind == CSI_METRIC_FULL || peer || band);      // (peer: the flag protocol lives in this kernel only; band: its cut tile)
    {
        if (!c->host_ring) {
            HIP_TRY(c, hipHostMalloc((void**)&c->host_ring, sizeof(FusedTable) * (NSINGLE + NPAIR) * csi_context::kRing, hipHostMallocDefault));
            for (int q = 0; q < csi_context::kRing; ++q) HIP_TRY(c, hipEventCreateWithFlags(&c->ring_ev[q], hipEventDisableTiming));
        }
        const int slot = (int)(c->ring_pos++ % csi_context::kRing);
        if (c->ring_used[slot]) HIP_TRY(c, hipEventSynchronize(c->ring_ev[slot]));
        FusedTable* host = c->host_ring + (size_t)slot * (NSINGLE + NPAIR);
        for (int m = 0; m < kb; ++m) {
            const int V = tiled ? W - 2 * m : 2;
            G[m] = fused_geom(c, V);
            if (single_by_pair) {
                const ImageSpec ims11 = image_spec(c, CSI_F_S11), ims22 = image_spec(c, CSI_F_S22), ims12 = image_spec(c, CSI_F_S12);
                const SideV vs = pair_side_v(c, V, 2);
                const Range dec = v_stress_range(c, vs);
                G[m] = pair_geom(c, dec);
                for (int cur = 0; cur < 2; ++cur)
                    for (int uf = 0; uf < 2; ++uf) {
                        Range rs = clip_store(c, dec, true), r1 = clip_store(c, v_first_range(c, vs, uf != 0), false),
                              r2 = clip_store(c, v_second_range(c, vs), false);
                        if (band) { rs.j1 = std::min(rs.j1, c->Ny); r1.j1 = std::min(r1.j1, c->Ny); r2.j1 = std::min(r2.j1, c->Ny); }
                        FusedTable* t = &host[(m * 2 + cur) * 2 + uf];
                        fused_fill_table(P, fc, cur == 0 ? orig : alt, cur == 0 ? alt : orig, rs, r1, r1, r2, imu, imv, t);
                        fused_fill_pair_extra(dec, dec.j0, dec.j1, ims11, ims22, ims12, t);
                        if (force) { fused_fill_forcing(P, ubar_v, vbar_u, t); if (wind) fused_fill_forcing_top(P, tbar_v, tbar_u, t); }
                        if (extra) fused_fill_extra(P, xd_u, xd_v, t);
                        if (peer && (rc = peer_fill_table(c, G[m], cur == 0, t))) return rc;
                    }
                continue;
            }
            for (int cur = 0; cur < 2; ++cur)
                for (int uf = 0; uf < 2; ++uf) {
                    Range r1, r1c, r2;
                    velocity_ranges(c, uf != 0, V, r1, r1c, r2);
                    fused_fill_table(P, fc, cur == 0 ? orig : alt, cur == 0 ? alt : orig, G[m].rs, r1, r1c, r2, imu, imv,
                                     &host[(m * 2 + cur) * 2 + uf]);
                }
        }
        if (pairs) {
            const ImageSpec ims11 = image_spec(c, CSI_F_S11), ims22 = image_spec(c, CSI_F_S22), ims12 = image_spec(c, CSI_F_S12);
            for (int mp = 0; 2 * mp + 1 < kb; ++mp) {
                const SideV va = pair_side_v(c, W - 4 * mp, 4), vb = pair_side_v(c, W - 4 * mp - 2, 2);
                const Range dec = v_stress_range(c, vb), ra = v_stress_range(c, va);
                GP[mp] = pair_geom(c, dec);
                for (int cur = 0; cur < 2; ++cur)
                    for (int auf = 0; auf < 2; ++auf) {
                        const bool buf = auf == 0;                  // the second sub-step has the other order
                        Range rs = clip_store(c, dec, true), r1 = clip_store(c, v_first_range(c, vb, buf), false),
                              r2 = clip_store(c, v_second_range(c, vb), false);
                        if (band) {         // rows above M are the band's: it stores them into the same buffer meanwhile
                            rs.j1 = std::min(rs.j1, c->Ny); r1.j1 = std::min(r1.j1, c->Ny); r2.j1 = std::min(r2.j1, c->Ny);
                        }
                        FusedTable* t = &host[NSINGLE + (mp * 2 + cur) * 2 + auf];
                        fused_fill_table(P, fc, cur == 0 ? orig : alt, cur == 0 ? alt : orig, rs, r1, r1, r2, imu, imv, t);
                        fused_fill_pair_extra(dec, ra.j0, ra.j1, ims11, ims22, ims12, t);
                        if (force) { fused_fill_forcing(P, ubar_v, vbar_u, t); if (wind) fused_fill_forcing_top(P, tbar_v, tbar_u, t); }
                        if (extra) fused_fill_extra(P, xd_u, xd_v, t);
                        if (peer && (rc = peer_fill_table(c, GP[mp], cur == 0, t))) return rc;
                    }
            }
        }
        if (trios) {
            const ImageSpec ims11 = image_spec(c, CSI_F_S11), ims22 = image_spec(c, CSI_F_S22), ims12 = image_spec(c, CSI_F_S12);
            const SideV va = pair_side_v(c, 6, 6), vc = pair_side_v(c, 2, 2);
            const Range dec = v_stress_range(c, vc), ra = v_stress_range(c, va);
            GT = trio_geom(c, dec);
            for (int cur = 0; cur < 2; ++cur)
                for (int auf = 0; auf < 2; ++auf) {
                    const bool cuf = auf != 0;                      // the third sub-step has the first one's order
                    const Range rs = clip_store(c, dec, true), r1 = clip_store(c, v_first_range(c, vc, cuf), false),
                                r2 = clip_store(c, v_second_range(c, vc), false);
                    FusedTable* t = &host[TRIO0 + cur * 2 + auf];
                    fused_fill_table(P, fc, cur == 0 ? orig : alt, cur == 0 ? alt : orig, rs, r1, r1, r2, imu, imv, t);
                    fused_fill_pair_extra(dec, ra.j0, ra.j1, ims11, ims22, ims12, t);
                }
        }
        HIP_TRY(c, hipMemcpyAsync(c->dev_tables, host, sizeof(FusedTable) * (NSINGLE + NPAIR), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipEventRecord(c->ring_ev[slot], c->stream));
        c->ring_used[slot] = true;
    }
    unsigned long long peer_dld_bit = 0ull;      // neighbours with other row strides: the DLD instantiation (bit 63 of the launch number)
    if (peer)
        for (int d = 0; d < 8; ++d) if (c->peer.dld[d][0] | c->peer.dld[d][1]) peer_dld_bit = 1ull << 63;
    int cur = 0;   // 0: the caller's arrays hold the current state
    int m = 0, nex = 0, nlaunch = 0;
    c->last_trios = 0;
    const int end = first + substeps;
    HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    if (band) {
        HIP_TRY(c, hipEventRecord(c->band_ev_pair, c->stream));      // the first band starts behind everything queued so far
        HIP_TRY(c, hipEventRecord(c->band_ev_band, c->stream));      // (nothing for the first pair launch to wait for)
    }
    for (int s = first; s < end;) {
        const bool ufirst = (s % 2) == 0;                  // split_explicit_momentum_equations.jl:178
        if (trios && end - s >= 3 && end - s != 4) {       // (4 = 2 + 2: never leave a single sub-step behind)
            launch_fused_trio(c->dev_tables + TRIO0 + cur * 2 + (ufirst ? 1 : 0), c->coef.uniform != 0, ufirst, common_forcing,
                              GT.nstrips, GT.nchunks, GT.rows, s + 3 == end, c->stream);
            s += 3; m = 0;
            ++c->last_trios;
        } else if (pairs && end - s >= 2 && m + 1 < kb) {
            const int mp = m / 2;
            if (band) {
                HIP_TRY(c, hipStreamWaitEvent(c->stream, c->band_ev_band, 0));       // the previous band: this launch's rows M + 1 .. M + 4
                if ((rc = band_substeps(c, *band, fc, cur, s, 2, s + 2 == end))) return rc;
                nlaunch += 8;
            }
            launch_fused_pair(c->dev_tables + NSINGLE + ((mp * 2 + cur) * 2 + (ufirst ? 1 : 0)),
                              c->metric_kind == CSI_METRIC_FULL ? 2 : (c->coef.uniform != 0 ? 0 : 1), ufirst,
                              has_walls(c) || masked || force || peer_dld_bit != 0, masked, force, P.free_drift != 0, extra_kind, common_forcing, GP[mp].nstrips, GP[mp].nchunks, GP[mp].rows, s + 2 == end,
                              peer ? (++c->peer.seq | peer_dld_bit) : (c->tune.peer_kernel > 0 ? 1ull : 0ull), c->stream);
            if (band) HIP_TRY(c, hipEventRecord(c->band_ev_pair, c->stream));
            m += 2; s += 2;
        } else if (single_by_pair) {
            // one sub-step through the two-sub-steps kernel (write_diag bit 1): masks, array forcing, per-point metrics
            if (band) {
                HIP_TRY(c, hipStreamWaitEvent(c->stream, c->band_ev_band, 0));
                if ((rc = band_substeps(c, *band, fc, cur, s, 1, s + 1 == end))) return rc;
                nlaunch += 5;
            }
            launch_fused_pair(c->dev_tables + ((m * 2 + cur) * 2 + (ufirst ? 1 : 0)),
                              c->metric_kind == CSI_METRIC_FULL ? 2 : (c->coef.uniform != 0 ? 0 : 1), ufirst,
                              has_walls(c) || masked || force || peer_dld_bit != 0, masked, force, P.free_drift != 0, extra_kind, common_forcing, G[m].nstrips, G[m].nchunks, G[m].rows,
                              2 | (s + 1 == end ? 1 : 0), peer ? (++c->peer.seq | peer_dld_bit) : 0ull, c->stream);
            if (band) HIP_TRY(c, hipEventRecord(c->band_ev_pair, c->stream));
            m += 1; s += 1;
        } else if (masked || force || c->metric_kind == CSI_METRIC_FULL) {
            // (no pair kernel for this grid -- halo < 4, tiny tiles: the three kernels in place on whichever buffer is current)
            EvpDev Q = P;
            const FRef* b = cur == 0 ? orig : alt;
            Q.u = b[0]; Q.v = b[1]; Q.s11 = b[2]; Q.s22 = b[3]; Q.s12 = b[4];
            Q.write_diag = (s + 1 == end);
            const int V = tiled ? W - 2 * m : 2;
            launch_fast_stress(Q, stress_range(c, V), fc, c->stream);
            if (ufirst) { launch_fast_ustep(Q, first_u_range(c, V), imu, fc, c->stream); launch_fast_vstep(Q, second_range(c, V), imv, fc, c->stream); }
            else { launch_fast_vstep(Q, first_v_range(c, V), imv, fc, c->stream); launch_fast_ustep(Q, second_range(c, V), imu, fc, c->stream); }
            m += 1; s += 1;
            cur ^= 1;           // undone below: this sub-step did not switch buffers
            nlaunch += 2;
        } else {
            launch_fused_substep(c->dev_tables + ((m * 2 + cur) * 2 + (ufirst ? 1 : 0)), c->coef.uniform != 0, ufirst,
                                 G[m].nstrips, G[m].nchunks, G[m].rows, s + 1 == end, c->stream);
            m += 1; s += 1;
        }
        cur ^= 1;
        ++nlaunch;
        if (tiled && (m == kb || s == end)) {
            if ((rc = exchange_tile(cur == 0 ? orig : alt))) return rc;
            m = 0;
            ++nex;
        } else if (m >= kb) {
            m = 0;
        }
    }
    if (band) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->band_ev_band, 0));
    if (peer) {
        // the neighbours' last launch wrote into this rank's halos: wait for all of it before anything later on this stream
        // (the copy back, finalize_rheology!, the next exchange) reads them
        launch_wait_peers(c->peer.slots, c->peer.sync_rank, csi_context::Peer::SLOTS, c->peer.nbr_wait, c->peer.seq, c->peer.err, c->stream);
        HIP_TRY(c, hipMemcpyAsync(c->peer.err_host, c->peer.err, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    if (cur == 1)   // the result sits in the library's buffers
        for (int q = 0; q < 5; ++q) {
            const Bound& b = c->f[kPing[q]];
            HIP_TRY(c, hipMemcpyAsync(b.p, c->alt[q], c->alt_elems[q] * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        }
    HIP_TRY(c, hipGetLastError());
    c->last_exchanges = nex;
    c->last_k = k;
    c->last_launches = nlaunch;
    c->last_substeps = substeps;
    c->last_used_pairs = pairs && substeps >= 2;
    return CSI_OK;
}

int32_t peer_check_entry(csi_context* c);      // (defined with csi_sync)

// One sub-cycle on the peer transport: an RCCL exchange of u, v, sigma brings the halos up to date (and orders this rank behind
// whatever its neighbours did last), then the connected sides count as periodic ones for the launch loop.
int32_t run_fused_peer(csi_context* c, double dt, const FastCoef& fc, int substeps, int first) {
    int32_t rc;
    FRef orig[5];
    for (int q = 0; q < 5; ++q) orig[q] = ref_of(c, kPing[q]);
    // Both ping-pong buffers start identical where no launch ever writes (cells beyond walls).  The copy comes BEFORE the
    // exchange: once a neighbour has received this rank's message it may start its first launch, whose halo images land in this
    // rank's second buffer -- they must not be overwritten by a copy that is still on its way.  (Halos beyond connected sides
    // need no copy: the neighbours' images rewrite all H layers at every launch.)
    if ((rc = ensure_alt(c))) return rc;
    const bool fold = c->g.yhi == SIDE_FOLD;
    if (has_walls(c) || fold)
        for (int q = 0; q < 5; ++q)
            HIP_TRY(c, hipMemcpyAsync(c->alt[q], c->f[kPing[q]].p, c->alt_elems[q] * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    const int W = std::min(std::min(c->Hx, c->Hy), 4);
    if ((rc = exchange_refs(c, orig, 5, W))) return rc;
    // the fold tile of a y partition: its three-kernel band works on the tile as it is (FoldBand); the pair launches see the
    // tile cut below the band, like every other tile with its connected sides turned into periodic ones (PeerView)
    FoldBand bd;
    const EvpDev Pfull = evp_dev(c, dt);
    if (fold) {
        bd.M = c->Ny - c->Hy - 4;
        bd.tiled = false; bd.k = 2;
        bd.g_full = c->g; bd.Ny_full = c->Ny;
        bd.P = Pfull;
        bd.imu = image_spec(c, CSI_F_U); bd.imv = image_spec(c, CSI_F_V);
        bd.rs = stress_range(c); bd.ru1 = first_u_range(c); bd.rv1 = first_v_range(c); bd.r2 = second_range(c);
    }
    PeerView view(c);
    bd.g_cut = c->g;
    EvpDev P = Pfull;           // (arrays and per-row pointers of the tile as it is; only the grid descriptor differs)
    P.g = c->g;
    rc = run_fused(c, P, fc, substeps, first, true, fold ? &bd : nullptr);
    c->last_exchanges = 1;
    return rc;
}

// A north fold on an untiled grid (RightFolded y, Periodic x): see FoldBand.
struct FoldCut {        // RAII: the tile with the band cut off (rows 1 .. M, north side "connected")
    csi_context* c; GridDev g; int Ny, band;
    FoldCut(csi_context* cc, int M) : c(cc), g(cc->g), Ny(cc->Ny), band(cc->geom_band) { c->Ny = M; c->g.Ny = M; c->g.yhi = SIDE_CONNECTED; c->geom_band = 1; }
    ~FoldCut() { c->g = g; c->Ny = Ny; c->geom_band = band; }
};
bool fold_band_supported(csi_context* c, const EvpDev& Pfull, int substeps) {
    const GridDev& g = c->g;
    if (g.yhi != SIDE_FOLD || g.xlo != SIDE_PERIODIC || g.xhi != SIDE_PERIODIC) return false;
    if (c->mode != CSI_MODE_FAST || !c->fusion || !c->pairing || substeps < 2 || c->Hy < 4) return false;
    if (g.ylo == SIDE_CONNECTED) {
        // the fold tile of a y partition: the k-batched message exchange with the tile below, pair launches need an even k
        const int k = exchange_interval(c);
        if (k % 2 != 0 || !has_comm(c) || !c->tile.set) return false;
    }
    const int M = c->Ny - c->Hy - 4;
    if (M < 2 * c->Hy + 8) return false;
    FoldCut cut(c, M);
    EvpDev P = Pfull;
    P.g = c->g;
    return pair_supported(c) && pair_forcing_kind(P) >= 0;
}
int32_t run_fused_fold(csi_context* c, const EvpDev& Pfull, const FastCoef& fc, int substeps, int first) {
    FoldBand bd;
    bd.M = c->Ny - c->Hy - 4;
    bd.tiled = c->g.ylo == SIDE_CONNECTED;
    bd.k = bd.tiled ? exchange_interval(c) : 2;
    bd.g_full = c->g; bd.Ny_full = c->Ny;
    bd.P = Pfull;
    bd.imu = image_spec(c, CSI_F_U); bd.imv = image_spec(c, CSI_F_V);
    bd.rs = stress_range(c); bd.ru1 = first_u_range(c); bd.rv1 = first_v_range(c); bd.r2 = second_range(c);
    FoldCut cut(c, bd.M);
    bd.g_cut = c->g;
    EvpDev P = Pfull;
    P.g = c->g;
    return run_fused(c, P, fc, substeps, first, false, &bd);
}

int32_t do_subcycle(csi_context* c, double dt, int substeps, int first) {
    int32_t rc;
    if ((rc = peer_check_entry(c))) return rc;
    {                                                // :170-171, both fields in one batch of two launches
        HaloBatch B{};
        B.f[0] = ref_of(c, CSI_F_U); B.im[0] = image_spec(c, CSI_F_U);
        B.f[1] = ref_of(c, CSI_F_V); B.im[1] = image_spec(c, CSI_F_V);
        B.n = 2;
        launch_fill_halo_batch(B, c->g, c->stream);
    }
    const bool tiled = is_tiled(c);
    const int uv[2] = {CSI_F_U, CSI_F_V};
    // halo exchange of u, v every k sub-steps with width 2k (k = 1: every sub-step; the reference is the
    // k = substeps extreme with its 2*substeps+3 halo, split_explicit_momentum_equations.jl:51-64)
    const int k = exchange_interval(c);
    const int W = 2 * k;
    // sigma is history dependent (sigma += (sigma' - sigma) / gamma): with k = 1 the ring-1 values are
    // recomputed every sub-step and stay identical to the neighbour's; with k > 1 the outer rings skip
    // updates inside a batch, so sigma travels with u, v.  alpha is recomputed before every use.
    const int uvs[5] = {CSI_F_U, CSI_F_V, CSI_F_S11, CSI_F_S22, CSI_F_S12};
    const int nxf = k > 1 ? 5 : 2;
    (void)uv;
    if (c->free_drift) {
        // free-drift velocities of marginal ice depend on the forcing only: once per sub-cycle, every point whose
        // four-point averages stay inside the parent arrays
        const int src[2] = {CSI_F_U, CSI_F_V};
        for (int q = 0; q < 2; ++q) {
            const size_t n = (size_t)c->f[src[q]].ld * (size_t)c->f[src[q]].nj;
            if (c->fd_elems[q] != n) {
                if (c->fd[q]) { HIP_TRY(c, hipStreamSynchronize(c->stream)); hipFree(c->fd[q]); c->fd[q] = nullptr; }
                HIP_TRY(c, hipMalloc((void**)&c->fd[q], n * sizeof(double)));
                HIP_TRY(c, hipMemsetAsync(c->fd[q], 0, n * sizeof(double), c->stream));
                c->fd_elems[q] = n;
            }
        }
        launch_free_drift(evp_dev(c, dt), Range{2 - c->Hx, c->Nx + c->Hx - 1, 2 - c->Hy, c->Ny + c->Hy - 1}, c->stream);
    }
    EvpDev P = evp_dev(c, dt);
    const ImageSpec imu = image_spec(c, CSI_F_U), imv = image_spec(c, CSI_F_V);
    const bool fast = c->mode == CSI_MODE_FAST;
    FastCoef fc = c->coef;
    {
        const double ie = 1.0 / P.ecc;
        fc.em2 = ie * ie;
        fc.ca_dt = 0.5 * (P.ca * dt); fc.hkc = fc.ca_dt * fc.uni[FC_RAZC]; fc.hkf = fc.ca_dt * fc.uni[FC_RAZF]; fc.hk1 = 0.5 * (1.0 - ie * ie);
        fc.rdt = 1.0 / dt;
        fc.Dmin2 = P.Dmin * P.Dmin; fc.rDmin = 1.0 / P.Dmin;
        fc.amin2 = P.amin * P.amin; fc.amax2 = P.amax * P.amax; fc.ramin = 1.0 / P.amin; fc.ramax = 1.0 / P.amax;
    }
    if (fast && !fast_supported(P)) return fail(c, CSI_ERR_UNSUPPORTED, "CSI_MODE_FAST does not support this configuration yet; use CSI_MODE_STRICT");
    // immersed masks: only the two-sub-steps-per-launch kernel takes them (a trailing odd sub-step falls back to the
    // three kernels inside run_fused)
    const int pfk = pair_forcing_kind(P);
    const bool pair_only = P.g.has_mask || pfk == 1 || c->metric_kind == CSI_METRIC_FULL;      // configurations only the two-sub-steps kernel takes
    // the peer halo transport (tiles) needs none of the RCCL batching constraints (an even exchange interval): decide it first
    bool peer = false;
    if (fast && c->fusion && substeps > 0 && (rc = peer_decide(c, P, substeps, &peer))) return rc;
    if (!peer && fast && fold_band_supported(c, P, substeps)) {
        c->peer.last = 0;
        if ((rc = run_fused_fold(c, P, fc, substeps, first))) return rc;
        c->timed = true;
        c->launches_per_substep = 1;
        c->last_fused = 2;
        return CSI_OK;
    }
    const bool fuse = peer || (fast && c->fusion && substeps > 0 &&
                               (pair_only ? (pfk >= 0 && pair_supported(c) && (!tiled || k % 2 == 0) && substeps >= 2)
                                          : fused_supported(P)));
    if (fuse) {
        c->peer.last = peer ? 1 : 0;
        if ((rc = peer ? run_fused_peer(c, dt, fc, substeps, first) : run_fused(c, P, fc, substeps, first))) return rc;
        c->timed = true;
        c->launches_per_substep = 1 + ((tiled && k == 1) ? 3 : 0);
        c->last_fused = c->last_trios > 0 ? 3 : (c->last_used_pairs ? 2 : 1);
        return CSI_OK;
    }
    c->last_fused = 0;
    c->peer.last = 0;
    if (tiled && (rc = exchange(c, uvs, nxf, W))) return rc;
    HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    int m = 0, nex = 0;   // position inside the exchange batch
    for (int s = first; s < first + substeps; ++s) {
        const int V = W - 2 * m;
        const Range rs = stress_range(c, V), ru1 = first_u_range(c, V), rv1 = first_v_range(c, V), r2 = second_range(c, V);
        if (fast) {
            P.write_diag = (s == first + substeps - 1);
            launch_fast_stress(P, rs, fc, c->stream);
        } else {
            launch_strict_visc(P, rs, c->stream);          // compute_stresses!, evp:222-234
            launch_strict_stress(P, rs, c->stream);
        }
        if ((s % 2) == 0) {                                // :178-182
            if (fast) { launch_fast_ustep(P, ru1, imu, fc, c->stream); launch_fast_vstep(P, r2, imv, fc, c->stream); }
            else { launch_strict_ustep(P, ru1, imu, c->stream); launch_strict_vstep(P, r2, imv, c->stream); }
        } else {                                           // :184-187
            if (fast) { launch_fast_vstep(P, rv1, imv, fc, c->stream); launch_fast_ustep(P, r2, imu, fc, c->stream); }
            else { launch_strict_vstep(P, rv1, imv, c->stream); launch_strict_ustep(P, r2, imu, c->stream); }
        }
        ++m;
        if (tiled && (m == k || s == first + substeps - 1)) {   // RCCL send/recv of the u, v halos
            if ((rc = exchange(c, uvs, nxf, W))) return rc;
            m = 0;
            ++nex;
        }
    }
    HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    HIP_TRY(c, hipGetLastError());
    c->timed = true;
    c->launches_per_substep = (fast ? 3 : 4) + ((tiled && k == 1) ? 3 : 0);
    c->last_exchanges = nex;
    c->last_k = k;
    return CSI_OK;
}

int32_t do_finalize(csi_context* c) {
    HaloBatch B{};
    for (int fid : {CSI_F_S11, CSI_F_S12, CSI_F_S22}) { B.f[B.n] = ref_of(c, fid); B.im[B.n] = image_spec(c, fid); ++B.n; }
    launch_fill_halo_batch(B, c->g, c->stream);
    HIP_TRY(c, hipGetLastError());
    // fill_halo_regions!(sigma) across tiles.  After a sub-cycle on the peer transport there is nothing to move: the neighbours' last
    // launch stored the images of their sigma into all H halo layers beyond the connected sides (and k_wait_peers has seen them
    // land) -- exactly the values an exchange would bring; what it would ALSO bring are the neighbours' own y fills in the corners
    // (beyond a wall next to a connected x side: nobody stores mirror images of sigma): the same fill on this tile's halo columns
    if (c->peer.last && is_tiled(c)) {
        launch_fill_halo_xcolumns(B, c->g, c->stream);
        HIP_TRY(c, hipGetLastError());
        return CSI_OK;
    }
    const int sg[3] = {CSI_F_S11, CSI_F_S12, CSI_F_S22};
    return exchange(c, sg, 3, c->Hx < c->Hy ? c->Hx : c->Hy);
}

int32_t need_evp(csi_context* c) {
    int32_t rc = need(c, {CSI_F_U, CSI_F_V, CSI_F_H, CSI_F_A, CSI_F_S11, CSI_F_S22, CSI_F_S12, CSI_F_UN, CSI_F_VN,
                          CSI_F_P, CSI_F_ALPHA, CSI_F_DELTA, CSI_F_ZETA_F, CSI_F_ZETA_C});
    if (rc) return rc;
    if (!c->evp_set) return fail(c, CSI_ERR_NOT_BOUND, "csi_evp_params_set has not been called");
    if ((rc = check_stress_fields(c, CSI_STRESS_TOP))) return rc;
    if ((rc = check_stress_fields(c, CSI_STRESS_BOTTOM))) return rc;
    if (c->Hx < 2 || c->Hy < 2) return fail(c, CSI_ERR_INVALID_ARGUMENT, "EVP needs halo >= 2");
    if (c->free_drift) {   // stress_balance_free_drift.jl:21-35: exactly one of the two stresses is a SemiImplicitStress
        const bool ts = c->stress[CSI_STRESS_TOP].kind == CSI_STRESS_SEMI_IMPLICIT, bs = c->stress[CSI_STRESS_BOTTOM].kind == CSI_STRESS_SEMI_IMPLICIT;
        if (ts == bs) return fail(c, CSI_ERR_INVALID_ARGUMENT, "StressBalanceFreeDrift needs exactly one SemiImplicitStress (top or bottom)");
    }
    if (c->Nx < c->Hx || c->Ny < c->Hy) return fail(c, CSI_ERR_UNSUPPORTED, "tile smaller than its halo");
    if (is_tiled(c) && !c->tile.set) return fail(c, CSI_ERR_NOT_BOUND, "connected topology but csi_tile_set has not been called");
    return sync_coriolis(c);
}

int32_t do_time_step_momentum(csi_context* c, double dt, int substeps, int rk_reset) {
    int32_t rc;
    if (rk_reset) {                                         // reset_velocities! :89-93
        if ((rc = need(c, {CSI_F_UM, CSI_F_VM}))) return rc;
        if ((rc = copy_parent(c, CSI_F_U, CSI_F_UM))) return rc;
        if ((rc = copy_parent(c, CSI_F_V, CSI_F_VM))) return rc;
    }
    if ((rc = do_initialize(c))) return rc;                 // :130
    // update_external_stress! :133-134: halos of the forcing fields (local boundary conditions, then tiles)
    // ... and of model.forcing.u / .v when they are arrays: inside an exchange batch the velocity kernels run on ranges that
    // extend into the halo and read the forcing there (elasto_visco_plastic_rheology.jl:391-401 is evaluated at every point
    // the step updates), so beyond a connected side the halo must hold the neighbour's values
    if ((c->f[CSI_F_FORCING_U].p != nullptr) != (c->f[CSI_F_FORCING_V].p != nullptr))
        return fail(c, CSI_ERR_NOT_BOUND, "model.forcing arrays: bind both CSI_F_FORCING_U and CSI_F_FORCING_V or neither");
    const int forcing_ids[6] = {CSI_F_TOP_U, CSI_F_TOP_V, CSI_F_BOT_U, CSI_F_BOT_V, CSI_F_FORCING_U, CSI_F_FORCING_V};
    for (int id : forcing_ids)
        if (c->f[id].p && (rc = fill_halo(c, id))) return rc;
    if (is_tiled(c)) {
        int ff[6], n = 0;
        for (int id : forcing_ids) if (c->f[id].p) ff[n++] = id;
        if (n && (rc = exchange(c, ff, n, c->Hx < c->Hy ? c->Hx : c->Hy))) return rc;
    }
    if ((rc = do_subcycle(c, dt, substeps, 1))) return rc;  // :170-189
    return do_finalize(c);                                  // :192
}

AdvDev adv_dev(const csi_context* c, int scheme, double dt, int from_cache) {
    AdvDev A{};
    A.g = c->g;
    A.u = ref_of(c, CSI_F_U); A.v = ref_of(c, CSI_F_V); A.h = ref_of(c, CSI_F_H); A.a = ref_of(c, CSI_F_A);
    A.Gh = ref_of(c, CSI_F_GH); A.Ga = ref_of(c, CSI_F_GA); A.hm = ref_of(c, CSI_F_HM); A.am = ref_of(c, CSI_F_AM);
    A.has_snow = c->f[CSI_F_HS].p != nullptr && c->f[CSI_F_GHS].p != nullptr;     // snow thickness: the third tracer
    if (A.has_snow) { A.hs = ref_of(c, CSI_F_HS); A.Ghs = ref_of(c, CSI_F_GHS); A.hsm = ref_of(c, CSI_F_HSM); }
    A.scheme = scheme; A.dt = dt; A.from_cache = from_cache;
    A.fill_images = 0; A.im = image_spec(c, CSI_F_H);
    return A;
}

// in_step: called from csi_time_step_*.  tracers_filled: the tracer update of this stage already wrote the halo images of
// h, aice [, hs] with its stores (no mask, no thermodynamic step after it).  Inside a step the velocities are prognostic
// fields only with dynamics (sea_ice_model.jl:230,373-377): prescribed velocities keep the halos set! gave them.
int32_t do_update_state(csi_context* c, bool in_step = false, bool tracers_filled = false) {
    int32_t rc;
    if ((rc = need(c, {CSI_F_H, CSI_F_A}))) return rc;
    // mask_immersed_field_xy! of every prognostic field, then their local halo fills in one batch (two launches)
    const bool snow = c->f[CSI_F_HS].p != nullptr;
    const bool vel = c->f[CSI_F_U].p && c->f[CSI_F_V].p && (!in_step || c->evp_set);
    launch_mask_center(ref_of(c, CSI_F_H), c->g, c->stream);
    launch_mask_center(ref_of(c, CSI_F_A), c->g, c->stream);
    if (snow) launch_mask_center(ref_of(c, CSI_F_HS), c->g, c->stream);
    for (int id : {CSI_F_MASS_FLUX, CSI_F_MASS_FLUX_SNOW, CSI_F_SNOWFALL_INTERCEPTED})       // sea_ice_model.jl:387-390
        if (c->f[id].p) launch_mask_center(ref_of(c, id), c->g, c->stream);
    if (vel) {
        launch_mask_u(ref_of(c, CSI_F_U), c->g, c->stream);
        launch_mask_v(ref_of(c, CSI_F_V), c->g, c->stream);
    }
    HaloBatch B{};
    auto add = [&](int fid) { B.f[B.n] = ref_of(c, fid); B.im[B.n] = image_spec(c, fid); ++B.n; };
    if (!tracers_filled) {
        add(CSI_F_H); add(CSI_F_A);
        if (snow) add(CSI_F_HS);
    }
    if (vel) { add(CSI_F_U); add(CSI_F_V); }
    launch_fill_halo_batch(B, c->g, c->stream);
    HIP_TRY(c, hipGetLastError());
    if (is_tiled(c)) {                                      // the MPI part of fill_halo_regions!, sea_ice_model.jl:383
        int ff[5] = {CSI_F_H, CSI_F_A, CSI_F_U, CSI_F_V, CSI_F_HS};
        int n = vel ? 4 : 2;
        if (snow) ff[n++] = CSI_F_HS;
        if ((rc = exchange(c, ff, n, c->Hx < c->Hy ? c->Hx : c->Hy))) return rc;
    }
    return CSI_OK;
}

int32_t do_tendencies(csi_context* c, int scheme) {
    int32_t rc;
    if ((rc = need(c, {CSI_F_U, CSI_F_V, CSI_F_H, CSI_F_A, CSI_F_GH, CSI_F_GA}))) return rc;
    const bool third = scheme == CSI_ADVECT_WENO3 || scheme == CSI_ADVECT_UPWIND3;
    int need_h = scheme == CSI_ADVECT_WENO7 ? 4 : (scheme == CSI_ADVECT_UPWIND1 ? 1 : (third ? 2 : 3));
    if (scheme != CSI_ADVECT_UPWIND1 && scheme != CSI_ADVECT_WENO5 && scheme != CSI_ADVECT_WENO7 && scheme != CSI_ADVECT_UPWIND5 && !third)
        return fail(c, CSI_ERR_INVALID_ARGUMENT, "unknown advection scheme");
    if (c->Hx < need_h || c->Hy < need_h) return fail(c, CSI_ERR_INVALID_ARGUMENT, "halo too small for the advection scheme");
    launch_tracer_tendencies(adv_dev(c, scheme, 0.0, 0), c->mode, c->stream);
    HIP_TRY(c, hipGetLastError());
    return CSI_OK;
}
// advection = nothing: zero tendencies (horizontal_div_Uc(..., ::Nothing, ...) = zero(grid), sea_ice_advection.jl:50); the
// tracer update still runs -- dynamic_time_step! launches unconditionally -- and resets h, aice [, hs] to Psi^- at every
// RK stage (what makes the stage-wise thermodynamic steps of an RK3 step non-cumulative)
int32_t do_tendencies_or_zero(csi_context* c, int scheme) {
    if (scheme) return do_tendencies(c, scheme);
    int32_t rc;
    if ((rc = need(c, {CSI_F_GH, CSI_F_GA}))) return rc;
    for (int id : {CSI_F_GH, CSI_F_GA, CSI_F_GHS}) {
        const Bound& b = c->f[id];
        if (b.p) HIP_TRY(c, hipMemsetAsync(b.p, 0, (size_t)b.ld * (size_t)b.nj * sizeof(double), c->stream));
    }
    return CSI_OK;
}
// fill_images: the stores also write the local halo images (periodic wrap / no-flux mirror) of h, aice [, hs]
int32_t do_tracer_step(csi_context* c, double dt, int from_cache, bool fill_images = false) {
    int32_t rc;
    if ((rc = need(c, {CSI_F_H, CSI_F_A, CSI_F_GH, CSI_F_GA}))) return rc;
    if (from_cache && (rc = need(c, {CSI_F_HM, CSI_F_AM}))) return rc;
    if (from_cache && c->f[CSI_F_HS].p && c->f[CSI_F_GHS].p && (rc = need(c, {CSI_F_HSM}))) return rc;
    AdvDev A = adv_dev(c, 0, dt, from_cache);
    A.fill_images = fill_images ? 1 : 0;
    launch_tracer_step(A, c->stream);
    HIP_TRY(c, hipGetLastError());
    return CSI_OK;
}

}  // namespace

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
        c->tune.pair_common = env_int("CSI_PAIR_COMMON"); c->tune.trio_tiles = env_int("CSI_TRIO_TILES");
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
    if (c->host_ring) hipHostFree(c->host_ring);
    for (auto& e : c->ring_ev) if (e) hipEventDestroy(e);
    if (c->dev_coef) hipFree(c->dev_coef);
    for (int k = 0; k < 5; ++k) if (c->alt[k]) hipFree(c->alt[k]);
    for (int k = 0; k < 4; ++k) if (c->adv_buf[k]) hipFree(c->adv_buf[k]);
    for (int k = 0; k < 9; ++k) if (c->band[k]) hipFree(c->band[k]);
    if (c->band_ev_pair) hipEventDestroy(c->band_ev_pair);
    if (c->band_ev_band) hipEventDestroy(c->band_ev_band);
    if (c->band_stream) hipStreamDestroy(c->band_stream);
    for (int k = 0; k < 2; ++k) if (c->fbar[k]) hipFree(c->fbar[k]);
    for (int k = 0; k < 2; ++k) if (c->fbar_top[k]) hipFree(c->fbar_top[k]);
    for (int k = 0; k < 2; ++k) if (c->fd[k]) hipFree(c->fd[k]);
    for (int k = 0; k < 2; ++k) if (c->xd[k]) hipFree(c->xd[k]);
    if (c->dev_tables) hipFree(c->dev_tables);
    if (c->sendbuf) hipFree(c->sendbuf);
    if (c->recvbuf) hipFree(c->recvbuf);
    if (c->comm) ncclCommDestroy(c->comm);
    if (c->hostg) hostgroup_leave(c->hostg);
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    if (c->own_stream) hipStreamDestroy(c->stream);
    delete c;
    return CSI_OK;
}

// A wait of the peer halo transport has given up (this rank's error word, copied to pinned memory behind every sub-cycle): the
// sub-cycle that saw it is invalid on this rank and -- through the abort words -- on its neighbours.  Reported by EVERY entry
// point that advances the model (at its start and at its end) and by csi_sync, whichever comes first; the words are cleared so
// that the caller may go on (the launch numbers stay in step on all ranks: they are counted on the host).
static int32_t peer_check(csi_context* c);
namespace { int32_t peer_check_entry(csi_context* c) { return peer_check(c); } }
static int32_t peer_check(csi_context* c) {
    if (c->peer.err_host && *c->peer.err_host) {
        *c->peer.err_host = 0;
        hipMemsetAsync(c->peer.err, 0, sizeof(unsigned), c->stream);
        for (int d = 0; d < 8; ++d)
            hipMemsetAsync(c->peer.slots + (size_t)d * csi_context::Peer::SLOTS + (csi_context::Peer::SLOTS - 1), 0, sizeof(unsigned long long), c->stream);
        return fail(c, CSI_ERR_COMM, "peer halo transport: a tile waited 3 s for its neighbour's flags and gave up (or a neighbouring rank did) -- the results of "
                                     "that sub-cycle are invalid (a rank that fell behind or died; csi_set_halo_transport(ctx, CSI_TRANSPORT_RCCL) selects "
                                     "the RCCL exchange)");
    }
    return CSI_OK;
}
int32_t csi_sync(csi_context* c) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return peer_check(c);
}

int32_t csi_set_mode(csi_context* c, int32_t mode) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    if (mode != CSI_MODE_STRICT && mode != CSI_MODE_FAST) return fail(c, CSI_ERR_INVALID_ARGUMENT, "unknown mode");
    c->mode = mode;
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
    }
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
    // measured (scripts/r03_adv_sizes.sh, WENO7, us per RK3 step, separate launches -> one per stage): 256^2 38 -> 23,
    // 512^2 71 -> 55, 1024^2 219 -> 187, 1536^2 427 -> 384, 2048^2 768 -> 794: the separate update is a pure streaming kernel,
    // which wins once the grid is large enough for launch latencies not to matter
    if ((long)c->Nx * c->Ny > 3000000L) return false;
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
    peer_release(c); c->peer.failed = false;           // (the neighbours may be other ranks now)
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
    peer_release(c); c->peer.failed = false;           // (mappings of the previous communicator's neighbours)
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
    peer_release(c); c->peer.failed = false;
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
    return CSI_OK;
}

int32_t csi_comm_init_host(csi_context* c, const char* shm_name, int32_t world_size, int32_t rank) {
    if (!c || !shm_name) return CSI_ERR_INVALID_ARGUMENT;
    if (world_size < 2 || rank < 0 || rank >= world_size) return fail(c, CSI_ERR_INVALID_ARGUMENT, "rank / world_size out of range (a host-channel group has at least two ranks)");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->comm) { ncclCommDestroy(c->comm); c->comm = nullptr; }
    if (c->hostg) { hostgroup_leave(c->hostg); c->hostg = nullptr; }
    c->local = nullptr;
    peer_release(c); c->peer.failed = false;
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

int32_t csi_set_fusion(csi_context* c, int32_t on) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    c->fusion = on != 0;
    c->pairing = on != 1;         // 1: one sub-step per launch only; any other non-zero value: pairs where supported
    c->trios = on == 3;            // 3: three sub-steps per launch where supported (evp_fused3.hip), two elsewhere
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
    return CSI_OK;
}
int32_t csi_halo_transport(csi_context* c, int32_t* kind) {
    if (!c || !kind) return CSI_ERR_INVALID_ARGUMENT;
    *kind = c->peer.last ? CSI_TRANSPORT_PEER : CSI_TRANSPORT_RCCL;
    return CSI_OK;
}
int32_t csi_set_peer_tier(csi_context* c, int32_t tier) {
    if (!c) return CSI_ERR_INVALID_ARGUMENT;
    if (tier < 0 || tier > 2) return fail(c, CSI_ERR_INVALID_ARGUMENT, "peer protocol tier must be 0, 1 or 2");
    c->peer.tier = tier;
    return CSI_OK;
}
int32_t csi_peer_tier(csi_context* c, int32_t* tier) {
    if (!c || !tier) return CSI_ERR_INVALID_ARGUMENT;
    *tier = c->peer.tier;
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
