// csi_ctx.h -- the host side of libcsi_hip.so: the context, the tile groups and what the translation units of the host code share.
//   csi_core.hip    context helpers, index ranges, launch geometries of the fused kernels
//   csi_group.hip   halo exchange: RCCL, the in-process tile group, the host-channel group (csi_hostgroup.hip)
//   csi_peer.hip    peer halo transport: set-up (IPC mapping, flag arrays), tile sets, kernel tables
//   csi_fold.hip    north fold: the three-kernel band beside the pair launches
//   csi_launch.hip  launch loops: sub-cycle, finalize, time_step_momentum!, tracer steps, update_state!
//   csi_abi.hip     the C ABI (include/csi.h)
#pragma once
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
#include <algorithm>
#include <initializer_list>

using namespace csi;

namespace csi_host {


struct Bound {
    double* p = nullptr;
    int64_t ld = 0;
    int ni = 0, nj = 0;
};

// (x, y) location of every field slot
static const int kLoc[CSI_F_COUNT][2] = {
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
static const char* const kName[CSI_F_COUNT] = {"u", "v", "h", "aice", "sigma11", "sigma22", "sigma12", "un", "vn", "P", "alpha",
                                  "Delta", "zeta_f", "zeta_c", "Gh", "Gaice", "h-", "aice-", "u-", "v-",
                                  "top_u", "top_v", "bottom_u", "bottom_v", "mass_flux",
                                  "hs", "Ghs", "hs-", "mass_flux_snow", "intercepted_snowfall", "Tu", "Tu_snow", "forcing_u", "forcing_v"};

extern std::string g_create_error;      // csi_context_create failures (no context to hold the message)


}  // namespace csi_host
using namespace csi_host;

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
    std::vector<int> device;                       // per rank: the HIP device of its context (-1: not joined); peer_effective_tier
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
    // csi_subcycle_stats_begin / _end: one more event pair around every sub-step loop while `on` (bench.py's timed region)
    struct Stats { bool on = false; std::vector<hipEvent_t> ev; std::vector<int> launches; } stats;
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
        int set_sig[8] = {};                 // ... and the sizes of this rank's tile sets then (the launch geometry depends on the forcing kinds too)
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
        int tier = -1;                       // protocol tier asked for (csi_set_peer_tier; -1: automatic -- peer_effective_tier: 1 as soon as a
                                             // neighbour lives in another process or on another device, 0 for a tile connected to itself / an in-process group)
        bool aborted = false;                // a wait of the flag protocol gave up (here or at a neighbour): the transport is refused until every rank has
                                             // called csi_set_halo_transport / csi_comm_init* again (sticky: the flags cannot recover, csi_abi.hip peer_check)
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
    hipEvent_t exp_ev[2] = {nullptr, nullptr};      // (CSI_EXP_OVERLAP: the second stream is band_stream; band_cus: order between stream and pair_stream)
    hipStream_t pair_stream = nullptr;              // fold band with reserved CUs (tune.band_cus): the pair launches' stream, masked to the other CUs
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
    int weno_w32 = 0;     // csi_set_weno_weight_dtype: 1 = WENO weights in single precision (upstream's FT2 = Float32, recalled)
    int fusion = 1;       // 1: use the fused sub-step kernel when the configuration allows it
    int pairing = 1;      // 1: two sub-steps per launch where supported (csi_set_fusion level 2)
    int last_launches = 0, last_substeps = 0, last_used_pairs = 0;   // kernel launches / sub-steps of the last fused sub-cycle
    int last_fused = 0;
    double ibc[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};   // csi_immersed_flux_bc_set: [u | v][west, east, south, north]
    int exch_k = 0;       // sub-steps per halo exchange (0 = auto: the largest k with 2k <= halo, at most 4)
    // Tile activity (csi_activity.hip, run_fused): the live-tile list of the current sub-cycle and, read back asynchronously, how many
    // tiles were live in earlier ones -- the launch geometry of the NEXT sub-cycles is refined so that the live tiles fill one round
    // (correctness never depends on that estimate: the list is made afresh, on the device, before every sub-cycle's first launch)
    struct Activity {
        static constexpr int kSamples = 4;
        int enabled = 1;                     // csi_set_tile_skipping
        int* flags = nullptr;                // device: one int per tile
        int* list = nullptr;                 // device: {live, tiles, the live tiles' numbers}
        int* list0 = nullptr;                // device: the same for the first two launches (all but the tiles quiescent from the start)
        int* host = nullptr;                 // pinned, device-visible: {seqlock, live, tiles, sample id} written by k_activity_compact
        int* host_dev = nullptr;             // ... as the device addresses it
        double sample_scale[kSamples] = {1, 1, 1, 1};      // the geometry scale of sample id % kSamples
        unsigned seq = 0;                    // sample ids handed out
        int seen_id = -1;                    // the newest sample the host has taken
        double scale = 1.0;                  // tiles of a live launch relative to the one-round geometry (pair_geom's tile_scale)
        int last_live = -1, last_tiles = 0;  // the newest sample that has arrived
        int last_used = 0;                   // the last sub-cycle ran live launches
        int since_probe = 0;                 // sub-cycles since the last test while nothing is quiescent (run_fused: probed every 32nd)
    } act;
    // CSI_METRIC_FULL: rows whose twelve coefficient planes (and per-point Coriolis planes) hold one value per row (ensure_row_constant)
    std::vector<double> coef2_host, fcor2_host;      // host copies of the planes the marks are made from
    double* dev_c2row = nullptr;         // (C2_COUNT + 2) vectors of nj doubles
    int* dev_rcsum = nullptr;            // nj + 1 prefix sums
    bool rc_dirty = true;
    int rc_enabled = 1;                  // csi_set_row_constant(ctx, on, rtol)
    double rc_rtol = 0.0;                // 0: bitwise-equal columns only (results unchanged); > 0: columns within this relative distance of column 1 count as equal -- CHANGES results at that level
    int rc_rows = 0;                     // rows marked
    int geom_band = 0;    // the pair launches being laid out run beside a fold band (FoldCut / PeerView of a fold tile): see pair_geom
    int geom_peer = 0;    // ... are launches of the peer transport (PeerView): shorter chunks next to the connected y sides (bit 0; bits 1 / 2: both x / y sides connected)
    // tuning aids (A/B runs), read from the environment ONCE, when the context is created; -1 = not set
    struct Tuning { int fused_rows = -1, pair_tiles = -1, pair_target = -1, row_target_1024 = 0, pair_minrows = -1, pair_rows = -1, pair_common = -1, peer_kernel = -1, peer_edge = -1, write_through = -1,
                    adv_nt = -1,           // CSI_ADV_NT: tracers per thread of the advection tendency kernel (1 / 2; default by grid size)
                    band_fused = -1,      // CSI_BAND_FUSED=0: the fold band on the three kernels + two copies per step (rounds 4-6a); default: six launches per step without copies, loads hoisted (csi_fold.hip band_substeps_fused)
                    band_event_flags = -1,   // CSI_BAND_EVENT_FLAGS (csi_fold.hip ensure_band)
                    band_cus = -1,        // CSI_BAND_CUS: CUs per XCD reserved for the fold band's launches (the pair launches beside them run on the others)
                    band_cus_share = -1,  // CSI_BAND_CUS_SHARE=1: the band may use every CU (only the pair launches are masked)
                    exp_band_only = -1,   // TIMING EXPERIMENT (CSI_EXP_BAND_ONLY=1, wrong results): fold grids run the band's launches without the pair launches beside them
                    exp_overlap = -1,     // EXPERIMENT (CSI_EXP_OVERLAP, profiles/r06_tile_overlap.txt): bit 0 every tile of a peer-connected launch in the sets of both sides of a connected axis, bit 1 consecutive launches on two streams
                    no_geom_sig = -1;      // debugging aid (CSI_DEBUG_NO_GEOM_SIG=1): skip the launch-geometry check of the peer set-up (tests/test_gpu_local_tiles.py)
      long adv_stage_max_cells = 1L << 40;      // advection-only models: one launch per RK stage up to this many cells (advect_stage_supported; no cut since round 6)
    } tune;
};

namespace csi_host {

#define HIP_TRY(c, expr)                                                                        \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(c, CSI_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));     \
    } while (0)

#define NCCL_TRY(c, expr)                                                                       \
    do {                                                                                        \
        ncclResult_t r_ = (expr);                                                               \
        if (r_ != ncclSuccess)                                                                  \
            return fail(c, CSI_ERR_COMM, std::string(#expr) + ": " + ncclGetErrorString(r_));   \
    } while (0)

static inline int nxf_of(int k) { return k > 1 ? 5 : 2; }     // sigma travels with u, v when k > 1 (see do_subcycle)

struct FoldBand;
// ---- functions shared by the translation units (definitions: see the list at the top) ----
static const int kPing[5] = {CSI_F_U, CSI_F_V, CSI_F_S11, CSI_F_S22, CSI_F_S12};
// (elo / ehi: rows of the FIRST / LAST chunk where they differ from `rows` -- shorter tiles next to a peer-connected y side,
//  pair_geom; elo == rows and ehi == 0: every chunk `rows` rows, the last one what is left)
struct FusedGeom { Range rs; int nstrips, nchunks, rows; int elo = 0, ehi = 0; int wt = 0; };      // wt: write-through result stores (FI_WT)
// rows [ja, jb] of chunk q (the kernels' formula: evp_fused2.hip)
static inline void chunk_rows(const FusedGeom& G, int q, int* ja, int* jb) {
    const int elo = G.elo > 0 ? G.elo : G.rows;
    const bool last_short = G.ehi > 0 && q == G.nchunks - 1 && q > 0;
    *ja = last_short ? G.rs.j1 - G.ehi + 1 : G.rs.j0 + (q == 0 ? 0 : elo + (q - 1) * G.rows);
    *jb = q == 0 ? std::min(*ja + elo - 1, G.rs.j1) : (q == G.nchunks - 1 ? G.rs.j1 : std::min(*ja + G.rows - 1, G.rs.j1 - G.ehi));
}
constexpr int kMaxExchangeInterval = 16;
struct SideV { int xlo, xhi, ylo, yhi; };
static const int kPeerDx[8] = {-1, 1, 0, 0, -1, 1, -1, 1}, kPeerDy[8] = {0, 0, -1, 1, -1, -1, 1, 1};
static const int kPeerOpp[8] = {1, 0, 3, 2, 7, 6, 5, 4};
struct PeerSets { int nW, nE, nS, nN, size[8], n[8]; };
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
    explicit PeerView(csi_context* cc) : c(cc), g(cc->g), Ny(cc->Ny), peer_was(cc->geom_peer) {
        c->geom_peer = 1 | (c->g.xlo == SIDE_CONNECTED && c->g.xhi == SIDE_CONNECTED ? 2 : 0) | (c->g.ylo == SIDE_CONNECTED && c->g.yhi == SIDE_CONNECTED ? 4 : 0);
        for (int* side : {&c->g.xlo, &c->g.xhi, &c->g.ylo, &c->g.yhi}) if (*side == SIDE_CONNECTED) *side = SIDE_PERIODIC;
        if (c->g.yhi == SIDE_FOLD) { const int M = c->Ny - c->Hy - 4; c->Ny = M; c->g.Ny = M; c->g.yhi = SIDE_CONNECTED; band = c->geom_band; c->geom_band = 1; }
    }
    int band = -1, peer_was = 0;
    ~PeerView() { c->g = g; c->Ny = Ny; if (band >= 0) c->geom_band = band; c->geom_peer = peer_was; }
};
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
static const int kBandDiag[4] = {CSI_F_ALPHA, CSI_F_ZETA_C, CSI_F_ZETA_F, CSI_F_DELTA};
struct FoldCut {        // RAII: the tile with the band cut off (rows 1 .. M, north side "connected")
    csi_context* c; GridDev g; int Ny, band;
    FoldCut(csi_context* cc, int M) : c(cc), g(cc->g), Ny(cc->Ny), band(cc->geom_band) { c->Ny = M; c->g.Ny = M; c->g.yhi = SIDE_CONNECTED; c->geom_band = 1; }
    ~FoldCut() { c->g = g; c->Ny = Ny; c->geom_band = band; }
};

int32_t fail(csi_context* c, int32_t code, const std::string& msg);
int side_lo(int topo);
int side_hi(int topo);
int img_of(int side, int loc);
ImageSpec image_spec(const csi_context* c, int fid);
FRef ref_of(const csi_context* c, int fid);
int32_t need(csi_context* c, std::initializer_list<int> ids);
StressDev stress_dev(const csi_context* c, int side);
int32_t check_stress_fields(csi_context* c, int side);
int32_t sync_coriolis(csi_context* c);
EvpDev evp_dev(const csi_context* c, double dt);
Range stress_range(const csi_context* c, int V = 2);
Range first_u_range(const csi_context* c, int V = 2);
Range first_v_range(const csi_context* c, int V = 2);
Range second_range(const csi_context* c, int V = 2);
bool is_tiled(const csi_context* c);
int32_t exchange(csi_context* c, const int* fids, int nf, int W);
int32_t local_allgather(csi_context* c, const void* mine, size_t nb, std::vector<uint8_t>& out);
int32_t local_allreduce_min(csi_context* c, int* v);
int32_t comm_allreduce_max(csi_context* c, int* v);      // over whatever joins the ranks (RCCL communicator, in-process group, host-channel group)
int peer_effective_tier(const csi_context* c);
int32_t local_wait_consumed(csi_context* c);
int32_t local_sendrecv(csi_context* c, const long* soff, const long* scnt, const int* speer, const long* roff, const long* rcnt, const int* rpeer);
int32_t exchange_refs(csi_context* c, const FRef* fr, int nf, int W);
int32_t fill_halo(csi_context* c, int fid);
int32_t copy_parent(csi_context* c, int dst, int src);
int32_t do_initialize(csi_context* c);
FRef alt_ref(const csi_context* c, int k);
FusedGeom fused_geom(const csi_context* c, int V);
void velocity_ranges(const csi_context* c, bool ufirst, int V, Range& r1, Range& r1c, Range& r2);
int32_t ensure_alt(csi_context* c);
int exchange_interval(const csi_context* c);
SideV pair_side_v(const csi_context* c, int v_connected, int v_periodic);
Range v_first_range(const csi_context* c, const SideV& v, bool ufirst);
Range clip_store(const csi_context* c, Range r, bool sigma);
bool has_walls(const csi_context* c);
bool offsets_fit_32bit(int Nx, int Ny, int Hx, int Hy, int64_t max_ld);
int64_t max_bound_ld(const csi_context* c);
bool pair_supported(const csi_context* c);
FusedGeom pair_geom(const csi_context* c, const Range& dec, double tile_scale = 1.0);
int32_t ensure_row_constant(csi_context* c);
bool activity_sample(csi_context* c);
PeerSets peer_wait_counts(const csi_context* c, const FusedGeom& G);
void peer_local_arrays(const csi_context* c, const void* out[csi_context::Peer::NARR]);
void peer_release(csi_context* c);
bool fold_cut_possible(const csi_context* c);
bool peer_tile_supported(csi_context* c, const EvpDev& Pfull);
PeerSets peer_my_sets(csi_context* c);
int32_t peer_setup(csi_context* c, bool local_ok);
int32_t peer_decide(csi_context* c, const EvpDev& P, int substeps, bool* use);
int32_t peer_fill_table(csi_context* c, const FusedGeom& G, bool out_is_alt, FusedTable* t);
FRef band_ref(const csi_context* c, int q);
int32_t ensure_band(csi_context* c);
int32_t band_substep(csi_context* c, const FoldBand& bd, const FastCoef& fc, const FRef* b, const FRef* d, bool ufirst, int jlo, bool last, hipStream_t st);
int32_t band_substeps(csi_context* c, const FoldBand& bd, const FastCoef& fc, int cur, int s, int n, bool last);
int band_launches(const csi_context* c, int n);
int32_t run_fused(csi_context* c, const EvpDev& P, const FastCoef& fc, int substeps, int first, bool peer = false, const FoldBand* band = nullptr);
int32_t run_fused_peer(csi_context* c, double dt, const FastCoef& fc, int substeps, int first);
bool fold_band_supported(csi_context* c, const EvpDev& Pfull, int substeps);
int32_t run_fused_fold(csi_context* c, const EvpDev& Pfull, const FastCoef& fc, int substeps, int first);
int32_t do_subcycle(csi_context* c, double dt, int substeps, int first);
int32_t do_finalize(csi_context* c);
int32_t need_evp(csi_context* c);
int32_t do_time_step_momentum(csi_context* c, double dt, int substeps, int rk_reset);
AdvDev adv_dev(const csi_context* c, int scheme, double dt, int from_cache);
int32_t do_update_state(csi_context* c, bool in_step = false, bool tracers_filled = false);
int32_t do_tendencies(csi_context* c, int scheme);
int32_t do_tendencies_or_zero(csi_context* c, int scheme);
int32_t do_tracer_step(csi_context* c, double dt, int from_cache, bool fill_images = false);
int extra_x(const csi_context* c, int fid);
int extra_y(const csi_context* c, int fid);
Range interior_range(const csi_context* c);
Range parent_range(const csi_context* c);
bool has_comm(const csi_context* c);
Range v_stress_range(const csi_context* c, const SideV& v);
Range v_second_range(const csi_context* c, const SideV& v);
const Bound& band_bound(const csi_context* c, int q);
int32_t peer_check_entry(csi_context* c);      // (csi_abi.hip: the error word of the peer transport, checked at every entry point)

}  // namespace csi_host

