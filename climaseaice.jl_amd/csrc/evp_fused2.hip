// evp_fused2.hip -- TWO EVP sub-steps in one launch (FAST arithmetic, bit-identical to evp_fast.hip and to
// evp_fused.hip applied twice).
//
// The one-sub-step kernel (evp_fused.hip) moves 120 B per cell-update and runs at the bandwidth this access
// pattern can reach.  The only way further down is fewer bytes: this kernel carries the state of sub-step s into
// sub-step s + 1 without a trip through HBM (temporal blocking), so u, v, sigma cross HBM once per PAIR of sub-steps
// (60 B per cell-update).  The dependency ring doubles (radius 2 per sub-step):
//
//   * a wave tile owns 56 columns (lanes 4..59) x `rows` rows; lanes 0..3 / 60..63 and 3 + 3 rows above and
//     below are recomputed redundantly (bit-identical to their owner's results: same code, same inputs);
//   * PRODUCER / CONSUMER waves: a workgroup is two 64-lane waves working on the same tile.  Wave 0 runs stage A
//     (sub-step s) down the rows; wave 1 runs stage B (sub-step s + 1) two rows behind it.  A's new u, v, sigma rows
//     travel through a four-row ring in LDS (10 KB per workgroup, one s_barrier per row); P, h, aice, u^n, v^n are
//     read by both waves (B's reads hit L2: A fetched those rows two iterations earlier).  Each wave holds ONE stage's
//     row window, constants and delay lines, so the kernel needs ~1/2 of the registers of a single wave running both
//     stages (round 1: 212-234 VGPRs -> 2 waves per SIMD, VALU busy 62-68 %): three or more waves per SIMD, half the
//     instructions per wave-row (a wave issues at most one instruction per 4 cycles), and a tile is finished by two
//     waves at once -- small tiles (multi-GPU decompositions) are twice as tall for the same number of waves;
//   * the halo cells B's neighbours need next are refreshed by the owner's store (periodic sides: halo images
//     of u, v AND sigma -- sigma is history dependent, so its halo copies must follow the owner) or by the tile
//     exchange (connected sides, every k sub-steps, k even); walls: see below.
//
// Ranges follow csi_abi.hip (stress [2-V, N+V-1], ...) with V = 4 for A and V = 2 for B on periodic sides and
// the batch position's V on connected sides; every in-range value depends on in-range values only, so the
// stages compute unconditionally and only B's stores are predicated.
// Reference: SeaIceDynamics/split_explicit_momentum_equations.jl:173-189 (two trips of the sub-step loop).
#include "evp_pair_stage.h"

#include <cstdio>
#include <cstdlib>

#ifndef CSI_PAIR_WAVES
#define CSI_PAIR_WAVES 3        // waves per SIMD the register allocation aims at (512 / 3 -> 168 VGPRs)
#endif
#ifndef CSI_PAIR_UNI_WAVES
#define CSI_PAIR_UNI_WAVES 2    // ... of the uniform-coefficient instantiations: pair_geom (csi_core.hip) gives them 1024 tiles = two waves per SIMD
#endif                          //     (round 5), so the velocity coefficients they hold in vector registers (Stage::hoist_uniform) cost no occupancy
#ifndef CSI_PAIR_FULL_WAVES
#define CSI_PAIR_FULL_WAVES 2   // ... of the per-point-coefficient (CSI_METRIC_FULL) instantiations
#endif
#ifndef CSI_PAIR_PRIO
#define CSI_PAIR_PRIO 2         // 2: the consumer wave of every pair above the producers (see k_pair); 1: rotate the priority of the
                                // resident workgroups of a CU every row; 0: none
#endif
#ifndef CSI_PAIR_HOIST
#define CSI_PAIR_HOIST 1        // 1: let the compiler keep the table constants in SGPRs across rows (no per-row reload fence); not in the
                                // array-forcing and per-point-metric instantiations (scalar spills: curvilinear 30.3 -> 27.1 G with it)
#endif
#ifndef CSI_FULL_FENCE
#define CSI_FULL_FENCE 1        // per-point-metric instantiations: reload the table constants every row (1) or let the compiler keep them (0: scalar spills)
#endif
#ifndef CSI_PAIR_STORES
#define CSI_PAIR_STORES 7       // which of stage B's results the CONSUMER stores itself (bit 0: the stresses, bit 1: the first velocity, bit 2: the
                                // second); the producer stores the rest, handed over through the out ring, two iterations later
#endif
#ifndef CSI_PEER_EXP
#define CSI_PEER_EXP 3          // how an edge tile publishes its halo images (PEER instantiations).  3 (default): the images are write-through
                                // stores at system scope (sc0 sc1) and the flags follow a drained store queue (vmcnt(0)); 0: plain stores and a
                                // system-scope release fence before the flags -- the fence writes the XCD's whole L2 back (buffer_wbl2), once per
                                // edge tile: +6 us per launch on a 1024 x 512 tile, +20 us at 2048^2 (profiles/r03_peer_protocol.md); bit 0 alone:
                                // no fence, bit 2: no wait at the start of an edge tile (timing experiments, not valid protocols); bit 3: wait for the fifteen tiles
                                // around this one only (experiment on a tile connected to itself, CSI_EXP_OVERLAP=3: profiles/r06_tile_overlap.txt)
#endif
#ifndef CSI_PAIR_SKIPB
#define CSI_PAIR_SKIPB 0       // 1: the consumer wave skips its first iterations of a tile (pure pipeline lag; see bodyB).  Measured round 5, same-box A/B: 2048 x 256 43.0 G with, 43.4 without; 1024 x 512 43.1 / 43.5 -- the test in every iteration costs more than two idle iterations save: off
#endif
#ifndef CSI_PAIR_PD
#define CSI_PAIR_PD 1           // rows the producer prefetches ahead (1 or 2; 2 costs 20 more VGPRs)
#endif

namespace csi {
namespace fused {

#ifdef CSI_PAIR_PROBE
static __device__ unsigned long long g_probe[8192 * 16];
#define PROBE_DECL unsigned long long pacc0 = 0, pacc1 = 0, pacc2 = 0, ptprev = 0, pit = 0; const unsigned long long pwall0 = wall_clock64()
#define PROBE_START do { __builtin_amdgcn_sched_barrier(0); ptprev = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); ++pit; } while (0)
#define PROBE(acc) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_readcyclecounter(); \
                        __builtin_amdgcn_sched_barrier(0); acc += t_ - ptprev; ptprev = t_; } while (0)
#define PROBE_END(slot) do { if (lane == 0 && (slot) < 8192 && !write_diag) { unsigned long long* dbg = g_probe + (size_t)(slot) * 16; \
                        dbg[0] = pacc0; dbg[1] = pacc1; dbg[2] = pacc2; dbg[6] = pit; dbg[8] = pwall0; dbg[9] = wall_clock64(); \
                        dbg[12] = PSTAGE.sp0; dbg[13] = PSTAGE.sp1; dbg[14] = PSTAGE.sp2; dbg[15] = PSTAGE.sp3; \
                        dbg[10] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4); dbg[11] = (unsigned long long)blockIdx.x; } } while (0)
#else
#define PROBE_DECL do { } while (0)
#define PROBE_START do { } while (0)
#define PROBE(acc) do { } while (0)
#define PROBE_END(slot) do { } while (0)
#endif

template <int V> struct Idx { static constexpr int value = V; };

// hand-off ring: rows x {sigma11, sigma22, sigma12, u, v of sub-step s; P, ice mass, aice, u^n, v^n} x 64 lanes (+ the 2-bit mask
// code of each row): everything stage B consumes, so that the consumer wave issues no global loads at all -- its vmcnt
// queue holds stores only and is never waited for inside the row loop
constexpr int RING_ROWS = 4;
enum : int { RF_S11 = 0, RF_S22, RF_S12, RF_U, RF_V, RF_P, RF_M, RF_A, RF_UN, RF_VN, RF_PF, RF_RMC, RF_RMF };
// The last three -- ice strength at the corner, 1 / m at the cell and the corner, of the producer's row -- spare the consumer
// two reciprocals, a lane shift and four more operations per row (it is the longer wave of the pair); 26 KB of LDS per
// workgroup instead of 20: six workgroups per CU still fit, except with the mask rows of the immersed-boundary
// instantiations, which therefore keep the ten-field ring (CSI_PAIR_PRE).
#ifndef CSI_PAIR_LDSC_NOPRE
#define CSI_PAIR_LDSC_NOPRE 1   // per-row coefficients through the LDS window (Stage::pc): the ten-field ring, so that six workgroups per CU still fit
#endif
#ifndef CSI_PAIR_PRE
#define CSI_PAIR_PRE 1
#endif
// FULL (per-point metric planes): nine of the twelve plane values of a stage-row travel through the ring as well (fields 10 .. 18:
// what the producer's step consumed for row r is what the consumer's step needs for the same row two iterations later), the
// consumer loads the other three itself (1 / Az at the cell and at the velocity points).  Loaded by both waves -- rounds 3 / 4 --
// the consumer, the longer wave of the pair, issued twelve to fourteen vector loads per row, every one of them a miss in the XCD's
// L2 (128 tiles x 27 streams x 512 B per row iteration push a row out before the consumer reaches it: FETCH_SIZE 635 -> 416 MB
// per launch with those loads redirected to resident rows), and the producer waited at the row barrier for half of every
// iteration (in-kernel probe: 4235 of 8061 cycles).  Without the consumer's plane loads: 35.6 -> 41.8 G (timing experiment).
// 19 fields x 4 rows x 512 B = 38 KB per workgroup: four workgroups per CU still fit (no PRE fields here: the consumer forms
// the corner ice strength and the reciprocal masses itself).
// Array forcing on uniform / per-row metrics (FRING): the eight forcing values of a stage-row -- top stress, own and cross component of
// the ocean velocity, free-drift velocity, at the u point and at the v point -- travel through the ring as well (fields 10 .. 17, no
// PRE fields: 18 x 4 x 512 B = 36 KB per workgroup, four workgroups per CU: these instantiations are compiled for two waves per
// SIMD and 256 registers, which also ends their scratch spills).  The consumer loaded them itself -- six to eight vector loads per
// row on the longer wave of the pair, each a miss in the XCD's L2 like the metric planes above: the producer waited at the row
// barrier for 28 % of every iteration (in-kernel probe).  Timing experiment without those loads at two waves per SIMD: OMIP
// style 40.5 -> 48-49 G.  The wind-drag / bottom-stress-array instantiations (EXTRA 2, no free drift there) map their own eight
// values onto the same slots, and so do model.forcing arrays on unmasked grids (EXTRA 1 without MASK: in the free-drift slots).  Not
// with immersed-flux-BC divergences (EXTRA 1 with MASK: ten values and the mask rows -- more than four workgroups per CU hold).
#ifndef CSI_PAIR_FRING
#define CSI_PAIR_FRING 1
#endif
constexpr int RF_FORCING = 8;
enum : int { RF_FU_TAU = 10, RF_FU_WE, RF_FU_WB, RF_FU_FD, RF_FV_TAU, RF_FV_WE, RF_FV_WB, RF_FV_FD };
// Round 5: ALL TWELVE plane values of a stage-row travel from the producer to the consumer, through a ring of their own that is
// THREE rows deep -- 12 x 3 x 512 B = 18 KB, exactly what nine fields in the four-row ring took, so still four workgroups per CU.
// Three rows suffice because the consumer reads a row's planes one iteration EARLIER than it uses them (between the phases of its
// previous step, into the registers its memory prefetch of 1 / Az used to fill): written at producer iteration r (slot r mod 3), read
// during consumer iteration r + 1, which runs beside producer iteration r + 2 (slot r + 2 mod 3).  The three planes the consumer
// still loaded itself -- 1 / Az at the cell and at the two velocity points -- were a SECOND fetch of 24 B per cell, every one a miss in
// the XCD's L2 (counted traffic 1.23 x the compulsory bytes).  Timing experiment without those loads: curvilinear 40.6 -> 44.1 G.
constexpr int RP_ROWS = 3, RP_FIELDS = 12;
enum : int { RP_DXV = 0, RP_RDYV, RP_RDXU, RP_DXF2, RP_DYF2, RP_RAZF, RP_DYU, RP_DYC2, RP_DXC2, RP_RAZC, RP_RAZU, RP_RAZV };

// FULL (orthogonal curvilinear grids, per-point metric planes, csi_fast_coef.h): 14 more loads per stage-row
// in flight -- compiled for 2 waves per SIMD (256 VGPRs); the kernel is bound by the planes' traffic and load count there.
// The kernel's body is a function of its own (not inlined: its registers are allocated for it alone), so that the PEER kernels can
// run their INTERIOR tiles -- no neighbour to wait for, no image into another rank's memory -- through the body of the untiled
// instantiation: compiled into one function with the flag protocol and the redirected image stores, the row loops of every tile
// were 6 % slower (1.7 us per launch on a 1024 x 512 tile with no neighbour at all; round 4, profiles/r04_tile_1024x512.md).
template <bool UNI, bool AUF, bool WALLS, bool MASK, bool FORCE, bool FD, int CF, bool FULL = false, bool PEER = false, int EXTRA = 0, bool DLD = false>
__device__ __forceinline__ void pair_body(const FusedTable* __restrict__ table, const int w, int nstrips, int nchunks, int rows,
                                       int blocks_per_xcd, int write_diag, unsigned long long seq,
                                       double* __restrict__ ring, unsigned* __restrict__ ringm, double* __restrict__ outr, unsigned* __restrict__ peer_abort_p, double* __restrict__ ringp,
                                       double* __restrict__ ringc) {
    constexpr bool LDSC = !UNI && !FULL && !(EXTRA == 1 && MASK) && (CSI_PAIR_LDSC != 0);
    constexpr bool FRING = FORCE && !FULL && (EXTRA != 1 || !MASK) && CSI_PAIR_FRING;      // the forcing values of a stage-row travel through the ring too (below)
    constexpr bool PRE = CSI_PAIR_PRE && !MASK && !FULL && !FRING && !(CSI_PAIR_LDSC_NOPRE && !UNI && !(EXTRA == 1 && MASK) && (CSI_PAIR_LDSC != 0));
    constexpr int RING_FIELDS = FRING ? 10 + RF_FORCING : (PRE ? 13 : 10);      // (FULL: ten fields; its plane values have a ring of their own, ringp)
#define peer_abort (*peer_abort_p)
    const int b = (int)blockIdx.x;      // (w: this workgroup's tile, from k_pair -- dealt over the XCDs there)
    // Roles: wave 0 produces, wave 1 consumes.  (Measured placement of the 12 waves of a CU's six workgroups, in dispatch
    // order, on its SIMDs a..d: a b | b c | c d | d a | a b | c d -- every SIMD gets producers and consumers, the two waves
    // of a workgroup never share a SIMD; swapping the roles in some workgroups changed nothing measurable.)
#ifndef CSI_EXP_ROLESWAP
#define CSI_EXP_ROLESWAP 0     // experiment: workgroups of odd arrival rank on their CU swap the roles of their waves (SIMDs then hold two producers or two consumers)
#endif
    const bool consumer = (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) != 0) != (CSI_EXP_ROLESWAP && ((((b >> 3) / CSI_EXP_ROLESWAP) & 1) != 0));
    const int chunk = w / nstrips, strip = w - chunk * nstrips;
    const int lane = (int)(threadIdx.x & 63);
    tptr_t T = (tptr_t)table;
    // write_diag bit 1: ONE sub-step (the odd trailing sub-step of a sub-cycle): the producer runs its stage, the consumer does no
    // second sub-step and stores the producer's results instead (below).  Bit 0: store the diagnostics (last launch).
    const bool single = (write_diag & 2) != 0;
    write_diag &= 1;

    int ja, jb, rstart, rend;
    unsigned loff, sc, sf;
    unsigned lm = 0, sm = 0;   // MASK: column offset / row stride into the uint8 activity mask
    int i;
    unsigned flags;
    int dx;                  // byte offset of this column's halo image on a periodic side (0: none): every field
    int dxv;                 // the same for v, which also has mirror images across x walls (Center in x)
    bool wave_valx_b = false;  // WALLS: some lane of the wave reflects v about a ValueBoundaryCondition value (L_VAL_LO / L_VAL_HI)
    bool wave_has_dx_b;      // any lane of the wave has one
    bool lanes_uniform;      // fast store path allowed (see flush)
    bool lanes_same;         // every lane stores all three kinds of results or none (x images allowed)
    int row0;                // first row of the parent arrays (1 - Hy)
    enum : unsigned { L_RS = 1, L_R1 = 2, L_R2 = 4, L_WALL_U = 8, L_WALL_V = 16, L_MIR_LO = 32, L_MIR_HI = 64, L_VAL_LO = 128, L_VAL_HI = 256, L_LOW = 512 };
    {
        const int Nx = T->I[FI_NX], Hx = T->I[FI_HX], Hy = T->I[FI_HY];
        const int i0s = T->I[FI_DEC + 0] - P_LO + strip * P_W;
        i = i0s + lane;
        // chunk 0 may be shorter (FI_ELO rows), the last chunk is what FI_EHI keeps for it: tiles next to a peer-connected y
        // side wait for / signal a neighbour and store its halo images -- fewer rows take them off the launch's critical path
        {
            const int elo = T->I[FI_ELO] > 0 ? T->I[FI_ELO] : rows;
            ja = (T->I[FI_EHI] > 0 && chunk == nchunks - 1 && chunk > 0) ? T->I[FI_DEC + 3] - T->I[FI_EHI] + 1
                                                                         : T->I[FI_DEC + 2] + (chunk == 0 ? 0 : elo + (chunk - 1) * rows);
            jb = chunk == 0 ? min(ja + elo - 1, T->I[FI_DEC + 3])
                            : (chunk == nchunks - 1 ? T->I[FI_DEC + 3] : min(ja + rows - 1, T->I[FI_DEC + 3] - T->I[FI_EHI]));
        }
        const int ic = min(max(i, 1 - Hx), Nx + Hx);
        loff = (unsigned)(ic - (1 - Hx)) * 8u;
        row0 = 1 - Hy;
        const bool own = (lane >= P_LO) & (lane <= P_HI) & (i <= T->I[FI_DEC + 1]);
        flags = 0;
        if (own & (i >= T->I[FI_RS + 0]) & (i <= T->I[FI_RS + 1])) flags |= L_RS;
        if (own & (i >= T->I[FI_R1 + 0]) & (i <= T->I[FI_R1 + 1])) flags |= L_R1;
        if (own & (i >= T->I[FI_R2 + 0]) & (i <= T->I[FI_R2 + 1])) flags |= L_R2;
        // halo images of a periodic side (the pair kernel only runs on periodic / connected / wall sides, N >= 2H):
        // column i in [1, H] is also stored at i + N, column in (N - H, N] at i - N; rows likewise
        dx = 0;
        // (per side: on the peer transport a tile may have a wall on one side and a neighbour -- "periodic" -- on the other; the
        //  images of the low columns serve the neighbour beyond the LOW side, whose high halo they fill)
        if ((T->I[FI_XLO] == SIDE_PERIODIC) & (i >= 1) & (i <= Hx)) dx = Nx * 8;
        else if ((T->I[FI_XHI] == SIDE_PERIODIC) & (i > Nx - Hx) & (i <= Nx)) dx = -Nx * 8;
        dxv = dx;
        if (WALLS) {
            // Walls (Bounded sides): faces on / beyond the wall are peripheral nodes (velocity 0); v, Center in x, is
            // mirrored across an x wall: column i in [1, H] -> 1 - i, column in (N - H, N] -> 2N + 1 - i.  Lanes at
            // columns 0 and N + 1 are the first mirror cells: stage B takes their v from the neighbouring lane.
            const bool xlo_wall = T->I[FI_XLO] == SIDE_WALL, xhi_wall = T->I[FI_XHI] == SIDE_WALL;
            if ((xlo_wall & (i <= 1)) | (xhi_wall & (i > Nx))) flags |= L_WALL_U;
            if ((xlo_wall & (i < 1)) | (xhi_wall & (i > Nx))) flags |= L_WALL_V;
            if (xlo_wall & (i == 0)) flags |= L_MIR_LO;
            if (xhi_wall & (i == Nx + 1)) flags |= L_MIR_HI;
            // v at an x wall: no-flux mirror of H columns, or ValueBoundaryCondition (IMG_VALUE): ONE halo cell, 2 val - v
            const bool vval_lo = T->I[FI_IMV + 0] == IMG_VALUE, vval_hi = T->I[FI_IMV + 1] == IMG_VALUE;
            if (xlo_wall & (i >= 1) & (i <= Hx)) dxv = vval_lo ? ((i == 1) ? -8 : 0) : (1 - 2 * i) * 8;
            if (xhi_wall & (i > Nx - Hx) & (i <= Nx)) dxv = vval_hi ? ((i == Nx) ? 8 : 0) : (2 * Nx + 1 - 2 * i) * 8;
            if (xlo_wall & vval_lo & ((i == 1) | (i == 0))) flags |= L_VAL_LO;      // columns 0 / 1: reflection about 2 * K[FK_BCV]
            if (xhi_wall & vval_hi & ((i == Nx) | (i == Nx + 1))) flags |= L_VAL_HI;
            wave_valx_b = __builtin_amdgcn_ballot_w64((flags & (L_VAL_LO | L_VAL_HI)) != 0) != 0;
        }
        wave_has_dx_b = __builtin_amdgcn_ballot_w64((dx != 0) | (dxv != 0)) != 0;
        // every lane stores all three kinds or none, and no lane has an image: the common store path
        const bool same = ((flags & L_RS) != 0) == ((flags & L_R1) != 0) && ((flags & L_RS) != 0) == ((flags & L_R2) != 0);
        lanes_same = __builtin_amdgcn_ballot_w64(!same) == 0;
        lanes_uniform = !wave_has_dx_b && lanes_same;
        rstart = max(ja - 3, T->I[FI_AJ0]);
#ifndef CSI_EXP_RINGCUT
#define CSI_EXP_RINGCUT 0       // TIMING EXPERIMENT ONLY (wrong results): so many of a tile's top ring rows are not run -- what a launch would
                                // cost if vertically adjacent tiles shared their seam rows instead of recomputing them (profiles/r05_tile.md)
#endif
        rend = min(jb + 3 - CSI_EXP_RINGCUT, T->I[FI_AJ1]);
        if (i <= Hx) flags |= L_LOW;                         // this column's x image lies beyond the HIGH side of the low-side neighbour
        sc = (unsigned)T->I[FI_LD_C] * 8u;
        sf = (unsigned)T->I[FI_LD_F] * 8u;
        if (MASK) {
            sm = (unsigned)T->I[FI_MASK_LD];
            lm = (unsigned)(ic - (1 - Hx));
        }
    }
    // ---- peer-connected sides (halo transport "peer", csi_abi.hip): flags instead of a halo exchange ----------------------------
    // The neighbours' tiles of launch seq - 1 stored the images this tile is about to read (its halo beyond a connected side),
    // and read the halo cells this tile's images are about to overwrite; each of them publishes seq - 1 in a slot of THIS rank's
    // flag array when it is done (below).  Only tiles next to a connected side wait -- the interior of the launch runs while the
    // edges of the neighbours' previous launch finish, which is all the overlap of communication and computation there is to have.
    // pdirs: the direction sets (W E S N SW SE NW NE) this tile belongs to, restricted to directions with a neighbour.
    unsigned pdirs = 0;
    if constexpr (PEER) {
        const bool pw = strip < T->I[FI_PSET], pe = strip >= nstrips - T->I[FI_PSET + 1], ps = chunk < T->I[FI_PSET + 2],
                   pn = chunk >= nchunks - T->I[FI_PSET + 3];
        pdirs = ((pw ? 1u : 0u) | (pe ? 2u : 0u) | (ps ? 4u : 0u) | (pn ? 8u : 0u) | ((ps & pw) ? 16u : 0u) | ((ps & pe) ? 32u : 0u) |
                 ((pn & pw) ? 64u : 0u) | ((pn & pe) ? 128u : 0u)) & (unsigned)T->I[FI_PMASK];
        pdirs = (unsigned)__builtin_amdgcn_readfirstlane((int)pdirs);
        if (pdirs) {
            if (!consumer && !(CSI_PEER_EXP & 4)) {
                // One poll = the slots of ALL this tile's directions in flight at once (lane l reads slot l of each direction; the
                // sets are a few dozen tiles).  No cache invalidate afterwards: every tile whose footprint shares a 128-byte line
                // with halo cells of a direction is in that direction's set and loads nothing before it has seen the flags, and the
                // vector L1 starts every launch empty -- so no line of these halos can have been fetched before the neighbour's
                // stores landed in this GPU's memory (whose L2 is kept coherent with incoming writes by the memory-side probes).
                // A wait that gives up (3 s at 100 MHz; or this rank's error word / a neighbour's abort word is already set) never
                // hangs and never lets the tile run on stale halos: the workgroup leaves WITHOUT storing or publishing anything, sets
                // this rank's error word (the host reports it at the next ABI call, csi_sync included) and the abort word of every
                // neighbour's flag array (the last slot of each direction's block), so that the ranks around stop waiting at once
                // instead of three seconds later each.  The sub-cycle is lost on every rank that sees it.
                const unsigned long long t0 = wall_clock64();
                bool dead = false;
                for (;;) {
                    bool behind = false, poison = false;
                    if constexpr ((CSI_PEER_EXP & 8) != 0) {
                        // TIMING EXPERIMENT (a tile connected to ITSELF in y with every tile in its S and N sets, CSI_EXP_OVERLAP=3:
                        // profiles/r06_tile_overlap.txt): wait for the tiles of the previous launch around this one only (3 strips x 5 chunks) -- the
                        // tiles whose stores this one reads and whose loads its stores would overtake -- instead of for all of them
                        const unsigned long long* slots = (const unsigned long long*)T->P[FP_SLOT_IN + 2];      // (direction S)
                        if (lane < 15) {      // (two chunks up and down: the chunks next to a connected side may be shorter than a footprint's reach)
                            const int cq = (chunk + lane / 3 - 2 + 2 * nchunks) % nchunks, cs = (strip + lane % 3 - 1 + nstrips) % nstrips;
                            behind |= __hip_atomic_load(slots + (cq * nstrips + cs), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + 1ull < seq;
                        }
                    } else
#pragma unroll
                    for (int d = 0; d < 8; ++d) {
                        if (!((pdirs >> d) & 1u)) continue;
                        const int nslots = T->I[FI_PWAIT + d];
                        const unsigned long long* slots = (const unsigned long long*)T->P[FP_SLOT_IN + d];
#pragma unroll 1
                        for (int b0 = 0; b0 < nslots; b0 += 64) {
                            const int idx = b0 + lane;
                            if (idx < nslots) behind |= __hip_atomic_load(slots + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + 1ull < seq;   // v < seq - 1
                        }
                        if (lane == 63) poison |= __hip_atomic_load(slots + (kPeerSlots - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0ull;
                    }
                    if (__builtin_amdgcn_ballot_w64(behind) == 0) break;
                    if (__builtin_amdgcn_ballot_w64(poison | (*(volatile unsigned*)T->P[FP_PERR] != 0u)) != 0 || wall_clock64() - t0 > 300000000ull) { dead = true; break; }
                    __builtin_amdgcn_s_sleep(8);
                }
                if (dead) {
                    if (lane == 0) *(volatile unsigned*)T->P[FP_PERR] = 1u;
                    if (lane < 8) {
                        typedef const __attribute__((address_space(4))) unsigned long* sptr_t;
                        unsigned long long* out = (unsigned long long*)((sptr_t)&T->P[FP_SLOT_OUT])[lane];       // (the whole block of direction `lane` at the neighbour)
                        if (out) __hip_atomic_store(out + (kPeerSlots - 1), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                }
                if (lane == 0) peer_abort = dead ? 1u : 0u;
                // Run-time protocol tiers (FI_PTIER, csi_set_peer_tier; chosen by the host for ALL ranks).  0: the reasoning above.
                // >= 1: the textbook acquire as well -- a system-scope fence once the flags have been seen (invalidates this
                // CU's vector L1 and the L2 lines that are not kept coherent by hardware), should that reasoning not hold
                // between two devices.  (>= 2: the publishing side fences too, see publish.)
                if (T->I[FI_PTIER] >= 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();                                         // (uniform over the workgroup) releases the consumer's stores
            if (peer_abort) return;                                  // (LDS word: both waves take the same way)
        }
    }
    // byte offset of (this lane's column, row j) in a Center-x / Face-x parent, and into the mask
    auto offc = [&](int j) __attribute__((always_inline)) { return loff + (unsigned)(j - row0) * sc; };
    auto offf = [&](int j) __attribute__((always_inline)) { return loff + (unsigned)(j - row0) * sf; };
    auto offm = [&](int j) __attribute__((always_inline)) { return lm + (unsigned)(j - row0) * sm; };
    const int NyW = T->I[FI_NY], HyW = T->I[FI_HY];
    // FULL, round 6: ROW-CONSTANT tiles.  A real tripolar grid is a latitude-longitude grid south of its bipolar cap: there the
    // twelve planes (and a per-point Coriolis parameter) have the same value in every column of a row.  csi_abi.hip marks such
    // rows (bitwise comparison of all columns, FP_RCSUM: prefix sums over parent rows) and keeps the planes' values of those rows as
    // per-row vectors (FP_C2ROW_0 ...).  A tile whose rows -- window warm-up and look-ahead included -- are all marked reads every
    // plane value from the vectors: same instructions, same operands, bit for bit the same results, but the twelve loads of a
    // stage-row all hit one cached line instead of streaming 96 B per cell from HBM (the per-point-metric kernel runs at its byte
    // roof: 216 B per cell and launch, of which these are 96).
    int rcd = 0;
    if constexpr (FULL) {
        typedef const __attribute__((address_space(4))) int* iptr_t;
        iptr_t rcs = (iptr_t)T->P[FP_RCSUM];
        if (rcs) {
            const int t0 = max(rstart - 4, row0) - row0, t1 = min(rend + 2, NyW + HyW + 1) - row0;
            if (rcs[t1 + 1] - rcs[t0] == t1 - t0 + 1) rcd = FP_C2ROW_0 - FP_C2_0;
        }
    }
    const unsigned c2s = FULL ? (rcd ? 8u : (unsigned)T->I[FI_C2_LD] * 8u) : 0u;         // row stride of the per-point coefficient planes / of their per-row vectors
    const unsigned loff2 = (FULL && rcd) ? 0u : loff;
    auto off2 = [&](int j) __attribute__((always_inline)) { return loff2 + (unsigned)(j - row0) * c2s; };
    const int NyLoW = PEER ? T->I[FI_NYLO] : NyW;       // rows from a low row to its image: the height of the tile BELOW (a fold tile's own is cut)
    // u, Center in y, is also mirrored across y walls: row j in [1, H] -> 1 - j, row in (N - H, N] -> 2N + 1 - j
    const bool wrap_lo_b = T->I[FI_YLO] == SIDE_PERIODIC, wrap_hi_b = T->I[FI_YHI] == SIDE_PERIODIC;      // (per side, as in x)
    const bool ylo_wall_b = WALLS && T->I[FI_YLO] == SIDE_WALL, yhi_wall_b = WALLS && T->I[FI_YHI] == SIDE_WALL;
    // ... unless that wall carries a ValueBoundaryCondition (IMG_VALUE): then ONE halo row, 2 val - u
    const bool uval_lo_b = WALLS && T->I[FI_IMU + 2] == IMG_VALUE, uval_hi_b = WALLS && T->I[FI_IMU + 3] == IMG_VALUE;
    // the wave-uniform switches of the row loop, packed into one scalar register (as separate bools each is a 64-bit lane mask)
    enum : unsigned { U_WRAPLO = 1, U_YLO = 2, U_YHI = 4, U_UVLO = 8, U_UVHI = 16, U_HASDX = 32, U_VALX = 64, U_WRAPHI = 128, U_MIR = 256 };
    // WALLS: some lane of the wave is the first mirror cell beyond an x wall (L_MIR_LO / L_MIR_HI): only the two edge strips of a grid
    const bool wave_mir_b = WALLS && __builtin_amdgcn_ballot_w64((flags & (L_MIR_LO | L_MIR_HI)) != 0) != 0;
    const unsigned UF = (unsigned)__builtin_amdgcn_readfirstlane((int)((wrap_lo_b ? U_WRAPLO : 0u) | (wrap_hi_b ? U_WRAPHI : 0u) | (ylo_wall_b ? U_YLO : 0u) | (yhi_wall_b ? U_YHI : 0u) |
                                                                         (uval_lo_b ? U_UVLO : 0u) | (uval_hi_b ? U_UVHI : 0u) |
                                                                         (wave_has_dx_b ? U_HASDX : 0u) | (wave_valx_b ? U_VALX : 0u) | (wave_mir_b ? U_MIR : 0u)));
#define wrap_lo ((UF & U_WRAPLO) != 0)
#define wrap_hi ((UF & U_WRAPHI) != 0)
#define ylo_wall (WALLS && (UF & U_YLO) != 0)
#define yhi_wall (WALLS && (UF & U_YHI) != 0)
#define uval_lo (WALLS && (UF & U_UVLO) != 0)
#define uval_hi (WALLS && (UF & U_UVHI) != 0)
#define wave_has_dx ((UF & U_HASDX) != 0)
#define wave_valx (WALLS && (UF & U_VALX) != 0)
#define wave_mir (WALLS && (UF & U_MIR) != 0)
#define has_dld (PEER && DLD)      // (an instantiation of its own: compiled into the common PEER one it cost 3 - 4 % of its launch time, not taken)
    // peripheral nodes of a row (walls only): u faces of rows beyond a y wall, v faces on / beyond it
    auto wall_row = [&](int j) __attribute__((always_inline)) { return (ylo_wall & (j < 1)) | (yhi_wall & (j > NyW)); };
    auto wall_vrow = [&](int j) __attribute__((always_inline)) { return (ylo_wall & (j <= 1)) | (yhi_wall & (j > NyW)); };
    const bool lane_wu = WALLS && (flags & L_WALL_U) != 0, lane_wv = WALLS && (flags & L_WALL_V) != 0;
    // External stresses.  Numbers come from the table; FORCE: an array-valued top stress (tau at the u / v points)
    // and / or a bottom SemiImplicitStress whose ocean velocities are arrays (own component at the point, cross
    // component pre-averaged to the point once per sub-cycle: csi_abi.hip) are loaded one value per lane for the rows a
    // stage updates.
    auto numbers = [&](Forcing& F) __attribute__((always_inline)) {
        F.t_tau_u = T->K[FK_TOP_TAU_U]; F.t_we_u = T->K[FK_TOP_UE]; F.t_wb_u = T->K[FK_TOP_VE];
        F.b_tau_u = T->K[FK_BOT_TAU_U]; F.b_we_u = T->K[FK_BOT_UE]; F.b_wb_u = T->K[FK_BOT_VE];
        F.t_tau_v = T->K[FK_TOP_TAU_V]; F.t_we_v = T->K[FK_TOP_VE]; F.t_wb_v = T->K[FK_TOP_UE];
        F.b_tau_v = T->K[FK_BOT_TAU_V]; F.b_we_v = T->K[FK_BOT_VE]; F.b_wb_v = T->K[FK_BOT_UE];
        F.fd_u = 0.0; F.fd_v = 0.0; F.fd = FD;
        F.xc_u = 0.0; F.xd_u = 0.0; F.xc_v = 0.0; F.xd_v = 0.0; F.extra = EXTRA == 1 ? T->I[FI_EXTRA] : 0;
    };
    auto arrays = [&](Forcing& F, unsigned ou, unsigned ov) __attribute__((always_inline)) {
        if (T->I[FI_TOP_KIND] == 2) { F.t_tau_u = ldg(T->P[FP_FT_U], ou); F.t_tau_v = ldg(T->P[FP_FT_V], ov); }
        if constexpr (EXTRA == 2) {
            // wind drag (air velocities as arrays, like the ocean's below) and an explicit bottom stress given as arrays: instantiations
            // of their own -- as run-time branches of the common array-forcing ones their six per-lane values (uniform numbers
            // there) pushed those over the register budget (scratch spills: model.forcing arrays 55 -> 42 G, measured)
            if (T->I[FI_TOP_UEK] == 2) { F.t_we_u = ldg(T->P[FP_FT_U], ou); F.t_wb_v = ldg(T->P[FP_FT_UBAR], ov); }
            if (T->I[FI_TOP_VEK] == 2) { F.t_we_v = ldg(T->P[FP_FT_V], ov); F.t_wb_u = ldg(T->P[FP_FT_VBAR], ou); }
            if (T->I[FI_BOT_KIND] == 2) { F.b_tau_u = ldg(T->P[FP_FB_U], ou); F.b_tau_v = ldg(T->P[FP_FB_V], ov); }
        }
        if (T->I[FI_BOT_UEK] == 2) { F.b_we_u = ldg(T->P[FP_FB_U], ou); F.b_wb_v = ldg(T->P[FP_FB_UBAR], ov); }   // u_e: own component at u points, averaged to v points
        if (T->I[FI_BOT_VEK] == 2) { F.b_we_v = ldg(T->P[FP_FB_V], ov); F.b_wb_u = ldg(T->P[FP_FB_VBAR], ou); }   // v_e: own component at v points, averaged to u points
        if (FD) { F.fd_u = ldg(T->P[FP_FD_U], ou); F.fd_v = ldg(T->P[FP_FD_V], ov); }                             // StressBalanceFreeDrift (once per sub-cycle, csi_abi.hip)
        if (EXTRA == 1) {
            if (F.extra & 1) { F.xc_u = ldg(T->P[FP_XC_U], ou); F.xc_v = ldg(T->P[FP_XC_V], ov); }              // model.forcing.u / .v
            if (F.extra & 2) { F.xd_u = ldg(T->P[FP_XD_U], ou); F.xd_v = ldg(T->P[FP_XD_V], ov); }              // immersed flux boundary conditions
        }
    };
    auto stress_consts = [&](fm::StressConst& ks) __attribute__((always_inline)) {
        ks.em2 = T->K[FK_EM2]; ks.Dmin = T->K[FK_DMIN]; ks.Dmin2 = T->K[FK_DMIN2]; ks.rDmin = T->K[FK_RDMIN];
        ks.amin = T->K[FK_AMIN]; ks.amax = T->K[FK_AMAX]; ks.amin2 = T->K[FK_AMIN2]; ks.amax2 = T->K[FK_AMAX2];
        ks.ramin = T->K[FK_RAMIN]; ks.ramax = T->K[FK_RAMAX]; ks.hk1 = T->K[FK_HK1]; ks.pressure_kind = T->I[FI_PRESSURE_KIND];
        ks.em2_8 = T->K[FK_PK_EM2_8]; ks.Dmin2_16 = T->K[FK_PK_DMIN2_16];
    };
    auto vel_consts = [&](fm::VelConst& kv) __attribute__((always_inline)) {
        // doubled dt and thresholds: Stage updates the velocities from sums over the two cells of a face (fm::vel_update_sum)
        kv.dt = T->K[FK_DT2]; kv.rdt = T->K[FK_RDT]; kv.fcor = T->K[FK_FCOR]; kv.min_mass = T->K[FK_MIN_MASS2];
        kv.min_conc = T->K[FK_MIN_CONC2]; kv.has_cor = T->I[FI_HAS_COR];
    };
    // Issue arbitration favours the OLDEST wave of a SIMD: without help the workgroups dispatched first finish after ~2/3 of
    // the launch and the youngest run on alone (measured: tile lifetimes 100 .. 180 us in one launch; by age rank on the CU
    // 115 .. 136 us even with a rotation keyed on the dispatch order, because the allocator does not place ranks r, r + 3 on
    // different SIMDs).  The waves therefore rotate the user priority every row, keyed on their own slot in the SIMD
    // (HW_ID.wave_id: the three resident waves of a SIMD hold slots 0, 1, 2): slot + row (mod 3) gives each of them the
    // top priority one row in three.
    // (Measured with static priorities: the two youngest workgroups of a CU at a higher priority finish after 77 us, the others
    // after 108 .. 138 us, the launch still takes 150 us -- the SIMDs are work-conserving, the launch time is set by the
    // work per SIMD and by how often its three waves stall at the same time, not by the order they finish in.)
    const int prio_rank = (int)(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) % 3u);      // HW_REG_HW_ID bits [3:0]
    auto set_prio = [&](int k) __attribute__((always_inline)) {
        if (CSI_PAIR_PRIO == 2) {                    // the longer wave of the pair (since the stores moved: the consumer) first
            if (consumer) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
        } else if (CSI_PAIR_PRIO == 3) {             // (the other way round: measured worse)
            if (consumer) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(2);
        } else if (CSI_PAIR_PRIO == 4) {
            if (consumer) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
        } else if (CSI_PAIR_PRIO) {
            int p = prio_rank + k;
            p = p >= 3 ? p - 3 : p;
            if (p == 0) __builtin_amdgcn_s_setprio(0); else if (p == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(2);
        }
    };
    // rows of this tile each kind of store covers (wave-uniform, fixed for the whole march)
    const int rs_lo = max(ja, T->I[FI_RS + 2]), rs_hi = min(jb, T->I[FI_RS + 3]);
    const int r1_lo = max(ja, T->I[FI_R1 + 2]), r1_hi = min(jb, T->I[FI_R1 + 3]);
    const int r2_lo = max(ja, T->I[FI_R2 + 2]), r2_hi = min(jb, T->I[FI_R2 + 3]);
    // rows (uniform): +Ny / -Ny / 0 rows to the halo image of row j
    auto yimg = [&](int j) __attribute__((always_inline)) {
        return (wrap_lo & (j >= 1) & (j <= HyW)) ? NyLoW : ((wrap_hi & (j > NyW - HyW) & (j <= NyW)) ? -NyW : 0);
    };
    auto yimg_u = [&](int j) __attribute__((always_inline)) {
        int d = yimg(j);
        if (WALLS) {
            if (ylo_wall & (j >= 1) & (j <= HyW)) d = uval_lo ? ((j == 1) ? -1 : 0) : 1 - 2 * j;
            if (yhi_wall & (j > NyW - HyW) & (j <= NyW)) d = uval_hi ? ((j == NyW) ? 1 : 0) : 2 * NyW + 1 - 2 * j;
        }
        return d;
    };
    // one value -> its cell and the halo images of that cell (same semantics as store_with_images); valy / valx: the
    // value of the y / x image (a ValueBoundaryCondition reflection differs from the cell's own value).
    // WHERE the images go: into the halo of THIS tile's arrays on an untiled grid and beyond walls (mirrors); on a
    // peer-connected side (csi_abi.hip, halo transport "peer": the neighbour's arrays are mapped into this process and the side
    // behaves like a periodic one whose halo lives on another GPU) into the NEIGHBOUR's arrays, over xGMI.  The table holds, per
    // output array k (sigma11, sigma22, sigma12, u, v, then the diagnostics alpha, zeta_c, zeta_f, Delta) and direction d, the
    // parent address that receives the image: low columns (1 .. H) are imaged beyond the high side of the WEST neighbour, high
    // columns beyond the low side of the EAST one, rows likewise (S / N), corners diagonally.
    enum : int { D_W = 0, D_E, D_S, D_N, D_SW, D_SE, D_NW, D_NE };
    auto own = [&](int k) __attribute__((always_inline)) { return T->P[k < 5 ? FP_S11_OUT + k : FP_AL + (k - 5)]; };
    // (direction-major: the bases of one direction are adjacent table entries -- one wide scalar load per row of an edge tile)
    auto img = [&](int k, int d) __attribute__((always_inline)) { return PEER ? T->P[FP_IMG0 + d * 9 + k] : own(k); };
    // PEER: the lanes of a wave that store x images all lie on one side of the tile (csi_abi.hip admits tiles of 128 columns or
    // more), so the neighbour -- the base address -- is wave-uniform
    const bool wave_low = PEER && __builtin_amdgcn_ballot_w64(((flags & L_LOW) != 0) & ((dx != 0) | (dxv != 0))) != 0;
    // a result into this tile's own arrays: write-through on small grids (FI_WT; csi_core.hip pair_geom) -- at the end of a launch
    // the XCDs' L2s write their dirty lines back before the next launch may start (+ bytes / 6 TB/s at the boundary,
    // MI355X_MICROARCH.md): on a 1024 x 512 tile that is 4 - 5 % of the launch; at 2048^2 the 8-byte write-through stores cost 11 %
    const bool wt = T->I[FI_WT] != 0;
    auto sto = [&](unsigned long b_, unsigned o_, double v_) __attribute__((always_inline)) {
        if (wt) stg_agent(b_, o_, v_); else stg(b_, o_, v_);
    };
    auto put4 = [&](int k, unsigned off, unsigned dy, bool ylow, int dxl, double val, double valy, double valx, double valxy, int j, int yr) __attribute__((always_inline)) {
        const unsigned long base = own(k);
        auto sti = [&](unsigned long b_, unsigned o_, double v_) __attribute__((always_inline)) {
            if constexpr (PEER && (CSI_PEER_EXP & 2) != 0)
                __scoped_atomic_store_n((__attribute__((address_space(1))) long*)((gptr_t)b_ + o_), __builtin_bit_cast(long, v_), __ATOMIC_RELAXED, __MEMORY_SCOPE_SYSTEM);
            else sto(b_, o_, v_);
        };
        sto(base, off, val);
        if constexpr (PEER) {
            // neighbours whose arrays have another row stride (FI_PDLD): the image of parent row p sits p * (their stride - ours)
            // bytes further on (j: the row of `off`, wave-uniform -- scalar arithmetic, in a branch of its own)
            if (has_dld) {
                const int cls = (k == 2 || k == 3 || k == 7) ? 1 : 0;          // sigma12, u, zeta_f: Face in x
                const int p0 = j + HyW - 1, p1 = p0 + yr;                      // parent rows of the cell / of its y image (yr rows away)
                const int dY = ylow ? D_S : D_N, dX = wave_low ? D_W : D_E;
                const int dXY = ylow ? (wave_low ? D_SW : D_SE) : (wave_low ? D_NW : D_NE);
                if (dy != 0u) sti(img(k, dY), off + dy + (unsigned)(p1 * T->I[FI_PDLD + dY * 2 + cls]), valy);
                if (wave_has_dx && dxl != 0) {
                    sti(img(k, dX), off + (unsigned)dxl + (unsigned)(p0 * T->I[FI_PDLD + dX * 2 + cls]), valx);
                    if (dy != 0u) sti(img(k, dXY), off + (unsigned)dxl + dy + (unsigned)(p1 * T->I[FI_PDLD + dXY * 2 + cls]), valxy);
                }
                return;
            }
        }
        if (dy != 0u) sti(PEER ? (ylow ? img(k, D_S) : img(k, D_N)) : base, off + dy, valy);
        if (wave_has_dx) {
            if (dxl != 0) {
                sti(PEER ? (wave_low ? img(k, D_W) : img(k, D_E)) : base, off + (unsigned)dxl, valx);
                if (dy != 0u) sti(PEER ? (ylow ? (wave_low ? img(k, D_SW) : img(k, D_SE)) : (wave_low ? img(k, D_NW) : img(k, D_NE))) : base, off + (unsigned)dxl + dy, valxy);
            }
        }
    };
    auto put = [&](int k, unsigned off, unsigned dy, bool ylow, int dxl, double val, int j, int yr) __attribute__((always_inline)) {
        put4(k, off, dy, ylow, dxl, val, val, val, val, j, yr);
    };
    // u of row j: its y image is a reflection about 2 val on a ValueBoundaryCondition wall
    auto put_u = [&](unsigned off, int j, unsigned dy, double val, int yr) __attribute__((always_inline)) {
        double vy = val;
        if (WALLS) {
            if (uval_lo & ylo_wall & (j == 1)) vy = 2 * T->K[FK_BCU] - val;
            if (uval_hi & yhi_wall & (j == NyW)) vy = 2 * T->K[FK_BCU + 1] - val;
        }
        put4(3, off, dy, j <= HyW, dx, val, vy, val, vy, j, yr);
    };
    auto put_v = [&](unsigned off, unsigned dy, bool ylow, double val, int j, int yr) __attribute__((always_inline)) {
        double vx = val;
        if (WALLS && wave_valx) {
            if (flags & L_VAL_LO) vx = 2 * T->K[FK_BCV] - val;
            if (flags & L_VAL_HI) vx = 2 * T->K[FK_BCV + 1] - val;
        }
        put4(4, off, dy, ylow, dxv, val, val, vx, vx, j, yr);
    };
    // rows q for which every kind of store is due and no row has a y image: with lanes_uniform this is the common
    // store path (two scalar compares per row instead of the full bookkeeping)
    int fast_lo, fast_hi;
    {
        const int d1 = AUF ? 0 : 1;                                            // first velocity row = q - d1, second = q - 1
        fast_lo = max(rs_lo, max(r1_lo + d1, r2_lo + 1));
        fast_hi = min(rs_hi, min(r1_hi + d1, r2_hi + 1));
        if (wrap_lo | ylo_wall) fast_lo = max(fast_lo, HyW + 2);                // rows q-1 .. q clear of the low image rows 1 .. H
        if (wrap_hi | yhi_wall) fast_hi = min(fast_hi, NyW - HyW);              // ... and of the high ones N-H+1 .. N
    }
    const bool fast_plain = lanes_uniform;                // no lane of the wave has an x image either
    // Stage B's results of row q (sigma(q); first velocity v(q) / u(q-1); second velocity u(q-1) / v(q-1)) go to memory from
    // whichever wave CSI_PAIR_STORES names: the consumer itself, straight from its registers (the default), or the producer,
    // which gets them through a two-row LDS ring and stores them two iterations later.  The stores and their bookkeeping
    // are ~20 % of a wave's row time, so they decide which wave of the pair waits for the other at the barrier: while the
    // arithmetic was 282 instructions per stage-row the consumer was the longer wave and the producer stored (+3 %); at 234
    // the producer -- loads, address arithmetic, eleven LDS writes -- became the longer one (probe: 36 + 4 waiting against
    // 30 + 12 waiting) and the stores went back to the consumer: +5 % at 2048^2, +7 % on a 1024 x 512 tile (28 + 13 against
    // 39 + 5 now; splitting the five stores between the waves balances them better and measures the same within noise).
    auto flush = [&](int q, double v11, double v22, double v12, double vfirst, double vsecond, auto WH, auto VFT) __attribute__((always_inline)) {
        constexpr int which = decltype(WH)::value;
        constexpr bool VF = decltype(VFT)::value != 0;         // the stored stage is v-first (pairs: stage B, i.e. A is u-first; single sub-step: stage A itself)           // bit 0: stresses, bit 1: first velocity, bit 2: second velocity
        if (which == 0) return;
        if ((q >= fast_lo) & (q <= fast_hi)) {
            // interior rows (nearly every call): every kind of store is due, no row has a y image
            const unsigned ocq = offc(q), ofq = offf(q);
            if (fast_plain) {
                // interior tile: the owned lanes store five values, no images
                if (flags & L_RS) {
                    if (which & 1) { sto(T->P[FP_S11_OUT], ocq, v11); sto(T->P[FP_S22_OUT], ocq, v22); sto(T->P[FP_S12_OUT], ofq, v12); }
                    if (which & 2) sto(T->P[VF ? FP_V_OUTP : FP_U_OUTP], VF ? ocq : ofq - sf, vfirst);
                    if (which & 4) sto(T->P[VF ? FP_U_OUTP : FP_V_OUTP], VF ? ofq - sf : ocq - sc, vsecond);
                }
            } else {
                // tile on an x edge of the domain: some lanes also store the x image of their column (periodic wrap / the
                // neighbouring tile's halo; v mirrors / reflects across an x wall), none of the row bookkeeping of the general path.
                // Each kind of store under its own lane flag: next to an x wall the three column ranges differ (the face ON the
                // wall is no u to store), and such a strip took the general path for every row -- the east strip of a Bounded grid
                // 25 % longer than the others, which set the launch's end (bounded 2048^2: 147 us against 125 for the interior strips)
                const bool ls = (flags & L_RS) != 0, l1 = (flags & L_R1) != 0, l2 = (flags & L_R2) != 0;
                if ((which & 1) && ls) { put(0, ocq, 0u, false, dx, v11, q, 0); put(1, ocq, 0u, false, dx, v22, q, 0); put(2, ofq, 0u, false, dx, v12, q, 0); }
                if (VF) { if ((which & 2) && l1) put_v(ocq, 0u, false, vfirst, q, 0); if ((which & 4) && l2) put(3, ofq - sf, 0u, false, dx, vsecond, q - 1, 0); }
                else { if ((which & 2) && l1) put(3, ofq - sf, 0u, false, dx, vfirst, q - 1, 0); if ((which & 4) && l2) put_v(ocq - sc, 0u, false, vsecond, q - 1, 0); }
            }
            return;
        }
        const int j1 = VF ? q : q - 1, j2 = q - 1;            // rows of the first / second velocity
        const bool do_s = ((which & 1) != 0) & (q >= rs_lo) & (q <= rs_hi), do_1 = ((which & 2) != 0) & (j1 >= r1_lo) & (j1 <= r1_hi),
                   do_2 = ((which & 4) != 0) & (j2 >= r2_lo) & (j2 <= r2_hi);
        if (!(do_s | do_1 | do_2)) return;
        const unsigned ocq = offc(q), ofq = offf(q);
        const unsigned o1 = VF ? ocq : ofq - sf;                 // first velocity: v(q) / u(q-1)
        const unsigned o2 = VF ? ofq - sf : ocq - sc;            // second velocity: u(q-1) / v(q-1)
        // rows of the images: sigma wraps only; u (the first velocity when B is u-first) may mirror
        const int yq = yimg(q), y1 = VF ? yimg(j1) : yimg_u(j1), y2 = VF ? yimg_u(j2) : yimg(j2);
        if (do_s & ((flags & L_RS) != 0)) {
            put(0, ocq, (unsigned)yq * sc, q <= HyW, dx, v11, q, yq);
            put(1, ocq, (unsigned)yq * sc, q <= HyW, dx, v22, q, yq);
            put(2, ofq, (unsigned)yq * sf, q <= HyW, dx, v12, q, yq);
        }
        if (do_1 & ((flags & L_R1) != 0)) {
            if (VF) put_v(o1, (unsigned)y1 * sc, j1 <= HyW, vfirst, j1, y1); else put_u(o1, j1, (unsigned)y1 * sf, vfirst, y1);
        }
        if (do_2 & ((flags & L_R2) != 0)) {
            if (VF) put_u(o2, j2, (unsigned)y2 * sf, vsecond, y2); else put_v(o2, (unsigned)y2 * sc, j2 <= HyW, vsecond, j2, y2);
        }
    };
    // ring slot of row j (lane-private column): element f of row j sits at ring[((j - rstart) & 3) * 5 + f][lane]
    // MASK instantiations (their row loops are a few registers over the 168 of three waves per SIMD): the lane index is formed
    // again in every iteration (two v_mbcnt, volatile so that it is not hoisted) instead of being kept -- kept, it was spilled to
    // scratch memory and came back behind an s_waitcnt vmcnt(0) that also drained the producer's row prefetch (round 5, ISA listing)
    unsigned lane_it = (unsigned)lane;
    auto fresh_lane = [&]() __attribute__((always_inline)) {
        if constexpr (MASK && !FULL) {
            unsigned l;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
            lane_it = l;
        }
    };
    auto rslot = [&](int j) __attribute__((always_inline)) { return (unsigned)((j - rstart) & (RING_ROWS - 1)) * (RING_FIELDS * 64) + lane_it; };

#ifndef CSI_PAIR_UNROLL
#define CSI_PAIR_UNROLL 3
#endif

    if (!consumer) {
        // ===== PRODUCER: stage A = sub-step s, rows rstart .. rend ===================================================
        PROBE_DECL;
        Stage<UNI, AUF, MASK, FORCE, CF, FULL, !(EXTRA == 1 && MASK)> A;    // (TIGHT scalar live ranges in the array-forcing variants)
        A.lc = ringc;
        // LDSC: the address of lane k's entry (k < FC_COUNT; the other lanes re-read the last one) of coefficient row j of the table
        const int cjmin = LDSC ? T->I[FI_COEF_JMIN] : 0, cjmax = LDSC ? T->I[FI_COEF_JMAX] : 0;
        const unsigned ck_lane = (unsigned)min(lane, FC_COUNT - 1) * 8u;
        auto coef_row_base = [&](int j) __attribute__((always_inline)) {
            return T->P[FP_PCOEF_VEC] + (unsigned long)((long)min(max(j, cjmin), cjmax) * (long)(FC_COUNT * 8));
        };
#ifndef CSI_EXP_RINGALIAS
#define CSI_EXP_RINGALIAS 0     // TIMING EXPERIMENT ONLY (wrong results): a tile's ring rows are read from the nearest rows it owns -- same instructions,
                                // same arithmetic, but no ring row ever comes from HBM: what it would be worth if every ring-row re-read hit the L2
#endif
        auto alias_row = [&](int j) __attribute__((always_inline)) { return CSI_EXP_RINGALIAS ? min(max(j, ja + 1), max(jb - 1, ja + 1)) : j; };
        unsigned oc = offc(alias_row(rstart)), of = offf(alias_row(rstart)), om = MASK ? offm(rstart) : 0u;
        {
            const double rho0 = T->K[FK_RHO];
            A.u_m = ldg(T->P[FP_U_IN], of - sf); A.v_m = ldg(T->P[FP_V_IN], oc - sc);
            A.u_0 = ldg(T->P[FP_U_IN], of); A.v_0 = ldg(T->P[FP_V_IN], oc);
            A.a_mm = 0.0; A.a_m = ldg(T->P[FP_A], oc - sc);
            A.m_mm = 0.0; A.m_m = ldg(T->P[FP_H], oc - sc) * rho0 * A.a_m;
            const double P_m = ldg(T->P[FP_P], oc - sc);
            A.XP_m = fm::sum2(from_left(P_m), P_m);
            A.Xm_m = fm::sum2(from_left(A.m_m), A.m_m);
            A.Xa_m = fm::sum2(from_left(A.a_m), A.a_m);
            A.Xv_m = fm::sum2(from_left(A.v_m), A.v_m);
            A.Xv_0 = fm::sum2(from_left(A.v_0), A.v_0);
            double e11_m, e22_m;
            const int jm = rstart - 1;
            A.hoist_uniform(T);
            if constexpr (FULL) {
                const unsigned om2 = off2(jm), o02 = off2(rstart);
                full_cell(T, om2, c2s, A.u_m, A.v_m, A.v_0, e11_m, e22_m, rcd);
                A.e12_0 = 8.0 * full_corner(T, o02, c2s, A.u_0, A.u_m, A.v_0, rcd);      // (the stage carries 8 e12: evp_pair_stage.h)
                A.hoist_planes(T, rcd);
                A.full_init(T, o02, om2, c2s);
            } else {
            fm::strain_cell<UNI>(pcoef<UNI>(T, FC_A, jm), pcoef<UNI>(T, FC_BN, jm), pcoef<UNI>(T, FC_BS, jm), pcoef<UNI>(T, FC_CN, jm),
                            pcoef<UNI>(T, FC_CS, jm), from_right(A.u_m), A.u_m, A.v_0, A.v_m, e11_m, e22_m);
            A.e12_0 = fm::strain_corner<UNI>(pcoef<UNI>(T, FC_SN, rstart), pcoef<UNI>(T, FC_SS, rstart), pcoef<UNI>(T, FC_SV, rstart), A.u_0, A.u_m, A.v_0, from_left(A.v_0));
            }
            A.Xe11_m = fm::sum2(from_left(e11_m), e11_m);
            A.Xe22_m = fm::sum2(from_left(e22_m), e22_m);
            A.Ye12_0 = fm::sum2(A.e12_0, from_right(A.e12_0));
            A.XAL_m = 0; A.XS11L_m = 0; A.XS22L_m = 0; A.XW = 0; A.Wprev = 0;
            A.S11_mm = 0; A.S22_mm = 0; A.S12_mm = 0; A.AL_mm = 0; A.S11_m = 0; A.S22_m = 0; A.S12_m = 0; A.AL_m = 0;
            A.S11_0 = 0; A.S22_0 = 0; A.S12_0 = 0; A.AL_0 = 0; A.first = 0; A.second = 0;
        }
        // the ring starts clean: the consumer's first iterations read rows the producer never wrote (their results only
        // fill B's window and are never used -- but they must not be NaN patterns left in LDS by an earlier workgroup)
#pragma unroll
        for (int q = 0; q < RING_ROWS * RING_FIELDS; ++q) ring[q * 64 + lane] = 0.0;
        if constexpr (FULL) {
#pragma unroll
            for (int q = 0; q < RP_ROWS * RP_FIELDS; ++q) ringp[q * 64 + lane] = 1.0;      // (plane values: finite reciprocals)
            A.RAZU_m = 1.0; A.RAZV_x = 1.0; A.RAZC_0 = 1.0;      // (the first step runs no velocity phase and loads none of the first two)
        }
        // MASK: two bits per row (bit 0 inactive, bit 1 beyond a wall), newest row in bits 1:0; row rstart-1 from memory,
        // older rows count as beyond the domain (their results are never used)
        unsigned mhist = 0xffffffffu;
        if (MASK) {
            const bool wi = lane_wv | wall_row(rstart - 1);
            const unsigned act = ldub(T->P[FP_MASK], om - sm);
            mhist = (mhist << 2) | (wi ? 3u : (act ? 0u : 1u));
            ringm[(unsigned)((-1) & (RING_ROWS - 1)) * 64 + (unsigned)lane] = mhist & 3u;      // row rstart - 1, for the consumer
        }
        // Prefetch CSI_PAIR_PD rows ahead (three rotating register sets; the loop is unrolled by three, so the set of every
        // iteration is fixed at compile time).  The producer issues no stores and loads return in order, so the wait at the
        // top of an iteration leaves exactly the younger rows in flight.  (Measured: one row ahead already hides the
        // memory latency -- 9 cycles of 2500 per iteration are spent in that wait.)
        RowIn R[3];
#ifndef CSI_FULL_HOISTP
#define CSI_FULL_HOISTP 1
#endif
        // FULL (no hoisting of the table by the compiler, see body): the ten input addresses read ONCE (CSI_FULL_HOISTP >= 2)
#ifndef CSI_FORCE_HOIST
#define CSI_FORCE_HOIST 0
#endif
        constexpr bool HIN = (FULL && CSI_FULL_HOISTP >= 2) || (FORCE && !FULL && (CSI_FORCE_HOIST & 1));
        const unsigned long hU = HIN ? T->P[FP_U_IN] : 0ul, hV = HIN ? T->P[FP_V_IN] : 0ul, hP = HIN ? T->P[FP_P] : 0ul, hH = HIN ? T->P[FP_H] : 0ul,
                            hA = HIN ? T->P[FP_A] : 0ul, h11 = HIN ? T->P[FP_S11_IN] : 0ul, h22 = HIN ? T->P[FP_S22_IN] : 0ul, h12 = HIN ? T->P[FP_S12_IN] : 0ul,
                            hUN = HIN ? T->P[FP_UN] : 0ul, hVN = HIN ? T->P[FP_VN] : 0ul;
        int rnext = rstart;                               // the row the NEXT load_row fetches (advance())
        auto load_row = [&](RowIn& Q) __attribute__((always_inline)) {
            if constexpr (HIN) {
                Q.u_p = ldg(hU, of + sf); Q.v_p = ldg(hV, oc + sc);
                Q.P_0 = ldg(hP, oc); Q.h_0 = ldg(hH, oc); Q.a_0 = ldg(hA, oc);
                Q.s11 = ldg(h11, oc); Q.s22 = ldg(h22, oc); Q.s12 = ldg(h12, of);
                Q.un_m = ldg(hUN, of - sf); Q.vn_x = ldg(hVN, AUF ? oc - sc : oc);
            } else {
            Q.u_p = ldg(T->P[FP_U_IN], of + sf); Q.v_p = ldg(T->P[FP_V_IN], oc + sc);
            Q.P_0 = ldg(T->P[FP_P], oc); Q.h_0 = ldg(T->P[FP_H], oc); Q.a_0 = ldg(T->P[FP_A], oc);
            Q.s11 = ldg(T->P[FP_S11_IN], oc); Q.s22 = ldg(T->P[FP_S22_IN], oc); Q.s12 = ldg(T->P[FP_S12_IN], of);
            Q.un_m = ldg(T->P[FP_UN], of - sf); Q.vn_x = ldg(T->P[FP_VN], AUF ? oc - sc : oc);
            }
            Q.mk = MASK ? ldub(T->P[FP_MASK], om) : 1u;
            if constexpr (LDSC) Q.ck = ldg(coef_row_base(rnext + 2), ck_lane);      // coefficient row (state row) + 2: in LDS one iteration before its first use
        };
        // oc / of / om: offsets of the row the NEXT load_row fetches; advance by one row, stopping at rend (the last
        // iterations re-read row rend: an unconditional prefetch keeps the number of loads in flight static)
        auto advance = [&]() __attribute__((always_inline)) {
            const bool more = rnext < rend;
            if (CSI_EXP_RINGALIAS) { rnext += more ? 1 : 0; oc = offc(alias_row(rnext)); of = offf(alias_row(rnext)); return; }
            oc += more ? sc : 0u; of += more ? sf : 0u;
            if (MASK) om += more ? sm : 0u;
            rnext += more ? 1 : 0;
        };
        int r = rstart;
        auto body = [&](auto KK) __attribute__((always_inline)) {
            constexpr int k = decltype(KK)::value;
            if (!CSI_PAIR_HOIST || FORCE || (FULL && CSI_FULL_FENCE)) asm volatile("" : "+s"(T));
            set_prio(k);
            fresh_lane();
            PROBE_START;
            // rows r and r + 1 are in flight (loads return in order; FORCE: the array loads of the previous iteration were
            // consumed there): wait until only row r + 1's remain
            if (CSI_PAIR_PD == 1) __builtin_amdgcn_s_waitcnt(0x0F70);
            else if (MASK) __builtin_amdgcn_s_waitcnt(0x0F70 | 11);
            else __builtin_amdgcn_s_waitcnt(0x0F70 | 10);
            PROBE(pacc0);
            const RowIn& C = R[k];
            // LDSC: this row's loads brought lane k's entry of coefficient row r + 2: into the window, behind this iteration's barrier
            // it is visible to both waves (first use: the corner coefficients of the producer's iteration r + 1)
            if constexpr (LDSC) { if (lane < FC_COUNT) ringc[(unsigned)((r + 2) & 7) * FC_COUNT + (unsigned)lane] = C.ck; }
            // stage B's results of two iterations ago (rows r - 4 / r - 5): read them now, store them after the prefetch
            const unsigned so = (unsigned)((r - rstart) & 1) * (5 * 64) + (unsigned)lane;
            constexpr int PW = 7 & ~CSI_PAIR_STORES;          // what the producer stores
            double o11 = 0, o22 = 0, o12 = 0, ofirst = 0, osecond = 0;
            if (PW & 1) { o11 = outr[so]; o22 = outr[so + 64]; o12 = outr[so + 128]; }
            if (PW & 2) ofirst = outr[so + 192];
            if (PW & 4) osecond = outr[so + 256];
            // FORCE: this row's forcing values are issued BEFORE the next row's prefetch -- the vector-memory counter is in order, so
            // the velocity phase, which consumes them, waits for them alone and not for the prefetched row behind them.  (Measured
            // at 2048^2, round 4: no change, 39.8 / 44.1 / 52.7 G on the OMIP-style / coupled / model.forcing configurations either
            // way -- those instantiations run at their own HBM traffic, 1.3x the plain kernel's.)
            Forcing FA;
            numbers(FA);
            if (FORCE) arrays(FA, offf(r - 1), AUF ? offc(r - 1) : offc(r));      // u points of row r-1, v points of row r-1 / r
            // (FULL: the next row's loads are issued between the phases of the step below, with the plane prefetch)
            if constexpr (!FULL) {
                advance();
                load_row(R[(k + CSI_PAIR_PD) % 3]);           // row r + CSI_PAIR_PD (clamped to rend)
            }
            flush(r - 4, o11, o22, o12, ofirst, osecond, Idx<PW>{}, Idx<AUF ? 1 : 0>{});
            fm::StressConst ks; stress_consts(ks);
            fm::VelConst kv; vel_consts(kv);
            const double m_0 = C.h_0 * T->K[FK_RHO] * C.a_0;
            // first / second velocity of a stage that ran row rr: u-first: u(rr-1), v(rr-1); v-first: v(rr), u(rr-1)
            const bool pa1 = WALLS && (AUF ? (lane_wu | wall_row(r - 1)) : (lane_wv | wall_vrow(r)));
            const bool pa2 = WALLS && (AUF ? (lane_wv | wall_vrow(r - 1)) : (lane_wu | wall_row(r - 1)));
            if (MASK) {
                const bool wi = lane_wv | wall_row(r);
                mhist = (mhist << 2) | (wi ? 3u : (C.mk ? 0u : 1u));
            }
            if constexpr (FULL) {
                static_assert(CSI_PAIR_PD == 1, "the per-point-metric producer prefetches one row ahead");
                auto mid = [&]() __attribute__((always_inline)) { advance(); load_row(R[(k + 1) % 3]); };
                A.template step<false>(T, ks, kv, r, C.u_p, C.v_p, C.P_0, m_0, C.a_0, C.s11, C.s22, C.s12, C.un_m, C.vn_x, true, r > rstart, pa1, pa2, mhist, FA,
                                       off2(r), c2s, off2(min(r + 1, rend)), off2(max(r - 1, row0)), mid);
            } else
            A.step(T, ks, kv, r, C.u_p, C.v_p, C.P_0, m_0, C.a_0, C.s11, C.s22, C.s12, C.un_m, C.vn_x, true, r > rstart, pa1, pa2, mhist, FA);
            // ---- hand-off to the consumer: sigma(r); u(r-1); v(r-1) [u first] or v(r) [v first] --------------------------
            {
                const unsigned s0 = rslot(r), s1 = rslot(r - 1);
                ring[s0 + RF_S11 * 64] = A.S11_0; ring[s0 + RF_S22 * 64] = A.S22_0; ring[s0 + RF_S12 * 64] = A.S12_0;
                if (AUF) { ring[s1 + RF_U * 64] = A.first; ring[s1 + RF_V * 64] = A.second; }
                else { ring[s0 + RF_V * 64] = A.first; ring[s1 + RF_U * 64] = A.second; }
                // the static fields of the rows B reaches two iterations from now: P, m, aice of row r; u^n of row r-1;
                // v^n of row r-1 (A u-first: B v-first reads it as row q) / row r (B u-first reads it as row q-1 one iteration later)
                if (single) {
                    // no second sub-step: the consumer stores this stage's results; the diagnostics of row r ride in the slots
                    // the static fields would use (alpha, zeta_f, zeta_c, Delta; the stage carries 2 zeta, Delta^2, 1 / Delta)
                    ring[s0 + RF_P * 64] = A.AL_0; ring[s0 + RF_M * 64] = 0.5 * A.zf; ring[s0 + RF_A * 64] = 0.5 * A.zc; ring[s0 + RF_VN * 64] = A.Dc * A.rDc;
                } else {
                ring[s0 + RF_P * 64] = C.P_0; ring[s0 + RF_M * 64] = m_0; ring[s0 + RF_A * 64] = C.a_0;
                if (PRE) { ring[s0 + RF_PF * 64] = A.Pf_0; ring[s0 + RF_RMC * 64] = A.rmc_0; ring[s0 + RF_RMF * 64] = A.rmf_0; }
                ring[s1 + RF_UN * 64] = C.un_m; ring[(AUF ? s1 : s0) + RF_VN * 64] = C.vn_x;
                }
                if constexpr (FRING) {
                    // the forcing values this step used: u points of row r - 1, v points of row r - 1 (A u-first) / r (A v-first)
                    if constexpr (EXTRA == 2) {
                        // wind drag / bottom stress arrays (no free drift there): top stress OR the air velocity's own component, its cross
                        // component, bottom stress OR the ocean velocity's own component, its cross component
                        const bool tt = T->I[FI_TOP_KIND] == 2, bt = T->I[FI_BOT_KIND] == 2;
                        ring[s0 + RF_FU_TAU * 64] = tt ? FA.t_tau_u : FA.t_we_u; ring[s0 + RF_FU_FD * 64] = FA.t_wb_u;
                        ring[s0 + RF_FU_WE * 64] = bt ? FA.b_tau_u : FA.b_we_u; ring[s0 + RF_FU_WB * 64] = FA.b_wb_u;
                        ring[s0 + RF_FV_TAU * 64] = tt ? FA.t_tau_v : FA.t_we_v; ring[s0 + RF_FV_FD * 64] = FA.t_wb_v;
                        ring[s0 + RF_FV_WE * 64] = bt ? FA.b_tau_v : FA.b_we_v; ring[s0 + RF_FV_WB * 64] = FA.b_wb_v;
                    } else {
                    // (EXTRA 1 without a mask: model.forcing arrays -- no free drift, no immersed-flux term there -- in the free-drift slots)
                    ring[s0 + RF_FU_TAU * 64] = FA.t_tau_u; ring[s0 + RF_FU_WE * 64] = FA.b_we_u; ring[s0 + RF_FU_WB * 64] = FA.b_wb_u; ring[s0 + RF_FU_FD * 64] = EXTRA == 1 ? FA.xc_u : FA.fd_u;
                    ring[s0 + RF_FV_TAU * 64] = FA.t_tau_v; ring[s0 + RF_FV_WE * 64] = FA.b_we_v; ring[s0 + RF_FV_WB * 64] = FA.b_wb_v; ring[s0 + RF_FV_FD * 64] = EXTRA == 1 ? FA.xc_v : FA.fd_v;
                    }
                }
                if constexpr (FULL) {
                    // the plane values this step consumed (v-point and corner planes of row r + 1, u-point and cell planes of row r)
                    // + 1 / Az at the cell (row r) and at the velocity points (u: row r - 1; v: row r - 1 for a u-first producer, r for a v-first one)
                    const unsigned p0 = (unsigned)k * (RP_FIELDS * 64) + lane_it;      // (k = (r - rstart) mod 3: the body is unrolled three times)
                    ringp[p0 + RP_DXV * 64] = A.DXV_p; ringp[p0 + RP_RDYV * 64] = A.RDYV_p; ringp[p0 + RP_RDXU * 64] = A.RDXU_p;
                    ringp[p0 + RP_DXF2 * 64] = A.DXF2_p; ringp[p0 + RP_DYF2 * 64] = A.DYF2_p; ringp[p0 + RP_RAZF * 64] = A.RAZF_p;
                    ringp[p0 + RP_DYU * 64] = A.DYU_0; ringp[p0 + RP_DYC2 * 64] = A.DYC2_0; ringp[p0 + RP_DXC2 * 64] = A.DXC2_0;
                    ringp[p0 + RP_RAZC * 64] = A.RAZC_0; ringp[p0 + RP_RAZU * 64] = A.RAZU_m; ringp[p0 + RP_RAZV * 64] = A.RAZV_x;
                }
                if (MASK) ringm[(unsigned)((r - rstart) & (RING_ROWS - 1)) * 64 + lane_it] = mhist & 3u;
            }
            A.shift(C.u_p, C.v_p, m_0, C.a_0);
            PROBE(pacc1);
#ifndef CSI_EXP_HALFBARRIER
#define CSI_EXP_HALFBARRIER 0      // timing experiment (RACY, wrong results): the two waves meet at every second row only
#endif
            if (!(CSI_EXP_HALFBARRIER && ((r - rstart) & 1)))
            __syncthreads();                              // row r is complete: the consumer may run its iteration r
            PROBE(pacc2);
        };
        if constexpr (LDSC) {
            // the window's first rows: rstart - 3 .. rstart + 1 (the consumer's first iteration reads down to rstart - 3, the producer's
            // first up to rstart + 1); the same wave reads them back in order, the consumer behind the first row barrier
#pragma unroll
            for (int d = -3; d <= 1; ++d) {
                const double v = ldg(coef_row_base(rstart + d), ck_lane);
                if (lane < FC_COUNT) ringc[(unsigned)((rstart + d) & 7) * FC_COUNT + (unsigned)lane] = v;
            }
        }
        load_row(R[0]);                                   // row rstart
        if (CSI_PAIR_PD == 2) {
            advance();
            load_row(R[1]);                               // row rstart + 1 (clamped)
        }
        for (;;) {
            body(Idx<0>{});
            if (++r > rend) break;
            body(Idx<1>{});
            if (++r > rend) break;
            body(Idx<2>{});
            if (++r > rend) break;
        }
        // drain: the consumer's last two rows (r = rend + 1: its iteration rend - 1 is complete; one more barrier for rend)
        for (int d = 0; d < 2; ++d) {
            const unsigned so = (unsigned)((r - rstart) & 1) * (5 * 64) + (unsigned)lane;
            flush(r - 4, outr[so], outr[so + 64], outr[so + 128], outr[so + 192], outr[so + 256], Idx<(7 & ~CSI_PAIR_STORES)>{}, Idx<AUF ? 1 : 0>{});
            if (d == 0) __syncthreads();
            ++r;
        }
#define PSTAGE A
        PROBE_END(w * 2);
#undef PSTAGE
        return;
    }

    auto publish = [&]() __attribute__((always_inline)) {
        if (PEER && pdirs) {
            // this tile is done with launch seq: its images are stored (release at system scope) and its halo reads are complete
            // (the producer's loads were consumed before its last barrier).  One lane per direction publishes seq in this tile's
            // slot of that neighbour's flag array.
            // tier >= 2: a system-scope release fence (writes this XCD's L2 back: + 6 us per launch on a 1024 x 512 tile) on top of
            // the write-through image stores and the drained store queue
            if ((CSI_PEER_EXP & 1) && T->I[FI_PTIER] < 2) __builtin_amdgcn_s_waitcnt(0x0F70);
            else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, ""); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }   // system scope: this wave's stores are in their -- possibly remote -- memory
            if ((lane < 8) && ((pdirs >> lane) & 1u)) {
                const int nW = T->I[FI_PSET], nE = T->I[FI_PSET + 1], nN = T->I[FI_PSET + 3];
                const bool xw = (lane == D_W) | (lane == D_SW) | (lane == D_NW), xe = (lane == D_E) | (lane == D_SE) | (lane == D_NE);
                const bool yn = (lane == D_N) | (lane == D_NW) | (lane == D_NE);
                const int xs = xe ? strip - (nstrips - nE) : strip, xn = xw ? nW : (xe ? nE : nstrips);
                const int ys = yn ? chunk - (nchunks - nN) : chunk;
                typedef const __attribute__((address_space(4))) unsigned long* sptr_t;
                unsigned long long* out = (unsigned long long*)((sptr_t)&T->P[FP_SLOT_OUT])[lane];
                __hip_atomic_store(out + (ys * xn + xs), seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    };

    if (single) {
        // ===== ONE sub-step: the consumer stores stage A's results of row r behind the barrier of row r ==================
        // (sigma(r); A u-first: u(r-1), v(r-1); A v-first: v(r), u(r-1) -- the row conventions flush() has for a stored stage
        // of that order; the table's store ranges are those of a single sub-step, csi_abi.hip)
        {
            const int d1 = AUF ? 1 : 0;
            fast_lo = max(rs_lo, max(r1_lo + d1, r2_lo + 1));
            fast_hi = min(rs_hi, min(r1_hi + d1, r2_hi + 1));
            if (wrap_lo | ylo_wall) fast_lo = max(fast_lo, HyW + 2);
            if (wrap_hi | yhi_wall) fast_hi = min(fast_hi, NyW - HyW);
            }
        for (int r = rstart; r <= rend; ++r) {
            __syncthreads();                              // the producer has finished row r
            const unsigned s0 = rslot(r), s1 = rslot(r - 1);
            const double v11 = ring[s0 + RF_S11 * 64], v22 = ring[s0 + RF_S22 * 64], v12 = ring[s0 + RF_S12 * 64];
            const double vfirst = AUF ? ring[s1 + RF_U * 64] : ring[s0 + RF_V * 64];
            const double vsecond = AUF ? ring[s1 + RF_V * 64] : ring[s1 + RF_U * 64];
            const double dal = ring[s0 + RF_P * 64], dzf = ring[s0 + RF_M * 64], dzc = ring[s0 + RF_A * 64], ddl = ring[s0 + RF_VN * 64];
            flush(r, v11, v22, v12, vfirst, vsecond, Idx<7>{}, Idx<AUF ? 0 : 1>{});
            if (write_diag) {
                if (((flags & L_RS) != 0) & (r >= rs_lo) & (r <= rs_hi)) {
                    const unsigned ocq = offc(r), ofq = offf(r);
                    const int yq = yimg(r);
                    put(5, ocq, (unsigned)yq * sc, r <= HyW, dx, dal, r, yq);
                    put(7, ofq, (unsigned)yq * sf, r <= HyW, dx, dzf, r, yq);
                    put(6, ocq, (unsigned)yq * sc, r <= HyW, dx, dzc, r, yq);
                    put(8, ocq, (unsigned)yq * sc, r <= HyW, dx, ddl, r, yq);
                }
            }
        }
        __syncthreads();
        publish();
        return;
    }

    // ===== CONSUMER: stage B = sub-step s + 1, rows q = r - 2, one iteration behind the producer ====================
    PROBE_DECL;
    Stage<UNI, !AUF, MASK, FORCE, CF, FULL, !(EXTRA == 1 && MASK)> B;
    B.lc = ringc;
    B.u_m = 0; B.u_0 = 0; B.v_m = 0; B.v_0 = 0; B.Xv_m = 0; B.Xv_0 = 0;
    B.a_mm = 0; B.a_m = 0; B.m_mm = 0; B.m_m = 0;
    B.XP_m = 0; B.Xm_m = 0; B.Xa_m = 0; B.Xe11_m = 0; B.Xe22_m = 0; B.Ye12_0 = 0; B.e12_0 = 0;
    B.XAL_m = 0; B.XS11L_m = 0; B.XS22L_m = 0; B.XW = 0; B.Wprev = 0;
    B.S11_mm = 0; B.S22_mm = 0; B.S12_mm = 0; B.AL_mm = 0; B.S11_m = 0; B.S22_m = 0; B.S12_m = 0; B.AL_m = 0;
    B.S11_0 = 0; B.S22_0 = 0; B.S12_0 = 0; B.AL_0 = 0; B.first = 0; B.second = 0;
    B.zc = 0; B.zf = 0; B.Dc = 0; B.rDc = 0; B.Pf_0 = 0; B.rmc_0 = 0; B.rmf_0 = 0;
    B.hoist_uniform(T);
    if constexpr (FULL) {
        B.hoist_planes(T, rcd);
        B.full_init(T, off2(max(rstart - 2, row0)), off2(max(rstart - 3, row0)), c2s);
        B.full_prefetch_vel(T, off2(max(rstart - 2, row0)), off2(max(rstart - 3, row0)));
    }
    // FULL: stage B's results of row q are stored at the START of the next iteration (fq, f11 .. fsecond): the wave's vector-memory
    // queue then holds, when the prefetched plane values are waited for at the top of an iteration, stores that are a whole
    // iteration old and loads that are half an iteration old -- nothing younger (the counter is in order: a younger store would
    // have to be waited for too)
    int fq = 0;
    bool fhave = false;
    double f11 = 0, f22 = 0, f12 = 0, ffirst = 0, fsecond = 0;
    // Stage B's row inputs all come from the ring (the producer read them from memory two or three iterations earlier):
    // the consumer issues no global loads (FORCE: except its forcing arrays), so it never waits for its own stores.
    const int rlo = rstart - 1;
    unsigned mhistB = 0xffffffffu;
    double vn_delay = 0.0;                                // B u-first: v^n of row q - 1 (read one iteration earlier as row q)
    double razv_delay = 1.0;                              // FULL, B u-first: 1 / Az at the v points of row q - 1 likewise (midB)
    double fvd_tau = 0.0, fvd_we = 0.0, fvd_wb = 0.0, fvd_fd = 0.0;      // FRING, B u-first: the v-point forcing values of row q - 1 likewise
    int r = rstart;
    auto bodyB = [&](auto KK) __attribute__((always_inline)) {
        if (!CSI_PAIR_HOIST || FORCE || (FULL && CSI_FULL_FENCE)) asm volatile("" : "+s"(T));
        set_prio(decltype(KK)::value);
        const int q = r - 2;
        PROBE_START;
        if (!(CSI_EXP_HALFBARRIER && ((r - rstart) & 1)))
        __syncthreads();                                  // the producer has finished row r
        PROBE(pacc0);
        // The consumer's first iterations of a tile (q < ja - 3: q = rstart - 2, rstart - 1 unless the tile starts at the low end of
        // the first sub-step's range) read ring rows the producer has not written yet: pure pipeline lag, whose results the next two
        // iterations -- which fill the window from rows ja - 2, ja - 1 -- overwrite completely (the first stress row is ja - 1, the
        // first velocity row ja).  They are skipped (round 5): the wave only keeps the barrier count, and the SIMD's other wave has
        // the vector ALU to itself meanwhile.  (Not with per-point metrics: their steps carry the plane prefetch of the next one.)
        if constexpr (!FULL && CSI_PAIR_SKIPB) { if (q < ja - 3) return; }
        fresh_lane();
        // ---- stage A's results from the ring: u, v of row q + 1, sigma of row q; P, m, aice of row q, u^n of row q - 1, v^n
        const unsigned s1 = rslot(r - 1), s2 = rslot(r - 2), s3 = rslot(r - 3);
        double bu_p = ring[s1 + RF_U * 64], bv_p = ring[s1 + RF_V * 64];
        const double s11 = ring[s2 + RF_S11 * 64], s22 = ring[s2 + RF_S22 * 64], s12 = ring[s2 + RF_S12 * 64];
        const double bP_0 = ring[s2 + RF_P * 64], bm_0 = ring[s2 + RF_M * 64], ba_0 = ring[s2 + RF_A * 64];
        if (PRE) { B.Pf_0 = ring[s2 + RF_PF * 64]; B.rmc_0 = ring[s2 + RF_RMC * 64]; B.rmf_0 = ring[s2 + RF_RMF * 64]; }
        const double bun = ring[s3 + RF_UN * 64], vn_new = ring[s2 + RF_VN * 64];
        const double bvn = AUF ? vn_new : vn_delay;       // B v-first: v^n(q); B u-first: v^n(q - 1)
        vn_delay = vn_new;
        const unsigned bmk = MASK ? ringm[(unsigned)((r - 2 - rstart) & (RING_ROWS - 1)) * 64 + lane_it] : 0u;
        if constexpr (FULL) {
            if (fhave) flush(fq, f11, f22, f12, ffirst, fsecond, Idx<(CSI_PAIR_STORES & 7)>{}, Idx<AUF ? 1 : 0>{});
        }
        PROBE(pacc1);
        fm::StressConst ks; stress_consts(ks);
        fm::VelConst kv; vel_consts(kv);
        if (WALLS) {
            // What the reference reads from mirror halos, stage B reads from A's results: v of the first cell
            // beyond an x wall is its neighbour's; u of the first row beyond a y wall is the wall row's (row 0 is
            // patched when row 1 arrives, row N + 1 copies row N).  Deeper halo cells only feed halo results.
            if (wave_mir) {      // (wave-uniform: interior strips skip the two lane shifts and the selects)
            const double vl = from_left(bv_p), vr = from_right(bv_p);
            // (a ValueBoundaryCondition wall reflects about 2 val instead: 2 val - v, 2 val - u)
            double ml = vr, mh = vl;
            if (wave_valx) {
                if (flags & L_VAL_LO) ml = 2 * T->K[FK_BCV] - vr;
                if (flags & L_VAL_HI) mh = 2 * T->K[FK_BCV + 1] - vl;
            }
            bv_p = (flags & L_MIR_LO) ? ml : ((flags & L_MIR_HI) ? mh : bv_p);
            }
            if (ylo_wall & (q == 0)) B.u_0 = uval_lo ? 2 * T->K[FK_BCU] - bu_p : bu_p;
            if (yhi_wall & (q == NyW)) bu_p = uval_hi ? 2 * T->K[FK_BCU + 1] - B.u_0 : B.u_0;
        }
        if (MASK) mhistB = (mhistB << 2) | ((q >= rlo) ? bmk : 3u);           // the producer's 2-bit code of row q
        Forcing FB;
        numbers(FB);
        if constexpr (FRING) {
            // From the ring.  u points of row q - 1: the producer's step q (slot s2).  v points: B v-first needs row q = the producer's
            // (u-first) step q + 1 (slot s1); B u-first needs row q - 1, which the producer (v-first) had at its step q - 1 -- three
            // iterations ago, its slot is being rewritten -- so the values of its step q are read now and used one iteration later
            // (fvd_*), like v^n above
            const double u0 = ring[s2 + RF_FU_TAU * 64], u1 = ring[s2 + RF_FU_WE * 64], u2 = ring[s2 + RF_FU_WB * 64], u3 = ring[s2 + RF_FU_FD * 64];
            double v0, v1, v2, v3;
            if (AUF) {
                v0 = ring[s1 + RF_FV_TAU * 64]; v1 = ring[s1 + RF_FV_WE * 64]; v2 = ring[s1 + RF_FV_WB * 64]; v3 = ring[s1 + RF_FV_FD * 64];
            } else {
                const double n0 = ring[s2 + RF_FV_TAU * 64], n1 = ring[s2 + RF_FV_WE * 64], n2 = ring[s2 + RF_FV_WB * 64], n3 = ring[s2 + RF_FV_FD * 64];
                v0 = fvd_tau; v1 = fvd_we; v2 = fvd_wb; v3 = fvd_fd;
                fvd_tau = n0; fvd_we = n1; fvd_wb = n2; fvd_fd = n3;
            }
            if constexpr (EXTRA == 2) {      // (the producer's mapping, above)
                if (T->I[FI_TOP_KIND] == 2) { FB.t_tau_u = u0; FB.t_tau_v = v0; } else { FB.t_we_u = u0; FB.t_we_v = v0; }
                FB.t_wb_u = u3; FB.t_wb_v = v3;
                if (T->I[FI_BOT_KIND] == 2) { FB.b_tau_u = u1; FB.b_tau_v = v1; } else { FB.b_we_u = u1; FB.b_we_v = v1; }
                FB.b_wb_u = u2; FB.b_wb_v = v2;
            } else {
                FB.t_tau_u = u0; FB.b_we_u = u1; FB.b_wb_u = u2;
                FB.t_tau_v = v0; FB.b_we_v = v1; FB.b_wb_v = v2;
                if (EXTRA == 1) { FB.xc_u = u3; FB.xc_v = v3; } else { FB.fd_u = u3; FB.fd_v = v3; }
            }
        } else if (FORCE) {
            // u points of row q-1, v points of row q (B v-first) / q-1 (B u-first); B's first rows of a tile only fill its
            // window: clamp their row to the array instead of running off it
            const int qm = max(q - 1, row0), qa = max(q, row0);
#ifndef CSI_EXP_NOCONSLOADS      // (TIMING EXPERIMENT ONLY, wrong results: the consumer's forcing loads left out -- the upper bound of what forcing
                                 //  values handed over through the ring could buy the per-point-metric instantiations)
            arrays(FB, offf(qm), AUF ? offc(qa) : offc(qm));
#endif
        }
        const bool pb1 = WALLS && (!AUF ? (lane_wu | wall_row(q - 1)) : (lane_wv | wall_vrow(q)));
        const bool pb2 = WALLS && (!AUF ? (lane_wv | wall_vrow(q - 1)) : (lane_wu | wall_row(q - 1)));
        if constexpr (FULL) {
            // (rows below the planes only fill the window: clamped)
            // The plane values of the NEXT step (row q + 1 = r - 1), read between the phases of this one: slot (r - 1 - rstart) mod 3 =
            // (k + 2) mod 3 -- the producer, one iteration ahead, is writing slot (k + 1) mod 3.  Rows it never ran (the first iterations
            // only fill the window): the ring's initial 1.0.  1 / Az at the v point: a v-first producer's slot X holds row X, the u-first
            // consumer needs row q of its step q + 1 = the slot read ONE iteration ago (razv_delay); a u-first producer's slot X holds
            // row X - 1, the v-first consumer needs row q + 1 = slot r, which the producer finished before this iteration's barrier.
            constexpr int kk = decltype(KK)::value;
            auto midB = [&]() __attribute__((always_inline)) {
                const unsigned pn = (unsigned)((kk + 2) % 3) * (RP_FIELDS * 64) + lane_it;
                B.N_DXV = ringp[pn + RP_DXV * 64]; B.N_RDYV = ringp[pn + RP_RDYV * 64]; B.N_RDXU = ringp[pn + RP_RDXU * 64];
                B.N_DXF2 = ringp[pn + RP_DXF2 * 64]; B.N_DYF2 = ringp[pn + RP_DYF2 * 64]; B.N_RAZF = ringp[pn + RP_RAZF * 64];
                B.N_DYU = ringp[pn + RP_DYU * 64]; B.N_DYC2 = ringp[pn + RP_DYC2 * 64]; B.N_DXC2 = ringp[pn + RP_DXC2 * 64];
                B.N_RAZC = ringp[pn + RP_RAZC * 64]; B.N_RAZU = ringp[pn + RP_RAZU * 64];
                if (AUF) B.N_RAZV = ringp[(unsigned)kk * (RP_FIELDS * 64) + lane_it + RP_RAZV * 64];
                else { const double x = ringp[pn + RP_RAZV * 64]; B.N_RAZV = razv_delay; razv_delay = x; }
            };
            B.template step<PRE, decltype(midB), true, true>(T, ks, kv, q, bu_p, bv_p, bP_0, bm_0, ba_0, s11, s22, s12, bun, bvn, q >= ja - 1, q >= ja, pb1, pb2, mhistB, FB,
                   off2(max(q, row0)), c2s, off2(max(q + 1, row0)), off2(max(q - 1, row0)), midB, off2(max(q, row0)));
            fq = q; f11 = B.S11_0; f22 = B.S22_0; f12 = B.S12_0; ffirst = B.first; fsecond = B.second; fhave = true;
        } else {
        B.template step<PRE>(T, ks, kv, q, bu_p, bv_p, bP_0, bm_0, ba_0, s11, s22, s12, bun, bvn, q >= ja - 1, q >= ja, pb1, pb2, mhistB, FB);
        flush(q, B.S11_0, B.S22_0, B.S12_0, B.first, B.second, Idx<(CSI_PAIR_STORES & 7)>{}, Idx<AUF ? 1 : 0>{});
        }
        if ((CSI_PAIR_STORES & 7) != 7) {
            const unsigned so = (unsigned)((r - rstart) & 1) * (5 * 64) + (unsigned)lane;
            if (!(CSI_PAIR_STORES & 1)) { outr[so] = B.S11_0; outr[so + 64] = B.S22_0; outr[so + 128] = B.S12_0; }
            if (!(CSI_PAIR_STORES & 2)) outr[so + 192] = B.first;
            if (!(CSI_PAIR_STORES & 4)) outr[so + 256] = B.second;
        }
        // diagnostics: last launch of the sub-cycle only, stored at once (with their halo images on periodic sides,
        // where the reference computes them from halo data: same values)
        if (write_diag) {
            if (((flags & L_RS) != 0) & (q >= rs_lo) & (q <= rs_hi)) {
                const unsigned ocq = offc(q), ofq = offf(q);
                const int yq = yimg(q);
                put(5, ocq, (unsigned)yq * sc, q <= HyW, dx, B.AL_0, q, yq);
                put(7, ofq, (unsigned)yq * sf, q <= HyW, dx, 0.5 * B.zf, q, yq);       // the stage carries 2 zeta
                put(6, ocq, (unsigned)yq * sc, q <= HyW, dx, 0.5 * B.zc, q, yq);
                put(8, ocq, (unsigned)yq * sc, q <= HyW, dx, B.Dc * B.rDc, q, yq);         // ... and Delta^2, 1 / Delta
            }
        }
        B.shift(bu_p, bv_p, bm_0, ba_0);
        PROBE(pacc2);
    };
    for (;;) {
        bodyB(Idx<0>{});
        if (++r > rend) break;
        bodyB(Idx<1>{});
        if (++r > rend) break;
        bodyB(Idx<2>{});
        if (++r > rend) break;
    }
    if constexpr (FULL) {
        if (fhave) flush(fq, f11, f22, f12, ffirst, fsecond, Idx<(CSI_PAIR_STORES & 7)>{}, Idx<AUF ? 1 : 0>{});
    }
    __syncthreads();                                      // the last row's results are in the out ring: the producer drains them
    publish();
#define PSTAGE B
    PROBE_END(w * 2 + 1);
#undef PSTAGE
#undef peer_abort
}

template <bool UNI, bool AUF, bool WALLS, bool MASK, bool FORCE, bool FD, int CF, bool FULL = false, bool PEER = false, int EXTRA = 0, bool DLD = false>
__global__ void __launch_bounds__(128, (FULL || (FORCE && (EXTRA != 1 || !MASK) && CSI_PAIR_FRING)) ? CSI_PAIR_FULL_WAVES
                                       : (UNI && !(EXTRA == 1 && MASK)) ? CSI_PAIR_UNI_WAVES : CSI_PAIR_WAVES) k_pair(const FusedTable* __restrict__ table, int nstrips, int nchunks, int rows,
                                                              int blocks_per_xcd, int write_diag, unsigned long long seq) {
    constexpr bool FRING = FORCE && !FULL && (EXTRA != 1 || !MASK) && CSI_PAIR_FRING;
    constexpr bool PRE = CSI_PAIR_PRE && !MASK && !FULL && !FRING && !(CSI_PAIR_LDSC_NOPRE && !UNI && !(EXTRA == 1 && MASK) && (CSI_PAIR_LDSC != 0));
    constexpr int RING_FIELDS = FRING ? 10 + RF_FORCING : (PRE ? 13 : 10);      // (FULL: ten fields; its plane values have a ring of their own, ringp)
    __shared__ double ring[RING_ROWS * RING_FIELDS * 64];
    __shared__ double ringp[FULL ? RP_ROWS * RP_FIELDS * 64 : 1];      // FULL: the plane values of three rows
    __shared__ unsigned ringm[MASK ? RING_ROWS * 64 : 1];
    __shared__ double outr[(CSI_PAIR_STORES & 7) != 7 ? 2 * 5 * 64 : 1];      // stage B's results on their way to the producer's stores
    __shared__ unsigned peer_abort_w;                      // PEER: the producer's wait has given up
#ifdef CSI_EXP_LDSPAD      // TIMING EXPERIMENT ONLY: so many more bytes of LDS per workgroup of the per-point-metric + array-forcing instantiations -- what a
                           // forcing ring there would cost in occupancy (12 KB more: three workgroups per CU instead of four; profiles/r06_full_force_ring_experiment.txt)
    __shared__ double lds_pad[(FULL && FORCE) ? CSI_EXP_LDSPAD / 8 : 1];
    if (FULL && FORCE) { lds_pad[threadIdx.x] = 0.0; asm volatile("" :: "v"(lds_pad[threadIdx.x])); }
#endif
    constexpr bool LDSC = !UNI && !FULL && !(EXTRA == 1 && MASK) && (CSI_PAIR_LDSC != 0);
    __shared__ double ringc[LDSC ? 8 * FC_COUNT : 1];      // per-row coefficients: an eight-row window (Stage::pc)
    // This workgroup's tile.  XCD-aware: blocks are dealt round-robin over the XCDs, each XCD walks one band of consecutive tiles.
    const int b = (int)blockIdx.x;
    int w = (b & 7) * blocks_per_xcd + (b >> 3);
    if (write_diag & 12) {
        // Tile activity (csi_activity.hip; write_diag bit 2: not one of the first two launches of a sub-cycle, not its last): only the LIVE
        // tiles run -- a quiescent tile (no ice mass in or around it) would store exactly what it stored two launches ago.  The live
        // tiles' numbers come from a list, dealt over the XCDs like the tiles themselves; the workgroups beyond the list leave at once.
        // (Peer-connected launches: the tiles of the direction sets are always on the list -- they publish their flags whatever they hold.)
        typedef const __attribute__((address_space(4))) int* iptr_t;
        // (bit 3: one of the first two launches -- the list that leaves out only the tiles quiescent from the start)
        iptr_t act = (iptr_t)((tptr_t)table)->P[(write_diag & 8) ? FP_ACT_LIVE0 : FP_ACT_LIVE];
        const int live = act[0], per = (live + 7) >> 3, k = b >> 3, p = (b & 7) * per + k;
        if ((k >= per) | (p >= live)) return;
        w = act[2 + p];
    } else if (w >= nstrips * nchunks) return;               // (uniform over the workgroup: both waves leave)
    if constexpr (PEER) {
        // an interior tile of a peer-connected launch (in no direction's set): the untiled instantiation's body
        const int chunk = w / nstrips, strip = w - chunk * nstrips;
        tptr_t T = (tptr_t)table;
        const bool pw = strip < T->I[FI_PSET], pe = strip >= nstrips - T->I[FI_PSET + 1], ps = chunk < T->I[FI_PSET + 2],
                   pn = chunk >= nchunks - T->I[FI_PSET + 3];
        const unsigned pd = ((pw ? 1u : 0u) | (pe ? 2u : 0u) | (ps ? 4u : 0u) | (pn ? 8u : 0u) | ((ps & pw) ? 16u : 0u) | ((ps & pe) ? 32u : 0u) |
                             ((pn & pw) ? 64u : 0u) | ((pn & pe) ? 128u : 0u)) & (unsigned)T->I[FI_PMASK];
        if (__builtin_amdgcn_readfirstlane((int)pd) == 0) {
            pair_body<UNI, AUF, WALLS, MASK, FORCE, FD, CF, FULL, false, EXTRA, false>(table, w, nstrips, nchunks, rows, blocks_per_xcd, write_diag, seq, ring, ringm, outr, &peer_abort_w, ringp, ringc);
            return;
        }
    }
    pair_body<UNI, AUF, WALLS, MASK, FORCE, FD, CF, FULL, PEER, EXTRA, DLD>(table, w, nstrips, nchunks, rows, blocks_per_xcd, write_diag, seq, ring, ringm, outr, &peer_abort_w, ringp, ringc);
}

}  // namespace fused

#if defined(CSI_PAIR_PROBE) && (!defined(CSI_PAIR_VARIANT) || CSI_PAIR_VARIANT == 0)
extern "C" int csi_debug_probe(unsigned long long* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(fused::g_probe), sizeof(unsigned long long) * 8192 * 16);
}
#endif
#if defined(CSI_PAIR_PROBE) && defined(CSI_PAIR_VARIANT) && CSI_PAIR_VARIANT >= 1
#define CSI_PROBE_NAME_(v) csi_debug_probe_v##v
#define CSI_PROBE_NAME(v) CSI_PROBE_NAME_(v)
extern "C" int CSI_PROBE_NAME(CSI_PAIR_VARIANT)(unsigned long long* dst) {      // (the probe buffer of this variant's translation unit)
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(fused::g_probe), sizeof(unsigned long long) * 8192 * 16);
}
#endif

// One translation unit per variant so that the instantiations compile in parallel.  CSI_PAIR_VARIANT:
// 0 plain, 1 walls, 2 walls + immersed mask, 3 walls + array-valued forcing, 4 walls + mask + array-valued forcing,
// 5 / 6: 3 / 4 with StressBalanceFreeDrift (free-drift velocity arrays); 7 / 8: 3 / 4 with model.forcing arrays / immersed flux BCs.
#ifndef CSI_PAIR_VARIANT
#define CSI_PAIR_VARIANT 0
#endif
#if CSI_PAIR_VARIANT == 0
#define CSI_PAIR_NAME launch_fused_pair_plain
#define CSI_PAIR_FLAGS false, false, false, false
#elif CSI_PAIR_VARIANT == 1
#define CSI_PAIR_NAME launch_fused_pair_walls
#define CSI_PAIR_FLAGS true, false, false, false
#elif CSI_PAIR_VARIANT == 2
#define CSI_PAIR_NAME launch_fused_pair_mask
#define CSI_PAIR_FLAGS true, true, false, false
#elif CSI_PAIR_VARIANT == 3
#define CSI_PAIR_NAME launch_fused_pair_force
#define CSI_PAIR_FLAGS true, false, true, false
#elif CSI_PAIR_VARIANT == 4
#define CSI_PAIR_NAME launch_fused_pair_mask_force
#define CSI_PAIR_FLAGS true, true, true, false
#elif CSI_PAIR_VARIANT == 5
#define CSI_PAIR_NAME launch_fused_pair_force_fd
#define CSI_PAIR_FLAGS true, false, true, true
#elif CSI_PAIR_VARIANT == 6
#define CSI_PAIR_NAME launch_fused_pair_mask_force_fd
#define CSI_PAIR_FLAGS true, true, true, true
#elif CSI_PAIR_VARIANT == 7      // 7 / 8: 3 / 4 with model.forcing arrays and / or immersed flux boundary conditions (EXTRA)
#define CSI_PAIR_NAME launch_fused_pair_force_x
#define CSI_PAIR_FLAGS true, false, true, false
#define CSI_PAIR_EXTRA 1
#elif CSI_PAIR_VARIANT == 8
#define CSI_PAIR_NAME launch_fused_pair_mask_force_x
#define CSI_PAIR_FLAGS true, true, true, false
#define CSI_PAIR_EXTRA 1
#elif CSI_PAIR_VARIANT == 9      // 9 / 10: 3 / 4 with array-valued wind drag and / or an explicit bottom stress given as arrays (EXTRA == 2)
#define CSI_PAIR_NAME launch_fused_pair_force_w
#define CSI_PAIR_FLAGS true, false, true, false
#define CSI_PAIR_EXTRA 2
#else
#define CSI_PAIR_NAME launch_fused_pair_mask_force_w
#define CSI_PAIR_FLAGS true, true, true, false
#define CSI_PAIR_EXTRA 2
#endif
#ifndef CSI_PAIR_EXTRA
#define CSI_PAIR_EXTRA 0
#endif
// common: number-valued top stress + bottom SemiImplicitStress with number-valued ocean velocities (Stage's CF; 2: zero ocean velocities); the
// array-forcing variants have one instantiation (kinds read from the table)
void CSI_PAIR_NAME(const FusedTable* dev_table, int metric, bool a_ufirst, int common, int nstrips, int nchunks, int rows,
                   int write_diag, unsigned long long seq, hipStream_t s) {
    const int nblocks = nstrips * nchunks;              // one workgroup (producer wave + consumer wave) per tile
    const int per_xcd = (nblocks + 7) / 8;
    dim3 grid((unsigned)(per_xcd * 8)), block(128);
    // seq != 0: the instantiation with the peer-flag protocol (tiles next to a connected side wait for / signal their neighbours)
    // ... bit 63 of seq: neighbours with other row strides (FI_PDLD; walls somewhere, so not in the plain variant)
    const bool dld = (seq >> 63) != 0;
    seq &= ~(1ull << 63);
    (void)dld;
#if CSI_PAIR_VARIANT >= 1
#define CSI_LAUNCH_PAIR_(U, A, C, F) do { if (seq && dld) hipLaunchKernelGGL((fused::k_pair<U, A, CSI_PAIR_FLAGS, C, F, true, CSI_PAIR_EXTRA, true>), grid, block, 0, s, dev_table, nstrips, nchunks, rows, per_xcd, write_diag, seq); \
                                          else if (seq) hipLaunchKernelGGL((fused::k_pair<U, A, CSI_PAIR_FLAGS, C, F, true, CSI_PAIR_EXTRA>), grid, block, 0, s, dev_table, nstrips, nchunks, rows, per_xcd, write_diag, seq); \
                                          else hipLaunchKernelGGL((fused::k_pair<U, A, CSI_PAIR_FLAGS, C, F, false, CSI_PAIR_EXTRA>), grid, block, 0, s, dev_table, nstrips, nchunks, rows, per_xcd, write_diag, seq); } while (0)
#else
#define CSI_LAUNCH_PAIR_(U, A, C, F) do { if (seq) hipLaunchKernelGGL((fused::k_pair<U, A, CSI_PAIR_FLAGS, C, F, true, CSI_PAIR_EXTRA>), grid, block, 0, s, dev_table, nstrips, nchunks, rows, per_xcd, write_diag, seq); \
                                          else hipLaunchKernelGGL((fused::k_pair<U, A, CSI_PAIR_FLAGS, C, F, false, CSI_PAIR_EXTRA>), grid, block, 0, s, dev_table, nstrips, nchunks, rows, per_xcd, write_diag, seq); } while (0)
#endif
#if CSI_PAIR_VARIANT <= 2
#define CSI_LAUNCH_PAIR(U, A) do { if (common == 2) CSI_LAUNCH_PAIR_(U, A, 2, false); else if (common) CSI_LAUNCH_PAIR_(U, A, 1, false); else CSI_LAUNCH_PAIR_(U, A, 0, false); } while (0)
#else
#define CSI_LAUNCH_PAIR(U, A) do { (void)common; CSI_LAUNCH_PAIR_(U, A, 0, false); } while (0)
#endif
    if (metric == 2) {
#if CSI_PAIR_VARIANT >= 1 && CSI_PAIR_VARIANT <= 2
        // per-point metrics: the forcing kinds fixed at compile time for the reference's default (number-valued top stress, ocean at
        // rest) as well: four wave-uniform branches, the explicit parts and a dozen scalar loads per stage-row less (+6 % measured on
        // the uniform-grid walls variant, CSI_PAIR_COMMON=0 / 2)
        if (common == 2) { if (a_ufirst) CSI_LAUNCH_PAIR_(false, true, 2, true); else CSI_LAUNCH_PAIR_(false, false, 2, true); }
        else { if (a_ufirst) CSI_LAUNCH_PAIR_(false, true, 0, true); else CSI_LAUNCH_PAIR_(false, false, 0, true); }
#elif CSI_PAIR_VARIANT >= 1
        if (a_ufirst) CSI_LAUNCH_PAIR_(false, true, 0, true); else CSI_LAUNCH_PAIR_(false, false, 0, true);
#endif
    } else if (metric == 0) { if (a_ufirst) CSI_LAUNCH_PAIR(true, true); else CSI_LAUNCH_PAIR(true, false); }
    else { if (a_ufirst) CSI_LAUNCH_PAIR(false, true); else CSI_LAUNCH_PAIR(false, false); }
#undef CSI_LAUNCH_PAIR
#undef CSI_LAUNCH_PAIR_
}

#if CSI_PAIR_VARIANT == 0
void launch_fused_pair_walls(const FusedTable*, int, bool, int, int, int, int, int, unsigned long long, hipStream_t);
void launch_fused_pair_mask(const FusedTable*, int, bool, int, int, int, int, int, unsigned long long, hipStream_t);
void launch_fused_pair_force(const FusedTable*, int, bool, int, int, int, int, int, unsigned long long, hipStream_t);
void launch_fused_pair_mask_force(const FusedTable*, int, bool, int, int, int, int, int, unsigned long long, hipStream_t);
void launch_fused_pair_force_fd(const FusedTable*, int, bool, int, int, int, int, int, unsigned long long, hipStream_t);
void launch_fused_pair_mask_force_fd(const FusedTable*, int, bool, int, int, int, int, int, unsigned long long, hipStream_t);
void launch_fused_pair_force_x(const FusedTable*, int, bool, int, int, int, int, int, unsigned long long, hipStream_t);
void launch_fused_pair_mask_force_x(const FusedTable*, int, bool, int, int, int, int, int, unsigned long long, hipStream_t);
void launch_fused_pair_force_w(const FusedTable*, int, bool, int, int, int, int, int, unsigned long long, hipStream_t);
void launch_fused_pair_mask_force_w(const FusedTable*, int, bool, int, int, int, int, int, unsigned long long, hipStream_t);
void launch_fused_pair(const FusedTable* dev_table, int metric, bool a_ufirst, bool walls, bool mask, bool force, bool free_drift, int extra,
                       int common, int nstrips, int nchunks, int rows, int write_diag, unsigned long long seq, hipStream_t s) {
    if (metric == 2) walls = true;      // per-point coefficients: the general variants only (none built without walls)
    if (extra == 2 && mask) launch_fused_pair_mask_force_w(dev_table, metric, a_ufirst, common, nstrips, nchunks, rows, write_diag, seq, s);
    else if (extra == 2) launch_fused_pair_force_w(dev_table, metric, a_ufirst, common, nstrips, nchunks, rows, write_diag, seq, s);
    else if (extra && mask) launch_fused_pair_mask_force_x(dev_table, metric, a_ufirst, common, nstrips, nchunks, rows, write_diag, seq, s);
    else if (extra) launch_fused_pair_force_x(dev_table, metric, a_ufirst, common, nstrips, nchunks, rows, write_diag, seq, s);
    else if (free_drift && mask) launch_fused_pair_mask_force_fd(dev_table, metric, a_ufirst, common, nstrips, nchunks, rows, write_diag, seq, s);
    else if (free_drift) launch_fused_pair_force_fd(dev_table, metric, a_ufirst, common, nstrips, nchunks, rows, write_diag, seq, s);
    else if (force && mask) launch_fused_pair_mask_force(dev_table, metric, a_ufirst, common, nstrips, nchunks, rows, write_diag, seq, s);
    else if (force) launch_fused_pair_force(dev_table, metric, a_ufirst, common, nstrips, nchunks, rows, write_diag, seq, s);
    else if (mask) launch_fused_pair_mask(dev_table, metric, a_ufirst, common, nstrips, nchunks, rows, write_diag, seq, s);
    else if (walls) launch_fused_pair_walls(dev_table, metric, a_ufirst, common, nstrips, nchunks, rows, write_diag, seq, s);
    else launch_fused_pair_plain(dev_table, metric, a_ufirst, common, nstrips, nchunks, rows, write_diag, seq, s);
}
#endif

}  // namespace csi
