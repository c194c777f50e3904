// csi_kernels.h -- host-side launchers of the HIP kernels (one translation unit per kernel family).
#pragma once
#include "csi_dev.h"
#include "csi_fast_coef.h"

namespace csi {

// STRICT mode: one kernel per reference @kernel (evp_strict.hip)
void launch_strict_init(const EvpDev& P, const Range& r, hipStream_t s);
void launch_strict_visc(const EvpDev& P, const Range& r, hipStream_t s);
void launch_strict_stress(const EvpDev& P, const Range& r, hipStream_t s);
void launch_strict_ustep(const EvpDev& P, const Range& r, const ImageSpec& im, hipStream_t s);
void launch_strict_vstep(const EvpDev& P, const Range& r, const ImageSpec& im, hipStream_t s);

// free-drift velocities of marginal ice (StressBalanceFreeDrift), once per sub-cycle, both modes (evp_strict.hip)
void launch_free_drift(const EvpDev& P, const Range& r, hipStream_t s);

// FAST mode (evp_fast.hip): fused viscosity + stress phase, u step, v step
void launch_fast_init(const EvpDev& P, const Range& r, hipStream_t s);
void launch_fast_stress(const EvpDev& P, const Range& r, const FastCoef& c, hipStream_t s);
void launch_fast_ustep(const EvpDev& P, const Range& r, const ImageSpec& im, const FastCoef& c, hipStream_t s);
void launch_fast_vstep(const EvpDev& P, const Range& r, const ImageSpec& im, const FastCoef& c, hipStream_t s);
// the fold band's launches (evp_fast.hip, k_band_stress / k_band_vel): inputs from P, outputs into other arrays.  stress: a, b, c take
// sigma11, sigma22, sigma12 and da, db, dc (may be null) rows >= j0 a second time; a velocity component: a (and da, rows >= j0).
struct BandOut { FRef a, b, c, da, db, dc; int j0; };
void launch_band_stress(const EvpDev& P, const Range& r, const FastCoef& c, const BandOut& o, hipStream_t s);
void launch_band_vel(const EvpDev& P, const Range& r, const ImageSpec& im, const FastCoef& c, bool u, const BandOut& o, hipStream_t s);
bool fast_supported(const EvpDev& P);

// fused sub-step (evp_fused.hip): stress + both velocities in one launch, double-buffered u, v, sigma.
// All uniform inputs live in a device table read through the constant address space.
// (order of use: adjacent entries are read by merged wide scalar loads)
enum : int { FK_EM2 = 0, FK_DMIN, FK_DMIN2, FK_AMIN2, FK_AMAX2, FK_HK1, FK_HKC, FK_HKF,      // stress phase
             FK_DT, FK_RDT, FK_MIN_MASS, FK_MIN_CONC, FK_RHO,                                     // velocity phases
             FK_DT2, FK_MIN_MASS2, FK_MIN_CONC2,      // 2 dt, 2 minimum_mass, 2 minimum_concentration (fm::vel_update_sum)
             FK_TOP_TAU_U, FK_TOP_TAU_V, FK_TOP_RHOCD, FK_TOP_UE, FK_TOP_VE,
             FK_BOT_TAU_U, FK_BOT_TAU_V, FK_BOT_RHOCD, FK_BOT_UE, FK_BOT_VE,
             FK_BCU, FK_BCV = FK_BCU + 2,      // ValueBoundaryCondition values: u at the y walls (low, high), v at the x walls
             FK_CA_DT = FK_BCV + 2, FK_RDMIN, FK_AMIN, FK_AMAX, FK_RAMIN, FK_RAMAX, FK_FCOR,
             FK_COEF0,
             FK_PCOEF0 = FK_COEF0 + FC_COUNT,      // the pair kernel's scaled copy of the uniform-grid coefficients (pair_coef_scale)
             FK_PK_EM2_8 = FK_PCOEF0 + FC_COUNT, FK_PK_DMIN2_16, FK_PK_HKF4, FK_PK_CA_DT4,      // e^-2 / 8, 16 Delta_min^2, 4 hkf, 4 x (c_alpha dt / 2): fm::stress_update_s
             FK_COUNT };
enum : int { FP_U_IN = 0, FP_V_IN, FP_S11_IN, FP_S22_IN, FP_S12_IN, FP_P, FP_H, FP_A, FP_UN, FP_VN,     // 10 inputs, contiguous
             FP_S11_OUT, FP_S22_OUT, FP_S12_OUT, FP_U_OUTP, FP_V_OUTP,                                // 5 outputs (parent addresses), contiguous
             FP_U_OUT, FP_V_OUT, FP_S11_OUT0, FP_S22_OUT0, FP_S12_OUT0,                               // (0,0)-offset addresses for stores with halo images
             FP_AL, FP_ZC, FP_ZF, FP_DL, FP_COEF_VEC, FP_PCOEF_VEC, FP_MASK,
             FP_FT_U, FP_FT_V, FP_FB_U, FP_FB_V, FP_FB_UBAR, FP_FB_VBAR, FP_FD_U, FP_FD_V,
             FP_FT_UBAR, FP_FT_VBAR,      // wind drag (a SemiImplicitStress on top with array-valued air velocities, in FP_FT_U / _V): their cross averages
             FP_FROW_U, FP_FROW_V,      // CSI_METRIC_FULL: per-row Coriolis parameter (device pointers, ptr[j] = row j)
             FP_F2U, FP_F2V,            // ... per-point Coriolis planes (parent addresses)
             FP_XC_U, FP_XC_V, FP_XD_U, FP_XD_V,   // EXTRA: model.forcing arrays; immersed-flux-BC stress divergence arrays (parent addresses)
             FP_C2_0,
             // halo images of the two-sub-steps kernel: for each of 9 arrays (sigma11, sigma22, sigma12, u, v, alpha, zeta_c, zeta_f, Delta)
             // and 8 directions (W E S N SW SE NW NE) the parent address of the array that receives the image -- the array itself
             // on untiled periodic sides and walls, the neighbouring tile's (mapped into this process) on peer-connected sides
             FP_IMG0 = FP_C2_0 + C2_COUNT,
             FP_SLOT_IN = FP_IMG0 + 72,     // peer flags: per direction, this rank's slots the neighbour there writes ...
             FP_SLOT_OUT = FP_SLOT_IN + 8,  // ... and that neighbour's slots this rank writes (device addresses)
             FP_PERR = FP_SLOT_OUT + 8,     // error word (a wait timed out)
             FP_ACT_LIVE,                   // tile activity (csi_activity.hip): device int array {live tiles, all tiles, list of the live tiles' numbers ...};
                                            // 0: every tile runs.  Read by launches whose write_diag has bit 2 set (run_fused: not the first two, not the last)
             FP_ACT_LIVE0,                  // ... the same for the FIRST TWO launches (write_diag bit 3): every tile but those that are quiescent from the
                                            // start -- u = v = +0.0 as well, no halo image to store --, 0: every tile runs
             FP_ROWPAD_ = FP_ACT_LIVE0 + (FP_C2_0 - FP_F2U),      // (room for FP_F2ROW_U / _V below FP_C2ROW_0)
             FP_C2ROW_0,                    // CSI_METRIC_FULL: the C2_COUNT planes as per-ROW vectors (ptr[parent row]), valid for the rows FP_RCSUM marks constant
             FP_F2ROW_U = FP_F2U + (FP_C2ROW_0 - FP_C2_0), FP_F2ROW_V,      // ... and the per-point Coriolis planes likewise (the SAME distance from FP_F2U / _V as the vectors from the planes: Stage::rcd_)
             FP_RCSUM = FP_C2ROW_0 + C2_COUNT,      // int prefix sums over parent rows: rcsum[t] = rows among parent rows [0, t) whose plane values are the same in every column (0: none marked)
             FP_COUNT };   // (FP_FT_* .. FP_FD_*: forcing arrays; FP_MASK: the uint8 mask; FP_C2_0 ..: the C2_COUNT per-point coefficient planes -- parent addresses)
enum : int { FI_NX = 0, FI_NY, FI_HX, FI_HY, FI_XLO, FI_XHI, FI_YLO, FI_YHI, FI_LD_C, FI_LD_F,
             FI_RS, FI_R1 = FI_RS + 4, FI_R1C = FI_R1 + 4, FI_R2 = FI_R1C + 4, FI_IMU = FI_R2 + 4, FI_IMV = FI_IMU + 4,
             FI_PRESSURE_KIND = FI_IMV + 4, FI_HAS_COR, FI_TOP_KIND, FI_BOT_KIND, FI_COEF_STRIDE, FI_COEF_JMIN, FI_COEF_JMAX,
             FI_DEC, FI_AJ0 = FI_DEC + 4, FI_AJ1, FI_IMS11, FI_IMS22 = FI_IMS11 + 4, FI_IMS12 = FI_IMS22 + 4, FI_MASK_LD = FI_IMS12 + 4, FI_BOT_UEK, FI_BOT_VEK, FI_FREE_DRIFT, FI_TOP_UEK, FI_TOP_VEK,
             FI_C2_LD, FI_FKIND,        // CSI_METRIC_FULL: leading dimension of the coefficient / Coriolis planes; f kind 0 number, 1 rows, 2 points
             FI_EXTRA,                  // EXTRA instantiations: bit 0 model.forcing arrays, bit 1 immersed flux boundary conditions
             FI_PEER,                   // 1: peer-connected sides (flags instead of a halo exchange, evp_fused2.hip)
             FI_PSET, FI_PMASK = FI_PSET + 4,   // tile sets next to the W / E / S / N side: strips, strips, chunks, chunks; directions with a neighbour
             FI_PWAIT,                  // slots to wait for per direction (= the size of the neighbour's opposite set)
             FI_PDLD = FI_PWAIT + 8,    // per direction x {Center in x, Face in x}: the neighbour's row stride minus this tile's, in bytes
             FI_PHASDLD = FI_PDLD + 16, // any of them non-zero (a Bounded x direction partitioned in x: the easternmost tile's Face fields are one column wider)
             FI_NYLO,                   // PEER: Ny of the neighbour beyond the LOW y side (rows of the image shift of this tile's low rows; a fold tile's own Ny is cut)
             FI_ELO, FI_EHI,            // rows of the first chunk / rows kept for the last chunk (0: `rows` / what is left): shorter tiles next to peer-connected y sides
             FI_WT,                     // 1: the pair kernel stores its results write-through (small grids: the launch does not end on a write-back of its dirty lines)
             FI_PTIER,                  // peer protocol tier (csi_set_peer_tier): 0 write-through images + drained stores + flags; 1: + a system-scope acquire
                                        // fence once the flags have been seen; 2: + a system-scope release fence before the flags are published
             FI_COUNT };
// peer flag arrays: slots per direction block; the LAST slot of a block is the abort word (a neighbour whose wait has given up sets
// it: this rank's waits stop at once)
constexpr int kPeerSlots = 1024;
struct FusedTable {
    double K[FK_COUNT];
    unsigned long P[FP_COUNT];
    int I[FI_COUNT + (FI_COUNT & 1)];
};
bool fused_supported(const EvpDev& P);
bool fused_supported_forcing(const EvpDev& P);   // everything but the mask restriction
void fused_fill_table(const EvpDev& P, const FastCoef& c, const FRef* in, const FRef* out,
                      const Range& rs, const Range& r1, const Range& r1c, const Range& r2,
                      const ImageSpec& imu, const ImageSpec& imv, FusedTable* host_table);
void launch_fused_substep(const FusedTable* dev_table, bool uniform, bool ufirst, int nstrips, int nchunks, int rows,
                          int write_diag, hipStream_t s);
// two sub-steps per launch (evp_fused2.hip); metric: 0 uniform coefficients, 1 per-row table, 2 per-point planes (CSI_METRIC_FULL).
// The table is one filled by fused_fill_table for the SECOND
// sub-step's store ranges (rs, r1, r2; r1c unused) plus: dec = the second sub-step's compute (stress) range that
// the wave tiles decompose, a_j0 / a_j1 = the first sub-step's stress rows, sigma image specs.
void fused_fill_pair_extra(const Range& dec, int a_j0, int a_j1, const ImageSpec& ims11, const ImageSpec& ims22,
                           const ImageSpec& ims12, FusedTable* host_table, int elo = 0, int ehi = 0, int write_through = 0);
// seq: launch number of the peer-flag protocol (0 on grids without peer-connected sides)
// extra: model.forcing arrays / immersed flux boundary conditions (the EXTRA instantiations; implies force)
// Tile activity (csi_activity.hip).  A tile of the two-sub-steps launch is QUIESCENT when the ice mass h rho aice is exactly zero in every cell
// of its owned box and one cell around it and no owned stress holds a negative zero: there sigma += (m > 0 ? ... : 0) and the velocity
// select's zero branch (elasto_visco_plastic_rheology.jl:343-347, split_explicit_momentum_equations.jl:217-228) make both sub-steps
// an exact fixed point of u, v, sigma -- every later launch would store what the launch two before it stored.  `act` receives
// {number of live tiles, number of tiles, the live tiles' numbers in ascending order}.
struct ActivityArgs {
    FRef h, a, s11, s22, s12, u, v;    // (0,0)-offset references; sigma, u, v: the CURRENT state
    double rho;
    Range dec;                         // the range the launch's tiles decompose (FI_DEC)
    int nstrips, nchunks, rows, elo, ehi;
    Range pc, pf;                      // index bounds of the Center-Center / Face-Face parents
    Range pu, pv;                      // ... of the u / v parents
    int Nx, Ny, Hx, Hy;                // the launch's grid: tiles within H of a side store halo images (never quiescent from the start)
    int pset[4], pmask;                // peer-connected launches (FI_PSET, FI_PMASK): the tiles of the direction sets wait for / publish flags
                                       // and store halo images into the neighbours' arrays -- always live
};
constexpr int kMaxActTiles = 16384;
// flags: 0 quiescent from the start (the first two launches may leave it out as well), 1 quiescent after two launches, 2 live.
// act / act0: {count, tiles, numbers ...} of the tiles with flag 2 / with flag >= 1.
// sample: device-visible pinned host words {seqlock, live, tiles, sample_id} the compaction writes for the host to read without an API call
void launch_tile_activity(const ActivityArgs& A, int* flags, int* act, int* act0, int* sample, int sample_id, hipStream_t s);
void launch_fused_pair(const FusedTable* dev_table, int metric, bool a_ufirst, bool walls, bool mask, bool force, bool free_drift, int extra,
                       int common_forcing, int nstrips, int nchunks, int rows, int write_diag, unsigned long long seq, hipStream_t s);
// stress divergence of the immersed FluxBoundaryConditions at every u / v point whose stencil stays inside the parents (evp_fast.hip)
void launch_immersed_div(const EvpDev& P, const FRef& xd_u, const FRef& xd_v, hipStream_t s);
// array-valued forcing the pair kernel takes: 0 none needed, 1 supported (FORCE variant), -1 not supported;
// StressBalanceFreeDrift (P.free_drift: free-drift velocity arrays P.ufd / P.vfd) also selects the FORCE variant
int pair_forcing_kind(const EvpDev& P);
bool evp_array_forcing(const EvpDev& P);
bool evp_ring_forcing(const EvpDev& P);
// bottom SemiImplicitStress with array-valued ocean velocities: the cross component averaged to the u / v points
// (ubar at v points from fu, vbar at u points from fv), once per sub-cycle (evp_fast.hip)
void launch_forcing_bars(const EvpDev& P, const FRef& ubar_v, const FRef& vbar_u, hipStream_t s, bool top = false);
void fused_fill_forcing_top(const EvpDev& P, const FRef& ubar_v, const FRef& vbar_u, FusedTable* t);
void fused_fill_forcing(const EvpDev& P, const FRef& ubar_v, const FRef& vbar_u, FusedTable* host_table);
void fused_fill_extra(const EvpDev& P, const FRef& xd_u, const FRef& xd_v, FusedTable* host_table);

// halo / masks / copies (halo.hip)
void launch_fill_halo(const FRef& f, const GridDev& g, const ImageSpec& im, hipStream_t s);
void launch_mask_center(const FRef& f, const GridDev& g, hipStream_t s);
void launch_mask_u(const FRef& f, const GridDev& g, hipStream_t s);
struct HaloBatch { FRef f[6]; ImageSpec im[6]; int n; };
void launch_fill_halo_batch(const HaloBatch& B, const GridDev& g, hipStream_t s);
void launch_fill_halo_xcolumns(const HaloBatch& B, const GridDev& g, hipStream_t s);
void launch_mask_v(const FRef& f, const GridDev& g, hipStream_t s);
struct CopyBatch { const double* src[6]; double* dst[6]; long n[6]; int count; int aligned16; };
void launch_copy_batch(const CopyBatch& B, hipStream_t s, int block = 256);      // block 64: beside a pair launch (one-wave workgroups find a slot)

// advection + tracer update (advect.hip)
struct AdvDev {
    GridDev g;
    FRef u, v, h, a, Gh, Ga, hm, am;
    FRef hs, Ghs, hsm;      // snow thickness as a third tracer (has_snow)
    int has_snow;
    int fill_images;        // tracer update: also store the local halo images (im) of h, aice [, hs]
    ImageSpec im;
    int scheme;
    double dt;
    int from_cache;
    // one RK stage in ONE launch (launch_advect_stage): tendencies of (h, a) -- the stage's input --, then the tracer update
    // base + dt G written to OTHER arrays (ho, ao; with their halo images), so that no block reads what another has updated
    FRef hb, ab, ho, ao;    // the update's base (Psi^-) and output
    int write_cache;        // first stage: also store the base values into hm, am (cache_current_fields!, with halo images)
    int nt;                 // tracers per thread of the tendency kernel: 0 by grid size, 1 / 2 forced (CSI_ADV_NT)
    int w32;                // WENO schemes: smoothness indicators / weights in single precision (csi_set_weno_weight_dtype; advect.hip)
};
void launch_tracer_tendencies(const AdvDev& A, int mode, hipStream_t s);
void launch_tracer_step(const AdvDev& A, hipStream_t s);
void launch_advect_stage(const AdvDev& A, int mode, hipStream_t s);

// slab thermodynamics (thermo.hip)
struct SlabDev {
    double k, rho_bulk, rho_pure, rho_l, c_l, c_i, L0, T0, liq_slope, liq_T0, S, hc, Tu, Qu, Qb;
    int top_flux_kind, bot_flux_kind;
    int top_bc_kind;         // 0 PrescribedTemperature, 1 MeltingConstrainedFluxBalance (numeric flux, closed form)
    double ice_salinity;
};
struct SnowDev {
    double k, rho, snowfall, Tu;
    int top_bc_kind;
};
struct LayeredOut { FRef mf_ice, mf_snow, mf_int, tu_ice, tu_snow; };   // optional outputs (p == nullptr: absent)
void launch_layered_step(const SlabDev& S, const SnowDev& W, const GridDev& g, const FRef& h, const FRef& a, const FRef& hs,
                         const LayeredOut& o, double dt, hipStream_t s);
void launch_slab_step(const SlabDev& S, const GridDev& g, const FRef& h, const FRef& a, const FRef& mf, int has_mf,
                      double dt, hipStream_t s);

}  // namespace csi
