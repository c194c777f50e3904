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

// FAST mode (evp_fast.hip): fused viscosity + stress phase, u step, v step
void launch_fast_init(const EvpDev& P, const Range& r, hipStream_t s);
void launch_fast_stress(const EvpDev& P, const Range& r, const FastCoef& c, hipStream_t s);
void launch_fast_ustep(const EvpDev& P, const Range& r, const ImageSpec& im, const FastCoef& c, hipStream_t s);
void launch_fast_vstep(const EvpDev& P, const Range& r, const ImageSpec& im, const FastCoef& c, hipStream_t s);
bool fast_supported(const EvpDev& P);

// fused sub-step (evp_fused.hip): stress + both velocities in one launch, double-buffered u, v, sigma
struct FusedArgs {
    FRef u_in, v_in, s11_in, s22_in, s12_in;
    FRef u_out, v_out, s11_out, s22_out, s12_out;
    FRef h, a, P, un, vn;
    FRef al, zc, zf, Dl;                 // diagnostics, stored when write_diag
    GridDev g;
    Range rs, r1, r1c, r2;               // stress; first velocity (stored / computed incl. image ring); second velocity
    ImageSpec imu, imv;
    const double* consts;                // device table of uniform scalars (evp_fused.hip, K_* indices)
    int pressure_kind, has_cor, top_kind, bot_kind, write_diag;
    int nstrips, nchunks, rows, blocks_per_xcd;
};
bool fused_supported(const EvpDev& P);
void launch_fused_substep(const FusedArgs& A, const FastCoef& c, bool ufirst, hipStream_t s);
constexpr int FUSED_NCONST = 27 + FC_COUNT;
void fused_fill_consts(const EvpDev& P, const FastCoef& c, double* host_table);

// halo / masks / copies (halo.hip)
void launch_fill_halo(const FRef& f, const GridDev& g, const ImageSpec& im, hipStream_t s);
void launch_mask_center(const FRef& f, const GridDev& g, hipStream_t s);
void launch_mask_u(const FRef& f, const GridDev& g, hipStream_t s);
void launch_mask_v(const FRef& f, const GridDev& g, hipStream_t s);

// advection + tracer update (advect.hip)
struct AdvDev {
    GridDev g;
    FRef u, v, h, a, Gh, Ga, hm, am;
    int scheme;
    double dt;
    int from_cache;
};
void launch_tracer_tendencies(const AdvDev& A, int mode, hipStream_t s);
void launch_tracer_step(const AdvDev& A, hipStream_t s);

// slab thermodynamics (thermo.hip)
struct SlabDev {
    double k, rho_bulk, rho_pure, rho_l, c_l, c_i, L0, T0, liq_slope, liq_T0, S, hc, Tu, Qu, Qb;
    int top_flux_kind, bot_flux_kind;
};
void launch_slab_step(const SlabDev& S, const GridDev& g, const FRef& h, const FRef& a, const FRef& mf, int has_mf,
                      double dt, hipStream_t s);

}  // namespace csi
