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
