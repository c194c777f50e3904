// evp_fused_common.h -- device helpers shared by the fused EVP kernels (evp_fused.hip: one sub-step per
// launch; evp_fused2.hip: two sub-steps per launch).
#pragma once
#include "csi_dev.h"
#include "csi_kernels.h"
#include "csi_fast_coef.h"
#include "evp_fast_math.h"

namespace csi {
namespace fused {

// value of `x` in lane - 1 / lane + 1 (DPP wave shifts).  bound_ctrl: the lane without a source (lane 0 / lane 63)
// reads 0 -- it is a ring lane whose results are never used -- so no copy of the old value is needed.
__device__ __forceinline__ double from_left(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x138, 0xf, 0xf, true);   // wave_shr:1
    hi = __builtin_amdgcn_mov_dpp(hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double from_right(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x130, 0xf, 0xf, true);   // wave_shl:1
    hi = __builtin_amdgcn_mov_dpp(hi, 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// Everything uniform (rheology constants, forcing, uniform-grid stencil coefficients, array bases, index ranges)
// lives in a small device table (FusedTable, csi_kernels.h) read through the constant address space (s_load,
// scalar cache) INSIDE the row loop.  As kernel arguments these ~250 dwords are hoisted into SGPRs, spill to
// VGPR lanes and come back as v_readlane (measured: 418 of 1015 VALU instructions per row iteration).
typedef const __attribute__((address_space(4))) FusedTable* tptr_t;

template <bool UNI>
__device__ __forceinline__ double coef(tptr_t T, int which, int j) {
    if (UNI) return T->K[FK_COEF0 + which];
    typedef const __attribute__((address_space(4))) double* vptr_t;
    vptr_t vec = (vptr_t)T->P[FP_COEF_VEC];
    return vec[(long)min(max(j, T->I[FI_COEF_JMIN]), T->I[FI_COEF_JMAX]) * FC_COUNT + which];   // row-major; ring rows may fall off the table
}

// Addressing: uniform base = parent array start, per-lane unsigned byte offset in a VGPR ->
// `global_load v, v_off, s[base]` with no 64-bit address arithmetic.  All Center-in-x fields share one leading
// dimension and all Face-in-x fields another (dense Oceananigans parents; checked on the host), so two running
// offsets (oc, of) address every field.
typedef __attribute__((address_space(1))) char* gptr_t;     // global address space: global_load / global_store, not flat
__device__ __forceinline__ double ldg(unsigned long base, unsigned off) {
    return *(const __attribute__((address_space(1))) double*)((gptr_t)base + off);
}
__device__ __forceinline__ unsigned ldub(unsigned long base, unsigned off) {
    return *(const __attribute__((address_space(1))) unsigned char*)((gptr_t)base + off);
}
__device__ __forceinline__ void stg(unsigned long base, unsigned off, double v) {
    *(__attribute__((address_space(1))) double*)((gptr_t)base + off) = v;
}

// CSI_METRIC_FULL: stencil coefficient `which` (C2_*, csi_fast_coef.h) of this lane's column at the row whose byte offset in
// a coefficient plane is `off` (planes have their own leading dimension: FI_C2_LD)
__device__ __forceinline__ double c2at(tptr_t T, int which, unsigned off) { return ldg(T->P[FP_C2_0 + which], off); }

// Velocity store: plain, or with halo images when this wave's tile touches an edge band (wave-uniform test).
__device__ __forceinline__ void store_vel(tptr_t T, int which_ptr, int which_ld, int img0, bool near_edge, int i, int j, double val) {
    FRef f;
    f.p = (double*)(__attribute__((address_space(1))) double*)T->P[which_ptr];      // element (0, 0), global memory
    f.ld = T->I[which_ld];
    if (near_edge) {
        GridDev g;
        g.Nx = T->I[FI_NX]; g.Ny = T->I[FI_NY]; g.Hx = T->I[FI_HX]; g.Hy = T->I[FI_HY];
        ImageSpec im;
        im.xlo = T->I[img0]; im.xhi = T->I[img0 + 1]; im.ylo = T->I[img0 + 2]; im.yhi = T->I[img0 + 3];
        im.ex = 0; im.ey = 0;        // velocities are never stored on the wall faces
        // ValueBoundaryCondition sides (IMG_VALUE): u on the y walls, v on the x walls
        im.vxlo = im.vxhi = im.vylo = im.vyhi = 0.0;
        if (img0 == FI_IMU) { im.vylo = T->K[FK_BCU]; im.vyhi = T->K[FK_BCU + 1]; }
        else { im.vxlo = T->K[FK_BCV]; im.vxhi = T->K[FK_BCV + 1]; }
        store_with_images(f, g, im, i, j, val);
    } else {
        f(i, j) = val;
    }
}


}  // namespace fused
}  // namespace csi
