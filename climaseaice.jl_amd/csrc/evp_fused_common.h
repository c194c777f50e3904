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

// the pair kernel's copy, some entries scaled (csi_fast_coef.h pair_coef_scale): it reads EVERY coefficient from this copy -- one base
// address, one row, the same few wide scalar loads per stage-row as before (read from both tables the per-row instantiations lost
// 17 %: lat-lon 2048^2 61.3 -> 51.6 G, measured and undone)
template <bool UNI>
__device__ __forceinline__ double pcoef(tptr_t T, int which, int j) {
    if (UNI) return T->K[FK_PCOEF0 + which];
    typedef const __attribute__((address_space(4))) double* vptr_t;
    vptr_t vec = (vptr_t)T->P[FP_PCOEF_VEC];
    return vec[(long)min(max(j, T->I[FI_COEF_JMIN]), T->I[FI_COEF_JMAX]) * FC_COUNT + which];
}

// Addressing: uniform base = parent array start, per-lane unsigned byte offset in a VGPR ->
// `global_load v, v_off, s[base]` with no 64-bit address arithmetic.  All Center-in-x fields share one leading
// dimension and all Face-in-x fields another (dense Oceananigans parents; checked on the host), so two running
// offsets (oc, of) address every field.
typedef __attribute__((address_space(1))) char* gptr_t;     // global address space: global_load / global_store, not flat
// Cache policy of the streaming loads -- an experiment knob, plain loads are the default.  Non-temporal loads
// (`global_load ... nt`) gain 3-5 % at exactly 2048^2 (62.4 -> 65.4 G on one box) and LOSE everywhere else measured:
// 1024 x 512 -4.7 %, 1024^2 -6 %, 1536^2 -5 %, 3072^2 -7 %, 4096^2 -13 % (the ring lanes a strip shares with its neighbour
// are evicted before the neighbour reads them); non-temporal stores lose 5.5 %, agent- / system-scope loads change
// nothing (profiles/r02d_experiments.md).  ldg_keep: loads that are re-read soon whatever the policy (the per-point
// coefficient planes: stage B reads what stage A read three rows earlier).
#ifndef CSI_NT_LOAD
#define CSI_NT_LOAD 0           // 0 plain, 1 non-temporal, 2 / 3: agent- / system-scope relaxed atomic loads (sc1 / sc0 sc1; experiments)
#endif
#ifndef CSI_NT_STORE
#define CSI_NT_STORE 0
#endif
__device__ __forceinline__ double ldg_keep(unsigned long base, unsigned off) {
    return *(const __attribute__((address_space(1))) double*)((gptr_t)base + off);
}
__device__ __forceinline__ double ldg(unsigned long base, unsigned off) {
#if CSI_NT_LOAD == 1
    return __builtin_nontemporal_load((const __attribute__((address_space(1))) double*)((gptr_t)base + off));
#elif CSI_NT_LOAD == 2
    return __builtin_bit_cast(double, __scoped_atomic_load_n((const __attribute__((address_space(1))) long*)((gptr_t)base + off), __ATOMIC_RELAXED, __MEMORY_SCOPE_DEVICE));
#elif CSI_NT_LOAD == 3
    return __builtin_bit_cast(double, __scoped_atomic_load_n((const __attribute__((address_space(1))) long*)((gptr_t)base + off), __ATOMIC_RELAXED, __MEMORY_SCOPE_SYSTEM));
#else
    return ldg_keep(base, off);
#endif
}
// write-through at agent scope (sc1): the line leaves the XCD's L2 when the store completes, not at the end of the launch
__device__ __forceinline__ void stg_agent(unsigned long base, unsigned off, double v) {
    __scoped_atomic_store_n((__attribute__((address_space(1))) long*)((gptr_t)base + off), __builtin_bit_cast(long, v), __ATOMIC_RELAXED, __MEMORY_SCOPE_DEVICE);
}
__device__ __forceinline__ unsigned ldub(unsigned long base, unsigned off) {
    return *(const __attribute__((address_space(1))) unsigned char*)((gptr_t)base + off);
}
__device__ __forceinline__ void stg(unsigned long base, unsigned off, double v) {
#if CSI_NT_STORE == 1
    __builtin_nontemporal_store(v, (__attribute__((address_space(1))) double*)((gptr_t)base + off));
#elif CSI_NT_STORE == 2      // write-through at agent scope (sc1): the line leaves the XCD's L2 during the launch, not at its end
    __scoped_atomic_store_n((__attribute__((address_space(1))) long*)((gptr_t)base + off), __builtin_bit_cast(long, v), __ATOMIC_RELAXED, __MEMORY_SCOPE_DEVICE);
#elif CSI_NT_STORE == 3      // ... at system scope (sc0 sc1)
    __scoped_atomic_store_n((__attribute__((address_space(1))) long*)((gptr_t)base + off), __builtin_bit_cast(long, v), __ATOMIC_RELAXED, __MEMORY_SCOPE_SYSTEM);
#else
    *(__attribute__((address_space(1))) double*)((gptr_t)base + off) = v;
#endif
}

// CSI_METRIC_FULL: stencil coefficient `which` (C2_*, csi_fast_coef.h) of this lane's column at the row whose byte offset in
// a coefficient plane is `off` (planes have their own leading dimension: FI_C2_LD)
__device__ __forceinline__ double c2at(tptr_t T, int which, unsigned off) { return ldg_keep(T->P[FP_C2_0 + which], off); }

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
