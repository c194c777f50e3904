// evp_pair_stage.h -- one EVP sub-step as a row pipeline ("stage"), shared by the kernels that chain two stages:
// evp_fused2.hip runs both stages in one wave, stage B two rows behind stage A.
#pragma once
#include "evp_fused_common.h"

namespace csi {
namespace fused {

constexpr int P_LO = 4, P_HI = 59, P_W = P_HI - P_LO + 1;

// what fm::ext_stress needs at a u point and at a v point, for the top (t_) and bottom (b_) stress: tau (constant /
// array-valued stress), we (external velocity, own component), wb (cross component averaged to the point)
struct Forcing {
    double t_tau_u, t_we_u, t_wb_u, b_tau_u, b_we_u, b_wb_u, t_tau_v, t_we_v, t_wb_v, b_tau_v, b_we_v, b_wb_v;
    // StressBalanceFreeDrift: velocity of marginal ice at the u / v point being updated (fd: enabled, wave-uniform)
    double fd_u, fd_v;
    bool fd;
    // EXTRA instantiations: model.forcing.u / .v at the point (xc: added to the Coriolis term, sum_of_forcing_u / _v of
    // elasto_visco_plastic_rheology.jl:391-401) and the stress divergence of the immersed FluxBoundaryConditions at the point
    // (xd: added to d_j sigma_ij, ice_stress_divergence.jl:65-123; evaluated once per sub-cycle into library arrays)
    double xc_u, xd_u, xc_v, xd_v;
    int extra;           // bit 0: xc present, bit 1: xd present (0 in every other instantiation: the terms vanish at compile time)
};

struct RowIn { double u_p, v_p, P_0, h_0, a_0, s11, s22, s12, un_m, vn_x; unsigned mk; double ck; };      // ck: lane k's entry of a per-row coefficient row (Stage::LDSC)

// One sub-step as a row pipeline.  step(r) consumes row r of P, m, a, sigma, rows r+1 of u, v and produces
// sigma(r) and  UFIRST: u(r-1) ["first"], v(r-1) ["second"]   /   v first: v(r) ["first"], u(r-1) ["second"].
// The arithmetic (operations and their order) is that of evp_fused.hip's loop body.
// MASK: immersed boundary (GridFittedBoundary).  mh carries, per lane, two bits per row -- bit 0: the cell is inactive
// (immersed or beyond a wall), bit 1: it is beyond a wall -- for rows r, r-1, r-2 at bit positions 0, 2, 4.  As in
// evp_fast.hip (k_ustep / k_vstep): stresses of immersed cells / corners are zero in the divergence
// (ice_stress_divergence.jl:57-123), faces next to an inactive cell are peripheral nodes.
__device__ __forceinline__ unsigned left_bits(unsigned x) { return (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0x138, 0xf, 0xf, true); }

// TIGHT: the constants of each phase (strain / stress / velocities) are (re)loaded from the table right before that phase
// (an empty asm ties the loads to that point): the variants with walls, masks and array forcing otherwise hold more
// scalars than there are scalar registers and spill them to vector lanes.
// CF ("common forcing"): the external stresses are known at compile time to be a number-valued top stress (kind 0 / 1) and
// a bottom SemiImplicitStress with number-valued ocean velocities (kind 3) -- the bench and most stand-alone runs --, so
// the four wave-uniform kind branches per stage-row disappear.  CF == 2: those ocean velocities are zero (the reference's
// default: ZeroField), fm::ext_stress_rest.
// FULL: orthogonal curvilinear grid (CSI_METRIC_FULL): the strain rates and stress divergences are the per-POINT stencils
// of evp_fast.hip's k_*2 kernels (same operations, same order: bit-identical), their coefficients loaded per lane from
// the C2_* planes at byte offset o2 (row r; o2 - s2 / o2 + s2: rows r - 1 / r + 1); UNI must be false.
// CSI_METRIC_FULL: strain rates of cell (i, row of `o`) from u(i, row), v(i, row), v(i, row + 1) / of corner (i, row of `o`) from
// u(i, row), u(i, row - 1), v(i, row): the operations of evp_fast.hip's strain_cell2 / strain_corner2 (the neighbouring
// column's products come over the lane shift); s2: row pitch of the planes in bytes
// rcd: 0, or the distance from the planes' table entries to their per-row vectors' (FP_C2ROW_0 - FP_C2_0: a tile whose rows are all
// marked row-constant reads the SAME values from vectors that stay in the caches; evp_fused2.hip `rowc`)
__device__ __forceinline__ void full_cell(tptr_t T, unsigned o, unsigned s2, double u_0, double v_0, double v_p, double& e11, double& e22, int rcd = 0) {
    const double dyu = c2at(T, C2_DYU + rcd, o), dxv_s = c2at(T, C2_DXV + rcd, o), dxv_n = c2at(T, C2_DXV + rcd, o + s2);
    const double Uy_w = dyu * u_0, Ur_w = fm::rcp(dyu) * u_0;
    const double Vx_s = dxv_s * v_0, Vr_s = fm::rcp(dxv_s) * v_0;
    const double Vx_n = dxv_n * v_p, Vr_n = fm::rcp(dxv_n) * v_p;
    fm::full_strain_cell(from_right(Uy_w), Uy_w, Vx_n, Vx_s, from_right(Ur_w), Ur_w, Vr_n, Vr_s,
                         c2at(T, C2_DYC2 + rcd, o), c2at(T, C2_DXC2 + rcd, o), c2at(T, C2_RAZC + rcd, o), e11, e22);
}
__device__ __forceinline__ double full_corner(tptr_t T, unsigned o, unsigned s2, double u_n, double u_s, double v_e, int rcd = 0) {
    const double Ux_n = c2at(T, C2_RDXU + rcd, o) * u_n, Ux_s = c2at(T, C2_RDXU + rcd, o - s2) * u_s;
    const double Vy_e = c2at(T, C2_RDYV + rcd, o) * v_e;
    return fm::full_strain_corner(Ux_n, Ux_s, Vy_e, from_left(Vy_e), c2at(T, C2_DXF2 + rcd, o), c2at(T, C2_DYF2 + rcd, o), c2at(T, C2_RAZF + rcd, o));
}

template <bool UNI, bool UFIRST, bool MASK, bool TIGHT = false, int CF = 0, bool FULL = false, bool HOLDK = true>
struct Stage {
    double u_m, u_0, v_m, v_0, Xv_m, Xv_0;
    double a_mm, a_m, m_mm, m_m;
    double XP_m, Xm_m, Xa_m, Xe11_m, Xe22_m, Ye12_0, e12_0;
    double XAL_m, XS11L_m, XW, Wprev;
    double XS22L_m;      // FULL: sigma22 of row r-1 from the left neighbour (the u equation sees sigma22 on such grids)
    // FULL: the metric planes of the rows in the window -- every plane value is loaded ONCE, when its row enters (the v-point
    // and corner planes as row r + 1, the u-point and cell planes as row r), and kept until the velocity phase has used it
    // as row r - 1 / r - 2 (the kernel on such grids is bound by the number of loads).  (The products
    // with u, v of row r are formed again each row: stage B patches u_0 at walls between two steps.)
    double DXV_0, RDXV_0, RDYV_0, RDXU_0, DXF2_0, DYF2_0, RAZF_0;
    double DXV_m, RDXV_m, RDYV_m, RDXU_m, DXF2_m, DYF2_m, DYU_m, RDYU_m, DYC2_m, DXC2_m, DXC2_mm;
    double DXV_p, RDXV_p, RDYV_p, RDXU_p, DXF2_p, DYF2_p, RAZF_p, DYU_0, RDYU_0, DYC2_0, DXC2_0;      // pending (shift)
#ifndef CSI_FULL_HOISTP
#define CSI_FULL_HOISTP 1
#endif
#if CSI_FULL_HOISTP >= 1
    // FULL: the twelve plane base addresses, read from the table ONCE (the row loop's reload fence on the table pointer does not reach
    // them): two wide scalar loads and their waits per stage-row less
    unsigned long c2p[12];
    // rcd: 0 (the planes), or FP_C2ROW_0 - FP_C2_0 for a tile whose rows are all row-constant (their per-row vectors; the per-point
    // Coriolis planes' vectors sit the same distance from FP_F2U / FP_F2V: csi_kernels.h)
    int rcd_ = 0;
    __device__ __forceinline__ void hoist_planes(tptr_t T, int rcd = 0) {
        rcd_ = rcd;
#pragma unroll
        for (int k = 0; k < 12; ++k) c2p[k] = T->P[FP_C2_0 + k + rcd];
    }
    __device__ __forceinline__ double c2m(tptr_t, int which, unsigned off) const { return ldg_keep(c2p[which], off); }
#else
    int rcd_ = 0;
    __device__ __forceinline__ void hoist_planes(tptr_t, int rcd = 0) { rcd_ = rcd; }
    __device__ __forceinline__ double c2m(tptr_t T, int which, unsigned off) const { return c2at(T, which + rcd_, off); }
#endif
    __device__ __forceinline__ unsigned long f2u(tptr_t T) const { return T->P[FP_F2U + rcd_]; }
    __device__ __forceinline__ unsigned long f2v(tptr_t T) const { return T->P[FP_F2V + rcd_]; }
    double RAZC_0, RAZU_m, RAZV_x, FU_m, FV_x;       // this step's 1 / Az at the cell (row r), the u point (row r - 1), the v point (row r - 1 / r); f likewise

    // Uniform coefficients, round 5: the velocity phase's seven coefficients and the bottom drag constant live in VECTOR registers.
    // The kernel has no scalar register left for them (106 in use, 36-55 spilled to lanes outside the loops), so the compiler
    // re-read them from the table in every row -- three scalar loads per stage-row, each waited for on the spot (24-42 ns,
    // profiles/r01_microbenchmarks.md) by a wave that shares its SIMD with ONE other since the tile-count rule.  The opaque
    // asm pins the VGPR copy (a value the compiler cannot rematerialise from the table).
#ifndef CSI_PAIR_VK
#define CSI_PAIR_VK 1
#endif
    static constexpr bool VKC = UNI && !FULL && HOLDK && (CSI_PAIR_VK != 0);       // (k_pair's register budget of these instantiations: CSI_PAIR_UNI_WAVES)
    double VK_E, VK_FN, VK_FS, VK_FU, VK_Q2N, VK_K, VK_FV, VK_BRHO;
    __device__ __forceinline__ void hoist_uniform(tptr_t T) {
        if constexpr (VKC) {
            VK_E = T->K[FK_PCOEF0 + FC_E]; VK_FN = T->K[FK_PCOEF0 + FC_FN]; VK_FS = T->K[FK_PCOEF0 + FC_FS]; VK_FU = T->K[FK_PCOEF0 + FC_FU];
            VK_Q2N = T->K[FK_PCOEF0 + FC_Q2N]; VK_K = T->K[FK_PCOEF0 + FC_K]; VK_FV = T->K[FK_PCOEF0 + FC_FV]; VK_BRHO = T->K[FK_BOT_RHOCD];
            asm volatile("" : "+v"(VK_E), "+v"(VK_FN), "+v"(VK_FS), "+v"(VK_FU), "+v"(VK_Q2N), "+v"(VK_K), "+v"(VK_FV), "+v"(VK_BRHO));
        }
    }
    // Per-row coefficients from an eight-row window in LDS (`lc`, filled by the producer wave one row ahead of its first use:
    // evp_fused2.hip) instead of the table's scalar loads -- built in round 5 and LEFT OFF (CSI_PAIR_LDSC=1 enables it).  The scalar
    // loads come in five or six bursts per stage-row, each waited for on the spot (no scalar register is left to hold a row's twenty
    // values), and lat-lon tiles run 10-14 % behind uniform ones of the same shape where the arithmetic explains 4 %; but the
    // broadcast `ds_read`s that replace them wait too (and share lgkmcnt with the ring traffic), and the window's 1.3 KB end the
    // six-workgroups-per-CU fit of the 13-field ring: tiles +0-3 %, 2048^2 -3 %, 4096^2 -7 % (profiles/r05_row_coef_lds.txt).
#ifndef CSI_PAIR_LDSC
#define CSI_PAIR_LDSC 0
#endif
    static constexpr bool LDSC = !UNI && !FULL && HOLDK && (CSI_PAIR_LDSC != 0);
    const double* lc = nullptr;
    __device__ __forceinline__ double pc(tptr_t T, int which, int j) const {
        if constexpr (LDSC) return lc[(unsigned)(j & 7) * FC_COUNT + which]; else return pcoef<UNI>(T, which, j);
    }
    __device__ __forceinline__ double vk(tptr_t T, int which, int j, double held) const { if constexpr (VKC) return held; else return pc(T, which, j); }
    __device__ __forceinline__ double brho(tptr_t T) const { if constexpr (VKC) return VK_BRHO; else return T->K[FK_BOT_RHOCD]; }

    // FULL, round 4: the plane values a step needs are loaded DURING THE PREVIOUS step, between its stress phase and its velocity
    // phase (N_*: a phase of arithmetic and the row barrier lie between a load and its use; loaded at use -- round 3 -- every
    // iteration of both waves exposed a full memory latency at the top of its strain phase and, through the in-order vmcnt, gave
    // up the state rows' prefetch: 54 % of all wave-cycles waiting, profiles/r03_full_metric_kernel.md).  Same values, same
    // operations.  N_DXV .. N_RAZF: row r + 1 of the NEXT step; N_DYU .. N_RAZC: its row r.  (1 / Az at the velocity points and the
    // per-point Coriolis planes are issued at the top of the step that uses them, a phase ahead too; prefetched with the rest
    // they cost 28 more registers across the velocity phase: scratch spills, measured.)
    double N_DXV, N_RDYV, N_RDXU, N_DXF2, N_DYF2, N_RAZF, N_DYU, N_DYC2, N_DXC2, N_RAZC;
    double N_RAZU, N_RAZV, N_FU, N_FV;       // ALLPRE steps only (the consumer wave: no vector-memory instruction at the top of its iteration)
    // on: plane offset of the next step's row
    // the velocity phase's plane values of the next step too (om: plane offset of the row below the next step's row, clamped)
    __device__ __forceinline__ void full_prefetch_vel(tptr_t T, unsigned on, unsigned om) {
        N_RAZU = c2m(T, C2_RAZU, om); N_RAZV = c2m(T, C2_RAZV, UFIRST ? om : on);
        N_FU = 0.0; N_FV = 0.0;
        if (T->I[FI_FKIND] == 2) { N_FU = ldg(f2u(T), om); N_FV = ldg(f2v(T), UFIRST ? om : on); }
    }
    __device__ __forceinline__ void full_prefetch_f(tptr_t T, unsigned on, unsigned om) {
        N_FU = 0.0; N_FV = 0.0;
        if (T->I[FI_FKIND] == 2) { N_FU = ldg(f2u(T), om); N_FV = ldg(f2v(T), UFIRST ? om : on); }
    }
    __device__ __forceinline__ void full_prefetch(tptr_t T, unsigned on, unsigned s2) {
        N_DXV = c2m(T, C2_DXV, on + s2); N_RDYV = c2m(T, C2_RDYV, on + s2); N_RDXU = c2m(T, C2_RDXU, on + s2);
        N_DXF2 = c2m(T, C2_DXF2, on + s2); N_DYF2 = c2m(T, C2_DYF2, on + s2); N_RAZF = c2m(T, C2_RAZF, on + s2);
        N_DYU = c2m(T, C2_DYU, on); N_DYC2 = c2m(T, C2_DYC2, on); N_DXC2 = c2m(T, C2_DXC2, on); N_RAZC = c2m(T, C2_RAZC, on);
    }
    // FULL: before the first step(r): o = plane offset of row r, u_0 / v_0 = u, v of row r; om: of row r - 1 (clamped)
    __device__ __forceinline__ void full_init(tptr_t T, unsigned o, unsigned om, unsigned s2) {
        full_prefetch(T, o, s2);
        DXV_0 = c2m(T, C2_DXV, o); RDXV_0 = fm::rcp(DXV_0); RDYV_0 = c2m(T, C2_RDYV, o); RDXU_0 = c2m(T, C2_RDXU, o);
        DXF2_0 = c2m(T, C2_DXF2, o); DYF2_0 = c2m(T, C2_DYF2, o); RAZF_0 = c2m(T, C2_RAZF, o);
        // rows r - 1, r - 2: multiplied by zero stresses until their real values have been shifted in (any finite number)
        DXV_m = DXV_0; RDXV_m = RDXV_0; RDYV_m = RDYV_0; RDXU_m = RDXU_0; DXF2_m = DXF2_0; DYF2_m = DYF2_0;
        DYU_m = c2m(T, C2_DYU, o); RDYU_m = fm::rcp(DYU_m); DYC2_m = c2m(T, C2_DYC2, o); DXC2_m = c2m(T, C2_DXC2, o); DXC2_mm = DXC2_m;
    }
    double S11_mm, S22_mm, S12_mm, AL_mm, S11_m, S22_m, S12_m, AL_m;
#ifdef CSI_PAIR_PROBE
    unsigned long long sp0 = 0, sp1 = 0, sp2 = 0, sp3 = 0, spt = 0;      // cycles in: strain rates | stress phase | prefetch issue | velocity phase
#define SPROBE_START do { __builtin_amdgcn_sched_barrier(0); spt = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define SPROBE(acc) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_readcyclecounter(); \
                         __builtin_amdgcn_sched_barrier(0); acc += t_ - spt; spt = t_; } while (0)
#else
#define SPROBE_START do { } while (0)
#define SPROBE(acc) do { } while (0)
#endif
    // results of the last step()
    double S11_0, S22_0, S12_0, AL_0, zc, zf, Dc, rDc, first, second;
    // pending window updates
    double Xv_p, Xa_0, Xm_0, e12_p, XW_next;
    // functions of the static fields of row r that the OTHER stage of the pair needs for the same row: Pf (ice strength at
    // the corner), 1 / m at the cell and the corner (results of a step<false>; inputs of a step<true>)
    double Pf_0, rmc_0, rmf_0;

    // do_stress / do_vel (wave-uniform): the rows at the start of a tile only fill the window (strain rates and
    // x-averages); their stresses / velocities would never be used
    // PRE: Pf_0, rmc_0, rmf_0 were set by the caller (from the producer's results for the same row) instead of being
    // formed here: they depend on P and the ice mass only, which a sub-cycle does not change
    struct NoMid { __device__ __forceinline__ void operator()() const {} };
    // mid: called between the stress phase and the velocity phase (FULL: the producer issues the NEXT row's state loads there,
    // next to the plane prefetch, instead of at the top of the iteration -- they are then live across the velocity phase only,
    // not across the stress phase, where the register pressure peaks)
    // ALLPRE (FULL): 1 / Az at the velocity points and the per-point Coriolis planes were prefetched by the previous step as well
    // (full_prefetch_vel; o2nm: the row below the next step's row) -- the consumer wave, which has the registers for it
    // PLR (FULL): every plane value of a step comes from the pair kernel's plane ring: the caller's mid() functor, called between the
    // phases of the PREVIOUS step, has set N_DXV .. N_RAZV for this one (round 4 handed nine of them over at the top of the step and
    // loaded 1 / Az at the cell and at the velocity points from memory: a second fetch of three planes, every one a miss in the XCD's L2)
    template <bool PRE = false, class MID = NoMid, bool ALLPRE = false, bool PLR = false>
    __device__ __forceinline__ void step(tptr_t T, const fm::StressConst& ks_in, const fm::VelConst& kv_in, int r,
                                         double u_p, double v_p, double P_0, double m_0, double a_0,
                                         double s11, double s22, double s12, double un_m, double vn_x,
                                         bool do_stress, bool do_vel, bool per_first, bool per_second, unsigned mh,
                                         const Forcing& F, unsigned o2 = 0u, unsigned s2 = 0u, unsigned o2n = 0u, unsigned o2m = 0u, MID mid = MID(), unsigned o2nm = 0u) {
        SPROBE_START;
        Xa_0 = fm::sum2(from_left(a_0), a_0);               // SUMS too (Xa, Xm, XAL): fm::vel_update_sum
        Xv_p = fm::sum2(from_left(v_p), v_p);               // x-SUMS (Xv, Xe11, Xe22, Ye12, XP, XW): scaled once, in quarter()
        double e11_0, e22_0;
        if constexpr (FULL) {
            // strain_cell2 / strain_corner2 of evp_fast.hip: cell (i, r), corner (i, r + 1)
            // (full_cell / full_corner above with the plane values of rows already in the window taken from registers)
            DXV_p = N_DXV; RDXV_p = fm::rcp(DXV_p); RDYV_p = N_RDYV; RDXU_p = N_RDXU;
            DXF2_p = N_DXF2; DYF2_p = N_DYF2; RAZF_p = N_RAZF;
            DYU_0 = N_DYU; RDYU_0 = fm::rcp(DYU_0); DYC2_0 = N_DYC2; DXC2_0 = N_DXC2;
            RAZC_0 = N_RAZC;
            // what the velocity phase reads at rows r - 1 / r: issued here, a phase ahead of their use (o2m: row r - 1, clamped)
            if constexpr (ALLPRE) {
                RAZU_m = N_RAZU; RAZV_x = N_RAZV; FU_m = N_FU; FV_x = N_FV;
            } else if (do_vel) {
                RAZU_m = c2m(T, C2_RAZU, o2m); RAZV_x = c2m(T, C2_RAZV, UFIRST ? o2m : o2);
                if (T->I[FI_FKIND] == 2) { FU_m = ldg(f2u(T), o2m); FV_x = ldg(f2v(T), UFIRST ? o2m : o2); }
            }
            const double Uy_w = DYU_0 * u_0, Ur_w = RDYU_0 * u_0;
            fm::full_strain_cell(from_right(Uy_w), Uy_w, DXV_p * v_p, DXV_0 * v_0, from_right(Ur_w), Ur_w, RDXV_p * v_p, RDXV_0 * v_0,
                                 DYC2_0, DXC2_0, RAZC_0, e11_0, e22_0);
            const double Vy_e = RDYV_p * v_p;
            e12_p = fm::full_strain_corner8(RDXU_p * u_p, RDXU_0 * u_0, Vy_e, from_left(Vy_e), DXF2_p, DYF2_p, RAZF_p);      // 8 e12 (see below)
        } else {
        fm::strain_cell<UNI>(pc(T, FC_A, r), pc(T, FC_BN, r), pc(T, FC_BS, r), pc(T, FC_CN, r),
                        pc(T, FC_CS, r), from_right(u_0), u_0, v_p, v_0, e11_0, e22_0);
        e12_p = fm::strain_corner<UNI>(pc(T, FC_SN, r + 1), pc(T, FC_SS, r + 1), pc(T, FC_SV, r + 1),
                                  u_p, u_0, v_p, from_left(v_p));
        }
        // Scaled quantities (round 4): the corner strain rate is carried times 8 (e12_0, e12_p, Ye12_*: scaled coefficients, pcoef /
        // full_strain_corner8), the averages over four points stay sums (S11f, S22f, M4, the ice strength XP), the stress divergences
        // come out doubled (pcoef / full_div*_x2) -- exact powers of two, undone inside constants (fm::stress_update_s,
        // fm::vel_update_sum): ten multiplications / additions per stage-row less, every stored value bit for bit what it was.
        {
            const double Xe11_0 = fm::sum2(from_left(e11_0), e11_0), Xe22_0 = fm::sum2(from_left(e22_0), e22_0);
            const double Ye12_p = fm::sum2(e12_p, from_right(e12_p));
            Xm_0 = fm::sum2(from_left(m_0), m_0);
            const double e11f = fm::sum2(Xe11_m, Xe11_0);                     // 4 e11f
            const double e22f = fm::sum2(Xe22_m, Xe22_0);                     // 4 e22f
            const double e12c = 0.0625 * fm::sum2(Ye12_0, Ye12_p);            // 2 e12c (the corners' values are 8 e12)
            const double mf = fm::sum2(Xm_m, Xm_0);                           // 4 mf
            if constexpr (!PRE) {
                const double XP_0 = fm::sum2(from_left(P_0), P_0);
                Pf_0 = fm::sum2(XP_m, XP_0);                                  // 4 Pf
                XP_m = XP_0;
                rmc_0 = fm::rcp(m_0); rmf_0 = fm::rcp(mf);                    // 1 / m, 1 / (4 mf)
            }
            const double Pf = Pf_0;
            Xe11_m = Xe11_0; Xe22_m = Xe22_0; Ye12_0 = Ye12_p;
            SPROBE(sp0);
            if (do_stress) {
                fm::StressConst ks = ks_in;
                if (CF) ks.pressure_kind = 0;          // ReplacementPressure is part of the common configuration
                if (TIGHT) {
                    asm volatile("" : "+s"(T));
                    ks.em2 = T->K[FK_EM2]; ks.Dmin = T->K[FK_DMIN]; ks.Dmin2 = T->K[FK_DMIN2]; ks.amin2 = T->K[FK_AMIN2];
                    ks.amax2 = T->K[FK_AMAX2]; ks.hk1 = T->K[FK_HK1]; ks.pressure_kind = T->I[FI_PRESSURE_KIND];
                    ks.em2_8 = T->K[FK_PK_EM2_8]; ks.Dmin2_16 = T->K[FK_PK_DMIN2_16];
                }
                double kc, kf;      // kf: FOUR times c_alpha dt / (2 Az) at the corner
                if constexpr (FULL) { kc = T->K[FK_CA_DT] * RAZC_0; kf = T->K[FK_PK_CA_DT4] * RAZF_0; }
                else { kc = UNI ? T->K[FK_HKC] : T->K[FK_CA_DT] * pc(T, FC_RAZC, r); kf = UNI ? T->K[FK_PK_HKF4] : T->K[FK_PK_CA_DT4] * pc(T, FC_RAZF, r); }
                const fm::StressOut o = fm::stress_update_s(ks, e11_0, e22_0, e12_0, e11f, e22f, e12c, P_0, Pf, m_0, mf, rmc_0, rmf_0, kc, kf, s11, s22, s12);
                S11_0 = o.s11; S22_0 = o.s22; S12_0 = o.s12; AL_0 = o.alpha; zc = o.zc2; zf = o.zf2; Dc = o.xc; rDc = o.rDc;      // zc, zf: 2 zeta; Dc: Delta^2
            }
        }
        // stresses as the divergence sees them, peripheral flags
        double d11_0 = S11_0, d22_0 = S22_0, d12_0 = S12_0, d11_m = S11_m, d22_m = S22_m, d12_m = S12_m, d11_mm = S11_mm, d22_mm = S22_mm;
        double d11_mL = XS11L_m, d22_mL = XS22L_m;
        if (MASK) {
            const unsigned m0 = mh & 3u, m1 = (mh >> 2) & 3u, m2 = (mh >> 4) & 3u;
            const bool pcc_0 = m0 == 1u, pcc_m = m1 == 1u, pcc_mm = m2 == 1u;        // immersed, not beyond a wall
            unsigned c0 = m0 | m1, c1 = m1 | m2;                                      // corners of rows r, r-1: 4 cells
            c0 |= left_bits(c0); c1 |= left_bits(c1);
            const bool pff_0 = (c0 & 3u) == 1u, pff_m = (c1 & 3u) == 1u;
            d11_0 = pcc_0 ? 0.0 : S11_0; d22_0 = pcc_0 ? 0.0 : S22_0;
            d11_m = pcc_m ? 0.0 : S11_m; d22_m = pcc_m ? 0.0 : S22_m;
            d11_mm = pcc_mm ? 0.0 : S11_mm; d22_mm = pcc_mm ? 0.0 : S22_mm;
            d12_0 = pff_0 ? 0.0 : S12_0; d12_m = pff_m ? 0.0 : S12_m;
            d11_mL = from_left(d11_m);
            if constexpr (FULL) d22_mL = from_left(d22_m);
            const bool ia_0 = (m0 & 1u) != 0, ia_m = (m1 & 1u) != 0, ia_mm = (m2 & 1u) != 0;
            const bool ia_mL = (left_bits(m1) & 1u) != 0;
            const bool per_u = ia_m | ia_mL;                                          // u(i, r-1): cells (i, r-1), (i-1, r-1)
            per_first = UFIRST ? per_u : (ia_0 | ia_m);                               // v(i, r): cells (i, r), (i, r-1)
            per_second = UFIRST ? (ia_m | ia_mm) : per_u;                             // v(i, r-1): cells (i, r-1), (i, r-2)
        }
        // FULL: d_j sigma_1j at the u point of row ju (k_ustep2) / d_j sigma_2j at the v point of row jv (k_vstep2); o = the
        // coefficient offset of that row; f at the point: a number, a per-row value or a per-point plane
        // d_j sigma_1j at the u point of row r - 1; d_j sigma_2j at the v point of row r - 1 (`low`) or r: k_ustep2 / k_vstep2 of
        // evp_fast.hip (a neighbouring column's S, T, Z' is the neighbour's own product: one lane shift each)
        auto div1_full = [&]() __attribute__((always_inline)) {
            const double S_e = d11_m + d22_m, T_e = DYC2_m * (d11_m - d22_m);
            const double S_w = MASK ? from_left(S_e) : d11_mL + d22_mL;
            const double T_w = MASK ? from_left(T_e) : from_left(DYC2_m) * (d11_mL - d22_mL);
            return fm::full_div1_x2(DYU_m, RDYU_m, RDXU_m, RAZU_m, S_e, S_w, T_e, T_w, DXF2_0 * d12_0, DXF2_m * d12_m);
        };
        auto div2_full = [&](bool low) __attribute__((always_inline)) {
            if (low) {
                const double Zw = DYF2_m * d12_m;
                return fm::full_div2_x2(DXV_m, RDXV_m, RDYV_m, RAZV_x, d11_m + d22_m, d11_mm + d22_mm,
                                     DXC2_m * (d11_m - d22_m), DXC2_mm * (d11_mm - d22_mm), from_right(Zw), Zw);
            }
            const double Zw = DYF2_0 * d12_0;
            return fm::full_div2_x2(DXV_0, RDXV_0, RDYV_0, RAZV_x, d11_0 + d22_0, d11_m + d22_m,
                                 DXC2_0 * (d11_0 - d22_0), DXC2_m * (d11_m - d22_m), from_right(Zw), Zw);
        };
        auto f_full = [&](int which_row, int which_plane, unsigned o, int j) __attribute__((always_inline)) {
            const int kind = T->I[FI_FKIND];
            if (kind == 2) return which_plane == FP_F2U ? FU_m : FV_x;        // (prefetched by the previous step: full_prefetch)
            if (kind == 1) { typedef const __attribute__((address_space(4))) double* vptr_t; return ((vptr_t)T->P[which_row])[j]; }
            return T->K[FK_FCOR];
        };
        SPROBE(sp1);
        if constexpr (FULL) {
            // the next step's plane values: issued here, behind the stress phase, consumed after the velocity phase and the row barrier
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!PLR) full_prefetch(T, o2n, s2);
            // PLR (round 5: the consumer wave): ALL twelve plane values of the next step come from the pair kernel's plane ring -- the
            // caller's mid() reads them --, only the per-point Coriolis planes, where a grid has them, are still loaded here
            if constexpr (ALLPRE) { if constexpr (PLR) full_prefetch_f(T, o2n, o2nm); else full_prefetch_vel(T, o2n, o2nm); }
            mid();
            __builtin_amdgcn_sched_barrier(0);
        }
        SPROBE(sp2);
        fm::VelConst kv = kv_in;
        if (TIGHT) {
            asm volatile("" : "+s"(T));
            kv.dt = T->K[FK_DT2]; kv.rdt = T->K[FK_RDT]; kv.min_mass = T->K[FK_MIN_MASS2]; kv.min_conc = T->K[FK_MIN_CONC2];
        }
        if (!do_vel) {
            XW_next = XW;
        } else if (UFIRST) {
            const int j = r - 1;
            double W_0;
            {
                const double vbar = fm::quarter(Xv_m, Xv_0);
                double div;
                if constexpr (FULL) div = div1_full();
                else div = fm::div1(vk(T, FC_E, j, VK_E), vk(T, FC_FN, j, VK_FN), vk(T, FC_FS, j, VK_FS),
                                            d11_m, d11_mL, d12_0, d12_m);
                double ext, imt, exb, imb;
                if (CF) { ext = F.t_tau_u; imt = 0.0; } else fm::ext_stress(T->I[FI_TOP_KIND], F.t_tau_u, T->K[FK_TOP_RHOCD], F.t_we_u, F.t_wb_u, u_m, vbar, ext, imt);
                if (CF == 2) fm::ext_stress_rest(brho(T), u_m, vbar, exb, imb);
                else fm::ext_stress(CF ? 3 : T->I[FI_BOT_KIND], F.b_tau_u, brho(T), F.b_we_u, F.b_wb_u, u_m, vbar, exb, imb);
                double cor;
                if constexpr (FULL) cor = f_full(FP_FROW_U, FP_F2U, o2 - s2, j) * vbar;
                else cor = vk(T, FC_FU, j, VK_FU) * vbar;   // f = 0 without Coriolis (csi_abi.hip)
                if (F.extra & 1) cor += F.xc_u;
                if (F.extra & 2) div = fm::fma_(2.0, F.xd_u, div);      // (div is twice the divergence)
                W_0 = F.fd ? fm::vel_update_sum_fd(kv, u_m, un_m, Xm_m, Xa_m, XAL_m, div, cor, ext, imt, exb, imb, per_first, F.fd_u)
                           : fm::vel_update_sum(kv, u_m, un_m, Xm_m, Xa_m, XAL_m, div, cor, ext, imt, exb, imb, per_first);
            }
            const double XW_0 = fm::sum2(W_0, from_right(W_0));
            {
                const double ubar = fm::quarter(XW, XW_0);
                double div;
                if constexpr (FULL) div = div2_full(true);
                else div = fm::div2<UNI>(pc(T, FC_Q1N, j), vk(T, FC_Q2N, j, VK_Q2N), pc(T, FC_Q1S, j),
                                            pc(T, FC_Q2S, j), vk(T, FC_K, j, VK_K),
                                            d11_m, d22_m, d11_mm, d22_mm, from_right(d12_m), d12_m);
                double ext, imt, exb, imb;
                if (CF) { ext = F.t_tau_v; imt = 0.0; } else fm::ext_stress(T->I[FI_TOP_KIND], F.t_tau_v, T->K[FK_TOP_RHOCD], F.t_we_v, F.t_wb_v, v_m, ubar, ext, imt);
                if (CF == 2) fm::ext_stress_rest(brho(T), v_m, ubar, exb, imb);
                else fm::ext_stress(CF ? 3 : T->I[FI_BOT_KIND], F.b_tau_v, brho(T), F.b_we_v, F.b_wb_v, v_m, ubar, exb, imb);
                double cor;
                if constexpr (FULL) cor = -f_full(FP_FROW_V, FP_F2V, o2 - s2, j) * ubar;
                else cor = -vk(T, FC_FV, j, VK_FV) * ubar;
                if (F.extra & 1) cor += F.xc_v;
                if (F.extra & 2) div = fm::fma_(2.0, F.xd_v, div);
                second = F.fd ? fm::vel_update_sum_fd(kv, v_m, vn_x, m_mm + m_m, a_mm + a_m, AL_mm + AL_m, div, cor, ext, imt, exb, imb, per_second, F.fd_v)
                              : fm::vel_update_sum(kv, v_m, vn_x, m_mm + m_m, a_mm + a_m, AL_mm + AL_m, div, cor, ext, imt, exb, imb, per_second);
            }
            first = W_0;
            XW_next = XW_0;
        } else {
            double W_0;
            {
                const double ubar = fm::avg4(u_m, from_right(u_m), u_0, from_right(u_0));
                double div;
                if constexpr (FULL) div = div2_full(false);
                else div = fm::div2<UNI>(pc(T, FC_Q1N, r), vk(T, FC_Q2N, r, VK_Q2N), pc(T, FC_Q1S, r),
                                            pc(T, FC_Q2S, r), vk(T, FC_K, r, VK_K),
                                            d11_0, d22_0, d11_m, d22_m, from_right(d12_0), d12_0);
                double ext, imt, exb, imb;
                if (CF) { ext = F.t_tau_v; imt = 0.0; } else fm::ext_stress(T->I[FI_TOP_KIND], F.t_tau_v, T->K[FK_TOP_RHOCD], F.t_we_v, F.t_wb_v, v_0, ubar, ext, imt);
                if (CF == 2) fm::ext_stress_rest(brho(T), v_0, ubar, exb, imb);
                else fm::ext_stress(CF ? 3 : T->I[FI_BOT_KIND], F.b_tau_v, brho(T), F.b_we_v, F.b_wb_v, v_0, ubar, exb, imb);
                double cor;
                if constexpr (FULL) cor = -f_full(FP_FROW_V, FP_F2V, o2, r) * ubar;
                else cor = -vk(T, FC_FV, r, VK_FV) * ubar;
                if (F.extra & 1) cor += F.xc_v;
                if (F.extra & 2) div = fm::fma_(2.0, F.xd_v, div);
                W_0 = F.fd ? fm::vel_update_sum_fd(kv, v_0, vn_x, m_m + m_0, a_m + a_0, AL_m + AL_0, div, cor, ext, imt, exb, imb, per_first, F.fd_v)
                           : fm::vel_update_sum(kv, v_0, vn_x, m_m + m_0, a_m + a_0, AL_m + AL_0, div, cor, ext, imt, exb, imb, per_first);
            }
            const double XW_0 = fm::sum2(from_left(W_0), W_0);
            {
                const int j = r - 1;
                const double vbar = fm::quarter(XW, XW_0);
                double div;
                if constexpr (FULL) div = div1_full();
                else div = fm::div1(vk(T, FC_E, j, VK_E), vk(T, FC_FN, j, VK_FN), vk(T, FC_FS, j, VK_FS),
                                            d11_m, d11_mL, d12_0, d12_m);
                double ext, imt, exb, imb;
                if (CF) { ext = F.t_tau_u; imt = 0.0; } else fm::ext_stress(T->I[FI_TOP_KIND], F.t_tau_u, T->K[FK_TOP_RHOCD], F.t_we_u, F.t_wb_u, u_m, vbar, ext, imt);
                if (CF == 2) fm::ext_stress_rest(brho(T), u_m, vbar, exb, imb);
                else fm::ext_stress(CF ? 3 : T->I[FI_BOT_KIND], F.b_tau_u, brho(T), F.b_we_u, F.b_wb_u, u_m, vbar, exb, imb);
                double cor;
                if constexpr (FULL) cor = f_full(FP_FROW_U, FP_F2U, o2 - s2, j) * vbar;
                else cor = vk(T, FC_FU, j, VK_FU) * vbar;   // f = 0 without Coriolis (csi_abi.hip)
                if (F.extra & 1) cor += F.xc_u;
                if (F.extra & 2) div = fm::fma_(2.0, F.xd_u, div);      // (div is twice the divergence)
                second = F.fd ? fm::vel_update_sum_fd(kv, u_m, un_m, Xm_m, Xa_m, XAL_m, div, cor, ext, imt, exb, imb, per_second, F.fd_u)
                              : fm::vel_update_sum(kv, u_m, un_m, Xm_m, Xa_m, XAL_m, div, cor, ext, imt, exb, imb, per_second);
            }
            first = W_0;
            XW_next = XW_0;
        }
        SPROBE(sp3);
    }

    // slide the row window: row r becomes row r-1 (inputs of the step just done are passed again)
    __device__ __forceinline__ void shift(double u_p, double v_p, double m_0, double a_0) {
        u_m = u_0; u_0 = u_p; v_m = v_0; v_0 = v_p;
        Xv_m = Xv_0; Xv_0 = Xv_p;
        a_mm = a_m; a_m = a_0; m_mm = m_m; m_m = m_0;
        Xm_m = Xm_0; Xa_m = Xa_0;
        e12_0 = e12_p;
        S11_mm = S11_m; S22_mm = S22_m; S12_mm = S12_m; AL_mm = AL_m;
        S11_m = S11_0; S22_m = S22_0; S12_m = S12_0; AL_m = AL_0;
        XAL_m = fm::sum2(from_left(AL_0), AL_0); XS11L_m = from_left(S11_0);
        if constexpr (FULL) {
            XS22L_m = from_left(S22_0);
            DXV_m = DXV_0; RDXV_m = RDXV_0; RDYV_m = RDYV_0; RDXU_m = RDXU_0; DXF2_m = DXF2_0; DYF2_m = DYF2_0;
            DXV_0 = DXV_p; RDXV_0 = RDXV_p; RDYV_0 = RDYV_p; RDXU_0 = RDXU_p; DXF2_0 = DXF2_p; DYF2_0 = DYF2_p; RAZF_0 = RAZF_p;
            DYU_m = DYU_0; RDYU_m = RDYU_0; DYC2_m = DYC2_0; DXC2_mm = DXC2_m; DXC2_m = DXC2_0;
        }
        XW = XW_next;
        Wprev = first;
    }
};


}  // namespace fused
}  // namespace csi
