// evp_fused3.hip -- THREE EVP sub-steps per launch on fully periodic, untiled grids with halo >= 6: the two-sub-steps
// kernel of evp_fused2.hip (csi::fused::k_pair) with one more stage in the chain.
//
// A workgroup is THREE 64-lane waves working on one (52-column strip) x (rows) tile: wave 0 runs stage A (sub-step s) down
// the rows from memory, wave 1 runs stage B (sub-step s + 1) two rows behind it on A's results, wave 2 runs stage C
// (sub-step s + 2) three more rows behind on B's and stores.  Everything a stage consumes -- the previous stage's new u, v,
// sigma rows and the static fields of the same rows (P, ice mass, aice, u^n, v^n, the ice strength at the corner and
// 1 / m at cell and corner) -- travels through a four-row LDS ring per hand-off (2 x 26 KB per workgroup: three workgroups
// per CU), one s_barrier per row.  u, v, sigma cross HBM once per THREE sub-steps: 120 B per cell per launch = 40 B per
// cell-update (k_pair: 60), for a ring of 6 + 6 lanes and 5 + 6 rows per tile instead of 4 + 4 and 3 + 3.
//
// Same stage code (evp_pair_stage.h) and arithmetic (evp_fast_math.h) as every other FAST path: bit-identical results.
// Periodic sides only (the owner of a cell stores its wrapped halo images, sigma included), no walls, masks, array forcing
// or per-point metrics: those configurations run k_pair.
#include "csi_dev.h"
#include "csi_kernels.h"
#include "evp_pair_stage.h"

namespace csi {
namespace fused {

constexpr int T_LO = 6, T_HI = 57, T_W = T_HI - T_LO + 1;
#ifndef CSI_TRIO_PRE
#define CSI_TRIO_PRE 0          // hand Pf, 1 / m over with the static fields (13 ring fields: 53 KB, 3 workgroups per CU) or not (10: 40 KB, 4)
#endif
constexpr int TR_ROWS = 4, TR_FIELDS = CSI_TRIO_PRE ? 13 : 10, TR_SIZE = TR_ROWS * TR_FIELDS * 64;
enum : int { TF_S11 = 0, TF_S22, TF_S12, TF_U, TF_V, TF_P, TF_M, TF_A, TF_UN, TF_VN, TF_PF, TF_RMC, TF_RMF };
template <int V> struct TIdx { static constexpr int value = V; };

template <bool UNI, bool AUF, int CF>
__global__ void __launch_bounds__(192, 3) k_trio(const FusedTable* __restrict__ table, int nstrips, int nchunks, int rows,
                                                 int blocks_per_xcd, int write_diag) {
    __shared__ double ring[2 * TR_SIZE];
    const int b = (int)blockIdx.x;
    const int w = (b & 7) * blocks_per_xcd + (b >> 3);      // XCD-aware, as k_pair
    if (w >= nstrips * nchunks) return;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // 0: stage A, 1: stage B, 2: stage C
    const int chunk = w / nstrips, strip = w - chunk * nstrips;
    const int lane = (int)(threadIdx.x & 63);
    tptr_t T = (tptr_t)table;

    const int Nx = T->I[FI_NX], Ny = T->I[FI_NY], Hx = T->I[FI_HX], Hy = T->I[FI_HY];
    const int i = T->I[FI_DEC + 0] - T_LO + strip * T_W + lane;
    const int ja = T->I[FI_DEC + 2] + chunk * rows;
    const int jb = min(ja + rows - 1, T->I[FI_DEC + 3]);
    const int ic = min(max(i, 1 - Hx), Nx + Hx);
    const unsigned loff = (unsigned)(ic - (1 - Hx)) * 8u;
    const int row0 = 1 - Hy;
    enum : unsigned { L_RS = 1, L_R1 = 2, L_R2 = 4 };
    unsigned flags = 0;
    {
        const bool own = (lane >= T_LO) & (lane <= T_HI) & (i <= T->I[FI_DEC + 1]);
        if (own & (i >= T->I[FI_RS + 0]) & (i <= T->I[FI_RS + 1])) flags |= L_RS;
        if (own & (i >= T->I[FI_R1 + 0]) & (i <= T->I[FI_R1 + 1])) flags |= L_R1;
        if (own & (i >= T->I[FI_R2 + 0]) & (i <= T->I[FI_R2 + 1])) flags |= L_R2;
    }
    // halo images of the periodic sides: column i in [1, H] is also stored at i + N, column in (N - H, N] at i - N; rows likewise
    int dx = 0;
    if ((i >= 1) & (i <= Hx)) dx = Nx * 8;
    else if ((i > Nx - Hx) & (i <= Nx)) dx = -Nx * 8;
    const bool wave_has_dx = __builtin_amdgcn_ballot_w64(dx != 0) != 0;
    const bool same = ((flags & L_RS) != 0) == ((flags & L_R1) != 0) && ((flags & L_RS) != 0) == ((flags & L_R2) != 0);
    const bool lanes_same = __builtin_amdgcn_ballot_w64(!same) == 0;
    const bool fast_plain = !wave_has_dx && lanes_same;
    const int rstart = max(ja - 5, T->I[FI_AJ0]);
    const int rend = min(jb + 5, T->I[FI_AJ1]);
    const int rlast = rend + 1;                              // stage C needs one more iteration than A and B
    const unsigned sc = (unsigned)T->I[FI_LD_C] * 8u, sf = (unsigned)T->I[FI_LD_F] * 8u;
    auto offc = [&](int j) __attribute__((always_inline)) { return loff + (unsigned)(j - row0) * sc; };
    auto offf = [&](int j) __attribute__((always_inline)) { return loff + (unsigned)(j - row0) * sf; };
    auto numbers = [&](Forcing& F) __attribute__((always_inline)) {
        F.t_tau_u = T->K[FK_TOP_TAU_U]; F.t_we_u = T->K[FK_TOP_UE]; F.t_wb_u = T->K[FK_TOP_VE];
        F.b_tau_u = T->K[FK_BOT_TAU_U]; F.b_we_u = T->K[FK_BOT_UE]; F.b_wb_u = T->K[FK_BOT_VE];
        F.t_tau_v = T->K[FK_TOP_TAU_V]; F.t_we_v = T->K[FK_TOP_VE]; F.t_wb_v = T->K[FK_TOP_UE];
        F.b_tau_v = T->K[FK_BOT_TAU_V]; F.b_we_v = T->K[FK_BOT_VE]; F.b_wb_v = T->K[FK_BOT_UE];
        F.fd_u = 0.0; F.fd_v = 0.0; F.fd = false;
    };
    auto stress_consts = [&](fm::StressConst& ks) __attribute__((always_inline)) {
        ks.em2 = T->K[FK_EM2]; ks.Dmin = T->K[FK_DMIN]; ks.Dmin2 = T->K[FK_DMIN2]; ks.rDmin = T->K[FK_RDMIN];
        ks.amin = T->K[FK_AMIN]; ks.amax = T->K[FK_AMAX]; ks.amin2 = T->K[FK_AMIN2]; ks.amax2 = T->K[FK_AMAX2];
        ks.ramin = T->K[FK_RAMIN]; ks.ramax = T->K[FK_RAMAX]; ks.hk1 = T->K[FK_HK1]; ks.pressure_kind = T->I[FI_PRESSURE_KIND];
    };
    auto vel_consts = [&](fm::VelConst& kv) __attribute__((always_inline)) {
        kv.dt = T->K[FK_DT2]; kv.rdt = T->K[FK_RDT]; kv.fcor = T->K[FK_FCOR]; kv.min_mass = T->K[FK_MIN_MASS2];
        kv.min_conc = T->K[FK_MIN_CONC2]; kv.has_cor = T->I[FI_HAS_COR];
    };
    auto rslot = [&](int j) __attribute__((always_inline)) { return (unsigned)((j - rstart) & (TR_ROWS - 1)) * (TR_FIELDS * 64) + (unsigned)lane; };
    // every stage hands the static fields of its rows on; the priorities follow the chain (the last stage stores: longest wave)
#ifndef CSI_TRIO_PRIO
#define CSI_TRIO_PRIO 1
#endif
#if CSI_TRIO_PRIO == 1
    if (wid == 2) __builtin_amdgcn_s_setprio(2); else if (wid == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#elif CSI_TRIO_PRIO == 2
    if (wid != 0) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
#elif CSI_TRIO_PRIO == 3
    if (wid == 2) __builtin_amdgcn_s_setprio(0); else if (wid == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(2);
#endif

    if (wid == 0) {
        // ===== stage A = sub-step s, rows rstart .. rend, from memory ==================================================
        Stage<UNI, AUF, false, false, CF, false> A;
        unsigned oc = offc(rstart), of = offf(rstart);
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
            fm::strain_cell<UNI>(coef<UNI>(T, FC_A, jm), coef<UNI>(T, FC_BN, jm), coef<UNI>(T, FC_BS, jm), coef<UNI>(T, FC_CN, jm),
                                 coef<UNI>(T, FC_CS, jm), from_right(A.u_m), A.u_m, A.v_0, A.v_m, e11_m, e22_m);
            A.e12_0 = fm::strain_corner<UNI>(coef<UNI>(T, FC_SN, rstart), coef<UNI>(T, FC_SS, rstart), coef<UNI>(T, FC_SV, rstart), A.u_0, A.u_m, A.v_0, from_left(A.v_0));
            A.Xe11_m = fm::sum2(from_left(e11_m), e11_m);
            A.Xe22_m = fm::sum2(from_left(e22_m), e22_m);
            A.Ye12_0 = fm::sum2(A.e12_0, from_right(A.e12_0));
            A.XAL_m = 0; A.XS11L_m = 0; A.XS22L_m = 0; A.XW = 0; A.Wprev = 0;
            A.S11_mm = 0; A.S22_mm = 0; A.S12_mm = 0; A.AL_mm = 0; A.S11_m = 0; A.S22_m = 0; A.S12_m = 0; A.AL_m = 0;
            A.S11_0 = 0; A.S22_0 = 0; A.S12_0 = 0; A.AL_0 = 0; A.first = 0; A.second = 0;
        }
        // both rings start clean: the later stages' first iterations read rows nobody wrote (their results only fill windows)
#pragma unroll
        for (int q = 0; q < 2 * TR_ROWS * TR_FIELDS; ++q) ring[q * 64 + lane] = 0.0;
        RowIn R[3];
        auto load_row = [&](RowIn& Q) __attribute__((always_inline)) {
            Q.u_p = ldg(T->P[FP_U_IN], of + sf); Q.v_p = ldg(T->P[FP_V_IN], oc + sc);
            Q.P_0 = ldg(T->P[FP_P], oc); Q.h_0 = ldg(T->P[FP_H], oc); Q.a_0 = ldg(T->P[FP_A], oc);
            Q.s11 = ldg(T->P[FP_S11_IN], oc); Q.s22 = ldg(T->P[FP_S22_IN], oc); Q.s12 = ldg(T->P[FP_S12_IN], of);
            Q.un_m = ldg(T->P[FP_UN], of - sf); Q.vn_x = ldg(T->P[FP_VN], AUF ? oc - sc : oc);
            Q.mk = 1u;
        };
        int rnext = rstart;
        auto advance = [&]() __attribute__((always_inline)) {
            const bool more = rnext < rend;
            oc += more ? sc : 0u; of += more ? sf : 0u;
            rnext += more ? 1 : 0;
        };
        int r = rstart;
        auto body = [&](auto KK) __attribute__((always_inline)) {
            constexpr int k = decltype(KK)::value;
#ifndef CSI_TRIO_PD
#define CSI_TRIO_PD 1
#endif
            if (CSI_TRIO_PD == 1) __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): row r has arrived (the prefetch of row r + 1 is issued below)
            else __builtin_amdgcn_s_waitcnt(0x0F70 | 10);                   // two rows ahead: row r + 1's ten loads may still be in flight
            const RowIn& C = R[k];
            advance();
            load_row(R[(k + CSI_TRIO_PD) % 3]);               // row r + CSI_TRIO_PD (clamped to rend)
            if (r <= rend) {
                fm::StressConst ks; stress_consts(ks);
                fm::VelConst kv; vel_consts(kv);
                const double m_0 = C.h_0 * T->K[FK_RHO] * C.a_0;
                Forcing FA;
                numbers(FA);
                A.step(T, ks, kv, r, C.u_p, C.v_p, C.P_0, m_0, C.a_0, C.s11, C.s22, C.s12, C.un_m, C.vn_x, true, r > rstart, false, false, 0u, FA);
                const unsigned s0 = rslot(r), s1 = rslot(r - 1);
                ring[s0 + TF_S11 * 64] = A.S11_0; ring[s0 + TF_S22 * 64] = A.S22_0; ring[s0 + TF_S12 * 64] = A.S12_0;
                if (AUF) { ring[s1 + TF_U * 64] = A.first; ring[s1 + TF_V * 64] = A.second; }
                else { ring[s0 + TF_V * 64] = A.first; ring[s1 + TF_U * 64] = A.second; }
                ring[s0 + TF_P * 64] = C.P_0; ring[s0 + TF_M * 64] = m_0; ring[s0 + TF_A * 64] = C.a_0;
                if (CSI_TRIO_PRE) { ring[s0 + TF_PF * 64] = A.Pf_0; ring[s0 + TF_RMC * 64] = A.rmc_0; ring[s0 + TF_RMF * 64] = A.rmf_0; }
                ring[s1 + TF_UN * 64] = C.un_m; ring[(AUF ? s1 : s0) + TF_VN * 64] = C.vn_x;        // slot j holds u^n, v^n of row j
                A.shift(C.u_p, C.v_p, m_0, C.a_0);
            }
            __syncthreads();                                  // row r is complete
        };
        load_row(R[0]);
        if (CSI_TRIO_PD == 2) { advance(); load_row(R[1]); }
        for (;;) {
            body(TIdx<0>{});
            if (++r > rlast) break;
            body(TIdx<1>{});
            if (++r > rlast) break;
            body(TIdx<2>{});
            if (++r > rlast) break;
        }
        return;
    }

    // ===== stages B (wave 1: reads A's ring, writes its own) and C (wave 2: reads B's ring, stores) =======================
    // UPF: the order of the stage whose results this one consumes (this stage has the other order).  A stage sees what its
    // upstream stage finished before the barrier: stage B at iteration r sees A's row r; stage C sees B's iteration r - 1,
    // i.e. B's row r - 3.  `up`: that newest upstream row; this stage's row is up - 2.
    auto consume = [&](auto UP, auto LAST) __attribute__((always_inline)) {
        constexpr bool UPF = decltype(UP)::value != 0;
        constexpr bool IS_LAST = decltype(LAST)::value != 0;
        double* rin = ring + (IS_LAST ? TR_SIZE : 0);
        double* rout = ring + TR_SIZE;
        Stage<UNI, !UPF, false, false, CF, false> B;
        B.u_m = 0; B.u_0 = 0; B.v_m = 0; B.v_0 = 0; B.Xv_m = 0; B.Xv_0 = 0;
        B.a_mm = 0; B.a_m = 0; B.m_mm = 0; B.m_m = 0;
        B.XP_m = 0; B.Xm_m = 0; B.Xa_m = 0; B.Xe11_m = 0; B.Xe22_m = 0; B.Ye12_0 = 0; B.e12_0 = 0;
        B.XAL_m = 0; B.XS11L_m = 0; B.XS22L_m = 0; B.XW = 0; B.Wprev = 0;
        B.S11_mm = 0; B.S22_mm = 0; B.S12_mm = 0; B.AL_mm = 0; B.S11_m = 0; B.S22_m = 0; B.S12_m = 0; B.AL_m = 0;
        B.S11_0 = 0; B.S22_0 = 0; B.S12_0 = 0; B.AL_0 = 0; B.first = 0; B.second = 0;
        B.zc = 0; B.zf = 0; B.Dc = 0; B.rDc = 0; B.Pf_0 = 0; B.rmc_0 = 0; B.rmf_0 = 0;
        // rows of this tile each kind of store covers, images of a row (last stage only)
        const int rs_lo = max(ja, T->I[FI_RS + 2]), rs_hi = min(jb, T->I[FI_RS + 3]);
        const int r1_lo = max(ja, T->I[FI_R1 + 2]), r1_hi = min(jb, T->I[FI_R1 + 3]);
        const int r2_lo = max(ja, T->I[FI_R2 + 2]), r2_hi = min(jb, T->I[FI_R2 + 3]);
        auto yimg = [&](int j) __attribute__((always_inline)) {
            return ((j >= 1) & (j <= Hy)) ? Ny : (((j > Ny - Hy) & (j <= Ny)) ? -Ny : 0);
        };
        auto put = [&](unsigned long base, unsigned off, unsigned dy, double val) __attribute__((always_inline)) {
            stg(base, off, val);
            if (dy != 0u) stg(base, off + dy, val);
            if (wave_has_dx) {
                if (dx != 0) {
                    stg(base, off + (unsigned)dx, val);
                    if (dy != 0u) stg(base, off + (unsigned)dx + dy, val);
                }
            }
        };
        int fast_lo, fast_hi;
        {
            const int d1 = UPF ? 0 : 1;                       // first velocity row = p - d1, second = p - 1
            fast_lo = max(max(rs_lo, max(r1_lo + d1, r2_lo + 1)), Hy + 2);
            fast_hi = min(min(rs_hi, min(r1_hi + d1, r2_hi + 1)), Ny - Hy);
            if (!lanes_same) fast_hi = fast_lo - 1;
        }
        // this stage's results of row p: sigma(p); first velocity v(p) / u(p-1); second velocity u(p-1) / v(p-1)
        auto flush = [&](int p, double v11, double v22, double v12, double vfirst, double vsecond) __attribute__((always_inline)) {
            if ((p >= fast_lo) & (p <= fast_hi)) {
                if (flags & L_RS) {
                    const unsigned ocq = offc(p), ofq = offf(p);
                    if (fast_plain) {
                        stg(T->P[FP_S11_OUT], ocq, v11); stg(T->P[FP_S22_OUT], ocq, v22); stg(T->P[FP_S12_OUT], ofq, v12);
                        stg(T->P[UPF ? FP_V_OUTP : FP_U_OUTP], UPF ? ocq : ofq - sf, vfirst);
                        stg(T->P[UPF ? FP_U_OUTP : FP_V_OUTP], UPF ? ofq - sf : ocq - sc, vsecond);
                    } else {
                        put(T->P[FP_S11_OUT], ocq, 0u, v11); put(T->P[FP_S22_OUT], ocq, 0u, v22); put(T->P[FP_S12_OUT], ofq, 0u, v12);
                        put(T->P[UPF ? FP_V_OUTP : FP_U_OUTP], UPF ? ocq : ofq - sf, 0u, vfirst);
                        put(T->P[UPF ? FP_U_OUTP : FP_V_OUTP], UPF ? ofq - sf : ocq - sc, 0u, vsecond);
                    }
                }
                return;
            }
            const int j1 = UPF ? p : p - 1, j2 = p - 1;
            const bool do_s = (p >= rs_lo) & (p <= rs_hi), do_1 = (j1 >= r1_lo) & (j1 <= r1_hi), do_2 = (j2 >= r2_lo) & (j2 <= r2_hi);
            if (!(do_s | do_1 | do_2)) return;
            const unsigned ocq = offc(p), ofq = offf(p);
            const unsigned o1 = UPF ? ocq : ofq - sf, o2 = UPF ? ofq - sf : ocq - sc;
            const int yq = yimg(p), y1 = yimg(j1), y2 = yimg(j2);
            if (do_s & ((flags & L_RS) != 0)) {
                put(T->P[FP_S11_OUT], ocq, (unsigned)yq * sc, v11);
                put(T->P[FP_S22_OUT], ocq, (unsigned)yq * sc, v22);
                put(T->P[FP_S12_OUT], ofq, (unsigned)yq * sf, v12);
            }
            if (do_1 & ((flags & L_R1) != 0)) {
                if (UPF) put(T->P[FP_V_OUTP], o1, (unsigned)y1 * sc, vfirst); else put(T->P[FP_U_OUTP], o1, (unsigned)y1 * sf, vfirst);
            }
            if (do_2 & ((flags & L_R2) != 0)) {
                if (UPF) put(T->P[FP_U_OUTP], o2, (unsigned)y2 * sf, vsecond); else put(T->P[FP_V_OUTP], o2, (unsigned)y2 * sc, vsecond);
            }
        };
        // rows whose stresses / velocities somebody downstream needs (B: two more rings of rows than C)
        const int margin = IS_LAST ? 0 : 2;
        double vn_delay = 0.0;
        for (int r = rstart; r <= rlast; ++r) {
            __syncthreads();                                  // the upstream stage has finished its iteration
            const int up = IS_LAST ? r - 3 : r;
            const int p = up - 2;
            if (!IS_LAST && r > rend) continue;               // (stage B has nothing to do in the extra iteration)
            const unsigned s1 = rslot(up - 1), s2 = rslot(up - 2), s3 = rslot(up - 3);
            const double bu_p = rin[s1 + TF_U * 64], bv_p = rin[s1 + TF_V * 64];
            const double s11 = rin[s2 + TF_S11 * 64], s22 = rin[s2 + TF_S22 * 64], s12 = rin[s2 + TF_S12 * 64];
            const double bP_0 = rin[s2 + TF_P * 64], bm_0 = rin[s2 + TF_M * 64], ba_0 = rin[s2 + TF_A * 64];
            if (CSI_TRIO_PRE) { B.Pf_0 = rin[s2 + TF_PF * 64]; B.rmc_0 = rin[s2 + TF_RMC * 64]; B.rmf_0 = rin[s2 + TF_RMF * 64]; }
            const double bun = rin[s3 + TF_UN * 64], vn_new = rin[s2 + TF_VN * 64];
            const double bvn = UPF ? vn_new : vn_delay;       // this stage v-first: v^n(p); u-first: v^n(p - 1)
            vn_delay = vn_new;
            fm::StressConst ks; stress_consts(ks);
            fm::VelConst kv; vel_consts(kv);
            Forcing FB;
            numbers(FB);
            B.template step<(CSI_TRIO_PRE != 0)>(T, ks, kv, p, bu_p, bv_p, bP_0, bm_0, ba_0, s11, s22, s12, bun, bvn, p >= ja - 1 - margin, p >= ja - margin,
                                  false, false, 0u, FB);
            if (IS_LAST) {
                flush(p, B.S11_0, B.S22_0, B.S12_0, B.first, B.second);
                if (write_diag) {
                    if (((flags & L_RS) != 0) & (p >= rs_lo) & (p <= rs_hi)) {
                        const unsigned ocq = offc(p), ofq = offf(p);
                        const int yq = yimg(p);
                        put(T->P[FP_AL], ocq, (unsigned)yq * sc, B.AL_0);
                        put(T->P[FP_ZF], ofq, (unsigned)yq * sf, 0.5 * B.zf);
                        put(T->P[FP_ZC], ocq, (unsigned)yq * sc, 0.5 * B.zc);
                        put(T->P[FP_DL], ocq, (unsigned)yq * sc, B.Dc * B.rDc);
                    }
                }
            } else {
                // hand-off to stage C, row-indexed like A's: sigma(p), static fields of row p -> slot p; u / v of row p - 1 (or
                // v of row p when this stage is v-first) and u^n(p - 1) -> slot p - 1; v^n(p) -> slot p
                const unsigned t0 = rslot(p), t1 = rslot(p - 1);
                rout[t0 + TF_S11 * 64] = B.S11_0; rout[t0 + TF_S22 * 64] = B.S22_0; rout[t0 + TF_S12 * 64] = B.S12_0;
                if (!UPF) { rout[t1 + TF_U * 64] = B.first; rout[t1 + TF_V * 64] = B.second; }
                else { rout[t0 + TF_V * 64] = B.first; rout[t1 + TF_U * 64] = B.second; }
                rout[t0 + TF_P * 64] = bP_0; rout[t0 + TF_M * 64] = bm_0; rout[t0 + TF_A * 64] = ba_0;
                if (CSI_TRIO_PRE) { rout[t0 + TF_PF * 64] = B.Pf_0; rout[t0 + TF_RMC * 64] = B.rmc_0; rout[t0 + TF_RMF * 64] = B.rmf_0; }
                rout[t1 + TF_UN * 64] = bun; rout[t0 + TF_VN * 64] = vn_new;
            }
            B.shift(bu_p, bv_p, bm_0, ba_0);
        }
    };
    if (wid == 1) consume(TIdx<AUF ? 1 : 0>{}, TIdx<0>{});
    else consume(TIdx<AUF ? 0 : 1>{}, TIdx<1>{});
}

}  // namespace fused

void launch_fused_trio(const FusedTable* dev_table, bool uniform, bool a_ufirst, int common, int nstrips, int nchunks, int rows,
                       int write_diag, hipStream_t s) {
    const int nblocks = nstrips * nchunks;
    const int per_xcd = (nblocks + 7) / 8;
    dim3 grid((unsigned)(per_xcd * 8)), block(192);
#define CSI_LAUNCH_TRIO_(U, A, C) hipLaunchKernelGGL((fused::k_trio<U, A, C>), grid, block, 0, s, dev_table, nstrips, nchunks, rows, per_xcd, write_diag)
#define CSI_LAUNCH_TRIO(U, A) do { if (common == 2) CSI_LAUNCH_TRIO_(U, A, 2); else if (common) CSI_LAUNCH_TRIO_(U, A, 1); else CSI_LAUNCH_TRIO_(U, A, 0); } while (0)
    if (uniform) { if (a_ufirst) CSI_LAUNCH_TRIO(true, true); else CSI_LAUNCH_TRIO(true, false); }
    else { if (a_ufirst) CSI_LAUNCH_TRIO(false, true); else CSI_LAUNCH_TRIO(false, false); }
#undef CSI_LAUNCH_TRIO
#undef CSI_LAUNCH_TRIO_
}

}  // namespace csi
