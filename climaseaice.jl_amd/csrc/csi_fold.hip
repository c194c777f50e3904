// csi_fold.hip -- north fold: the band of rows next to the fold on the three kernels, beside the pair launches  (split out of csi_abi.hip in round 4; see csi_ctx.h)
#include "csi_ctx.h"

namespace csi_host {

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
        // Reserved CUs (tune.band_cus = r per XCD): the band's launches are six dependent, latency-bound kernels per pair of sub-steps
        // that otherwise queue for wave slots behind the pair kernel's workgroups.  Mask bit k is a CU of XCD k % 8
        // (scripts/microbench/cu_mask_probe.hip): bits 0 .. 8 r - 1 are r CUs of every XCD.
        const int r = c->tune.band_cus;
        if (r > 0 && r < 16) {
            uint32_t band_mask[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pair_mask[8];
            for (int k = 0; k < 8 * r; ++k) band_mask[k >> 5] |= 1u << (k & 31);
            for (int w = 0; w < 8; ++w) pair_mask[w] = ~band_mask[w];
            if (c->tune.band_cus_share > 0) for (int w = 0; w < 8; ++w) band_mask[w] = 0xffffffffu;
            HIP_TRY(c, hipExtStreamCreateWithCUMask(&c->band_stream, 8, band_mask));
            HIP_TRY(c, hipExtStreamCreateWithCUMask(&c->pair_stream, 8, pair_mask));
            for (hipEvent_t& e : c->exp_ev) if (!e) HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        } else
        HIP_TRY(c, hipStreamCreateWithFlags(&c->band_stream, hipStreamNonBlocking));
        // the two events order kernels of ONE device against each other (pair launch <-> band step), sixty times per sub-cycle each: no
        // system-scope fence (what they guard reaches the host and other devices behind the kernels that follow on the context's stream).
        // CSI_BAND_EVENT_FLAGS: 0 default events (7.18-7.26 ms per tripolar step), 1 hipEventDisableSystemFence (7.04-7.06: the default),
        // 2 hipEventReleaseToDevice (7.30); profiles/r06_band.md section 6
        const int ef = c->tune.band_event_flags < 0 ? 1 : c->tune.band_event_flags;
        const unsigned evf = hipEventDisableTiming | (ef == 1 ? hipEventDisableSystemFence : (ef == 2 ? hipEventReleaseToDevice : 0u));
        HIP_TRY(c, hipEventCreateWithFlags(&c->band_ev_pair, evf));
        HIP_TRY(c, hipEventCreateWithFlags(&c->band_ev_band, evf));
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
// The same without the copy kernels (round 6b; evp_fast.hip, k_band_stress / k_band_vel): the band's launches are a chain of dependent,
// latency-bound kernels beside the pair launch -- eight per pair of sub-steps with 3-4 us between two of them; this is six, each with
// every load of a point in flight at once.  The first sub-step reads the current buffer and writes the band's copies; the second
// works on those in place and stores rows >= M + 1 into the other buffer as well.  Same bits as band_substep's three kernels (the same
// value functions; tests/test_gpu_activity.py compares the two paths).
int32_t band_substeps_fused(csi_context* c, const FoldBand& bd, const FastCoef& fc, int cur, int s, int n, bool last) {
    hipStream_t st = c->band_stream;
    HIP_TRY(c, hipStreamWaitEvent(st, c->band_ev_pair, 0));
    FRef src[5], dst[5], b[5], d[4];
    for (int q = 0; q < 5; ++q) {
        const FRef o = ref_of(c, kPing[q]), a = alt_ref(c, q);
        src[q] = cur == 0 ? o : a; dst[q] = cur == 0 ? a : o;
        b[q] = band_ref(c, q);
    }
    for (int q = 0; q < 4; ++q) d[q] = band_ref(c, 5 + q);
    auto from = [&](Range r, int j0) { if (j0 > r.j0) r.j0 = j0; return r; };
    const FRef none{nullptr, 0};
    // one sub-step: u, v, sigma from `in` (each launch reads the component it replaces at the point itself only), results into the
    // band's copies and -- `out` -- rows >= M + 1 into dst
    auto substep = [&](const FRef* in, bool ufirst, int jlo, bool diag, bool out) {
        EvpDev Q = bd.P;
        Q.u = in[0]; Q.v = in[1]; Q.s11 = in[2]; Q.s22 = in[3]; Q.s12 = in[4];
        Q.al = d[0]; Q.zc = d[1]; Q.zf = d[2]; Q.Dl = d[3];
        Q.write_diag = diag;
        const int j0 = bd.M + 1;
        launch_band_stress(Q, from(bd.rs, jlo), fc, BandOut{b[2], b[3], b[4], out ? dst[2] : none, out ? dst[3] : none, out ? dst[4] : none, j0}, st);
        Q.s11 = b[2]; Q.s22 = b[3]; Q.s12 = b[4];
        const BandOut ou{b[0], none, none, out ? dst[0] : none, none, none, j0}, ov{b[1], none, none, out ? dst[1] : none, none, none, j0};
        if (ufirst) {
            launch_band_vel(Q, from(bd.ru1, jlo + 1), bd.imu, fc, true, ou, st);
            Q.u = b[0];                                                              // (v reads the new u)
            launch_band_vel(Q, from(bd.r2, jlo + 1), bd.imv, fc, false, ov, st);
        } else {
            launch_band_vel(Q, from(bd.rv1, jlo + 1), bd.imv, fc, false, ov, st);
            Q.v = b[1];
            launch_band_vel(Q, from(bd.r2, jlo + 1), bd.imu, fc, true, ou, st);
        }
    };
    if (n == 2) {
        substep(src, (s % 2) == 0, bd.M - 5, false, false);
        substep(b, ((s + 1) % 2) == 0, bd.M - 2, last, true);
    } else substep(src, (s % 2) == 0, bd.M - 3, last, true);
    if (last) {
        CopyBatch diag{};
        for (int q = 5; q < 9; ++q) {
            const Bound& bb = band_bound(c, q);
            const size_t row = (size_t)(bd.M + 1 - 1 + c->Hy), off = row * (size_t)bb.ld;
            diag.src[diag.count] = c->band[q] + off; diag.dst[diag.count] = bb.p + off; diag.n[diag.count] = (long)(((size_t)bb.nj - row) * (size_t)bb.ld);
            ++diag.count;
        }
        launch_copy_batch(diag, st, 64);
    }
    HIP_TRY(c, hipEventRecord(c->band_ev_band, st));
    return CSI_OK;
}
// launches of one band step (the sub-cycle's launch statistics)
int band_launches(const csi_context* c, int n) { return c->tune.band_fused != 0 ? 3 * n : (n == 2 ? 8 : 5); }

// two sub-steps (or the trailing single one) of the band: buffer `cur` (0: the caller's arrays) -> the other one, on the band's stream
int32_t band_substeps(csi_context* c, const FoldBand& bd, const FastCoef& fc, int cur, int s, int n, bool last) {
    if (c->tune.band_fused != 0) return band_substeps_fused(c, bd, fc, cur, s, n, last);
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
    launch_copy_batch(in, st, 64);
    FRef b[5], d[4];
    for (int q = 0; q < 5; ++q) b[q] = band_ref(c, q);
    for (int q = 0; q < 4; ++q) d[q] = band_ref(c, 5 + q);
    // validity after the first sub-step: sigma from row M - 5, velocities from M - 3; after the second: sigma M - 2, velocities M
    int32_t rc;
    if (n == 2) {
        if ((rc = band_substep(c, bd, fc, b, d, (s % 2) == 0, bd.M - 5, false, st))) return rc;
        if ((rc = band_substep(c, bd, fc, b, d, ((s + 1) % 2) == 0, bd.M - 2, last, st))) return rc;
    } else if ((rc = band_substep(c, bd, fc, b, d, (s % 2) == 0, bd.M - 3, last, st))) return rc;
    launch_copy_batch(out, st, 64);
    if (last) {
        for (int q = 5; q < 9; ++q) rows_from(q, bd.M + 1, c->band[q], band_bound(c, q).p, diag);
        launch_copy_batch(diag, st, 64);
    }
    HIP_TRY(c, hipEventRecord(c->band_ev_band, st));
    return CSI_OK;
}


// A north fold on an untiled grid (RightFolded y, Periodic x): see FoldBand.
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


}  // namespace csi_host
