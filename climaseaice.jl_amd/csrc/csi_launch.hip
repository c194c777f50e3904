// csi_launch.hip -- launch loops: the EVP sub-cycle on every path, finalize_rheology!, time_step_momentum!, tracer steps, update_state!  (split out of csi_abi.hip in round 4; see csi_ctx.h)
#include "csi_ctx.h"

namespace csi_host {

// csi_subcycle_stats_*: an extra event behind ev0 / ev1 of every sub-step loop while the statistics are on
static int32_t stats_mark(csi_context* c) {
    if (!c->stats.on || c->stats.ev.size() >= 2 * 4096) return CSI_OK;
    hipEvent_t e;
    HIP_TRY(c, hipEventCreate(&e));
    HIP_TRY(c, hipEventRecord(e, c->stream));
    c->stats.ev.push_back(e);
    return CSI_OK;
}
static void stats_launches(csi_context* c, int n) { if (c->stats.on && c->stats.launches.size() * 2 < c->stats.ev.size()) c->stats.launches.push_back(n); }

// The newest activity sample the device has written into the pinned words (k_activity_compact's seqlock): true when a new one was taken.
bool activity_sample(csi_context* c) {
    csi_context::Activity& a = c->act;
    if (!a.host) return false;
    volatile int* w = a.host;
    const int s1 = w[0];
    if (s1 & 1) return false;
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    const int live = w[1], tiles = w[2], id = w[3];
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    if (w[0] != s1 || s1 == 0 || id == a.seen_id) return false;
    a.seen_id = id; a.last_live = live; a.last_tiles = tiles;
    return true;
}

// peer: the caller (run_fused_peer) has turned the connected sides of c->g / P.g into periodic ones: the launch loop is that of an
// untiled periodic grid, the halo images of those sides go to the neighbouring tiles' arrays and every pair launch carries a
// number of the flag protocol.  band: the caller (run_fused_fold) has cut the rows next to a north fold off c->g / P.g.
int32_t run_fused(csi_context* c, const EvpDev& P, const FastCoef& fc, int substeps, int first, bool peer, const FoldBand* band) {
    int32_t rc;
    if ((rc = ensure_alt(c))) return rc;
    if (band && (rc = ensure_band(c))) return rc;
    const bool tiled = band ? band->tiled : is_tiled(c);
    const int k = band ? band->k : exchange_interval(c), W = 2 * k;
    // (band: the halo exchange of the fold tile is that of the tile as it is -- its north side has no neighbour)
    auto exchange_tile = [&](const FRef* fr) -> int32_t {
        if (band) { c->g = band->g_full; c->Ny = band->Ny_full; }
        const int32_t r = exchange_refs(c, fr, nxf_of(k), W);
        if (band) { c->g = band->g_cut; c->Ny = band->M; }
        return r;
    };
    const bool masked = P.g.has_mask != 0;
    const bool force = pair_forcing_kind(P) == 1;           // array-valued forcing: two-sub-steps kernel only
    const bool pairs = peer || (pair_supported(c) && (!tiled || k % 2 == 0));
    // number-valued top stress (or none) and a bottom SemiImplicitStress with number-valued ocean velocities: the kernels'
    // compile-time forcing kinds
    auto ocean_at_rest = [](int kind, double value) { return kind == 0 || (kind == 1 && value == 0.0 && !std::signbit(value)); };   // (-0.0 would flip signed zeros)
    int common_forcing = !force && P.top.kind <= 1 && P.bot.kind == 3 && P.bot.ue_kind != 2 && P.bot.ve_kind != 2 &&
                         P.pressure_kind == 0;            // ... and the default ReplacementPressure
    if (common_forcing && ocean_at_rest(P.bot.ue_kind, P.bot.ue) && ocean_at_rest(P.bot.ve_kind, P.bot.ve))
        common_forcing = 2;                               // ZeroField ocean velocities (the reference's default)
    if (c->tune.pair_common >= 0 && common_forcing > c->tune.pair_common) common_forcing = c->tune.pair_common;   // A/B knob (CSI_PAIR_COMMON)
    FRef ubar_v{nullptr, 0}, vbar_u{nullptr, 0};
    if (force && P.bot.kind == 3 && (P.bot.ue_kind == 2 || P.bot.ve_kind == 2)) {
        // cross components of the ocean velocity averaged to the velocity points, once per sub-cycle
        const Bound* src[2] = {&c->f[CSI_F_V], &c->f[CSI_F_U]};       // shapes: ubar lives at v points, vbar at u points
        for (int q = 0; q < 2; ++q) {
            const size_t n = (size_t)src[q]->ld * (size_t)src[q]->nj;
            if (c->fbar_elems[q] != n) {
                if (c->fbar[q]) { HIP_TRY(c, hipStreamSynchronize(c->stream)); hipFree(c->fbar[q]); c->fbar[q] = nullptr; }
                HIP_TRY(c, hipMalloc((void**)&c->fbar[q], n * sizeof(double)));
                HIP_TRY(c, hipMemsetAsync(c->fbar[q], 0, n * sizeof(double), c->stream));
                c->fbar_elems[q] = n;
            }
        }
        ubar_v.p = c->fbar[0] + (c->Hx - 1) + (int64_t)(c->Hy - 1) * src[0]->ld; ubar_v.ld = (int)src[0]->ld;
        vbar_u.p = c->fbar[1] + (c->Hx - 1) + (int64_t)(c->Hy - 1) * src[1]->ld; vbar_u.ld = (int)src[1]->ld;
        launch_forcing_bars(P, ubar_v, vbar_u, c->stream);
    }
    FRef tbar_v{nullptr, 0}, tbar_u{nullptr, 0};              // wind drag: the air velocities' cross averages
    const bool wind = force && P.top.kind == 3 && (P.top.ue_kind == 2 || P.top.ve_kind == 2);
    const int extra_kind = P.extra ? 1 : ((wind || (force && P.bot.kind == 2)) ? 2 : 0);      // which family of array-forcing instantiations
    if (wind) {
        const Bound* src[2] = {&c->f[CSI_F_V], &c->f[CSI_F_U]};
        for (int q = 0; q < 2; ++q) {
            const size_t n = (size_t)src[q]->ld * (size_t)src[q]->nj;
            if (c->fbar_top_elems[q] != n) {
                if (c->fbar_top[q]) { HIP_TRY(c, hipStreamSynchronize(c->stream)); hipFree(c->fbar_top[q]); c->fbar_top[q] = nullptr; }
                HIP_TRY(c, hipMalloc((void**)&c->fbar_top[q], n * sizeof(double)));
                HIP_TRY(c, hipMemsetAsync(c->fbar_top[q], 0, n * sizeof(double), c->stream));
                c->fbar_top_elems[q] = n;
            }
        }
        tbar_v.p = c->fbar_top[0] + (c->Hx - 1) + (int64_t)(c->Hy - 1) * src[0]->ld; tbar_v.ld = (int)src[0]->ld;
        tbar_u.p = c->fbar_top[1] + (c->Hx - 1) + (int64_t)(c->Hy - 1) * src[1]->ld; tbar_u.ld = (int)src[1]->ld;
        launch_forcing_bars(P, tbar_v, tbar_u, c->stream, true);
    }
    // model.forcing arrays / immersed flux boundary conditions (the EXTRA instantiations of the pair kernel): the divergence of
    // the immersed fluxes is a function of the mask and the metrics only -- once per sub-cycle into two arrays
    const bool extra = P.extra != 0;
    FRef xd_u{nullptr, 0}, xd_v{nullptr, 0};
    if (extra && P.g.has_mask) {
        bool any_ibc = false;
        for (int q = 0; q < 4; ++q) any_ibc |= (P.ibc_u[q] != 0.0) | (P.ibc_v[q] != 0.0);
        if (any_ibc) {
            const Bound* src[2] = {&c->f[CSI_F_U], &c->f[CSI_F_V]};
            for (int q = 0; q < 2; ++q) {
                const size_t n = (size_t)src[q]->ld * (size_t)src[q]->nj;
                if (c->xd_elems[q] != n) {
                    if (c->xd[q]) { HIP_TRY(c, hipStreamSynchronize(c->stream)); hipFree(c->xd[q]); c->xd[q] = nullptr; }
                    HIP_TRY(c, hipMalloc((void**)&c->xd[q], n * sizeof(double)));
                    HIP_TRY(c, hipMemsetAsync(c->xd[q], 0, n * sizeof(double), c->stream));
                    c->xd_elems[q] = n;
                }
            }
            xd_u.p = c->xd[0] + (c->Hx - 1) + (int64_t)(c->Hy - 1) * src[0]->ld; xd_u.ld = (int)src[0]->ld;
            xd_v.p = c->xd[1] + (c->Hx - 1) + (int64_t)(c->Hy - 1) * src[1]->ld; xd_v.ld = (int)src[1]->ld;
            launch_immersed_div(P, xd_u, xd_v, c->stream);
        }
    }
    const int kb = tiled ? k : (pairs ? 2 : 1);             // batch length: positions 0 .. kb-1
    FRef orig[5], alt[5];
    for (int q = 0; q < 5; ++q) { orig[q] = ref_of(c, kPing[q]); alt[q] = alt_ref(c, q); }
    if (tiled && (rc = exchange_tile(orig))) return rc;
    // both buffers start identical, so cells no sub-step ever writes (wall halos, the outermost halo layer of sigma
    // under the one-sub-step kernel) agree in both.  A fully periodic, untiled grid advanced by pair launches only
    // rewrites every cell of the five parents -- interior and all halo images -- at every launch: no copy needed.
    const bool every_cell_written = pairs && !tiled && !has_walls(c) && substeps % 2 == 0;
    const ImageSpec imu = image_spec(c, CSI_F_U), imv = image_spec(c, CSI_F_V);
    // tables: singles (position in the exchange batch) x (which buffer is current) x (u first / v first), then
    // pairs (pair position) x (buffer) x (first sub-step u first / v first)
    constexpr int KMAX = kMaxExchangeInterval, NSINGLE = KMAX * 4, NPAIR = (KMAX / 2) * 4;
    if (k > KMAX) return fail(c, CSI_ERR_UNSUPPORTED, "exchange interval too large for the fused path");
    if (!c->dev_tables) HIP_TRY(c, hipMalloc((void**)&c->dev_tables, (NSINGLE + NPAIR) * sizeof(FusedTable)));
    FusedGeom G[KMAX], GP[KMAX / 2];
    // Tile activity (csi_activity.hip): untiled grids (a fold band beside them or not) advanced by pair launches.  The first two pair
    // launches and the last launch of the sub-cycle run every tile on the one-round geometry; the launches in between run the LIVE
    // tiles of a finer geometry GA -- so many more tiles as the newest sample of the live fraction says fit one round.
    FusedGeom GA{};
    // (peer-connected tiles: the tiles of the direction sets always run -- they publish their flags whatever they hold --, the interior
    //  ones may go quiet; the launch geometry stays the one the transport was set up for)
    bool act_on = c->act.enabled != 0 && pairs && !tiled && substeps >= 8;
    if (act_on) {
        csi_context::Activity& a = c->act;
        if (!a.list) {
            HIP_TRY(c, hipMalloc((void**)&a.flags, sizeof(int) * kMaxActTiles));
            HIP_TRY(c, hipMalloc((void**)&a.list, sizeof(int) * (kMaxActTiles + 2)));
            HIP_TRY(c, hipMalloc((void**)&a.list0, sizeof(int) * (kMaxActTiles + 2)));
            HIP_TRY(c, hipHostMalloc((void**)&a.host, sizeof(int) * 4, hipHostMallocMapped));
            memset(a.host, 0, sizeof(int) * 4);
            HIP_TRY(c, hipHostGetDevicePointer((void**)&a.host_dev, a.host, 0));
        }
        // the newest sample the device has written (seqlock; no API call, no synchronisation): live tiles / tiles on the geometry of its sub-cycle
        double s_scale = 1.0;
        bool fresh = false;
        if (csi_host::activity_sample(c)) { s_scale = a.sample_scale[(unsigned)a.seen_id % csi_context::Activity::kSamples]; fresh = true; }
        if (fresh && a.last_tiles > 0) {
            const double f = (double)std::max(a.last_live, 1) / (double)a.last_tiles;
            // keep the geometry while the live tiles fill 90 .. 100 % of a round; otherwise aim at 97 %
            if (f >= 0.97) a.scale = 1.0;
            else if (s_scale * f > 1.0 || s_scale * f < 0.90 || a.scale != s_scale) a.scale = std::min(4.0, 0.97 / f);
        }
    }
    // (Almost) nothing quiescent in the newest sample -- the headline's grid: no land, ice everywhere but one patch of open water, 987
    // of 999 tiles live --: the test, the scan, the copy below and the list-mapped launches cost 3 % of such a sub-cycle (measured: 7.12
    // against 6.89 ms at 2048^2, scripts/ab_headline_skip.sh) and leaving a dozen tiles out of a one-round launch buys nothing.  So
    // unless fewer than nine tiles in ten are live the grid is only PROBED, every kProbeEvery-th sub-cycle, and the sub-cycles in
    // between run the plain launches.  Ice-free ocean that spreads is found within that many sub-cycles; while the cut is in use the
    // test runs before every sub-cycle (it must: the list decides what is skipped).
    if (act_on && c->act.last_live >= 0 && 10L * c->act.last_live >= 9L * c->act.last_tiles) {
        constexpr int kProbeEvery = 32;
        if (++c->act.since_probe < kProbeEvery) act_on = false;
        else c->act.since_probe = 0;
    } else {
        c->act.since_probe = 0;
    }
    // Tiles quiescent FROM THE START (no ice mass, velocities +0.0 already, no halo image to store): with the two buffers equal before
    // the first launch even the first two launches may leave them out.  Worth the copy below only when the newest sample says that
    // something is quiescent at all; not on peer-connected tiles (their copy is made before the sub-cycle's exchange, run_fused_peer).
    const bool q0_on = act_on && !peer && c->act.last_live >= 0 && 10L * c->act.last_live < 9L * c->act.last_tiles;
    if ((!every_cell_written || q0_on) && !peer)           // (peer: run_fused_peer has made the copy, BEFORE its exchange)
        for (int q = 0; q < 5; ++q) {
            const Bound& b = c->f[kPing[q]];
            HIP_TRY(c, hipMemcpyAsync(c->alt[q], b.p, c->alt_elems[q] * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        }
    // configurations only the two-sub-steps kernel takes (masks, array forcing, per-point metrics): a single sub-step (the odd
    // trailing one) runs through that kernel too, its consumer wave storing stage A's results (evp_fused2.hip, `single`)
    const bool single_by_pair = pairs && (masked || force || c->metric_kind == CSI_METRIC_FULL || peer || band);      // (peer: the flag protocol lives in this kernel only; band: its cut tile)
    {
        if (!c->host_ring) {
            HIP_TRY(c, hipHostMalloc((void**)&c->host_ring, sizeof(FusedTable) * (NSINGLE + NPAIR) * csi_context::kRing, hipHostMallocDefault));
            for (int q = 0; q < csi_context::kRing; ++q) HIP_TRY(c, hipEventCreateWithFlags(&c->ring_ev[q], hipEventDisableTiming));
        }
        const int slot = (int)(c->ring_pos++ % csi_context::kRing);
        if (c->ring_used[slot]) HIP_TRY(c, hipEventSynchronize(c->ring_ev[slot]));
        FusedTable* host = c->host_ring + (size_t)slot * (NSINGLE + NPAIR);
        for (int m = 0; m < kb; ++m) {
            const int V = tiled ? W - 2 * m : 2;
            G[m] = fused_geom(c, V);
            if (single_by_pair) {
                const ImageSpec ims11 = image_spec(c, CSI_F_S11), ims22 = image_spec(c, CSI_F_S22), ims12 = image_spec(c, CSI_F_S12);
                const SideV vs = pair_side_v(c, V, 2);
                const Range dec = v_stress_range(c, vs);
                G[m] = pair_geom(c, dec);
                for (int cur = 0; cur < 2; ++cur)
                    for (int uf = 0; uf < 2; ++uf) {
                        Range rs = clip_store(c, dec, true), r1 = clip_store(c, v_first_range(c, vs, uf != 0), false),
                              r2 = clip_store(c, v_second_range(c, vs), false);
                        if (band) { rs.j1 = std::min(rs.j1, c->Ny); r1.j1 = std::min(r1.j1, c->Ny); r2.j1 = std::min(r2.j1, c->Ny); }
                        FusedTable* t = &host[(m * 2 + cur) * 2 + uf];
                        fused_fill_table(P, fc, cur == 0 ? orig : alt, cur == 0 ? alt : orig, rs, r1, r1, r2, imu, imv, t);
                        fused_fill_pair_extra(dec, dec.j0, dec.j1, ims11, ims22, ims12, t, G[m].elo, G[m].ehi, G[m].wt);
                        if (force) { fused_fill_forcing(P, ubar_v, vbar_u, t); if (wind) fused_fill_forcing_top(P, tbar_v, tbar_u, t); }
                        if (extra) fused_fill_extra(P, xd_u, xd_v, t);
                        if (peer && (rc = peer_fill_table(c, G[m], cur == 0, t))) return rc;
                    }
                continue;
            }
            for (int cur = 0; cur < 2; ++cur)
                for (int uf = 0; uf < 2; ++uf) {
                    Range r1, r1c, r2;
                    velocity_ranges(c, uf != 0, V, r1, r1c, r2);
                    fused_fill_table(P, fc, cur == 0 ? orig : alt, cur == 0 ? alt : orig, G[m].rs, r1, r1c, r2, imu, imv,
                                     &host[(m * 2 + cur) * 2 + uf]);
                }
        }
        if (pairs) {
            const ImageSpec ims11 = image_spec(c, CSI_F_S11), ims22 = image_spec(c, CSI_F_S22), ims12 = image_spec(c, CSI_F_S12);
            for (int mp = 0; 2 * mp + 1 < kb; ++mp) {
                const SideV va = pair_side_v(c, W - 4 * mp, 4), vb = pair_side_v(c, W - 4 * mp - 2, 2);
                const Range dec = v_stress_range(c, vb), ra = v_stress_range(c, va);
                GP[mp] = pair_geom(c, dec);
                if (act_on && mp == 0) {
                    GA = pair_geom(c, dec, peer ? 1.0 : c->act.scale);
                    if (GA.nstrips * GA.nchunks > kMaxActTiles || GA.elo != GP[0].elo || GA.ehi != GP[0].ehi || GA.wt != GP[0].wt) act_on = false;
                }
                for (int cur = 0; cur < 2; ++cur)
                    for (int auf = 0; auf < 2; ++auf) {
                        const bool buf = auf == 0;                  // the second sub-step has the other order
                        Range rs = clip_store(c, dec, true), r1 = clip_store(c, v_first_range(c, vb, buf), false),
                              r2 = clip_store(c, v_second_range(c, vb), false);
                        if (band) {         // rows above M are the band's: it stores them into the same buffer meanwhile
                            rs.j1 = std::min(rs.j1, c->Ny); r1.j1 = std::min(r1.j1, c->Ny); r2.j1 = std::min(r2.j1, c->Ny);
                        }
                        FusedTable* t = &host[NSINGLE + (mp * 2 + cur) * 2 + auf];
                        fused_fill_table(P, fc, cur == 0 ? orig : alt, cur == 0 ? alt : orig, rs, r1, r1, r2, imu, imv, t);
                        fused_fill_pair_extra(dec, ra.j0, ra.j1, ims11, ims22, ims12, t, GP[mp].elo, GP[mp].ehi, GP[mp].wt);
                        if (force) { fused_fill_forcing(P, ubar_v, vbar_u, t); if (wind) fused_fill_forcing_top(P, tbar_v, tbar_u, t); }
                        if (extra) fused_fill_extra(P, xd_u, xd_v, t);
                        if (peer && (rc = peer_fill_table(c, GP[mp], cur == 0, t))) return rc;
                        if (act_on) { t->P[FP_ACT_LIVE] = (unsigned long)c->act.list; t->P[FP_ACT_LIVE0] = (unsigned long)c->act.list0; }
                    }
            }
        }
        HIP_TRY(c, hipMemcpyAsync(c->dev_tables, host, sizeof(FusedTable) * (NSINGLE + NPAIR), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipEventRecord(c->ring_ev[slot], c->stream));
        c->ring_used[slot] = true;
    }
    unsigned long long peer_dld_bit = 0ull;      // neighbours with other row strides: the DLD instantiation (bit 63 of the launch number)
    if (peer)
        for (int d = 0; d < 8; ++d) if (c->peer.dld[d][0] | c->peer.dld[d][1]) peer_dld_bit = 1ull << 63;
    c->act.last_used = 0;
    if (act_on) {
        // which tiles of GA are live: h, aice (constant over the sub-cycle) and the sign bits of the current stresses
        ActivityArgs A{};
        A.h = ref_of(c, CSI_F_H); A.a = ref_of(c, CSI_F_A);
        A.s11 = orig[2]; A.s22 = orig[3]; A.s12 = orig[4]; A.u = orig[0]; A.v = orig[1];
        A.rho = P.rho;
        A.dec = GA.rs;
        A.nstrips = GA.nstrips; A.nchunks = GA.nchunks; A.rows = GA.rows; A.elo = GA.elo; A.ehi = GA.ehi;
        // (band: the arrays are those of the whole grid, c->Ny is the cut one -- the parents' own extents)
        const Bound &bc = c->f[CSI_F_H], &bf = c->f[CSI_F_S12];
        A.pc = Range{1 - c->Hx, bc.ni - c->Hx, 1 - c->Hy, bc.nj - c->Hy};
        A.pf = Range{1 - c->Hx, bf.ni - c->Hx, 1 - c->Hy, bf.nj - c->Hy};
        const Bound &bu = c->f[CSI_F_U], &bv = c->f[CSI_F_V];
        A.pu = Range{1 - c->Hx, bu.ni - c->Hx, 1 - c->Hy, bu.nj - c->Hy};
        A.pv = Range{1 - c->Hx, bv.ni - c->Hx, 1 - c->Hy, bv.nj - c->Hy};
        A.Nx = c->Nx; A.Ny = c->Ny; A.Hx = c->Hx; A.Hy = c->Hy;      // (band: the cut grid -- the tiles next to the band count as edge tiles)
        if (peer) {
            const FusedTable* t0 = &(c->host_ring + (size_t)((c->ring_pos - 1) % csi_context::kRing) * (NSINGLE + NPAIR))[NSINGLE];      // (the pair tables just filled)
            for (int q = 0; q < 4; ++q) A.pset[q] = t0->I[FI_PSET + q];
            A.pmask = t0->I[FI_PMASK];
        }
        csi_context::Activity& a = c->act;
        const int id = (int)(a.seq++ & 0x3fffffffu);
        a.sample_scale[(unsigned)id % csi_context::Activity::kSamples] = a.scale;
        launch_tile_activity(A, a.flags, a.list, q0_on ? a.list0 : nullptr, a.host_dev, id, c->stream);
        a.last_used = q0_on ? 2 : 1;
    }
    int cur = 0;   // 0: the caller's arrays hold the current state
    int m = 0, nex = 0, nlaunch = 0, npair = 0;
    const int end = first + substeps;
    HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    if ((rc = stats_mark(c))) return rc;
    if (band) {
        HIP_TRY(c, hipEventRecord(c->band_ev_pair, c->stream));      // the first band starts behind everything queued so far
        HIP_TRY(c, hipEventRecord(c->band_ev_band, c->stream));      // (nothing for the first pair launch to wait for)
    }
    // EXPERIMENT (CSI_EXP_OVERLAP bit 1, tiles connected to themselves: profiles/r06_tile_overlap.txt): consecutive launches of the peer
    // transport on two streams -- with every tile on the flag protocol (bit 0) nothing but the flags orders launch n + 1 behind launch n
    const bool two_streams = peer && !band && c->tune.exp_overlap > 0 && (c->tune.exp_overlap & 2);
    if (two_streams) {
        if (!c->band_stream) HIP_TRY(c, hipStreamCreateWithFlags(&c->band_stream, hipStreamNonBlocking));
        for (hipEvent_t& e : c->exp_ev) if (!e) HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIP_TRY(c, hipEventRecord(c->exp_ev[0], c->stream));
        HIP_TRY(c, hipStreamWaitEvent(c->band_stream, c->exp_ev[0], 0));
    }
    // fold band on reserved CUs (csi_fold.hip, ensure_band): the pair launches beside it on their own stream, masked to the other CUs
    hipStream_t ps = (band && c->pair_stream) ? c->pair_stream : c->stream;
    if (ps != c->stream) {
        HIP_TRY(c, hipEventRecord(c->exp_ev[0], c->stream));
        HIP_TRY(c, hipStreamWaitEvent(ps, c->exp_ev[0], 0));
    }
    const bool skip_pair = band && c->tune.exp_band_only > 0;      // TIMING EXPERIMENT (wrong results): the band's launches alone
    int nflip = 0;
    auto launch_stream = [&]() { return ps != c->stream ? ps : ((two_streams && (nflip++ & 1)) ? c->band_stream : c->stream); };
    for (int s = first; s < end;) {
        const bool ufirst = (s % 2) == 0;                  // split_explicit_momentum_equations.jl:178
        if (pairs && end - s >= 2 && m + 1 < kb) {
            const int mp = m / 2;
            if (band) {
                HIP_TRY(c, hipStreamWaitEvent(ps, c->band_ev_band, 0));       // the previous band: this launch's rows M + 1 .. M + 4
                if ((rc = band_substeps(c, *band, fc, cur, s, 2, s + 2 == end))) return rc;
                nlaunch += band_launches(c, 2);
            }
            // (write_diag bit 2: the live tiles of GA only -- not in the first two pair launches, which bring BOTH buffers to the
            //  quiescent tiles' fixed point, halo images included, nor in the last launch, which stores every tile's diagnostics)
            const bool live_only = act_on && npair >= 2 && s + 2 != end;
            const bool start_only = q0_on && npair < 2 && s + 2 != end;      // (the first two launches: all but the tiles quiescent from the start)
            const FusedGeom& GL = (live_only || start_only) ? GA : GP[mp];
            if (!skip_pair)
            launch_fused_pair(c->dev_tables + NSINGLE + ((mp * 2 + cur) * 2 + (ufirst ? 1 : 0)),
                              c->metric_kind == CSI_METRIC_FULL ? 2 : (c->coef.uniform != 0 ? 0 : 1), ufirst,
                              has_walls(c) || masked || force || peer_dld_bit != 0, masked, force, P.free_drift != 0, extra_kind, common_forcing, GL.nstrips, GL.nchunks, GL.rows,
                              (s + 2 == end ? 1 : 0) | (live_only ? 4 : 0) | (start_only ? 8 : 0),
                              peer ? (++c->peer.seq | peer_dld_bit) : (c->tune.peer_kernel > 0 ? 1ull : 0ull), launch_stream());
            ++npair;
            if (band) HIP_TRY(c, hipEventRecord(c->band_ev_pair, ps));
            m += 2; s += 2;
        } else if (single_by_pair) {
            // one sub-step through the two-sub-steps kernel (write_diag bit 1): masks, array forcing, per-point metrics
            if (band) {
                HIP_TRY(c, hipStreamWaitEvent(ps, c->band_ev_band, 0));
                if ((rc = band_substeps(c, *band, fc, cur, s, 1, s + 1 == end))) return rc;
                nlaunch += band_launches(c, 1);
            }
            if (!skip_pair)
            launch_fused_pair(c->dev_tables + ((m * 2 + cur) * 2 + (ufirst ? 1 : 0)),
                              c->metric_kind == CSI_METRIC_FULL ? 2 : (c->coef.uniform != 0 ? 0 : 1), ufirst,
                              has_walls(c) || masked || force || peer_dld_bit != 0, masked, force, P.free_drift != 0, extra_kind, common_forcing, G[m].nstrips, G[m].nchunks, G[m].rows,
                              2 | (s + 1 == end ? 1 : 0), peer ? (++c->peer.seq | peer_dld_bit) : 0ull, launch_stream());
            if (band) HIP_TRY(c, hipEventRecord(c->band_ev_pair, ps));
            m += 1; s += 1;
        } else if (masked || force || c->metric_kind == CSI_METRIC_FULL) {
            // (no pair kernel for this grid -- halo < 4, tiny tiles: the three kernels in place on whichever buffer is current)
            EvpDev Q = P;
            const FRef* b = cur == 0 ? orig : alt;
            Q.u = b[0]; Q.v = b[1]; Q.s11 = b[2]; Q.s22 = b[3]; Q.s12 = b[4];
            Q.write_diag = (s + 1 == end);
            const int V = tiled ? W - 2 * m : 2;
            launch_fast_stress(Q, stress_range(c, V), fc, c->stream);
            if (ufirst) { launch_fast_ustep(Q, first_u_range(c, V), imu, fc, c->stream); launch_fast_vstep(Q, second_range(c, V), imv, fc, c->stream); }
            else { launch_fast_vstep(Q, first_v_range(c, V), imv, fc, c->stream); launch_fast_ustep(Q, second_range(c, V), imu, fc, c->stream); }
            m += 1; s += 1;
            cur ^= 1;           // undone below: this sub-step did not switch buffers
            nlaunch += 2;
        } else {
            launch_fused_substep(c->dev_tables + ((m * 2 + cur) * 2 + (ufirst ? 1 : 0)), c->coef.uniform != 0, ufirst,
                                 G[m].nstrips, G[m].nchunks, G[m].rows, s + 1 == end, c->stream);
            m += 1; s += 1;
        }
        cur ^= 1;
        ++nlaunch;
        if (tiled && (m == kb || s == end)) {
            if ((rc = exchange_tile(cur == 0 ? orig : alt))) return rc;
            m = 0;
            ++nex;
        } else if (m >= kb) {
            m = 0;
        }
    }
    if (band) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->band_ev_band, 0));
    if (two_streams || ps != c->stream) {
        HIP_TRY(c, hipEventRecord(c->exp_ev[1], ps != c->stream ? ps : c->band_stream));
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->exp_ev[1], 0));
    }
    if (peer) {
        // the neighbours' last launch wrote into this rank's halos: wait for all of it before anything later on this stream
        // (the copy back, finalize_rheology!, the next exchange) reads them
        launch_wait_peers(c->peer.slots, c->peer.sync_rank, csi_context::Peer::SLOTS, c->peer.nbr_wait, c->peer.seq, c->peer.err, c->stream);
        HIP_TRY(c, hipMemcpyAsync(c->peer.err_host, c->peer.err, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    if ((rc = stats_mark(c))) return rc;
    stats_launches(c, nlaunch);
    if (cur == 1)   // the result sits in the library's buffers
        for (int q = 0; q < 5; ++q) {
            const Bound& b = c->f[kPing[q]];
            HIP_TRY(c, hipMemcpyAsync(b.p, c->alt[q], c->alt_elems[q] * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        }
    HIP_TRY(c, hipGetLastError());
    c->last_exchanges = nex;
    c->last_k = k;
    c->last_launches = nlaunch;
    c->last_substeps = substeps;
    c->last_used_pairs = pairs && substeps >= 2;
    return CSI_OK;
}


// One sub-cycle on the peer transport: an RCCL exchange of u, v, sigma brings the halos up to date (and orders this rank behind
// whatever its neighbours did last), then the connected sides count as periodic ones for the launch loop.
int32_t run_fused_peer(csi_context* c, double dt, const FastCoef& fc, int substeps, int first) {
    int32_t rc;
    FRef orig[5];
    for (int q = 0; q < 5; ++q) orig[q] = ref_of(c, kPing[q]);
    // Both ping-pong buffers start identical where no launch ever writes (cells beyond walls).  The copy comes BEFORE the
    // exchange: once a neighbour has received this rank's message it may start its first launch, whose halo images land in this
    // rank's second buffer -- they must not be overwritten by a copy that is still on its way.  (Halos beyond connected sides
    // need no copy: the neighbours' images rewrite all H layers at every launch.)
    if ((rc = ensure_alt(c))) return rc;
    const bool fold = c->g.yhi == SIDE_FOLD;
    if (has_walls(c) || fold)
        for (int q = 0; q < 5; ++q)
            HIP_TRY(c, hipMemcpyAsync(c->alt[q], c->f[kPing[q]].p, c->alt_elems[q] * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    const int W = std::min(std::min(c->Hx, c->Hy), 4);
    if ((rc = exchange_refs(c, orig, 5, W))) return rc;
    // the fold tile of a y partition: its three-kernel band works on the tile as it is (FoldBand); the pair launches see the
    // tile cut below the band, like every other tile with its connected sides turned into periodic ones (PeerView)
    FoldBand bd;
    const EvpDev Pfull = evp_dev(c, dt);
    if (fold) {
        bd.M = c->Ny - c->Hy - 4;
        bd.tiled = false; bd.k = 2;
        bd.g_full = c->g; bd.Ny_full = c->Ny;
        bd.P = Pfull;
        bd.imu = image_spec(c, CSI_F_U); bd.imv = image_spec(c, CSI_F_V);
        bd.rs = stress_range(c); bd.ru1 = first_u_range(c); bd.rv1 = first_v_range(c); bd.r2 = second_range(c);
    }
    PeerView view(c);
    bd.g_cut = c->g;
    EvpDev P = Pfull;           // (arrays and per-row pointers of the tile as it is; only the grid descriptor differs)
    P.g = c->g;
    rc = run_fused(c, P, fc, substeps, first, true, fold ? &bd : nullptr);
    c->last_exchanges = 1;
    return rc;
}


int32_t do_subcycle(csi_context* c, double dt, int substeps, int first) {
    int32_t rc;
    if ((rc = peer_check_entry(c))) return rc;
    if ((rc = ensure_row_constant(c))) return rc;
    {                                                // :170-171, both fields in one batch of two launches
        HaloBatch B{};
        B.f[0] = ref_of(c, CSI_F_U); B.im[0] = image_spec(c, CSI_F_U);
        B.f[1] = ref_of(c, CSI_F_V); B.im[1] = image_spec(c, CSI_F_V);
        B.n = 2;
        launch_fill_halo_batch(B, c->g, c->stream);
    }
    const bool tiled = is_tiled(c);
    const int uv[2] = {CSI_F_U, CSI_F_V};
    // halo exchange of u, v every k sub-steps with width 2k (k = 1: every sub-step; the reference is the
    // k = substeps extreme with its 2*substeps+3 halo, split_explicit_momentum_equations.jl:51-64)
    const int k = exchange_interval(c);
    const int W = 2 * k;
    // sigma is history dependent (sigma += (sigma' - sigma) / gamma): with k = 1 the ring-1 values are
    // recomputed every sub-step and stay identical to the neighbour's; with k > 1 the outer rings skip
    // updates inside a batch, so sigma travels with u, v.  alpha is recomputed before every use.
    const int uvs[5] = {CSI_F_U, CSI_F_V, CSI_F_S11, CSI_F_S22, CSI_F_S12};
    const int nxf = k > 1 ? 5 : 2;
    (void)uv;
    if (c->free_drift) {
        // free-drift velocities of marginal ice depend on the forcing only: once per sub-cycle, every point whose
        // four-point averages stay inside the parent arrays
        const int src[2] = {CSI_F_U, CSI_F_V};
        for (int q = 0; q < 2; ++q) {
            const size_t n = (size_t)c->f[src[q]].ld * (size_t)c->f[src[q]].nj;
            if (c->fd_elems[q] != n) {
                if (c->fd[q]) { HIP_TRY(c, hipStreamSynchronize(c->stream)); hipFree(c->fd[q]); c->fd[q] = nullptr; }
                HIP_TRY(c, hipMalloc((void**)&c->fd[q], n * sizeof(double)));
                HIP_TRY(c, hipMemsetAsync(c->fd[q], 0, n * sizeof(double), c->stream));
                c->fd_elems[q] = n;
            }
        }
        launch_free_drift(evp_dev(c, dt), Range{2 - c->Hx, c->Nx + c->Hx - 1, 2 - c->Hy, c->Ny + c->Hy - 1}, c->stream);
    }
    EvpDev P = evp_dev(c, dt);
    const ImageSpec imu = image_spec(c, CSI_F_U), imv = image_spec(c, CSI_F_V);
    const bool fast = c->mode == CSI_MODE_FAST;
    FastCoef fc = c->coef;
    {
        const double ie = 1.0 / P.ecc;
        fc.em2 = ie * ie;
        fc.ca_dt = 0.5 * (P.ca * dt); fc.hkc = fc.ca_dt * fc.uni[FC_RAZC]; fc.hkf = fc.ca_dt * fc.uni[FC_RAZF]; fc.hk1 = 0.5 * (1.0 - ie * ie);
        fc.rdt = 1.0 / dt;
        fc.Dmin2 = P.Dmin * P.Dmin; fc.rDmin = 1.0 / P.Dmin;
        fc.amin2 = P.amin * P.amin; fc.amax2 = P.amax * P.amax; fc.ramin = 1.0 / P.amin; fc.ramax = 1.0 / P.amax;
    }
    if (fast && !fast_supported(P)) return fail(c, CSI_ERR_UNSUPPORTED, "CSI_MODE_FAST does not support this configuration yet; use CSI_MODE_STRICT");
    // immersed masks: only the two-sub-steps-per-launch kernel takes them (a trailing odd sub-step falls back to the
    // three kernels inside run_fused)
    const int pfk = pair_forcing_kind(P);
    const bool pair_only = P.g.has_mask || pfk == 1 || c->metric_kind == CSI_METRIC_FULL;      // configurations only the two-sub-steps kernel takes
    // the peer halo transport (tiles) needs none of the RCCL batching constraints (an even exchange interval): decide it first
    bool peer = false;
    if (fast && c->fusion && substeps > 0 && (rc = peer_decide(c, P, substeps, &peer))) return rc;
    if (!peer && fast && fold_band_supported(c, P, substeps)) {
        c->peer.last = 0;
        if ((rc = run_fused_fold(c, P, fc, substeps, first))) return rc;
        c->timed = true;
        c->launches_per_substep = 1;
        c->last_fused = 2;
        return CSI_OK;
    }
    const bool fuse = peer || (fast && c->fusion && substeps > 0 &&
                               (pair_only ? (pfk >= 0 && pair_supported(c) && (!tiled || k % 2 == 0) && substeps >= 2)
                                          : fused_supported(P)));
    if (fuse) {
        c->peer.last = peer ? 1 : 0;
        if ((rc = peer ? run_fused_peer(c, dt, fc, substeps, first) : run_fused(c, P, fc, substeps, first))) return rc;
        c->timed = true;
        c->launches_per_substep = 1 + ((tiled && k == 1) ? 3 : 0);
        c->last_fused = c->last_used_pairs ? 2 : 1;
        return CSI_OK;
    }
    c->last_fused = 0;
    c->peer.last = 0;
    if (tiled && (rc = exchange(c, uvs, nxf, W))) return rc;
    HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    if ((rc = stats_mark(c))) return rc;
    int m = 0, nex = 0;   // position inside the exchange batch
    for (int s = first; s < first + substeps; ++s) {
        const int V = W - 2 * m;
        const Range rs = stress_range(c, V), ru1 = first_u_range(c, V), rv1 = first_v_range(c, V), r2 = second_range(c, V);
        if (fast) {
            P.write_diag = (s == first + substeps - 1);
            launch_fast_stress(P, rs, fc, c->stream);
        } else {
            launch_strict_visc(P, rs, c->stream);          // compute_stresses!, evp:222-234
            launch_strict_stress(P, rs, c->stream);
        }
        if ((s % 2) == 0) {                                // :178-182
            if (fast) { launch_fast_ustep(P, ru1, imu, fc, c->stream); launch_fast_vstep(P, r2, imv, fc, c->stream); }
            else { launch_strict_ustep(P, ru1, imu, c->stream); launch_strict_vstep(P, r2, imv, c->stream); }
        } else {                                           // :184-187
            if (fast) { launch_fast_vstep(P, rv1, imv, fc, c->stream); launch_fast_ustep(P, r2, imu, fc, c->stream); }
            else { launch_strict_vstep(P, rv1, imv, c->stream); launch_strict_ustep(P, r2, imu, c->stream); }
        }
        ++m;
        if (tiled && (m == k || s == first + substeps - 1)) {   // RCCL send/recv of the u, v halos
            if ((rc = exchange(c, uvs, nxf, W))) return rc;
            m = 0;
            ++nex;
        }
    }
    HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    if ((rc = stats_mark(c))) return rc;
    stats_launches(c, substeps * (fast ? 3 : 4));
    HIP_TRY(c, hipGetLastError());
    c->timed = true;
    c->launches_per_substep = (fast ? 3 : 4) + ((tiled && k == 1) ? 3 : 0);
    c->last_exchanges = nex;
    c->last_k = k;
    return CSI_OK;
}

int32_t do_finalize(csi_context* c) {
    HaloBatch B{};
    for (int fid : {CSI_F_S11, CSI_F_S12, CSI_F_S22}) { B.f[B.n] = ref_of(c, fid); B.im[B.n] = image_spec(c, fid); ++B.n; }
    launch_fill_halo_batch(B, c->g, c->stream);
    HIP_TRY(c, hipGetLastError());
    // fill_halo_regions!(sigma) across tiles.  After a sub-cycle on the peer transport there is nothing to move: the neighbours' last
    // launch stored the images of their sigma into all H halo layers beyond the connected sides (and k_wait_peers has seen them
    // land) -- exactly the values an exchange would bring; what it would ALSO bring are the neighbours' own y fills in the corners
    // (beyond a wall next to a connected x side: nobody stores mirror images of sigma): the same fill on this tile's halo columns
    if (c->peer.last && is_tiled(c)) {
        launch_fill_halo_xcolumns(B, c->g, c->stream);
        HIP_TRY(c, hipGetLastError());
        return CSI_OK;
    }
    const int sg[3] = {CSI_F_S11, CSI_F_S12, CSI_F_S22};
    return exchange(c, sg, 3, c->Hx < c->Hy ? c->Hx : c->Hy);
}

int32_t need_evp(csi_context* c) {
    int32_t rc = need(c, {CSI_F_U, CSI_F_V, CSI_F_H, CSI_F_A, CSI_F_S11, CSI_F_S22, CSI_F_S12, CSI_F_UN, CSI_F_VN,
                          CSI_F_P, CSI_F_ALPHA, CSI_F_DELTA, CSI_F_ZETA_F, CSI_F_ZETA_C});
    if (rc) return rc;
    if (!c->evp_set) return fail(c, CSI_ERR_NOT_BOUND, "csi_evp_params_set has not been called");
    if ((rc = check_stress_fields(c, CSI_STRESS_TOP))) return rc;
    if ((rc = check_stress_fields(c, CSI_STRESS_BOTTOM))) return rc;
    if (c->Hx < 2 || c->Hy < 2) return fail(c, CSI_ERR_INVALID_ARGUMENT, "EVP needs halo >= 2");
    if (c->free_drift) {   // stress_balance_free_drift.jl:21-35: exactly one of the two stresses is a SemiImplicitStress
        const bool ts = c->stress[CSI_STRESS_TOP].kind == CSI_STRESS_SEMI_IMPLICIT, bs = c->stress[CSI_STRESS_BOTTOM].kind == CSI_STRESS_SEMI_IMPLICIT;
        if (ts == bs) return fail(c, CSI_ERR_INVALID_ARGUMENT, "StressBalanceFreeDrift needs exactly one SemiImplicitStress (top or bottom)");
    }
    if (c->Nx < c->Hx || c->Ny < c->Hy) return fail(c, CSI_ERR_UNSUPPORTED, "tile smaller than its halo");
    if (is_tiled(c) && !c->tile.set) return fail(c, CSI_ERR_NOT_BOUND, "connected topology but csi_tile_set has not been called");
    return sync_coriolis(c);
}

int32_t do_time_step_momentum(csi_context* c, double dt, int substeps, int rk_reset) {
    int32_t rc;
    if (rk_reset) {                                         // reset_velocities! :89-93
        if ((rc = need(c, {CSI_F_UM, CSI_F_VM}))) return rc;
        if ((rc = copy_parent(c, CSI_F_U, CSI_F_UM))) return rc;
        if ((rc = copy_parent(c, CSI_F_V, CSI_F_VM))) return rc;
    }
    if ((rc = do_initialize(c))) return rc;                 // :130
    // update_external_stress! :133-134: halos of the forcing fields (local boundary conditions, then tiles)
    // ... and of model.forcing.u / .v when they are arrays: inside an exchange batch the velocity kernels run on ranges that
    // extend into the halo and read the forcing there (elasto_visco_plastic_rheology.jl:391-401 is evaluated at every point
    // the step updates), so beyond a connected side the halo must hold the neighbour's values
    if ((c->f[CSI_F_FORCING_U].p != nullptr) != (c->f[CSI_F_FORCING_V].p != nullptr))
        return fail(c, CSI_ERR_NOT_BOUND, "model.forcing arrays: bind both CSI_F_FORCING_U and CSI_F_FORCING_V or neither");
    const int forcing_ids[6] = {CSI_F_TOP_U, CSI_F_TOP_V, CSI_F_BOT_U, CSI_F_BOT_V, CSI_F_FORCING_U, CSI_F_FORCING_V};
    for (int id : forcing_ids)
        if (c->f[id].p && (rc = fill_halo(c, id))) return rc;
    if (is_tiled(c)) {
        int ff[6], n = 0;
        for (int id : forcing_ids) if (c->f[id].p) ff[n++] = id;
        if (n && (rc = exchange(c, ff, n, c->Hx < c->Hy ? c->Hx : c->Hy))) return rc;
    }
    if ((rc = do_subcycle(c, dt, substeps, 1))) return rc;  // :170-189
    return do_finalize(c);                                  // :192
}

AdvDev adv_dev(const csi_context* c, int scheme, double dt, int from_cache) {
    AdvDev A{};
    A.g = c->g;
    A.u = ref_of(c, CSI_F_U); A.v = ref_of(c, CSI_F_V); A.h = ref_of(c, CSI_F_H); A.a = ref_of(c, CSI_F_A);
    A.Gh = ref_of(c, CSI_F_GH); A.Ga = ref_of(c, CSI_F_GA); A.hm = ref_of(c, CSI_F_HM); A.am = ref_of(c, CSI_F_AM);
    A.has_snow = c->f[CSI_F_HS].p != nullptr && c->f[CSI_F_GHS].p != nullptr;     // snow thickness: the third tracer
    if (A.has_snow) { A.hs = ref_of(c, CSI_F_HS); A.Ghs = ref_of(c, CSI_F_GHS); A.hsm = ref_of(c, CSI_F_HSM); }
    A.scheme = scheme; A.dt = dt; A.from_cache = from_cache;
    A.w32 = c->weno_w32;
    A.nt = c->tune.adv_nt > 0 ? c->tune.adv_nt : 0;
    A.fill_images = 0; A.im = image_spec(c, CSI_F_H);
    return A;
}

// in_step: called from csi_time_step_*.  tracers_filled: the tracer update of this stage already wrote the halo images of
// h, aice [, hs] with its stores (no mask, no thermodynamic step after it).  Inside a step the velocities are prognostic
// fields only with dynamics (sea_ice_model.jl:230,373-377): prescribed velocities keep the halos set! gave them.
int32_t do_update_state(csi_context* c, bool in_step, bool tracers_filled) {
    int32_t rc;
    if ((rc = need(c, {CSI_F_H, CSI_F_A}))) return rc;
    // mask_immersed_field_xy! of every prognostic field, then their local halo fills in one batch (two launches)
    const bool snow = c->f[CSI_F_HS].p != nullptr;
    const bool vel = c->f[CSI_F_U].p && c->f[CSI_F_V].p && (!in_step || c->evp_set);
    launch_mask_center(ref_of(c, CSI_F_H), c->g, c->stream);
    launch_mask_center(ref_of(c, CSI_F_A), c->g, c->stream);
    if (snow) launch_mask_center(ref_of(c, CSI_F_HS), c->g, c->stream);
    for (int id : {CSI_F_MASS_FLUX, CSI_F_MASS_FLUX_SNOW, CSI_F_SNOWFALL_INTERCEPTED})       // sea_ice_model.jl:387-390
        if (c->f[id].p) launch_mask_center(ref_of(c, id), c->g, c->stream);
    if (vel) {
        launch_mask_u(ref_of(c, CSI_F_U), c->g, c->stream);
        launch_mask_v(ref_of(c, CSI_F_V), c->g, c->stream);
    }
    HaloBatch B{};
    auto add = [&](int fid) { B.f[B.n] = ref_of(c, fid); B.im[B.n] = image_spec(c, fid); ++B.n; };
    if (!tracers_filled) {
        add(CSI_F_H); add(CSI_F_A);
        if (snow) add(CSI_F_HS);
    }
    if (vel) { add(CSI_F_U); add(CSI_F_V); }
    launch_fill_halo_batch(B, c->g, c->stream);
    HIP_TRY(c, hipGetLastError());
    if (is_tiled(c)) {                                      // the MPI part of fill_halo_regions!, sea_ice_model.jl:383
        int ff[5] = {CSI_F_H, CSI_F_A, CSI_F_U, CSI_F_V, CSI_F_HS};
        int n = vel ? 4 : 2;
        if (snow) ff[n++] = CSI_F_HS;
        if ((rc = exchange(c, ff, n, c->Hx < c->Hy ? c->Hx : c->Hy))) return rc;
    }
    return CSI_OK;
}

int32_t do_tendencies(csi_context* c, int scheme) {
    int32_t rc;
    if ((rc = need(c, {CSI_F_U, CSI_F_V, CSI_F_H, CSI_F_A, CSI_F_GH, CSI_F_GA}))) return rc;
    const bool third = scheme == CSI_ADVECT_WENO3 || scheme == CSI_ADVECT_UPWIND3;
    int need_h = scheme == CSI_ADVECT_WENO7 ? 4 : (scheme == CSI_ADVECT_UPWIND1 ? 1 : (third ? 2 : 3));
    if (scheme != CSI_ADVECT_UPWIND1 && scheme != CSI_ADVECT_WENO5 && scheme != CSI_ADVECT_WENO7 && scheme != CSI_ADVECT_UPWIND5 && !third)
        return fail(c, CSI_ERR_INVALID_ARGUMENT, "unknown advection scheme");
    if (c->Hx < need_h || c->Hy < need_h) return fail(c, CSI_ERR_INVALID_ARGUMENT, "halo too small for the advection scheme");
    launch_tracer_tendencies(adv_dev(c, scheme, 0.0, 0), c->mode, c->stream);
    HIP_TRY(c, hipGetLastError());
    return CSI_OK;
}
// advection = nothing: zero tendencies (horizontal_div_Uc(..., ::Nothing, ...) = zero(grid), sea_ice_advection.jl:50); the
// tracer update still runs -- dynamic_time_step! launches unconditionally -- and resets h, aice [, hs] to Psi^- at every
// RK stage (what makes the stage-wise thermodynamic steps of an RK3 step non-cumulative)
int32_t do_tendencies_or_zero(csi_context* c, int scheme) {
    if (scheme) return do_tendencies(c, scheme);
    int32_t rc;
    if ((rc = need(c, {CSI_F_GH, CSI_F_GA}))) return rc;
    for (int id : {CSI_F_GH, CSI_F_GA, CSI_F_GHS}) {
        const Bound& b = c->f[id];
        if (b.p) HIP_TRY(c, hipMemsetAsync(b.p, 0, (size_t)b.ld * (size_t)b.nj * sizeof(double), c->stream));
    }
    return CSI_OK;
}
// fill_images: the stores also write the local halo images (periodic wrap / no-flux mirror) of h, aice [, hs]
int32_t do_tracer_step(csi_context* c, double dt, int from_cache, bool fill_images) {
    int32_t rc;
    if ((rc = need(c, {CSI_F_H, CSI_F_A, CSI_F_GH, CSI_F_GA}))) return rc;
    if (from_cache && (rc = need(c, {CSI_F_HM, CSI_F_AM}))) return rc;
    if (from_cache && c->f[CSI_F_HS].p && c->f[CSI_F_GHS].p && (rc = need(c, {CSI_F_HSM}))) return rc;
    AdvDev A = adv_dev(c, 0, dt, from_cache);
    A.fill_images = fill_images ? 1 : 0;
    launch_tracer_step(A, c->stream);
    HIP_TRY(c, hipGetLastError());
    return CSI_OK;
}


}  // namespace csi_host
