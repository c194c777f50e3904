// csi_peer.hip -- peer halo transport: tile sets, set-up (IPC mapping of the neighbours' arrays and flag words), kernel tables  (split out of csi_abi.hip in round 4; see csi_ctx.h)
#include "csi_ctx.h"

namespace csi_host {

// ---- peer halo transport (csi_context::Peer) -------------------------------------------------------------------------------
// Directions: 0 W, 1 E, 2 S, 3 N, 4 SW, 5 SE, 6 NW, 7 NE (the order evp_fused2.hip's D_* and the table's FP_IMG0 rows use).

// Which tiles of a pair launch touch the halo beyond each side -- read it, store images of their own cells into the neighbour's,
// or share a 128-byte line with it -- and therefore wait for / signal that neighbour: the first nW / last nE strips, the first
// nS / last nN chunks.  size[d]: tiles in this rank's set of direction d; n[d]: slots to wait for from the neighbour in direction
// d = the size of ITS set towards this rank (tiles of one decomposition have the same shape, hence the same sets).
PeerSets peer_wait_counts(const csi_context* c, const FusedGeom& G) {
    PeerSets ps{};
    constexpr int P_LO = 4, P_W = 56;                       // evp_pair_stage.h: a strip is 64 lanes wide and owns lanes 4 .. 59
    for (int st = 0; st < G.nstrips; ++st) {
        const int i0s = G.rs.i0 - P_LO + st * P_W;
        if (i0s <= c->Hx + 16) ++ps.nW;
        if (i0s + 63 + 16 > c->Nx - c->Hx) ++ps.nE;
    }
    for (int q = 0; q < G.nchunks; ++q) {
        int ja, jb;
        chunk_rows(G, q, &ja, &jb);
        // in a side's set: the chunk reads halo rows beyond it (its footprint reaches 4 rows beyond its own: ja - 4 <= 0) or owns
        // rows whose images it stores there (rows 1 .. Hy / Ny - Hy + 1 .. Ny).  (Until round 4 the test was Hy + 4 rows wide:
        // correct but one chunk more than needed once the chunk next to the side is shorter than that.)
        const int reach = std::max(c->Hy, 4);
        if (ja <= reach) ++ps.nS;
        if (jb + reach > c->Ny) ++ps.nN;
    }
    if (c->tune.exp_overlap > 0 && (c->tune.exp_overlap & 1)) {
        // EXPERIMENT (profiles/r06_tile_overlap.txt): every tile in the sets of both sides of a connected axis -- each publishes its
        // flag at every launch and waits before it loads anything, so that consecutive launches need no stream order between them
        if (c->geom_peer & 4) ps.nS = ps.nN = G.nchunks;
        if (c->geom_peer & 2) ps.nW = ps.nE = G.nstrips;
    }
    const int sz[8] = {ps.nW * G.nchunks, ps.nE * G.nchunks, ps.nS * G.nstrips, ps.nN * G.nstrips,
                       ps.nW * ps.nS, ps.nE * ps.nS, ps.nW * ps.nN, ps.nE * ps.nN};
    for (int d = 0; d < 8; ++d) ps.size[d] = sz[d];
    for (int d = 0; d < 8; ++d) ps.n[d] = sz[kPeerOpp[d]];
    return ps;
}

// the 14 local arrays a neighbour stores images into, in Peer::arr order
void peer_local_arrays(const csi_context* c, const void* out[csi_context::Peer::NARR]) {
    for (int q = 0; q < 5; ++q) { out[q] = c->f[kPing[q]].p; out[5 + q] = c->alt[q]; }
    out[10] = c->f[CSI_F_ALPHA].p; out[11] = c->f[CSI_F_ZETA_C].p; out[12] = c->f[CSI_F_ZETA_F].p; out[13] = c->f[CSI_F_DELTA].p;
}

void peer_release(csi_context* c) {
    for (void* m : c->peer.opened) hipIpcCloseMemHandle(m);
    c->peer.opened.clear();
    c->peer.ready = false;
}


bool fold_cut_possible(const csi_context* c) {
    const GridDev& g = c->g;
    return g.yhi == SIDE_FOLD && g.xlo == SIDE_PERIODIC && g.xhi == SIDE_PERIODIC && c->Hy >= 4 && c->Ny - c->Hy - 4 >= 2 * c->Hy + 8;
}
// does the two-sub-steps kernel take this tile on the peer transport?  (P: the tile as it is)
bool peer_tile_supported(csi_context* c, const EvpDev& Pfull) {
    if (c->g.yhi != SIDE_FOLD) return pair_supported(c) && pair_forcing_kind(Pfull) >= 0;      // (the tile as it is: connected sides)
    if (!fold_cut_possible(c)) return false;
    const GridDev g = c->g;
    const int Ny = c->Ny, M = c->Ny - c->Hy - 4;
    c->Ny = M; c->g.Ny = M; c->g.yhi = SIDE_CONNECTED;                                        // cut below the band
    EvpDev P = Pfull;
    P.g = c->g;
    const bool ok = pair_supported(c) && pair_forcing_kind(P) >= 0;
    c->g = g; c->Ny = Ny;
    return ok;
}
PeerSets peer_my_sets(csi_context* c) {
    PeerView view(c);
    const Range dec = v_stress_range(c, pair_side_v(c, 2, 2));
    return peer_wait_counts(c, pair_geom(c, dec));
}

// Collective over the context's communicator: every rank publishes IPC handles of its arrays and flags, maps its neighbours'.
// Failure anywhere (no IPC, strides that differ across a side, sets larger than the flag array) makes EVERY rank stay on RCCL.
// Protocol tier the kernels run (FI_PTIER).  Automatic (csi_set_peer_tier(-1), the default): tier 1 -- an acquire fence behind the
// flags of every waiting tile -- as soon as a neighbour lives in another PROCESS or on another DEVICE; tier 0 (no fence: it rests on
// the reasoning of evp_fused2.hip about what cannot be cached) only where that reasoning has been soaked: one tile connected to itself
// and the tiles of an in-process group on one device.  Tier 0 across devices is an explicit opt-in (a passing
// bitwise check cannot prove a protocol whose failure would be a rare, timing-dependent stale line).
int peer_effective_tier(const csi_context* c) {
    if (c->peer.tier >= 0) return c->peer.tier;
    if (c->world > 1 && c->local) {
        // an in-process group: tier 0 only while all its contexts share ONE device (a group that spans GPUs got the
        // fence-free tier silently)
        std::unique_lock<std::mutex> lk(c->local->mu);
        for (int d : c->local->device) if (d >= 0 && d != c->device) return 1;
        return 0;
    }
    return c->world > 1 ? 1 : 0;
}

// Collective over the context's communicator: every rank publishes IPC handles of its arrays and flags, maps its neighbours'.
// Failure anywhere (no IPC, strides that differ across a side, sets larger than the flag array) makes EVERY rank stay on RCCL.
// Every LOCAL failure before the collectives (a device call, an allocation, an IPC handle) turns into ok = 0 and this rank still
// takes part in both of them -- an early return here would strand the other ranks inside ncclAllGather / ncclAllReduce (ADVICE
// round 3 / 4).  The one exception is the staging buffer of the RCCL all-gather itself: without it this rank cannot take part.
int32_t peer_setup(csi_context* c, bool local_ok) {
    csi_context::Peer& pr = c->peer;
    int ok = local_ok ? 1 : 0;          // (a rank whose own configuration rules the transport out still takes part: every rank or none)
    std::string soft_err;
    auto soft = [&](hipError_t e, const char* what) {
        if (e == hipSuccess) return true;
        (void)hipGetLastError();
        if (soft_err.empty()) soft_err = std::string("peer set-up: ") + what + ": " + hipGetErrorString(e);
        ok = 0;
        return false;
    };
    soft(hipSetDevice(c->device), "hipSetDevice");           // (allocations and IPC mappings below belong to the context's device)
    soft(hipStreamSynchronize(c->stream), "hipStreamSynchronize");
    peer_release(c);
    const int me = c->tile.ry * c->tile.Rx + c->tile.rx;
    if (!pr.slots) {
        // fine-grained (uncached) device memory where the runtime offers it: the flags are polled while remote ranks write them
        if (hipExtMallocWithFlags((void**)&pr.slots, sizeof(unsigned long long) * 8 * csi_context::Peer::SLOTS, hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            pr.slots = nullptr;
            if (!soft(hipMalloc((void**)&pr.slots, sizeof(unsigned long long) * 8 * csi_context::Peer::SLOTS), "hipMalloc(flags)")) pr.slots = nullptr;
        }
    }
    if (!pr.err && !soft(hipMalloc((void**)&pr.err, sizeof(unsigned)), "hipMalloc(error word)")) pr.err = nullptr;
    if (!pr.err_host && !soft(hipHostMalloc((void**)&pr.err_host, sizeof(unsigned), hipHostMallocDefault), "hipHostMalloc(error word)")) pr.err_host = nullptr;
    {
        const size_t need_x = sizeof(PeerRec) * kPeerRecs * (size_t)(c->world + 1) + 64;      // (a later csi_comm_init may have a larger world)
        if (need_x > pr.xbuf_cap) {
            if (pr.xbuf) hipFree(pr.xbuf);
            pr.xbuf = nullptr; pr.xbuf_cap = 0;
            HIP_TRY(c, hipMalloc((void**)&pr.xbuf, need_x));      // (the all-gather's own staging: the one failure this rank cannot report collectively)
            pr.xbuf_cap = need_x;
        }
    }
    if (!pr.slots || !pr.err || !pr.err_host) ok = 0;
    pr.seq = 0;
    pr.ny_below = c->Ny;
    // neighbours: where this tile's images go (a periodic or wall component keeps the coordinate: wraps / mirrors are local
    // in that direction) and whom to wait for (connected components only)
    for (int d = 0; d < 8; ++d) {
        pr.sync_rank[d] = tile_neighbor(c->tile, kPeerDx[d], kPeerDy[d], c->g.xlo, c->g.xhi, c->g.ylo, c->g.yhi);
        int rx = c->tile.rx, ry = c->tile.ry;
        if (kPeerDx[d] < 0 && c->g.xlo == SIDE_CONNECTED) rx = (rx - 1 + c->tile.Rx) % c->tile.Rx;
        if (kPeerDx[d] > 0 && c->g.xhi == SIDE_CONNECTED) rx = (rx + 1) % c->tile.Rx;
        if (kPeerDy[d] < 0 && c->g.ylo == SIDE_CONNECTED) ry = (ry - 1 + c->tile.Ry) % c->tile.Ry;
        if (kPeerDy[d] > 0 && c->g.yhi == SIDE_CONNECTED) ry = (ry + 1) % c->tile.Ry;
        pr.img_rank[d] = ry * c->tile.Rx + rx;
    }
    const void* local[csi_context::Peer::NARR];
    peer_local_arrays(c, local);
    const int64_t lds[csi_context::Peer::NARR] = {c->f[CSI_F_U].ld, c->f[CSI_F_V].ld, c->f[CSI_F_S11].ld, c->f[CSI_F_S22].ld, c->f[CSI_F_S12].ld,
                                                  c->f[CSI_F_U].ld, c->f[CSI_F_V].ld, c->f[CSI_F_S11].ld, c->f[CSI_F_S22].ld, c->f[CSI_F_S12].ld,
                                                  c->f[CSI_F_ALPHA].ld, c->f[CSI_F_ZETA_C].ld, c->f[CSI_F_ZETA_F].ld, c->f[CSI_F_DELTA].ld};
    std::vector<PeerRec> mine(kPeerRecs), all((size_t)kPeerRecs * c->world);
    for (int q = 0; q < kPeerRecs; ++q) {
        const void* ptr = q < csi_context::Peer::NARR ? local[q] : (const void*)pr.slots;
        PeerRec& r = mine[q];
        memset(&r, 0, sizeof r);
        r.ld = q < csi_context::Peer::NARR ? lds[q] : 0;
        r.local_ptr = (uint64_t)ptr;
        if (q == 0 && local_ok) {      // (the launch geometry is host arithmetic: independent of the soft failures above)
            const PeerSets ps = peer_my_sets(c);
            for (int d = 0; d < 8; ++d) { r.set_size[d] = ps.size[d]; pr.set_sig[d] = ps.size[d]; if (ps.size[d] >= csi_context::Peer::SLOTS) ok = 0; }
        }
        if (ok && c->world > 1 && !c->local) {               // (a single rank / an in-process group addresses the arrays directly)
            hipDeviceptr_t base = nullptr; size_t size = 0;
            if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)ptr) != hipSuccess || hipIpcGetMemHandle(&r.handle, base) != hipSuccess) {
                (void)hipGetLastError();
                ok = 0;
            } else {
                r.offset = (uint64_t)((const char*)ptr - (const char*)base);
            }
        }
        r.ok = ok;
    }
    if (c->local) {
        std::vector<uint8_t> bytes;
        int32_t lrc;
        if ((lrc = local_allgather(c, mine.data(), sizeof(PeerRec) * kPeerRecs, bytes))) return lrc;
        memcpy(all.data(), bytes.data(), bytes.size());
    } else if (c->hostg) {
        std::vector<uint8_t> bytes;
        if (!hostgroup_allgather(c->hostg, mine.data(), sizeof(PeerRec) * kPeerRecs, bytes, &c->err)) return CSI_ERR_COMM;
        memcpy(all.data(), bytes.data(), bytes.size());
    } else if (c->world > 1) {
        const size_t nb = sizeof(PeerRec) * kPeerRecs;
        // (a local copy that fails must not keep this rank out of the collectives: it takes part and votes 0 below)
        soft(hipMemcpy(pr.xbuf, mine.data(), nb, hipMemcpyHostToDevice), "hipMemcpy(records)");
        NCCL_TRY(c, ncclAllGather(pr.xbuf, pr.xbuf + nb, nb, ncclUint8, c->comm, c->stream));      // (an RCCL error is fatal for the communicator anyway)
        if (!soft(hipStreamSynchronize(c->stream), "hipStreamSynchronize(all-gather)") ||
            !soft(hipMemcpy(all.data(), pr.xbuf + nb, nb * c->world, hipMemcpyDeviceToHost), "hipMemcpy(all records)"))
            for (PeerRec& r : all) r.ok = 0;
    } else {
        all = mine;
    }
    // Every rank has arrived (and drained its stream on entry): no kernel anywhere still publishes into this rank's flag array or
    // raises its abort words.  NOW the flags, the abort words and the error word are cleared -- cleared before the all-gather, a
    // neighbour still finishing its last launch (or its aborted sub-cycle) could write them again behind the memset and the
    // restarted launch numbers would meet stale flags.  No rank launches before the all-reduce below has returned.
    if (pr.slots) soft(hipMemset(pr.slots, 0, sizeof(unsigned long long) * 8 * csi_context::Peer::SLOTS), "hipMemset(flags)");
    if (pr.err) soft(hipMemset(pr.err, 0, sizeof(unsigned)), "hipMemset(error word)");
    if (pr.err_host) *pr.err_host = 0;
    // map the neighbours' buffers (one mapping per distinct allocation)
    struct Mapped { int rank; hipIpcMemHandle_t h; void* p; };
    std::vector<Mapped> cache;
    auto resolve = [&](int rank, int q, void** out) -> bool {
        const PeerRec& r = all[(size_t)rank * kPeerRecs + q];
        if (!r.ok) return false;
        if (c->local) { *out = (void*)r.local_ptr; return true; }
        for (const Mapped& m : cache)
            if (m.rank == rank && memcmp(&m.h, &r.handle, sizeof r.handle) == 0) { *out = (char*)m.p + r.offset; return true; }
        void* mp = nullptr;
        if (hipIpcOpenMemHandle(&mp, r.handle, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); return false; }
        cache.push_back(Mapped{rank, r.handle, mp});
        pr.opened.push_back(mp);
        *out = (char*)mp + r.offset;
        return true;
    };
    for (int d = 0; d < 8 && ok; ++d) {
        const int r = pr.img_rank[d];
        for (int q = 0; q < csi_context::Peer::NARR; ++q) {
            if (r == me) { pr.arr[d][q] = const_cast<void*>(local[q]); continue; }
            if (!resolve(r, q, &pr.arr[d][q])) { ok = 0; break; }
        }
        // the neighbour's row strides may differ from this tile's (a Bounded x direction partitioned in x gives the easternmost
        // tile one more column of Face points): one difference per direction and location in x, which the image stores add per
        // parent row -- provided the neighbour's arrays of one location share a stride, as this tile's do
        pr.dld[d][0] = pr.dld[d][1] = 0;
        if (ok && r != me) {
            static const int cls[csi_context::Peer::NARR] = {1, 0, 0, 0, 1, 1, 0, 0, 0, 1, 0, 0, 1, 0};      // Face in x: u, sigma12, zeta_f
            for (int q = 0; q < csi_context::Peer::NARR; ++q) {
                const int64_t diff = all[(size_t)r * kPeerRecs + q].ld - lds[q];
                if (diff != all[(size_t)r * kPeerRecs + (cls[q] ? 0 : 1)].ld - lds[cls[q] ? 0 : 1] || diff < -64 || diff > 64) { ok = 0; break; }
                pr.dld[d][cls[q]] = (int)diff * 8;
            }
        }
        pr.nbr_slots[d] = nullptr;
        pr.nbr_wait[d] = (ok && pr.sync_rank[d] >= 0) ? all[(size_t)pr.sync_rank[d] * kPeerRecs].set_size[kPeerOpp[d]] : 0;
        if (ok && pr.sync_rank[d] >= 0) {
            void* sp = pr.slots;
            if (pr.sync_rank[d] != me && !resolve(pr.sync_rank[d], csi_context::Peer::NARR, &sp)) ok = 0;
            pr.nbr_slots[d] = (unsigned long long*)sp;
        }
    }
    if (c->local) {
        int32_t lrc;
        if ((lrc = local_allreduce_min(c, &ok))) return lrc;
    } else if (c->hostg) {
        std::vector<uint8_t> bytes;
        if (!hostgroup_allgather(c->hostg, &ok, sizeof(int), bytes, &c->err)) return CSI_ERR_COMM;
        for (int r = 0; r < c->world; ++r) { int x; memcpy(&x, bytes.data() + (size_t)r * sizeof(int), sizeof(int)); if (x < ok) ok = x; }
    } else if (c->world > 1) {                               // every rank or none
        int* flag = (int*)pr.xbuf;
        // (a failing upload votes with a zeroed word: hipMemset is tried, and whatever the word holds this rank itself ends with ok = 0)
        const bool up = soft(hipMemcpy(flag, &ok, sizeof(int), hipMemcpyHostToDevice), "hipMemcpy(vote)");
        // (... and when the word can be neither written nor zeroed it still holds all-gather bytes: this rank would vote non-zero while ending
        //  with ok = 0 itself -- a hard error instead of ranks that disagree about the transport)
        if (!up && hipMemset(flag, 0, sizeof(int)) != hipSuccess)
            return fail(c, CSI_ERR_HIP, "peer set-up: the vote word can be neither uploaded nor cleared (the ranks could disagree about the halo transport)");
        NCCL_TRY(c, ncclAllReduce(flag, flag, 1, ncclInt32, ncclMin, c->comm, c->stream));
        int voted = 0;
        if (soft(hipStreamSynchronize(c->stream), "hipStreamSynchronize(vote)") && soft(hipMemcpy(&voted, flag, sizeof(int), hipMemcpyDeviceToHost), "hipMemcpy(vote back)"))
            ok = ok && voted;
        else ok = 0;
    }
    if (!ok) { peer_release(c); pr.failed = true; if (!soft_err.empty()) c->err = soft_err + " (every rank stays on the RCCL exchange)"; return CSI_OK; }
    pr.aborted = false;
    for (int q = 0; q < csi_context::Peer::NARR; ++q) pr.sig[q] = local[q];
    pr.ready = true;
    return CSI_OK;
}

// Does this sub-cycle run on the peer transport?  Every condition is the same on all ranks of a decomposition (they share the
// configuration and the tile shape), so the ranks agree without talking; the set-up itself is collective.
int32_t peer_decide(csi_context* c, const EvpDev& P, int substeps, bool* use) {
    *use = false;
    csi_context::Peer& pr = c->peer;
    if (!is_tiled(c) || !pr.want || pr.failed || !has_comm(c) || !c->tile.set) return CSI_OK;
    if (c->exch_k > 0) return CSI_OK;                        // an explicit exchange interval asks for the RCCL exchange
    if (c->mode != CSI_MODE_FAST || !c->fusion || !c->pairing || substeps < 1) return CSI_OK;      // (an odd count ends with one single-mode launch)
    // Everything above is the same on every rank; what follows may differ from tile to tile (the fold lives on the northernmost
    // tile of a y partition only, a Bounded x partition has tiles of unequal strides): those conditions go INTO the collective
    // set-up, which answers with the minimum over the ranks -- every rank or none.
    const bool local_ok = peer_tile_supported(c, P) &&
                          c->f[CSI_F_U].ld == c->f[CSI_F_S12].ld && c->f[CSI_F_V].ld == c->f[CSI_F_S11].ld &&
                          c->Nx >= 128;                      // (a wave's x images all go to ONE neighbour: evp_fused2.hip)
    int32_t rc;
    if ((rc = ensure_alt(c))) return rc;
    const void* local[csi_context::Peer::NARR];
    peer_local_arrays(c, local);
    bool same = pr.ready;
    for (int q = 0; q < csi_context::Peer::NARR && same; ++q) same = pr.sig[q] == local[q];
    if (same && local_ok && c->tune.no_geom_sig <= 0) {
        // the neighbours wait for as many flags as this rank's tile sets had at set-up: a launch geometry that has changed since
        // (the tile count follows the forcing kinds: array forcing bound after the first sub-cycle) needs a new set-up -- on every
        // rank, like a re-bound array (the model's configuration is the same on all of them)
        const PeerSets now = peer_my_sets(c);
        for (int d = 0; d < 8 && same; ++d) same = now.size[d] == pr.set_sig[d];
    }
    if (!same && (rc = peer_setup(c, local_ok))) return rc;
    *use = pr.ready && local_ok;
    return CSI_OK;
}

// redirect the halo images of a pair table to the neighbours and describe the flag protocol (G: the launch geometry)
int32_t peer_fill_table(csi_context* c, const FusedGeom& G, bool out_is_alt, FusedTable* t) {
    const csi_context::Peer& pr = c->peer;
    const PeerSets ps = peer_wait_counts(c, G);
    for (int d = 0; d < 8; ++d)
        if (ps.size[d] >= csi_context::Peer::SLOTS) return fail(c, CSI_ERR_UNSUPPORTED, "peer halo transport: more edge tiles than flag slots");
    static const int karr[9] = {2, 3, 4, 0, 1, 10, 11, 12, 13};      // kernel order (sigma11, sigma22, sigma12, u, v, alpha, zeta_c, zeta_f, Delta) -> Peer::arr
    for (int k = 0; k < 9; ++k)
        for (int d = 0; d < 8; ++d) {
            const int q = karr[k] < 5 ? karr[k] + (out_is_alt ? 5 : 0) : karr[k];
            t->P[FP_IMG0 + d * 9 + k] = (unsigned long)pr.arr[d][q];
        }
    int mask = 0;
    for (int d = 0; d < 8; ++d) {
        t->P[FP_SLOT_IN + d] = (unsigned long)(pr.slots + (size_t)d * csi_context::Peer::SLOTS);
        t->P[FP_SLOT_OUT + d] = pr.nbr_slots[d] ? (unsigned long)(pr.nbr_slots[d] + (size_t)kPeerOpp[d] * csi_context::Peer::SLOTS) : 0ul;
        t->I[FI_PWAIT + d] = pr.nbr_wait[d];      // (the neighbour's own set: a fold tile's launches have another geometry)
        if (pr.sync_rank[d] >= 0) mask |= 1 << d;
    }
    int any = 0;
    for (int d = 0; d < 8; ++d)
        for (int q = 0; q < 2; ++q) { t->I[FI_PDLD + d * 2 + q] = pr.dld[d][q]; any |= pr.dld[d][q] != 0; }
    t->I[FI_PHASDLD] = any;
    t->I[FI_NYLO] = c->peer.ny_below > 0 ? c->peer.ny_below : c->Ny;
    t->P[FP_PERR] = (unsigned long)pr.err;
    t->I[FI_PEER] = 1; t->I[FI_PMASK] = mask;
    t->I[FI_PTIER] = peer_effective_tier(c);
    t->I[FI_PSET] = ps.nW; t->I[FI_PSET + 1] = ps.nE; t->I[FI_PSET + 2] = ps.nS; t->I[FI_PSET + 3] = ps.nN;
    return CSI_OK;
}


}  // namespace csi_host
