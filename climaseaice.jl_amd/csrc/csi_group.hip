// csi_group.hip -- halo exchange of tiles: RCCL send / receive, the in-process tile group, the host-channel group  (split out of csi_abi.hip in round 4; see csi_ctx.h)
#include "csi_ctx.h"

namespace csi_host {

// Exchange `W` halo layers of the given fields with the neighbouring tiles (no-op on an untiled grid).
int32_t exchange_refs(csi_context* c, const FRef* fr, int nf, int W);

int32_t exchange(csi_context* c, const int* fids, int nf, int W) {
    if (!is_tiled(c)) return CSI_OK;
    if (nf > MAX_EX_FIELDS) return fail(c, CSI_ERR_INVALID_ARGUMENT, "too many fields in one exchange");
    FRef fr[MAX_EX_FIELDS];
    for (int k = 0; k < nf; ++k) {
        if (!c->f[fids[k]].p) return fail(c, CSI_ERR_NOT_BOUND, std::string("field not bound: ") + kName[fids[k]]);
        fr[k] = ref_of(c, fids[k]);
    }
    return exchange_refs(c, fr, nf, W);
}

bool has_comm(const csi_context* c) { return c->comm != nullptr || c->local != nullptr || c->hostg != nullptr; }

constexpr int kLocalTimeoutSeconds = 120;
// all ranks of the group: rank r's `nb` bytes end up in out[r * nb ...] everywhere
int32_t local_allgather(csi_context* c, const void* mine, size_t nb, std::vector<uint8_t>& out) {
    csi_local_group* G = c->local;
    std::unique_lock<std::mutex> lk(G->mu);
    const long gen = G->generation;
    G->payload[c->rank].assign((const uint8_t*)mine, (const uint8_t*)mine + nb);
    if (++G->arrived == G->world) {
        // the last one in publishes: the payloads stay untouched until everybody of the NEXT collective has arrived
        G->arrived = 0; ++G->generation;
        G->cv.notify_all();
    } else if (!G->cv.wait_for(lk, std::chrono::seconds(kLocalTimeoutSeconds), [&] { return G->generation != gen; })) {
        return fail(c, CSI_ERR_COMM, "in-process tile group: a collective timed out (a rank did not arrive)");
    }
    out.resize(nb * (size_t)G->world);
    for (int r = 0; r < G->world; ++r) {
        if (G->payload[r].size() != nb) return fail(c, CSI_ERR_COMM, "in-process tile group: payload sizes differ");
        memcpy(out.data() + (size_t)r * nb, G->payload[r].data(), nb);
    }
    // second phase: nobody overwrites its payload before all have read
    const long gen2 = G->generation;
    if (++G->arrived == G->world) { G->arrived = 0; ++G->generation; G->cv.notify_all(); }
    else if (!G->cv.wait_for(lk, std::chrono::seconds(kLocalTimeoutSeconds), [&] { return G->generation != gen2; }))
        return fail(c, CSI_ERR_COMM, "in-process tile group: a collective timed out (a rank did not arrive)");
    return CSI_OK;
}
int32_t local_allreduce_min(csi_context* c, int* v) {
    std::vector<uint8_t> all;
    int32_t rc;
    if ((rc = local_allgather(c, v, sizeof(int), all))) return rc;
    for (int r = 0; r < c->world; ++r) { int x; memcpy(&x, all.data() + (size_t)r * sizeof(int), sizeof(int)); if (x < *v) *v = x; }
    return CSI_OK;
}
// max of one int over all ranks of whatever joins them: the in-process group, the host-channel group or the RCCL communicator
// (one rank: nothing to do).  Host-synchronous on the RCCL path (a 4-byte all-reduce on the context's stream).
int32_t comm_allreduce_max(csi_context* c, int* v) {
    if (c->world <= 1 || !has_comm(c)) return CSI_OK;
    std::vector<uint8_t> all;
    if (c->local) {
        int32_t rc;
        if ((rc = local_allgather(c, v, sizeof(int), all))) return rc;
    } else if (c->hostg) {
        if (!hostgroup_allgather(c->hostg, v, sizeof(int), all, &c->err)) return CSI_ERR_COMM;
    } else {
        int* flag = nullptr;
        HIP_TRY(c, hipMalloc((void**)&flag, sizeof(int)));
        hipError_t e = hipMemcpy(flag, v, sizeof(int), hipMemcpyHostToDevice);
        ncclResult_t r = ncclSuccess;
        if (e == hipSuccess) r = ncclAllReduce(flag, flag, 1, ncclInt32, ncclMax, c->comm, c->stream);
        if (e == hipSuccess && r == ncclSuccess) e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess && r == ncclSuccess) e = hipMemcpy(v, flag, sizeof(int), hipMemcpyDeviceToHost);
        hipFree(flag);
        if (r != ncclSuccess) return fail(c, CSI_ERR_COMM, std::string("ncclAllReduce: ") + ncclGetErrorString(r));
        if (e != hipSuccess) return fail(c, CSI_ERR_HIP, std::string("all-reduce of a status word: ") + hipGetErrorString(e));
        return CSI_OK;
    }
    for (int r = 0; r < c->world; ++r) { int x; memcpy(&x, all.data() + (size_t)r * sizeof(int), sizeof(int)); if (x > *v) *v = x; }
    return CSI_OK;
}
// before the send buffer is packed again: every message posted from it has been copied out
int32_t local_wait_consumed(csi_context* c) {
    csi_local_group* G = c->local;
    std::unique_lock<std::mutex> lk(G->mu);
    if (!G->cv.wait_for(lk, std::chrono::seconds(kLocalTimeoutSeconds), [&] { return G->consumed[c->rank] == G->posted[c->rank]; }))
        return fail(c, CSI_ERR_COMM, "in-process tile group: a neighbour never received this rank's previous halo message");
    return CSI_OK;
}
// the grouped send / receive of exchange_refs
int32_t local_sendrecv(csi_context* c, const long* soff, const long* scnt, const int* speer, const long* roff, const long* rcnt, const int* rpeer) {
    csi_local_group* G = c->local;
    HIP_TRY(c, hipStreamSynchronize(c->stream));                         // the pack kernel has filled the send buffer
    int from[8], nfrom = 0;
    {
        std::unique_lock<std::mutex> lk(G->mu);
        for (int k = 0; k < 8; ++k)
            if (speer[k] >= 0 && scnt[k] > 0) {
                G->box[(size_t)c->rank * G->world + speer[k]].push_back(csi_local_group::Msg{c->sendbuf + soff[k], (size_t)scnt[k]});
                ++G->posted[c->rank];
            }
        G->cv.notify_all();
        for (int k = 0; k < 8; ++k)
            if (rpeer[k] >= 0 && rcnt[k] > 0) {
                std::deque<csi_local_group::Msg>& q = G->box[(size_t)rpeer[k] * G->world + c->rank];
                if (!G->cv.wait_for(lk, std::chrono::seconds(kLocalTimeoutSeconds), [&] { return !q.empty(); }))
                    return fail(c, CSI_ERR_COMM, "in-process tile group: a halo message never arrived (a rank fell behind or died)");
                const csi_local_group::Msg m = q.front();
                q.pop_front();
                if (m.count != (size_t)rcnt[k]) return fail(c, CSI_ERR_COMM, "in-process tile group: halo message of unexpected size (send / receive plans do not match)");
                HIP_TRY(c, hipMemcpyAsync(c->recvbuf + roff[k], m.ptr, m.count * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
                from[nfrom++] = rpeer[k];
            }
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));                         // the copies are done: the senders may repack
    {
        std::unique_lock<std::mutex> lk(G->mu);
        for (int q = 0; q < nfrom; ++q) ++G->consumed[from[q]];
        G->cv.notify_all();
    }
    return CSI_OK;
}

// the same on explicit array references (the fused path exchanges whichever ping-pong buffer is current): pack, one grouped
// send / receive, unpack, all on the context stream
int32_t exchange_refs(csi_context* c, const FRef* fr, int nf, int W) {
    if (!is_tiled(c)) return CSI_OK;
    if (!c->tile.set) return fail(c, CSI_ERR_NOT_BOUND, "connected topology but csi_tile_set has not been called");
    if (!has_comm(c)) return fail(c, CSI_ERR_NOT_BOUND, "connected topology but csi_comm_init has not been called");
    if (nf > MAX_EX_FIELDS) return fail(c, CSI_ERR_INVALID_ARGUMENT, "too many fields in one exchange");
    if (W < 1 || W > c->Hx || W > c->Hy || W > c->Nx || W > c->Ny) return fail(c, CSI_ERR_INVALID_ARGUMENT, "exchange width out of range");
    ExPlan sp;
    long soff[8], scnt[8], roff[8], rcnt[8];
    int speer[8], rpeer[8];
    build_plan(c->g, c->tile, fr, nf, W, 0, sp, soff, scnt, speer);
    build_plan(c->g, c->tile, fr, nf, W, 1, c->pending_rp, roff, rcnt, rpeer);
    const size_t need_elems = (size_t)(sp.total > c->pending_rp.total ? sp.total : c->pending_rp.total);
    int32_t lrc;
    if (c->local && (lrc = local_wait_consumed(c))) return lrc;      // (before the send buffer is repacked -- or freed)
    if (c->hostg && !hostgroup_wait_consumed(c->hostg, &c->err)) return CSI_ERR_COMM;
    if (need_elems > c->buf_cap) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (c->sendbuf) hipFree(c->sendbuf);
        if (c->recvbuf) hipFree(c->recvbuf);
        c->sendbuf = c->recvbuf = nullptr;
        const size_t cap = need_elems * 2;
        HIP_TRY(c, hipMalloc((void**)&c->sendbuf, cap * sizeof(double)));
        HIP_TRY(c, hipMalloc((void**)&c->recvbuf, cap * sizeof(double)));
        c->buf_cap = cap;
        if (c->hostg && !hostgroup_set_sendbuf(c->hostg, c->sendbuf, cap * sizeof(double), &c->err)) return CSI_ERR_COMM;
    }
    launch_pack(sp, c->sendbuf, 0, c->stream);
    if (c->local) {
        if ((lrc = local_sendrecv(c, soff, scnt, speer, roff, rcnt, rpeer))) return lrc;
        launch_pack(c->pending_rp, c->recvbuf, 1, c->stream);
        HIP_TRY(c, hipGetLastError());
        return CSI_OK;
    }
    if (c->hostg) {
        if (!hostgroup_sendrecv(c->hostg, c->stream, c->recvbuf, soff, scnt, speer, roff, rcnt, rpeer, &c->err)) return CSI_ERR_COMM;
        launch_pack(c->pending_rp, c->recvbuf, 1, c->stream);
        HIP_TRY(c, hipGetLastError());
        return CSI_OK;
    }
    NCCL_TRY(c, ncclGroupStart());
    for (int k = 0; k < 8; ++k)
        if (speer[k] >= 0 && scnt[k] > 0) NCCL_TRY(c, ncclSend(c->sendbuf + soff[k], (size_t)scnt[k], ncclDouble, speer[k], c->comm, c->stream));
    for (int k = 0; k < 8; ++k)
        if (rpeer[k] >= 0 && rcnt[k] > 0) NCCL_TRY(c, ncclRecv(c->recvbuf + roff[k], (size_t)rcnt[k], ncclDouble, rpeer[k], c->comm, c->stream));
    NCCL_TRY(c, ncclGroupEnd());
    launch_pack(c->pending_rp, c->recvbuf, 1, c->stream);
    HIP_TRY(c, hipGetLastError());
    return CSI_OK;
}


}  // namespace csi_host
