// csi_hostgroup.hip -- see csi_hostgroup.h.  Everything shared between the processes lives in one POSIX shared-memory segment:
// a two-phase arrival counter for the collectives, one payload area per rank, one single-producer / single-consumer message
// queue per ordered pair of ranks, and per rank the IPC handle of its current send buffer.  Lock-free (address-free atomics of
// `long` / `int`), every wait bounded by a time-out: a rank that never arrives makes the others fail, not hang.
#include "csi_hostgroup.h"

#include <errno.h>
#include <fcntl.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstring>

namespace csi {

namespace {

constexpr int kMaxWorld = 16, kQueueLen = 64, kPayload = 16384;
constexpr double kTimeoutSeconds = 120.0;

struct Msg { long off, count, gen; };
struct Queue {
    std::atomic<long> head, tail;      // consumed / posted so far
    Msg e[kQueueLen];
};
struct RankArea {
    std::atomic<long> posted, consumed;
    std::atomic<long> buf_gen;         // bumped after `handle` has been written
    hipIpcMemHandle_t handle;
    long buf_bytes;
    long payload_size;
    uint8_t payload[kPayload];
};
struct Shared {
    std::atomic<int> state;            // 0 fresh (zero-filled by the kernel), 2 ready
    int world;
    long creator_pid;                  // rank 0's pid: a segment whose creator is gone is a stale one (hostgroup_join)
    std::atomic<int> joined;           // ranks that have accepted this segment
    std::atomic<long> arrived, generation;
    RankArea rank[kMaxWorld];
    Queue box[kMaxWorld * kMaxWorld];  // [src * world + dst]
};

double now_seconds() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

template <class Pred>
bool spin_until(Pred done) {
    const double t0 = now_seconds();
    for (long it = 0; !done(); ++it) {
        if ((it & 1023) == 1023 && now_seconds() - t0 > kTimeoutSeconds) return false;
        if (it > 2000) usleep(50);
    }
    return true;
}

}  // namespace

struct HostGroup {
    Shared* sh = nullptr;
    int world = 0, rank = 0;
    // the neighbours' send buffers as this process addresses them: one mapping per rank, reopened when its generation changes
    void* mapped[kMaxWorld] = {};
    long mapped_gen[kMaxWorld] = {};
};

static bool fail(std::string* err, const std::string& msg) {
    if (err) *err = "host-channel tile group: " + msg;
    return false;
}

// one phase of a collective: everybody arrives, the last one in opens the next generation
static bool phase(HostGroup* g) {
    Shared* sh = g->sh;
    const long gen = sh->generation.load(std::memory_order_acquire);
    if (sh->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == g->world) {
        sh->arrived.store(0, std::memory_order_relaxed);
        sh->generation.fetch_add(1, std::memory_order_acq_rel);
        return true;
    }
    return spin_until([&] { return sh->generation.load(std::memory_order_acquire) != gen; });
}

// Joining.  Rank 0 OWNS the name: it removes whatever an earlier run may have left under it (a run that crashed or timed out
// inside its join never unlinked its segment, whose arrival counters, generation and payloads would release this run's barriers
// early or never and hand out garbage IPC handles -- ADVICE round 4), creates the segment with O_EXCL and stamps it with its pid.
// The other ranks never create: they open what exists and accept it only if it is complete (size), initialised (state 2), for this
// world size and stamped by a LIVING process -- a stale segment's creator is gone -- and otherwise drop it and look again, until
// the time-out.  The name is unlinked as soon as everybody has it mapped, and also on every failure path of rank 0.
static bool pid_alive(long pid) { return pid > 0 && (kill((pid_t)pid, 0) == 0 || errno == EPERM); }

HostGroup* hostgroup_join(const char* shm_name, int world, int rank, std::string* err) {
    if (!shm_name || world < 1 || world > kMaxWorld || rank < 0 || rank >= world) { fail(err, "bad name / world size / rank (at most 16 ranks)"); return nullptr; }
    Shared* sh = nullptr;
    if (rank == 0) {
        (void)shm_unlink(shm_name);                      // a stale segment of an earlier run, if any
        const int fd = shm_open(shm_name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0) { fail(err, std::string("shm_open(") + shm_name + ", O_CREAT | O_EXCL) failed: another job is using this name"); return nullptr; }
        if (ftruncate(fd, (off_t)sizeof(Shared)) != 0) { close(fd); shm_unlink(shm_name); fail(err, "ftruncate failed"); return nullptr; }
        void* p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) { shm_unlink(shm_name); fail(err, "mmap failed"); return nullptr; }
        sh = (Shared*)p;                                 // (a fresh segment is zero-filled by the kernel)
        sh->world = world;
        sh->creator_pid = (long)getpid();
        sh->state.store(2, std::memory_order_release);
    } else {
        const double t0 = now_seconds();
        for (long it = 0; !sh; ++it) {
            if (now_seconds() - t0 > kTimeoutSeconds) { fail(err, "rank 0 never created the segment (or only a stale one of an earlier run was found)"); return nullptr; }
            if (it) usleep(it < 200 ? 500 : 5000);
            const int fd = shm_open(shm_name, O_RDWR, 0600);
            if (fd < 0) continue;                        // not there yet
            struct stat st;
            if (fstat(fd, &st) != 0 || st.st_size != (off_t)sizeof(Shared)) { close(fd); continue; }      // rank 0 is between shm_open and ftruncate
            void* p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            close(fd);
            if (p == MAP_FAILED) continue;
            Shared* cand = (Shared*)p;
            const bool good = cand->state.load(std::memory_order_acquire) == 2 && cand->world == world && pid_alive(cand->creator_pid) &&
                              cand->joined.load(std::memory_order_acquire) < world;
            if (!good) { munmap(p, sizeof(Shared)); continue; }      // being initialised, or stale: look again (rank 0 replaces a stale one)
            sh = cand;
        }
    }
    sh->joined.fetch_add(1, std::memory_order_acq_rel);
    HostGroup* g = new HostGroup;
    g->sh = sh; g->world = world; g->rank = rank;
    // everybody has the segment mapped: the name can go (no leak if a rank dies later)
    if (!phase(g)) {
        fail(err, "a rank did not join within two minutes");
        if (rank == 0) shm_unlink(shm_name);
        munmap(sh, sizeof(Shared)); delete g;
        return nullptr;
    }
    if (rank == 0) shm_unlink(shm_name);
    return g;
}

void hostgroup_leave(HostGroup* g) {
    if (!g) return;
    for (int r = 0; r < kMaxWorld; ++r) if (g->mapped[r]) (void)hipIpcCloseMemHandle(g->mapped[r]);
    if (g->sh) munmap(g->sh, sizeof(Shared));
    delete g;
}

int hostgroup_world(const HostGroup* g) { return g ? g->world : 0; }

bool hostgroup_allgather(HostGroup* g, const void* mine, size_t nb, std::vector<uint8_t>& out, std::string* err) {
    if (nb > (size_t)kPayload) return fail(err, "collective payload too large");
    RankArea& me = g->sh->rank[g->rank];
    memcpy(me.payload, mine, nb);
    me.payload_size = (long)nb;
    if (!phase(g)) return fail(err, "a collective timed out (a rank did not arrive)");
    out.resize(nb * (size_t)g->world);
    for (int r = 0; r < g->world; ++r) {
        if (g->sh->rank[r].payload_size != (long)nb) return fail(err, "payload sizes differ between the ranks");
        memcpy(out.data() + (size_t)r * nb, g->sh->rank[r].payload, nb);
    }
    // second phase: nobody overwrites its payload before all have read
    if (!phase(g)) return fail(err, "a collective timed out (a rank did not arrive)");
    return true;
}

bool hostgroup_wait_consumed(HostGroup* g, std::string* err) {
    RankArea& me = g->sh->rank[g->rank];
    if (!spin_until([&] { return me.consumed.load(std::memory_order_acquire) == me.posted.load(std::memory_order_relaxed); }))
        return fail(err, "a neighbour never received this rank's previous halo message");
    return true;
}

bool hostgroup_set_sendbuf(HostGroup* g, void* dev_ptr, size_t bytes, std::string* err) {
    RankArea& me = g->sh->rank[g->rank];
    hipIpcMemHandle_t h;
    if (hipIpcGetMemHandle(&h, dev_ptr) != hipSuccess) { (void)hipGetLastError(); return fail(err, "hipIpcGetMemHandle of the send buffer failed"); }
    me.handle = h;
    me.buf_bytes = (long)bytes;
    me.buf_gen.fetch_add(1, std::memory_order_release);
    return true;
}

bool hostgroup_sendrecv(HostGroup* g, hipStream_t stream, double* recvbuf, const long* soff, const long* scnt, const int* speer,
                        const long* roff, const long* rcnt, const int* rpeer, std::string* err) {
    Shared* sh = g->sh;
    if (hipStreamSynchronize(stream) != hipSuccess) return fail(err, "hipStreamSynchronize failed (pack)");      // the pack kernel has filled the send buffer
    RankArea& me = sh->rank[g->rank];
    const long my_gen = me.buf_gen.load(std::memory_order_relaxed);
    for (int k = 0; k < 8; ++k)
        if (speer[k] >= 0 && scnt[k] > 0) {
            Queue& q = sh->box[(size_t)g->rank * g->world + speer[k]];
            const long t = q.tail.load(std::memory_order_relaxed);
            if (!spin_until([&] { return t - q.head.load(std::memory_order_acquire) < kQueueLen; })) return fail(err, "a message queue stayed full");
            q.e[t % kQueueLen] = Msg{soff[k], scnt[k], my_gen};
            q.tail.store(t + 1, std::memory_order_release);
            me.posted.fetch_add(1, std::memory_order_relaxed);
        }
    int from[8], nfrom = 0;
    for (int k = 0; k < 8; ++k)
        if (rpeer[k] >= 0 && rcnt[k] > 0) {
            const int src = rpeer[k];
            Queue& q = sh->box[(size_t)src * g->world + g->rank];
            const long h = q.head.load(std::memory_order_relaxed);
            if (!spin_until([&] { return q.tail.load(std::memory_order_acquire) > h; }))
                return fail(err, "a halo message never arrived (a rank fell behind or died)");
            const Msg m = q.e[h % kQueueLen];
            q.head.store(h + 1, std::memory_order_release);
            if (m.count != rcnt[k]) return fail(err, "halo message of unexpected size (send / receive plans do not match)");
            const double* base = nullptr;
            if (src == g->rank) {
                return fail(err, "a rank cannot be its own neighbour in a host-channel group (use one tile connected to itself over RCCL)");
            }
            if (!g->mapped[src] || g->mapped_gen[src] != m.gen) {
                if (g->mapped[src]) { (void)hipIpcCloseMemHandle(g->mapped[src]); g->mapped[src] = nullptr; }
                if (sh->rank[src].buf_gen.load(std::memory_order_acquire) != m.gen) return fail(err, "a sender replaced its buffer while a message was in flight");
                void* mp = nullptr;
                if (hipIpcOpenMemHandle(&mp, sh->rank[src].handle, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
                    (void)hipGetLastError();
                    return fail(err, "hipIpcOpenMemHandle of a neighbour's send buffer failed");
                }
                g->mapped[src] = mp; g->mapped_gen[src] = m.gen;
            }
            base = (const double*)g->mapped[src];
            if (hipMemcpyAsync(recvbuf + roff[k], base + m.off, (size_t)m.count * sizeof(double), hipMemcpyDeviceToDevice, stream) != hipSuccess)
                return fail(err, "hipMemcpyAsync from a neighbour's send buffer failed");
            from[nfrom++] = src;
        }
    if (hipStreamSynchronize(stream) != hipSuccess) return fail(err, "hipStreamSynchronize failed (copies)");      // the copies are done: the senders may repack
    for (int q = 0; q < nfrom; ++q) sh->rank[from[q]].consumed.fetch_add(1, std::memory_order_release);
    return true;
}

}  // namespace csi
