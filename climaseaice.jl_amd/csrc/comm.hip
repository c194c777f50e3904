// comm.hip -- multi-GPU halo exchange: one process per GPU, RCCL point-to-point over xGMI.
//
// Replaces the MPI halo passes hidden in the reference's fill_halo_regions! on distributed grids
// (sea_ice_model.jl:381-384, elasto_visco_plastic_rheology.jl:275-280,
// sea_ice_external_stress.jl:72-78) and -- unlike the reference, which widens the halo to
// 2*substeps+3 and never communicates inside the sub-cycle
// (split_explicit_momentum_equations.jl:51-64) -- exchanges u, v (width 2) once per sub-step
// (SURVEY.md A.5, 8e).  No collective: every message is neighbour-to-neighbour.
//
// One pack kernel gathers all strips of all fields for the (up to 8) neighbours into one
// contiguous send buffer, one grouped ncclSend/ncclRecv moves them, one unpack kernel scatters
// the received strips into the halos.
#include "csi_comm.h"

#include <cstdio>

namespace csi {

__global__ void __launch_bounds__(256) k_pack(ExPlan pl, double* buf, int unpack) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= pl.total) return;
    // the segment that holds element t: offsets are ascending, at most 8 directions x 5 fields -> binary search
    int s = 0, hi = pl.nseg - 1;
#pragma unroll 1
    while (s < hi) {
        const int mid = (s + hi + 1) >> 1;
        if (t >= pl.seg[mid].off) s = mid; else hi = mid - 1;
    }
    const ExSeg& g = pl.seg[s];
    const long r = t - g.off;
    const int jj = (int)(r / g.ni), ii = (int)(r - (long)jj * g.ni);
    double* fp = g.f.p + (g.i0 + ii) + (long)(g.j0 + jj) * g.f.ld;
    if (unpack) *fp = buf[t];
    else buf[t] = *fp;
}

void launch_pack(const ExPlan& pl, double* buf, int unpack, hipStream_t s) {
    if (pl.total == 0) return;
    hipLaunchKernelGGL(k_pack, dim3((unsigned)((pl.total + 255) / 256)), dim3(256), 0, s, pl, buf, unpack);
}

struct WaitSpec { int n[8]; };
__global__ void __launch_bounds__(64) k_wait_peers(const unsigned long long* slots, WaitSpec w, int slots_per_dir, unsigned long long seq, unsigned* err) {
    const int lane = (int)threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    for (int d = 0; d < 8; ++d) {
        const int n = w.n[d];
        const unsigned long long* q = slots + (size_t)d * slots_per_dir;
        for (;;) {
            bool behind = false;
            for (int b0 = 0; b0 < n; b0 += 64) {
                const int idx = b0 + lane;
                if (idx < n) behind |= __hip_atomic_load(q + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < seq;
            }
            if (__builtin_amdgcn_ballot_w64(behind) == 0) break;
            // (the abort word of the block: a neighbour's wait has given up, evp_fused2.hip)
            const bool poison = lane == 63 && __hip_atomic_load(q + (slots_per_dir - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0ull;
            if (__builtin_amdgcn_ballot_w64(poison | (*(volatile unsigned*)err != 0u)) != 0 || wall_clock64() - t0 > 300000000ull) { if (lane == 0) *err = 1u; return; }
            __builtin_amdgcn_s_sleep(16);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}

void launch_wait_peers(const unsigned long long* slots, const int* sync_rank, int slots_per_dir, const int* n, unsigned long long seq,
                       unsigned* err, hipStream_t s) {
    WaitSpec w;
    bool any = false;
    for (int d = 0; d < 8; ++d) { w.n[d] = sync_rank[d] >= 0 ? n[d] : 0; any |= w.n[d] > 0; }
    if (any) hipLaunchKernelGGL(k_wait_peers, dim3(1), dim3(64), 0, s, slots, w, slots_per_dir, seq, err);
}

// Neighbour in direction (dx, dy) of tile (rx, ry); -1 if none.
int tile_neighbor(const TileInfo& t, int dx, int dy, int xlo, int xhi, int ylo, int yhi) {
    if (dx == 0 && dy == 0) return -1;
    if (dx < 0 && xlo != SIDE_CONNECTED) return -1;
    if (dx > 0 && xhi != SIDE_CONNECTED) return -1;
    if (dy < 0 && ylo != SIDE_CONNECTED) return -1;
    if (dy > 0 && yhi != SIDE_CONNECTED) return -1;
    int nx = t.rx + dx, ny = t.ry + dy;
    if (nx < 0 || nx >= t.Rx) { if (!t.periodic_x) return -1; nx = (nx + t.Rx) % t.Rx; }
    if (ny < 0 || ny >= t.Ry) { if (!t.periodic_y) return -1; ny = (ny + t.Ry) % t.Ry; }
    return ny * t.Rx + nx;
}

// Build the send (halo = 0: owned strips) or receive (halo = 1: halo strips) plan for `nf` fields.
// Direction order is fixed (dy outer, dx inner); the receive plan lists, in the same order e, the
// strip that arrives from the neighbour in direction -e (that neighbour's send number e), which
// keeps multiple messages between the same pair of ranks in matching FIFO order.
void build_plan(const GridDev& g, const TileInfo& t, const FRef* fields, int nf, int W, int halo,
                ExPlan& pl, long* dir_off, long* dir_cnt, int* dir_peer) {
    pl.nseg = 0;
    pl.total = 0;
    int k = 0;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            if (dx == 0 && dy == 0) continue;
            // send: towards (dx, dy).  recv (same loop index): from the neighbour at (-dx, -dy)
            const int ndx = halo ? -dx : dx, ndy = halo ? -dy : dy;
            const int peer = tile_neighbor(t, ndx, ndy, g.xlo, g.xhi, g.ylo, g.yhi);
            dir_peer[k] = peer;
            dir_off[k] = pl.total;
            dir_cnt[k] = 0;
            if (peer >= 0) {
                int i0, ni, j0, nj;
                // the strip lies on my side `ndx`: owned cells next to that edge (send) or the halo beyond it (recv)
                if (ndx < 0) { i0 = halo ? 1 - W : 1; ni = W; }
                else if (ndx > 0) { i0 = halo ? g.Nx + 1 : g.Nx - W + 1; ni = W; }
                else {
                    const int a = (g.xlo == SIDE_CONNECTED) ? 1 : 1 - W, b = (g.xhi == SIDE_CONNECTED) ? g.Nx : g.Nx + W;
                    i0 = a; ni = b - a + 1;
                }
                if (ndy < 0) { j0 = halo ? 1 - W : 1; nj = W; }
                else if (ndy > 0) { j0 = halo ? g.Ny + 1 : g.Ny - W + 1; nj = W; }
                else {
                    const int a = (g.ylo == SIDE_CONNECTED) ? 1 : 1 - W, b = (g.yhi == SIDE_CONNECTED) ? g.Ny : g.Ny + W;
                    j0 = a; nj = b - a + 1;
                }
                for (int f = 0; f < nf; ++f) {
                    ExSeg& s = pl.seg[pl.nseg++];
                    s.f = fields[f]; s.i0 = i0; s.j0 = j0; s.ni = ni; s.nj = nj; s.off = pl.total;
                    pl.total += (long)ni * nj;
                }
                dir_cnt[k] = pl.total - dir_off[k];
            }
            ++k;
        }
}

}  // namespace csi
