// csi_activity.hip -- which tiles of the two-sub-steps launches have anything to do (round 6).
//
// The reference's flagship workload (tripolar / OMIP grids, test/distributed_tests_utils.jl:186-224) is mostly land and ice-free
// ocean.  Where the ice mass m = h rho aice is exactly zero the EVP sub-step is an exact no-op:
//   * sigma += ifelse(m > 0, sigma*, 0)                       Rheologies/elasto_visco_plastic_rheology.jl:343-347
//     (FAST: fma(sigma' - sigma, 0, sigma) = sigma bit for bit unless sigma is -0.0, evp_fast_math.h stress_update_s);
//   * u = ifelse(active_ice, uD, ifelse(marginal_ice, uF, 0)) * active with mi = 0: neither active (minimum_mass > 0, which FAST
//     mode requires) nor marginal -> +0.0 whatever u was    SeaIceDynamics/split_explicit_momentum_equations.jl:217-228, 251-263.
// A tile (56 columns x `rows` rows of one launch, evp_fused2.hip) stores sigma11, sigma22 at its cells, sigma12 at their
// south-west corners, u / v at their west / south faces -- results that depend on m at the cell and its west / south / south-west
// neighbours only.  So a tile whose owned box AND one cell around it hold m == 0, and whose owned stresses hold no -0.0, stores at
// every launch exactly what it stored two launches earlier (same ping-pong buffer) -- provided the launch's other inputs are finite,
// which a run that has not blown up satisfies.  run_fused (csi_launch.hip) therefore runs every tile in the first two launches of
// a sub-cycle (both buffers then hold the fixed point, halo images included) and in the last one (diagnostics), and only the live
// tiles in between.  h and aice do not change inside a sub-cycle (split_explicit_momentum_equations.jl:103-195), so the test
// is made once, before the first launch.
#include "csi_kernels.h"

namespace csi {

namespace {

// rows [ja, jb] of chunk q: the kernels' formula (evp_fused2.hip pair_body; csi_ctx.h chunk_rows)
__device__ __forceinline__ void act_chunk_rows(const ActivityArgs& A, int q, int& ja, int& jb) {
    const int elo = A.elo > 0 ? A.elo : A.rows;
    const bool last_short = A.ehi > 0 && q == A.nchunks - 1 && q > 0;
    ja = last_short ? A.dec.j1 - A.ehi + 1 : A.dec.j0 + (q == 0 ? 0 : elo + (q - 1) * A.rows);
    jb = q == 0 ? min(ja + elo - 1, A.dec.j1) : (q == A.nchunks - 1 ? A.dec.j1 : min(ja + A.rows - 1, A.dec.j1 - A.ehi));
}

// one workgroup (four waves) per tile: flags[w] = 2 live, 1 quiescent once both buffers hold its fixed point (after two launches),
// 0 quiescent from the start: its velocities are +0.0 already and it stores no halo image, so that with the two buffers equal at the
// start of the sub-cycle (run_fused copies them) even the first two launches may leave it out
__global__ void __launch_bounds__(256) k_tile_activity(ActivityArgs A, int* __restrict__ flags) {
    const int w = (int)blockIdx.x;
    const int chunk = w / A.nstrips, strip = w - chunk * A.nstrips;
    int ja, jb;
    act_chunk_rows(A, chunk, ja, jb);
    const int i0 = A.dec.i0 + strip * 56, i1 = min(i0 + 55, A.dec.i1);
    const int lane = (int)(threadIdx.x & 63), ty = (int)(threadIdx.x >> 6);
    if (A.pmask) {
        // a tile of a direction set of the peer transport (evp_fused2.hip, k_pair): it takes part in the flag protocol whatever it holds
        const bool pw = strip < A.pset[0], pe = strip >= A.nstrips - A.pset[1], ps = chunk < A.pset[2], pn = chunk >= A.nchunks - A.pset[3];
        const unsigned pd = ((pw ? 1u : 0u) | (pe ? 2u : 0u) | (ps ? 4u : 0u) | (pn ? 8u : 0u) | ((ps & pw) ? 16u : 0u) | ((ps & pe) ? 32u : 0u) |
                             ((pn & pw) ? 64u : 0u) | ((pn & pe) ? 128u : 0u)) & (unsigned)A.pmask;
        if (pd) { if (threadIdx.x == 0) flags[w] = 2; return; }
    }
    // the ice mass of the owned box and one cell around it (clipped to the parent: the launch reads nothing beyond it either)
    const int i = i0 - 1 + lane;
    const bool col = (i <= i1 + 1) & (i >= A.pc.i0) & (i <= A.pc.i1);
    const int jlo = max(ja - 1, A.pc.j0), jhi = min(jb + 1, A.pc.j1);
    bool live = false;
    if (col)
        for (int j = jlo + ty; j <= jhi; j += 4) {
            const double m = A.h.ld_(i, j) * A.rho * A.a.ld_(i, j);      // ice_mass (src/ClimaSeaIce.jl:42), the kernels' order
            live |= !(m == 0.0);                                         // (NaN counts as ice)
        }
    if (__syncthreads_or(live ? 1 : 0)) {
        if (threadIdx.x == 0) flags[w] = 2;
        return;
    }
    // no ice anywhere near: the owned stresses must hold no -0.0 (fma(x, 0, -0.0) is +0.0 or -0.0 with the sign of x)
    const bool own = (lane >= 1) & (i <= i1);
    if (own) {
        const unsigned long long neg0 = 0x8000000000000000ull;
        const bool in_c = (i >= A.pc.i0) & (i <= A.pc.i1), in_f = (i >= A.pf.i0) & (i <= A.pf.i1);
        for (int j = ja + ty; j <= jb; j += 4) {
            if (in_c & (j >= A.pc.j0) & (j <= A.pc.j1)) {
                live |= (unsigned long long)__double_as_longlong(A.s11.ld_(i, j)) == neg0;
                live |= (unsigned long long)__double_as_longlong(A.s22.ld_(i, j)) == neg0;
            }
            if (in_f & (j >= A.pf.j0) & (j <= A.pf.j1)) live |= (unsigned long long)__double_as_longlong(A.s12.ld_(i, j)) == neg0;
        }
    }
    if (__syncthreads_or(live ? 1 : 0)) {
        if (threadIdx.x == 0) flags[w] = 2;
        return;
    }
    // quiescent.  From the start?  Not a tile within H of a side of the grid (its stores carry halo images: the first two launches write
    // them), and every owned velocity must be +0.0 bit for bit already (the first sub-step would zero it; its neighbours read the old value)
    bool later = (i0 <= A.Hx) | (i1 > A.Nx - A.Hx) | (ja <= A.Hy) | (jb > A.Ny - A.Hy);
    if (!later && own) {
        const bool in_u = (i >= A.pu.i0) & (i <= A.pu.i1), in_v = (i >= A.pv.i0) & (i <= A.pv.i1);
        for (int j = ja + ty; j <= jb; j += 4) {
            if (in_u & (j >= A.pu.j0) & (j <= A.pu.j1)) later |= __double_as_longlong(A.u.ld_(i, j)) != 0ll;
            if (in_v & (j >= A.pv.j0) & (j <= A.pv.j1)) later |= __double_as_longlong(A.v.ld_(i, j)) != 0ll;
        }
    }
    const int any = __syncthreads_or(later ? 1 : 0);
    if (threadIdx.x == 0) flags[w] = any ? 1 : 0;
}

// act = {live, tiles, the live tiles' numbers in ascending order}: one workgroup, a ballot scan per 1024 tiles
// sample (may be null): device-visible pinned host words {seqlock, live, tiles, id}: the host reads them without any API call
__global__ void __launch_bounds__(1024) k_activity_compact(const int* __restrict__ flags, int tiles, int level, int* __restrict__ act,
                                                           int* sample, int sample_id) {
    __shared__ int wave_count[16];
    __shared__ int base;
    const int lane = (int)(threadIdx.x & 63), wv = (int)(threadIdx.x >> 6);
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (int t0 = 0; t0 < tiles; t0 += 1024) {
        const int t = t0 + (int)threadIdx.x;
        const bool on = t < tiles && flags[t] >= level;
        const unsigned long long bal = __ballot(on);
        if (lane == 0) wave_count[wv] = __popcll(bal);
        __syncthreads();
        int off = base;
        for (int q = 0; q < wv; ++q) off += wave_count[q];
        if (on) act[2 + off + __popcll(bal & ((1ull << lane) - 1ull))] = t;
        __syncthreads();
        if (threadIdx.x == 0) { int s = 0; for (int q = 0; q < 16; ++q) s += wave_count[q]; base += s; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        act[0] = base; act[1] = tiles;
        if (sample) {
            // seqlock: odd while the words change; system scope (the words live in host memory)
            const int s0 = __hip_atomic_load(sample, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(sample, s0 | 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __atomic_thread_fence(__ATOMIC_RELEASE);
            __hip_atomic_store(sample + 1, base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(sample + 2, tiles, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(sample + 3, sample_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __atomic_thread_fence(__ATOMIC_RELEASE);
            __hip_atomic_store(sample, (s0 | 1) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

}  // namespace

void launch_tile_activity(const ActivityArgs& A, int* flags, int* act, int* act0, int* sample, int sample_id, hipStream_t s) {
    const int tiles = A.nstrips * A.nchunks;
    hipLaunchKernelGGL(k_tile_activity, dim3((unsigned)tiles), dim3(256), 0, s, A, flags);
    hipLaunchKernelGGL(k_activity_compact, dim3(1), dim3(1024), 0, s, (const int*)flags, tiles, 2, act, sample, sample_id);
    if (act0) hipLaunchKernelGGL(k_activity_compact, dim3(1), dim3(1024), 0, s, (const int*)flags, tiles, 1, act0, (int*)nullptr, 0);
}

}  // namespace csi
