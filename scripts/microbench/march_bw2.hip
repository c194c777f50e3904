// march_bw2.hip -- round 3: does the row-march access pattern of csi::fused::k_pair draw more HBM bandwidth with CDNA4's
// wider per-lane accesses or with loads that land in LDS directly?    Standalone:
//   hipcc --offload-arch=gfx950 -O3 march_bw2.hip -o march_bw2 && ./march_bw2 [N]
// The pattern, with all arithmetic removed: 56-column x `rows`-row tiles, one workgroup of two waves per tile; the producer
// wave reads ten arrays row by row (rows - 4 .. rows + 4 around the tile: the pair kernel's ring rows), hands the row to the
// consumer through an LDS ring, the consumer writes five arrays.  Variants of the producer's loads:
//   b64      global_load_dwordx2, one column per lane (what the kernel does)
//   b128     global_load_dwordx4, TWO columns per lane: a wave covers 128 columns (120 owned), half as many strips
//   lds128   the five static arrays (P, h, aice, u^n, v^n) by global_load_lds_dwordx4 straight into the ring (LDS DMA), two
//            columns per lane; the five state arrays as b128
//   lds32x2  the same arrays by two global_load_lds_dword per 8 bytes is not a CDNA4 form; instead: lds64 is emulated by one
//            global_load_lds_dwordx4 per PAIR of lanes' columns (identical bytes to lds128, listed once)
// Streaming reference: 64 x 4 tiles over the same fifteen arrays.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Tab { const double* r[10]; double* w[5]; };
typedef double double2_t __attribute__((ext_vector_type(2)));

// MODE 0: b64 (1 column per lane); 1: b128 (2 columns per lane); 2: lds128 for arrays 5..9, b128 for 0..4
template <int MODE>
__global__ void __launch_bounds__(128, 3) k_march2(Tab t, int ld, int nx, int ny, int H, int nstrips, int nchunks, int rows, int per_xcd) {
    constexpr int CPL = MODE == 0 ? 1 : 2;                     // columns per lane
    constexpr int OWN = 64 * CPL - 8;                          // owned columns per strip (4 ring columns either side)
    __shared__ double ring[4][10][64 * CPL];
    const int b = blockIdx.x;
    const int w = (b & 7) * per_xcd + (b >> 3);
    if (w >= nstrips * nchunks) return;
    const bool consumer = (threadIdx.x >> 6) != 0;
    const int chunk = w / nstrips, strip = w - chunk * nstrips;
    const int lane = threadIdx.x & 63;
    int i = strip * OWN - 4 + lane * CPL;                       // first 0-based interior column of this lane
    if (i > nx + 4 - CPL) i = nx + 4 - CPL;
    const int ja = chunk * rows, jb = min(ny, ja + rows) - 1;
    const int r0 = ja - 4, r1 = jb + 4;
    size_t off = (size_t)(r0 + H) * ld + (i + H);
    if (!consumer) {
        for (int j = r0; j <= r1; ++j, off += ld) {
            const int slot = (j - r0) & 3;
            if (MODE == 0) {
                double v[10];
#pragma unroll
                for (int a = 0; a < 10; ++a) v[a] = t.r[a][off];
#pragma unroll
                for (int a = 0; a < 10; ++a) ring[slot][a][lane] = v[a];
            } else {
                double2_t v[10];
#pragma unroll
                for (int a = 0; a < (MODE == 2 ? 5 : 10); ++a) v[a] = *(const double2_t*)(t.r[a] + off);
                if (MODE == 2) {
#pragma unroll
                    for (int a = 5; a < 10; ++a)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(t.r[a] + off),
                                                         (__attribute__((address_space(3))) void*)&ring[slot][a][0], 16, 0, 0);
                }
#pragma unroll
                for (int a = 0; a < (MODE == 2 ? 5 : 10); ++a) *(double2_t*)&ring[slot][a][lane * 2] = v[a];
                if (MODE == 2) __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0): the LDS-DMA loads have landed
            }
            __syncthreads();
        }
        __syncthreads();
        return;
    }
    const bool own_lane = lane * CPL >= 4 && lane * CPL < 64 * CPL - 4 && i < nx;
    for (int j = r0; j <= r1; ++j, off += ld) {
        __syncthreads();
        const int slot = (j - r0) & 3;
        if (MODE == 0) {
            double s = 0;
#pragma unroll
            for (int a = 0; a < 10; ++a) s += ring[slot][a][lane];
            if (own_lane && j >= ja && j <= jb) {
#pragma unroll
                for (int a = 0; a < 5; ++a) t.w[a][off] = s + a;
            }
        } else {
            double2_t s = {0, 0};
#pragma unroll
            for (int a = 0; a < 10; ++a) s += *(const double2_t*)&ring[slot][a][lane * 2];
            if (own_lane && j >= ja && j <= jb) {
#pragma unroll
                for (int a = 0; a < 5; ++a) *(double2_t*)(t.w[a] + off) = s + (double)a;
            }
        }
    }
    __syncthreads();
}

template <int NR, int NW>
__global__ void __launch_bounds__(256) k_tile(Tab t, int ld, int nx, int ny, int H, int tiles_x, int per_xcd) {
    const int b = blockIdx.x;
    const int blk = (b & 7) * per_xcd + (b >> 3);
    if ((b >> 3) >= per_xcd) return;
    const int ty = blk / tiles_x, tx = blk - ty * tiles_x;
    const int i = tx * 64 + (threadIdx.x & 63), j = ty * 4 + (threadIdx.x >> 6);
    if (i >= nx || j >= ny) return;
    const size_t off = (size_t)(j + H) * ld + (i + H);
    double s = 0;
#pragma unroll
    for (int a = 0; a < NR; ++a) s += t.r[a][off];
#pragma unroll
    for (int a = 0; a < NW; ++a) t.w[a][off] = s + a;
}

int main(int argc, char** argv) {
    const int nx = argc > 1 ? atoi(argv[1]) : 2048, ny = nx, H = 8, ld = nx + 2 * H;   // ld * 8 bytes: a multiple of 16
    const size_t n = (size_t)ld * (ny + 2 * H + 8);
    Tab t;
    for (int a = 0; a < 10; ++a) { double* p; CK(hipMalloc(&p, n * 8)); CK(hipMemset(p, 0, n * 8)); t.r[a] = p; }
    for (int a = 0; a < 5; ++a) { double* p; CK(hipMalloc(&p, n * 8)); CK(hipMemset(p, 0, n * 8)); t.w[a] = p; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto launch, double bytes, const char* name) {
        for (int k = 0; k < 5; ++k) launch();
        CK(hipEventRecord(e0));
        const int reps = 50;
        for (int k = 0; k < reps; ++k) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("%-64s %8.4f ms  %7.0f GB/s (algorithmic %.0f MB)\n", name, ms, bytes / ms * 1e-6, bytes * 1e-6);
    };
    const double cell = (double)nx * ny * 8;
    char name[160];
    {
        const int tiles_x = (nx + 63) / 64, nb = tiles_x * ((ny + 3) / 4), per = (nb + 7) / 8;
        time([&] { hipLaunchKernelGGL((k_tile<10, 5>), dim3(per * 8), dim3(256), 0, 0, t, ld, nx, ny, H, tiles_x, per); }, cell * 15, "streaming 64x4 tiles, 10r/5w, b64");
    }
    const int targets[] = {768, 1024, 1536, 2048, 3072};
    for (int mode = 0; mode < 3; ++mode)
        for (int target : targets) {
            const int own = mode == 0 ? 56 : 120;
            const int nstrips = (nx + own - 1) / own;
            int max_chunks = target / nstrips; if (max_chunks < 1) max_chunks = 1;
            const int rows = (ny + max_chunks - 1) / max_chunks, nchunks = (ny + rows - 1) / rows;
            const int nb = nstrips * nchunks, per = (nb + 7) / 8;
            snprintf(name, sizeof name, "march2 %-7s strips=%d x chunks=%d (%d tiles, %d rows + 8)", mode == 0 ? "b64" : (mode == 1 ? "b128" : "lds128"),
                     nstrips, nchunks, nb, rows);
            if (mode == 0) time([&] { hipLaunchKernelGGL((k_march2<0>), dim3(per * 8), dim3(128), 0, 0, t, ld, nx, ny, H, nstrips, nchunks, rows, per); }, cell * 15, name);
            if (mode == 1) time([&] { hipLaunchKernelGGL((k_march2<1>), dim3(per * 8), dim3(128), 0, 0, t, ld, nx, ny, H, nstrips, nchunks, rows, per); }, cell * 15, name);
            if (mode == 2) time([&] { hipLaunchKernelGGL((k_march2<2>), dim3(per * 8), dim3(128), 0, 0, t, ld, nx, ny, H, nstrips, nchunks, rows, per); }, cell * 15, name);
        }
    return 0;
}
