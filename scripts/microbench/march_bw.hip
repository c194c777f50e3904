// march_bw.hip -- how much HBM bandwidth do the access patterns of the EVP kernels reach, with the
// arithmetic removed?  Standalone (hipcc --offload-arch=gfx950 -O3 march_bw.hip -o march_bw).
//   march : each wave owns a 60-column strip (64 lanes, 2-column overlap either side) and marches down
//           `rows` rows (+ `ring` extra rows each side), reading NR arrays and writing NW per row --
//           the pattern of csi::fused::k_substep
//   tile  : 64x4 tiles, one row per wave -- the pattern of csi::fast::k_stress
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Tab { const double* r[12]; double* w[8]; };

template <int NR, int NW, int PF>
__global__ void __launch_bounds__(256) k_march(Tab t, int ld, int nx, int ny, int H, int nstrips, int nchunks, int rows, int ring,
                                               int per_xcd, int own) {
    const int b = blockIdx.x;
    const int blk = (b & 7) * per_xcd + (b >> 3);
    if ((b >> 3) >= per_xcd) return;
    const int w = __builtin_amdgcn_readfirstlane(blk * 4 + (int)(threadIdx.x >> 6));
    if (w >= nstrips * nchunks) return;
    const int chunk = w / nstrips, strip = w - chunk * nstrips;
    const int lane = threadIdx.x & 63;
    const int half = (64 - own) / 2;
    int i = strip * own - half + lane;              // 0-based interior column
    if (i > nx + half - 1) i = nx + half - 1;
    const int j0 = chunk * rows - ring, j1 = min(ny, (chunk + 1) * rows) + ring;
    const bool owner = lane >= half && lane < 64 - half && i < nx;
    size_t off = (size_t)(j0 + H) * ld + (i + H);
    double cur[NR], nxt[NR];
#pragma unroll
    for (int a = 0; a < NR; ++a) cur[a] = t.r[a][off];
    for (int j = j0; j < j1; ++j) {
        if (PF) {
            const size_t offn = off + (j + 1 < j1 ? ld : 0);
#pragma unroll
            for (int a = 0; a < NR; ++a) nxt[a] = t.r[a][offn];
        }
        double s = 0;
#pragma unroll
        for (int a = 0; a < NR; ++a) s += cur[a];
        s += __shfl_up(s, 1) + __shfl_down(s, 1);
        if (owner && j >= chunk * rows && j < min(ny, (chunk + 1) * rows)) {
#pragma unroll
            for (int a = 0; a < NW; ++a) t.w[a][off] = s + a;
        }
        off += ld;
        if (PF) {
#pragma unroll
            for (int a = 0; a < NR; ++a) cur[a] = nxt[a];
        } else if (j + 1 < j1) {
#pragma unroll
            for (int a = 0; a < NR; ++a) cur[a] = t.r[a][off];
        }
    }
}

struct S5 { double a, b, c, d, e; };
// AoS variant: state (5 doubles) + constants (5 doubles) read, state (5 doubles) written
template <int PF>
__global__ void __launch_bounds__(256) k_march_aos(const S5* __restrict__ in, const S5* __restrict__ cst, S5* __restrict__ out,
                                                   int ld, int nx, int ny, int H, int nstrips, int nchunks, int rows, int ring,
                                                   int per_xcd, int own) {
    const int b = blockIdx.x;
    const int blk = (b & 7) * per_xcd + (b >> 3);
    if ((b >> 3) >= per_xcd) return;
    const int w = __builtin_amdgcn_readfirstlane(blk * 4 + (int)(threadIdx.x >> 6));
    if (w >= nstrips * nchunks) return;
    const int chunk = w / nstrips, strip = w - chunk * nstrips;
    const int lane = threadIdx.x & 63;
    const int half = (64 - own) / 2;
    int i = strip * own - half + lane;
    if (i > nx + half - 1) i = nx + half - 1;
    const int j0 = chunk * rows - ring, j1 = min(ny, (chunk + 1) * rows) + ring;
    const bool owner = lane >= half && lane < 64 - half && i < nx;
    size_t off = (size_t)(j0 + H) * ld + (i + H);
    S5 c0 = in[off], c1 = cst[off], n0, n1;
    for (int j = j0; j < j1; ++j) {
        if (PF) { const size_t offn = off + (j + 1 < j1 ? ld : 0); n0 = in[offn]; n1 = cst[offn]; }
        double s = c0.a + c0.b + c0.c + c0.d + c0.e + c1.a + c1.b + c1.c + c1.d + c1.e;
        s += __shfl_up(s, 1) + __shfl_down(s, 1);
        if (owner && j >= chunk * rows && j < min(ny, (chunk + 1) * rows)) {
            S5 o; o.a = s; o.b = s + 1; o.c = s + 2; o.d = s + 3; o.e = s + 4;
            out[off] = o;
        }
        off += ld;
        if (PF) { c0 = n0; c1 = n1; }
        else if (j + 1 < j1) { c0 = in[off]; c1 = cst[off]; }
    }
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
    const int nx = argc > 1 ? atoi(argv[1]) : 2048, ny = nx, H = 8, ld = nx + 2 * H;
    const size_t n = (size_t)ld * (ny + 2 * H + 8);
    Tab t;
    for (int a = 0; a < 12; ++a) { double* p; CK(hipMalloc(&p, n * 8)); CK(hipMemset(p, 0, n * 8)); t.r[a] = p; }
    for (int a = 0; a < 8; ++a) { double* p; CK(hipMalloc(&p, n * 8)); CK(hipMemset(p, 0, n * 8)); t.w[a] = p; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto launch, double bytes, const char* name) {
        for (int k = 0; k < 5; ++k) launch();
        CK(hipEventRecord(e0));
        const int reps = 50;
        for (int k = 0; k < reps; ++k) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("%-58s %8.4f ms  %7.0f GB/s (algorithmic %.0f MB)\n", name, ms, bytes / ms * 1e-6, bytes * 1e-6);
    };
    const double cell = (double)nx * ny * 8;
    char name[128];
    {
        const int tiles_x = (nx + 63) / 64, nb = tiles_x * ((ny + 3) / 4), per = (nb + 7) / 8;
        time([&] { hipLaunchKernelGGL((k_tile<10, 5>), dim3(per * 8), dim3(256), 0, 0, t, ld, nx, ny, H, tiles_x, per); }, cell * 15, "tile 10r/5w");
        time([&] { hipLaunchKernelGGL((k_tile<7, 8>), dim3(per * 8), dim3(256), 0, 0, t, ld, nx, ny, H, tiles_x, per); }, cell * 15, "tile 7r/8w");
        time([&] { hipLaunchKernelGGL((k_tile<10, 1>), dim3(per * 8), dim3(256), 0, 0, t, ld, nx, ny, H, tiles_x, per); }, cell * 11, "tile 10r/1w");
    }
    {
        S5 *in, *cst, *out;
        CK(hipMalloc(&in, n * 40)); CK(hipMalloc(&cst, n * 40)); CK(hipMalloc(&out, n * 40));
        CK(hipMemset(in, 0, n * 40)); CK(hipMemset(cst, 0, n * 40)); CK(hipMemset(out, 0, n * 40));
        const int rowss[] = {8, 16, 25, 32, 64};
        for (int own = 52; own <= 64; own += 4) for (int rows : rowss) for (int ring = 0; ring <= 6; ring += 3) for (int pf = 0; pf < 2; ++pf) {
            const int nstrips = (nx + own - 1) / own, nchunks = (ny + rows - 1) / rows;
            const int nw = nstrips * nchunks, nb = (nw + 3) / 4, per = (nb + 7) / 8;
            snprintf(name, sizeof name, "aos   own=%d rows=%d ring=%d pf=%d waves=%d 10r/5w", own, rows, ring, pf, nw);
            if (pf) time([&] { hipLaunchKernelGGL((k_march_aos<1>), dim3(per * 8), dim3(256), 0, 0, in, cst, out, ld, nx, ny, H, nstrips, nchunks, rows, ring, per, own); }, cell * 15, name);
            else time([&] { hipLaunchKernelGGL((k_march_aos<0>), dim3(per * 8), dim3(256), 0, 0, in, cst, out, ld, nx, ny, H, nstrips, nchunks, rows, ring, per, own); }, cell * 15, name);
        }
    }
    if (argc > 2) return 0;
    const int owns[] = {60, 64};
    const int rowss[] = {8, 16, 25, 32, 64, 128};
    for (int own : owns) for (int rows : rowss) for (int ring = 0; ring <= 3; ring += 3) for (int pf = 0; pf < 2; ++pf) {
        const int nstrips = (nx + own - 1) / own, nchunks = (ny + rows - 1) / rows;
        const int nw = nstrips * nchunks, nb = (nw + 3) / 4, per = (nb + 7) / 8;
        snprintf(name, sizeof name, "march own=%d rows=%d ring=%d pf=%d waves=%d 10r/5w", own, rows, ring, pf, nw);
        if (pf) time([&] { hipLaunchKernelGGL((k_march<10, 5, 1>), dim3(per * 8), dim3(256), 0, 0, t, ld, nx, ny, H, nstrips, nchunks, rows, ring, per, own); }, cell * 15, name);
        else time([&] { hipLaunchKernelGGL((k_march<10, 5, 0>), dim3(per * 8), dim3(256), 0, 0, t, ld, nx, ny, H, nstrips, nchunks, rows, ring, per, own); }, cell * 15, name);
    }
    return 0;
}
