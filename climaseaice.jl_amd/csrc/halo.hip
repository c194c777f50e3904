// halo.hip -- local halo fill and immersed masking kernels.
//
// fill_halo_regions!(field; only_local_halos = true) (upstream Oceananigans, semantics recorded
// in SURVEY.md App. B): Periodic sides wrap, Center-located fields on a wall mirror (no-flux),
// Face-located fields on a wall are untouched; connected (tile) sides are left to
// csi_halo_exchange.  Used once per stage (split_explicit_momentum_equations.jl:170-171,
// elasto_visco_plastic_rheology.jl:275-280, sea_ice_model.jl:381-384); inside the sub-cycle
// the fill is fused into the velocity kernels' stores.
#include "csi_dev.h"
#include "csi_kernels.h"

namespace csi {

// Only threads within H of an edge do anything; they copy their interior value to its images.
// strip 0: the rows within Hy of the y edges (all i); strip 1: the columns within Hx of the x
// edges for the remaining middle rows.
__global__ void k_fill_halo(FRef f, GridDev g, ImageSpec im, int strip) {
    const int Nx = g.Nx, Ny = g.Ny, Hx = g.Hx, Hy = g.Hy;
    int i, j;
    if (strip == 0) {
        i = 1 + blockIdx.x * blockDim.x + threadIdx.x;
        int jj = blockIdx.y * blockDim.y + threadIdx.y;      // 0 .. 2Hy-1 (+1: the wall-face row Ny+1 of a Face-y field)
        const int fo = im.yhi == IMG_FOLD ? 1 : 0;            // a fold also images row Ny - Hy (Center in y)
        if (i > Nx + im.ex || jj >= 2 * Hy + im.ey + fo) return;
        j = jj < Hy ? 1 + jj : Ny - 2 * Hy + 1 - fo + jj;      // rows 1..Hy and Ny-Hy+1(-1)..Ny(+1)
        if (j < 1 || j > Ny + im.ey || (jj >= Hy && j <= Hy)) return; // overlap when Ny < 2Hy
    } else {
        int ii = blockIdx.x * blockDim.x + threadIdx.x;      // 0 .. 2Hx-1
        j = 1 + Hy + blockIdx.y * blockDim.y + threadIdx.y;  // middle rows Hy+1 .. Ny-Hy
        if (ii >= 2 * Hx || j > Ny - Hy) return;
        i = ii < Hx ? 1 + ii : Nx - 2 * Hx + 1 + ii;
        if (i < 1 || i > Nx || (ii >= Hx && i <= Hx)) return;
    }
    store_with_images(f, g, im, i, j, f(i, j));
}

void launch_fill_halo(const FRef& f, const GridDev& g, const ImageSpec& im, hipStream_t s) {
    dim3 b(64, 4);
    unsigned gx0 = (unsigned)((g.Nx + im.ex + 63) / 64), gy0 = (unsigned)((2 * g.Hy + im.ey + 1 + 3) / 4);
    hipLaunchKernelGGL(k_fill_halo, dim3(gx0, gy0, 1), b, 0, s, f, g, im, 0);
    int mid = g.Ny - 2 * g.Hy;
    if (mid > 0) {
        unsigned gx1 = (unsigned)((2 * g.Hx + 63) / 64), gy1 = (unsigned)((mid + 3) / 4);
        hipLaunchKernelGGL(k_fill_halo, dim3(gx1, gy1, 1), b, 0, s, f, g, im, 1);
    }
}

// Several fields in one pair of launches (update_state! fills h, aice, [hs,] u, v back to back: at 512^2 the eight
// 5-us launches cost more than the advection kernel); blockIdx.z selects the field.
__global__ void k_fill_halo_batch(HaloBatch B, GridDev g, int strip) {
    const FRef f = B.f[blockIdx.z];
    const ImageSpec im = B.im[blockIdx.z];
    const int Nx = g.Nx, Ny = g.Ny, Hx = g.Hx, Hy = g.Hy;
    int i, j;
    if (strip == 0) {
        i = 1 + blockIdx.x * blockDim.x + threadIdx.x;
        int jj = blockIdx.y * blockDim.y + threadIdx.y;
        const int fo = im.yhi == IMG_FOLD ? 1 : 0;
        if (i > Nx + im.ex || jj >= 2 * Hy + im.ey + fo) return;
        j = jj < Hy ? 1 + jj : Ny - 2 * Hy + 1 - fo + jj;
        if (j < 1 || j > Ny + im.ey || (jj >= Hy && j <= Hy)) return;
    } else {
        int ii = blockIdx.x * blockDim.x + threadIdx.x;
        j = 1 + Hy + blockIdx.y * blockDim.y + threadIdx.y;
        if (ii >= 2 * Hx || j > Ny - Hy) return;
        i = ii < Hx ? 1 + ii : Nx - 2 * Hx + 1 + ii;
        if (i < 1 || i > Nx || (ii >= Hx && i <= Hx)) return;
    }
    store_with_images(f, g, im, i, j, f(i, j));
}
// The y images (mirror / wrap) of the halo COLUMNS beyond connected x sides: what a halo exchange of full-height strips brings
// along in the corners (the neighbour's own y fill), done locally where the columns themselves are already current -- after a
// sub-cycle on the peer halo transport (csi_abi.hip do_finalize).  Rows within Hy of the y edges, columns 1 - Hx .. 0 and Nx + 1 .. Nx + Hx.
__global__ void k_fill_halo_xcolumns(HaloBatch B, GridDev g, int rows) {
    const FRef f = B.f[blockIdx.z];
    const ImageSpec im = B.im[blockIdx.z];
    const int Nx = g.Nx, Ny = g.Ny, Hx = g.Hx, Hy = g.Hy;
    int ii = blockIdx.x * blockDim.x + threadIdx.x, jj = blockIdx.y * blockDim.y + threadIdx.y;
    int i, j;
    if (!rows) {
        // halo columns beyond connected x sides, rows within Hy of the y edges: their y images
        if (ii >= 2 * Hx + 1 || jj >= 2 * Hy + im.ey) return;
        i = ii < Hx ? 1 - Hx + ii : Nx + 1 + (ii - Hx);
        if ((i < 1 && g.xlo != SIDE_CONNECTED) || (i > Nx && g.xhi != SIDE_CONNECTED) || i > Nx + Hx) return;
        j = jj < Hy ? 1 + jj : Ny - 2 * Hy + 1 + jj;
        if (j < 1 || j > Ny + im.ey || (jj >= Hy && j <= Hy)) return;
    } else {
        // halo rows beyond connected y sides, columns within Hx of the x edges: their x images
        if (jj >= 2 * Hy + 1 || ii >= 2 * Hx + im.ex) return;
        j = jj < Hy ? 1 - Hy + jj : Ny + 1 + (jj - Hy);
        if ((j < 1 && g.ylo != SIDE_CONNECTED) || (j > Ny && g.yhi != SIDE_CONNECTED) || j > Ny + Hy) return;
        i = ii < Hx ? 1 + ii : Nx - 2 * Hx + 1 + ii;
        if (i < 1 || i > Nx + im.ex || (ii >= Hx && i <= Hx)) return;
    }
    store_with_images(f, g, im, i, j, f(i, j));                  // (outside the interior in one direction: the images of the other only)
}
void launch_fill_halo_xcolumns(const HaloBatch& B, const GridDev& g, hipStream_t s) {
    if (B.n <= 0) return;
    dim3 b(16, 16);
    const dim3 gr((unsigned)((2 * g.Hx + 1 + 15) / 16), (unsigned)((2 * g.Hy + 1 + 15) / 16), (unsigned)B.n);
    if (g.xlo == SIDE_CONNECTED || g.xhi == SIDE_CONNECTED) hipLaunchKernelGGL(k_fill_halo_xcolumns, gr, b, 0, s, B, g, 0);
    if (g.ylo == SIDE_CONNECTED || g.yhi == SIDE_CONNECTED) hipLaunchKernelGGL(k_fill_halo_xcolumns, gr, b, 0, s, B, g, 1);
}

void launch_fill_halo_batch(const HaloBatch& B, const GridDev& g, hipStream_t s) {
    if (B.n <= 0) return;
    dim3 b(64, 4);
    unsigned gx0 = (unsigned)((g.Nx + 1 + 63) / 64), gy0 = (unsigned)((2 * g.Hy + 2 + 3) / 4);
    hipLaunchKernelGGL(k_fill_halo_batch, dim3(gx0, gy0, (unsigned)B.n), b, 0, s, B, g, 0);
    int mid = g.Ny - 2 * g.Hy;
    if (mid > 0) {
        unsigned gx1 = (unsigned)((2 * g.Hx + 63) / 64), gy1 = (unsigned)((mid + 3) / 4);
        hipLaunchKernelGGL(k_fill_halo_batch, dim3(gx1, gy1, (unsigned)B.n), b, 0, s, B, g, 1);
    }
}

// cache_current_fields! (sea_ice_rk_substep.jl:29-42): whole parents of up to five fields copied by ONE launch (at 512^2 five
// hipMemcpyAsync launches of 2 MB each cost 25 us of an 87 us step); blockIdx.y selects the pair, 16-byte accesses.
__global__ void __launch_bounds__(256) k_copy_batch(CopyBatch B) {
    const double* __restrict__ src = B.src[blockIdx.y];
    double* __restrict__ dst = B.dst[blockIdx.y];
    const long n = B.n[blockIdx.y];
    const long stride = (long)gridDim.x * blockDim.x * 2;
    for (long t = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 2; t < n; t += stride) {
        if (t + 1 < n && B.aligned16) {
            *reinterpret_cast<double2*>(dst + t) = *reinterpret_cast<const double2*>(src + t);
        } else {
            dst[t] = src[t];
            if (t + 1 < n) dst[t + 1] = src[t + 1];
        }
    }
}
void launch_copy_batch(const CopyBatch& B, hipStream_t s, int block) {
    if (B.count <= 0) return;
    long nmax = 0;
    for (int k = 0; k < B.count; ++k) nmax = B.n[k] > nmax ? B.n[k] : nmax;
    unsigned gx = (unsigned)((nmax / 2 + block - 1) / block);
    if (gx > 2048) gx = 2048;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(k_copy_batch, dim3(gx, (unsigned)B.count), dim3((unsigned)block), 0, s, B);
}

// mask_immersed_field_xy!(field, k = Nz), sea_ice_model.jl:381-389: zero at peripheral nodes of an
// immersed grid (no-op without a mask).
__global__ void k_mask(FRef f, GridDev g, int kind) {
    int i = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    int j = 1 + blockIdx.y * blockDim.y + threadIdx.y;
    if (i > g.Nx || j > g.Ny) return;
    bool z = kind == 0 ? inactive_cell(g, i, j) : (kind == 1 ? peripheral_u(g, i, j) : peripheral_v(g, i, j));
    if (z) f(i, j) = 0.0;
}
static void launch_mask(const FRef& f, const GridDev& g, int kind, hipStream_t s) {
    if (!g.has_mask) return;
    dim3 b(64, 4);
    hipLaunchKernelGGL(k_mask, dim3((unsigned)((g.Nx + 63) / 64), (unsigned)((g.Ny + 3) / 4), 1), b, 0, s, f, g, kind);
}
void launch_mask_center(const FRef& f, const GridDev& g, hipStream_t s) { launch_mask(f, g, 0, s); }
void launch_mask_u(const FRef& f, const GridDev& g, hipStream_t s) { launch_mask(f, g, 1, s); }
void launch_mask_v(const FRef& f, const GridDev& g, hipStream_t s) { launch_mask(f, g, 2, s); }

}  // namespace csi
