"""Host-side grids mirroring the Oceananigans grids the reference's hot path is built on
(RectilinearGrid, LatitudeLongitudeGrid; SURVEY.md App. B for the metric definitions).

Only what the EVP / advection path needs: sizes, halos, horizontal topology and the horizontal
metrics (dx, dy, Az at the four staggered locations).  numpy only -- no device code here.
"""
import numpy as np


class Periodic:
    pass


class Bounded:
    pass


class Flat:
    pass


class FullyConnected:
    """Tile edge pair whose halos come from the neighbouring tiles on both sides."""


class LeftConnected:
    """Low side exchanged with a neighbour tile, high side is a wall (the last tile of a Bounded direction)."""


class RightConnected:
    """Low side is a wall (the first tile of a Bounded direction), high side exchanged."""


class RightFolded:
    """y topology of a TripolarGrid: the low side is a wall (southernmost latitude), the high side is the north fold,
    filled by the Zipper boundary condition (sea_ice_model.jl:57-64)."""


class LeftConnectedRightFolded:
    """The northernmost tile of a y partition of a RightFolded direction: low side exchanged, high side folded."""


class Center:
    pass


class Face:
    pass


def _topo2(topology):
    t = tuple(topology)[:2]
    for k, x in enumerate(t):
        if x not in (Periodic, Bounded) and not (k == 1 and x is RightFolded):
            raise ValueError("horizontal topology must be Periodic or Bounded (y may also be RightFolded)")
    if t[1] is RightFolded and t[0] is not Periodic:
        raise ValueError("a RightFolded y direction needs a Periodic x direction")
    return t


def hi_wall(T):
    """True when the high side of a direction with topology T is a wall (Face fields then hold N + 1 points)."""
    return T in (Bounded, LeftConnected)


def local_topology(T, r, R):
    """Topology of tile r of R along a direction of global topology T (one rank: unchanged)."""
    if R == 1:
        return T
    if T is Periodic:
        return FullyConnected
    if r == 0:
        return RightConnected
    if r == R - 1:
        return LeftConnectedRightFolded if T is RightFolded else LeftConnected
    return FullyConnected


class _Grid2D:
    Nx: int
    Ny: int
    Hx: int
    Hy: int

    def field_size(self, LX, LY):
        """Parent-array extents (ni, nj) of a field at (LX, LY): Oceananigans' rule, SURVEY.md A.0."""
        ni = self.Nx + 2 * self.Hx + (1 if (LX is Face and hi_wall(self.topology[0])) else 0)
        nj = self.Ny + 2 * self.Hy + (1 if (LY is Face and hi_wall(self.topology[1])) else 0)
        return ni, nj

    def interior_size(self, LX, LY):
        nx = self.Nx + (1 if (LX is Face and hi_wall(self.topology[0])) else 0)
        ny = self.Ny + (1 if (LY is Face and hi_wall(self.topology[1])) else 0)
        return nx, ny

    def ynodes_with_halo(self, LY):
        """y nodes of rows 1-Hy .. Ny+Hy+1 (the layout of the per-row metric vectors: row j at [j + Hy - 1]).
        Halo rows of a Periodic direction hold the node of the row they image, so that a tile recomputing its ring
        uses the owner's value; beyond a wall the spacing continues."""
        j = np.arange(1 - self.Hy, self.Ny + self.Hy + 2)
        if self.topology[1] is Periodic:
            j = (j - 1) % self.Ny + 1
        return self._ynode(j, LY)

    def stress_kernel_range(self):
        """KernelParameters(-Hx+2:Nx+Hx-1, -Hy+2:Ny+Hy-1), elasto_visco_plastic_rheology.jl:145."""
        return (-self.Hx + 2, self.Nx + self.Hx - 1, -self.Hy + 2, self.Ny + self.Hy - 1)


class RectilinearGrid(_Grid2D):
    """RectilinearGrid(size=(Nx, Ny), x=(x0, x1), y=(y0, y1), topology=(TX, TY), halo=(Hx, Hy)), regular spacing."""

    metric_kind = "uniform"

    def __init__(self, size, x=None, y=None, extent=None, topology=(Periodic, Periodic), halo=(3, 3)):
        self.Nx, self.Ny = int(size[0]), int(size[1])
        self.Hx, self.Hy = int(halo[0]), int(halo[1])
        self.topology = _topo2(topology)
        if extent is not None:
            x, y = (0.0, float(extent[0])), (0.0, float(extent[1]))
        self.x, self.y = (float(x[0]), float(x[1])), (float(y[0]), float(y[1]))
        self.dx = (self.x[1] - self.x[0]) / self.Nx
        self.dy = (self.y[1] - self.y[0]) / self.Ny

    def xnodes(self, LX):
        i = np.arange(1, self.Nx + 1 + (1 if (LX is Face and hi_wall(self.topology[0])) else 0))
        return self.x[0] + (i - 1) * self.dx if LX is Face else self.x[0] + (i - 0.5) * self.dx

    def ynodes(self, LY):
        j = np.arange(1, self.Ny + 1 + (1 if (LY is Face and hi_wall(self.topology[1])) else 0))
        return self.y[0] + (j - 1) * self.dy if LY is Face else self.y[0] + (j - 0.5) * self.dy

    def _ynode(self, j, LY):
        return self.y[0] + (j - 1) * self.dy if LY is Face else self.y[0] + (j - 0.5) * self.dy

    def metrics(self):
        return dict(kind="uniform", dx=self.dx, dy=self.dy)


class LatitudeLongitudeGrid(_Grid2D):
    """Regular LatitudeLongitudeGrid(size, longitude=(l0, l1), latitude=(p0, p1), topology, halo).

    Metrics (Oceananigans): dx(j) = R cos(phi) dlambda at the latitude of the
    location, dy = R dphi, Az^{cc}(j) = R^2 dlambda (sin phi_f[j+1] - sin phi_f[j]),
    Az^{cf/ff}(j) = R^2 dlambda (sin phi_c[j] - sin phi_c[j-1]).
    """

    metric_kind = "per_j"

    def __init__(self, size, longitude, latitude, topology=(Bounded, Bounded), halo=(3, 3), radius=6371e3):
        self.Nx, self.Ny = int(size[0]), int(size[1])
        self.Hx, self.Hy = int(halo[0]), int(halo[1])
        self.topology = _topo2(topology)
        self.longitude = (float(longitude[0]), float(longitude[1]))
        self.latitude = (float(latitude[0]), float(latitude[1]))
        self.radius = float(radius)
        self.dlam = (self.longitude[1] - self.longitude[0]) / self.Nx
        self.dphi = (self.latitude[1] - self.latitude[0]) / self.Ny
        self.dy = self.radius * np.deg2rad(self.dphi)
        n = self.Ny + 2 * self.Hy + 1
        j = np.arange(1 - self.Hy, 1 - self.Hy + n)                     # row index of entry t is j[t]
        phif = self.latitude[0] + (j - 1) * self.dphi                   # face latitudes
        phic = phif + 0.5 * self.dphi                                   # centre latitudes
        R, dl = self.radius, np.deg2rad(self.dlam)
        self.dxc = R * np.cos(np.deg2rad(phic)) * dl
        self.dxf = R * np.cos(np.deg2rad(phif)) * dl
        self.azc = R * R * dl * (np.sin(np.deg2rad(phif + self.dphi)) - np.sin(np.deg2rad(phif)))
        self.azf = R * R * dl * (np.sin(np.deg2rad(phic)) - np.sin(np.deg2rad(phic - self.dphi)))

    def xnodes(self, LX):
        i = np.arange(1, self.Nx + 1 + (1 if (LX is Face and hi_wall(self.topology[0])) else 0))
        return self.longitude[0] + (i - 1) * self.dlam if LX is Face else self.longitude[0] + (i - 0.5) * self.dlam

    def ynodes(self, LY):
        j = np.arange(1, self.Ny + 1 + (1 if (LY is Face and hi_wall(self.topology[1])) else 0))
        return self.latitude[0] + (j - 1) * self.dphi if LY is Face else self.latitude[0] + (j - 0.5) * self.dphi

    def _ynode(self, j, LY):
        return self.latitude[0] + (j - 1) * self.dphi if LY is Face else self.latitude[0] + (j - 0.5) * self.dphi

    def metrics(self):
        return dict(kind="per_j", dy=self.dy, dxc=self.dxc, dxf=self.dxf, azc=self.azc, azf=self.azf)


def fold_north(a, Nx, Ny, Hx, Hy, face_x, face_y, sign=1):
    """The Zipper (north fold) halo fill of one parent-shaped array, in numpy (upstream semantics as recalled in
    oracle/csi_oracle.c fold_north: c[i, Ny + j] = s c[i', Ny - j (+1 if Face in y)], i' = Nx - i + 1 (+1 if Face in x,
    column 1 onto itself without the sign change); x halos of the folded rows periodic).  a[j + Hy - 1, i + Hx - 1] is
    element (i, j).  Used for fold-consistent metric arrays and as an independent check of the C / HIP fills."""
    a = np.array(a, dtype=np.float64, copy=True)
    i = np.arange(1, Nx + 1)
    ip = Nx - i + (2 if face_x else 1)
    s = np.full(Nx, float(sign))
    s[ip > Nx] = abs(float(sign))
    ip = np.where(ip > Nx, ip - Nx, ip)
    for m in range(1, Hy + 1):
        js = Ny - m + (1 if face_y else 0)
        row = s * a[js + Hy - 1, ip + Hx - 1]
        a[Ny + m + Hy - 1, i + Hx - 1] = row
        for k in range(1, Hx + 1):
            a[Ny + m + Hy - 1, (1 - k) + Hx - 1] = a[Ny + m + Hy - 1, (Nx + 1 - k) + Hx - 1]
            a[Ny + m + Hy - 1, (Nx + k) + Hx - 1] = a[Ny + m + Hy - 1, k + Hx - 1]
    return a


# order of the twelve 2-D metric arrays of an orthogonal curvilinear grid (include/csi.h, CSI_METRIC_FULL)
METRIC_NAMES = [w + l for w in ("dx", "dy", "az") for l in ("cc", "fc", "cf", "ff")]


class OrthogonalCurvilinearGrid(_Grid2D):
    """A grid given by its twelve 2-D metric arrays (dx, dy, Az at (c,c), (f,c), (c,f), (f,f)), the form an
    OrthogonalSphericalShellGrid hands to the operators.  Arrays have (Ny + 2Hy + 1, Nx + 2Hx + 1) entries, element
    (i, j) at [j + Hy - 1, i + Hx - 1]; halo entries of a Periodic direction must repeat the interior.

    `from_grid(g)` spreads the metrics of a RectilinearGrid / LatitudeLongitudeGrid over 2-D arrays (same numbers, so
    the same results bit for bit in STRICT mode); `distort` perturbs them smoothly for tests of the general operators."""

    metric_kind = "full"

    def __init__(self, size, metrics, topology=(Periodic, Periodic), halo=(4, 4), nodes=None):
        self.Nx, self.Ny = int(size[0]), int(size[1])
        self.Hx, self.Hy = int(halo[0]), int(halo[1])
        self.topology = _topo2(topology)
        shape = (self.Ny + 2 * self.Hy + 1, self.Nx + 2 * self.Hx + 1)
        self._m = {}
        for name in METRIC_NAMES:
            a = np.ascontiguousarray(metrics[name], dtype=np.float64)
            if a.shape != shape:
                raise ValueError(f"metric {name}: shape {a.shape}, expected {shape}")
            self._m[name] = a
        self._nodes = nodes          # optional (xnodes(LX), ynodes(LY)) callables of the grid it was built from

    @classmethod
    def from_grid(cls, g, distort=0.0, seed=0):
        n, ni = g.Ny + 2 * g.Hy + 1, g.Nx + 2 * g.Hx + 1
        m = g.metrics()
        ones = np.ones((n, ni))
        if m["kind"] == "uniform":
            row = {k: np.full(n, m["dx"]) for k in ("dxc", "dxf")}
            row.update(azc=np.full(n, m["dx"] * m["dy"]), azf=np.full(n, m["dx"] * m["dy"]))
            dy = m["dy"]
        else:
            row, dy = m, m["dy"]
        out = {}
        for l in ("cc", "fc", "cf", "ff"):
            fy = l[1] == "f"
            out["dx" + l] = (row["dxf"] if fy else row["dxc"])[:, None] * ones
            out["az" + l] = (row["azf"] if fy else row["azc"])[:, None] * ones
            out["dy" + l] = dy * ones
        if distort:
            # smooth, location-dependent stretching (periodic in both directions so that Periodic halos stay images)
            ia, ja = np.arange(ni) - (g.Hx - 1), np.arange(n) - (g.Hy - 1)          # the index i, j of every entry
            if g.topology[0] is Periodic:
                ia = (ia - 1) % g.Nx + 1                                            # halo entries are exact images
            if g.topology[1] is Periodic:
                ja = (ja - 1) % g.Ny + 1
            ii, jj = ia[None, :] / g.Nx, ja[:, None] / g.Ny
            rng = np.random.default_rng(seed)
            for k, name in enumerate(METRIC_NAMES):
                ph = rng.random(2) * 2 * np.pi
                out[name] = out[name] * (1.0 + distort * np.sin(2 * np.pi * ii + ph[0]) * np.cos(2 * np.pi * jj + ph[1]))
        if g.topology[1] is RightFolded:
            # metric halos beyond the fold are the fold images of the interior metrics (a grid folded onto itself)
            for name in METRIC_NAMES:
                out[name] = fold_north(out[name], g.Nx, g.Ny, g.Hx, g.Hy, name[2] == "f", name[3] == "f", 1)
        return cls((g.Nx, g.Ny), out, topology=g.topology, halo=(g.Hx, g.Hy), nodes=(g.xnodes, g.ynodes, g._ynode))

    def xnodes(self, LX):
        return self._nodes[0](LX)

    def ynodes(self, LY):
        return self._nodes[1](LY)

    def _ynode(self, j, LY):
        return self._nodes[2](j, LY)

    def metrics(self):
        return dict(kind="full", **self._m)


class TileGrid(_Grid2D):
    """One tile of an Rx x Ry decomposition of a global grid (the analogue of an Oceananigans grid built on
    Distributed(arch; partition = Partition(Rx, Ry)), test/distributed_tests_utils.jl:60-62).

    Metrics are SLICES of the global grid's metric vectors, so a tiled run uses bit-identical metric values."""

    def __init__(self, global_grid, Rx, Ry, rx, ry, force_connected=False, local_group=None, host_group=None):
        G = global_grid
        self.local_group = local_group      # _lib.LocalGroup: the tiles of this process talk through it instead of RCCL
        self.host_group = host_group        # name of a POSIX shared-memory segment: the ranks are processes that talk over the host (csi_comm_init_host)
        if G.Nx % Rx or G.Ny % Ry:
            raise ValueError("grid size must be divisible by the partition")
        self.global_grid, self.Rx, self.Ry, self.rx, self.ry = G, Rx, Ry, rx, ry
        self.Nx, self.Ny, self.Hx, self.Hy = G.Nx // Rx, G.Ny // Ry, G.Hx, G.Hy
        self.topology = (local_topology(G.topology[0], rx, Rx), local_topology(G.topology[1], ry, Ry))
        if force_connected:   # testing aid: a Periodic direction with one tile exchanges with itself
            fc = force_connected if isinstance(force_connected, tuple) else (True, True)
            self.topology = tuple(FullyConnected if (T is Periodic and f) else t for T, t, f in zip(G.topology, self.topology, fc))
        self.periodic = (G.topology[0] is Periodic, G.topology[1] is Periodic)
        self.i_off, self.j_off = rx * self.Nx, ry * self.Ny
        self.metric_kind = G.metric_kind
        self.rank = ry * Rx + rx

    def _nodes(self, full, off, n, L, T):
        m = n + (1 if (L is Face and hi_wall(T)) else 0)
        return full[off:off + m]

    def xnodes(self, LX):
        G = self.global_grid
        x = G.xnodes(LX)
        if LX is Face and not hi_wall(G.topology[0]):
            x = np.append(x, x[-1] + (x[-1] - x[-2]))
        return self._nodes(x, self.i_off, self.Nx, LX, self.topology[0])

    def ynodes(self, LY):
        G = self.global_grid
        y = G.ynodes(LY)
        if LY is Face and not hi_wall(G.topology[1]):
            y = np.append(y, y[-1] + (y[-1] - y[-2]))
        return self._nodes(y, self.j_off, self.Ny, LY, self.topology[1])

    def ynodes_with_halo(self, LY):
        n = self.Ny + 2 * self.Hy + 1
        return np.ascontiguousarray(self.global_grid.ynodes_with_halo(LY)[self.j_off:self.j_off + n])

    def metrics(self):
        m = dict(self.global_grid.metrics())
        if m["kind"] == "per_j":
            n = self.Ny + 2 * self.Hy + 1
            for k in ("dxc", "dxf", "azc", "azf"):
                m[k] = np.ascontiguousarray(m[k][self.j_off:self.j_off + n])
        elif m["kind"] == "full":
            n, ni = self.Ny + 2 * self.Hy + 1, self.Nx + 2 * self.Hx + 1
            for k in METRIC_NAMES:
                m[k] = np.ascontiguousarray(m[k][self.j_off:self.j_off + n, self.i_off:self.i_off + ni])
        return m

    def local_interior(self, global_interior, LX, LY):
        """The part of a global (ny, nx) interior array this tile owns (plus the wall face, if any)."""
        nx, ny = self.interior_size(LX, LY)
        return np.ascontiguousarray(global_interior[self.j_off:self.j_off + ny, self.i_off:self.i_off + nx])
