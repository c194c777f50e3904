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


def _haversine(lam1, phi1, lam2, phi2, radius):
    """Great-circle distance (degrees in, metres out), the numerically stable half-angle form."""
    p1, p2 = np.deg2rad(phi1), np.deg2rad(phi2)
    dl, dp = np.deg2rad(lam2 - lam1), p2 - p1
    a = np.sin(0.5 * dp) ** 2 + np.cos(p1) * np.cos(p2) * np.sin(0.5 * dl) ** 2
    return 2.0 * radius * np.arcsin(np.sqrt(np.clip(a, 0.0, 1.0)))


class TripolarGrid(OrthogonalCurvilinearGrid):
    """TripolarGrid(size=(Nx, Ny), southernmost_latitude, north_poles_latitude, first_pole_longitude, halo): a global-like grid
    with the STRUCTURE of the reference's flagship grid (test/distributed_tests_utils.jl:170-183, 226-259 build
    `TripolarGrid(size = (60, 120, 1), southernmost_latitude = 60)` and immerse discs around its poles):

      * rows south of `north_poles_latitude`: a regular latitude-longitude grid -- every metric is a function of the row;
      * north of it: a bipolar cap.  In the stereographic plane of the north pole the cap is the disc |z| <= r0 =
        tan((90 - phi0) / 2); s = log((1 + z / r0) / (1 - z / r0)) maps it conformally onto the strip |Im s| <= pi / 2, the two
        poles z = +-r0 (on the cap's boundary circle, at first_pole_longitude and 180 degrees from it) to +-infinity.  Columns are
        the lines Re s = log cot(theta / 2) -- they leave the boundary circle where the meridian of longitude theta arrives --,
        rows the lines Im s = +-2 atan(r / r0), r = tan((90 - psi) / 2), psi uniform from phi0 to 90: along the meridian
        perpendicular to the pole axis psi IS the latitude (Murray 1996's bipolar projection, the form MOM's tripolar grids use).
        The row psi = 90 is the segment between the two poles: the north fold, through the centres of row Ny, column i onto
        column Nx - i + 1 (`RightFolded`); rows beyond it are the analytic continuation = the fold images.

    Both families of lines are level sets of a conformal map: the net is orthogonal.  Metrics are great-circle distances between
    neighbouring nodes of the staggered net (Az = dx dy), as upstream computes a TripolarGrid's -- except that a row of the
    latitude-longitude part takes the distance of its first interior column in every column, so that it is constant bit for bit
    (per-point evaluation leaves rounding noise there).  NOT upstream's node-for-node construction -- that cannot be recalled or run
    here --: the same topology, the same fold, the same split into a row-constant part and a curvilinear cap, singular poles.
    The (Face, .) nodes of columns 1 and Nx / 2 + 1 lie ON the pole axis, where every row of the cap passes through the pole: they
    are moved 1e-3 of a column off the axis so that no metric is exactly zero (the reference immerses the cells around the poles)."""

    def __init__(self, size, southernmost_latitude=-80.0, north_poles_latitude=55.0, first_pole_longitude=70.0, halo=(4, 4),
                 radius=6371e3):
        Nx, Ny = int(size[0]), int(size[1])
        Hx, Hy = int(halo[0]), int(halo[1])
        if Nx % 2:
            raise ValueError("a TripolarGrid needs an even number of columns (column i folds onto Nx - i + 1)")
        self.southernmost_latitude = float(southernmost_latitude)
        self.first_pole_longitude = float(first_pole_longitude)
        self.radius = float(radius)
        dth = 360.0 / Nx
        dpsi = (90.0 - self.southernmost_latitude) / (Ny - 0.5)        # the fold runs through the centres of row Ny
        Jc = int(round((north_poles_latitude - self.southernmost_latitude) / dpsi))
        Jc = min(max(Jc, 0), Ny - 2)
        phi0 = self.southernmost_latitude + Jc * dpsi
        self.north_poles_latitude = phi0            # (moved onto the nearest row face)
        self.cap_first_row = Jc + 1                 # rows 1 .. Jc: latitude-longitude; Jc + 1 .. Ny: the cap
        self.dtheta, self.dpsi = dth, dpsi
        r0 = np.tan(np.deg2rad(0.5 * (90.0 - phi0)))

        def nodes(xi, eta):
            """(longitude, latitude) of the net's points at column coordinate xi (face i at i - 1, centre at i - 1/2) and row
            coordinate eta (likewise), broadcast."""
            th = np.asarray(xi, dtype=np.float64) * dth
            psi = self.southernmost_latitude + np.asarray(eta, dtype=np.float64) * dpsi
            th, psi = np.broadcast_arrays(th, psi)
            lam, phi = self.first_pole_longitude + th, psi.copy()
            cap = psi > phi0
            if cap.any():
                t = np.deg2rad(th[cap])
                # off the pole axis by 1e-3 of a column (see the class docstring)
                eps = 1e-3 * np.deg2rad(dth)
                near = np.abs(np.sin(t)) < np.sin(eps)
                t = np.where(near, t + eps, t)
                r = np.tan(np.deg2rad(0.5 * (90.0 - psi[cap])))          # negative beyond the fold: the mirror image
                g = 2.0 * np.arctan(r / r0) * np.sign(np.sin(t))
                a = np.log(np.abs(1.0 / np.tan(0.5 * t)))
                z = r0 * np.tanh(0.5 * (a + 1j * g))
                lam = lam.copy()
                lam[cap] = self.first_pole_longitude + np.rad2deg(np.angle(z))
                phi[cap] = 90.0 - 2.0 * np.rad2deg(np.arctan(np.abs(z)))
            return lam, phi

        self._node_fn = nodes
        n, ni = Ny + 2 * Hy + 1, Nx + 2 * Hx + 1
        ia = np.arange(ni) - (Hx - 1)                # index i of every entry
        ja = np.arange(n) - (Hy - 1)
        xc, xf = (ia - 0.5)[None, :], (ia - 1.0)[None, :]
        yc, yf = (ja - 0.5)[:, None], (ja - 1.0)[:, None]
        R = self.radius
        d = lambda A, B: _haversine(A[0], A[1], B[0], B[1], R)      # noqa: E731
        m = {}
        m["dxcc"] = d(nodes(xf + 1.0, yc), nodes(xf, yc))
        m["dxfc"] = d(nodes(xc, yc), nodes(xc - 1.0, yc))
        m["dxcf"] = d(nodes(xf + 1.0, yf), nodes(xf, yf))
        m["dxff"] = d(nodes(xc, yf), nodes(xc - 1.0, yf))
        m["dycc"] = d(nodes(xc, yf + 1.0), nodes(xc, yf))
        m["dyfc"] = d(nodes(xf, yf + 1.0), nodes(xf, yf))
        m["dycf"] = d(nodes(xc, yc), nodes(xc, yc - 1.0))
        m["dyff"] = d(nodes(xf, yc), nodes(xf, yc - 1.0))
        # the two pole points are u points of row Ny (columns 1 and Nx / 2 + 1, where cells i - 1 and i are fold images of each other:
        # dx = 0) -- upstream's "north singularities"; the reference immerses the cells around them.  No metric below one metre, so
        # that every reciprocal stays finite.
        for k in list(m):
            m[k] = np.maximum(m[k], 1.0)
        for l in ("cc", "fc", "cf", "ff"):
            m["az" + l] = m["dx" + l] * m["dy" + l]
        # latitude-longitude rows: one value per row, bit for bit (rows j <= Jc of every location; parent row j + Hy - 1)
        tlat = Jc + Hy                               # parent rows [0, tlat) have j <= Jc
        for name in METRIC_NAMES:
            m[name][:tlat, :] = m[name][:tlat, Hx:Hx + 1]
        # exact images: periodic in x, folded in y (analytically they are; this removes the rounding noise)
        for name in METRIC_NAMES:
            a = m[name]
            for k in range(1, Hx + 1):
                a[:, (1 - k) + Hx - 1] = a[:, (Nx + 1 - k) + Hx - 1]
                a[:, (Nx + k) + Hx - 1] = a[:, k + Hx - 1]
            a[:, Nx + Hx + 1 + Hx - 1] = a[:, Hx + 1 + Hx - 1]      # (the extra column Nx + Hx + 1 images column Hx + 1)
            m[name] = fold_north(a, Nx, Ny, Hx, Hy, name[2] == "f", name[3] == "f", 1)
        super().__init__((Nx, Ny), m, topology=(Periodic, RightFolded), halo=(Hx, Hy), nodes=None)

    def nodes_2d(self, LX, LY, with_halo=False):
        """(longitude, latitude) of the interior points at (LX, LY) as (Ny, Nx) arrays (with_halo: the metric planes' shape)."""
        if with_halo:
            ia = np.arange(self.Nx + 2 * self.Hx + 1) - (self.Hx - 1)
            ja = np.arange(self.Ny + 2 * self.Hy + 1) - (self.Hy - 1)
        else:
            ia, ja = np.arange(1, self.Nx + 1), np.arange(1, self.Ny + 1)
        xi = (ia - 1.0 if LX is Face else ia - 0.5)[None, :]
        eta = (ja - 1.0 if LY is Face else ja - 0.5)[:, None]
        lam, phi = self._node_fn(xi, eta)
        return (lam + 180.0) % 360.0 - 180.0, phi

    def coriolis_planes(self, rotation_rate=7.292115e-5):
        """f = 2 Omega sin(latitude) at the u and the v points, in the metric planes' layout (PointwiseCoriolis): per row in the
        latitude-longitude part (one value per row, bit for bit), per point in the cap, fold images beyond the fold."""
        out = []
        for LX, LY in ((Face, Center), (Center, Face)):
            _, phi = self.nodes_2d(LX, LY, with_halo=True)
            f = 2.0 * rotation_rate * np.sin(np.deg2rad(phi))
            tlat = self.cap_first_row - 1 + self.Hy
            f[:tlat, :] = f[:tlat, self.Hx:self.Hx + 1]
            f = fold_north(f, self.Nx, self.Ny, self.Hx, self.Hy, LX is Face, LY is Face, 1)
            out.append(np.ascontiguousarray(f))
        return tuple(out)

    def analytic_land(self, radius=5.0):
        """The reference's `analytical_immersed_tripolar_grid` (test/distributed_tests_utils.jl:170-183): land within `radius`
        degrees (in longitude AND latitude) of the two north poles and within `radius` degrees of the southern edge.  Returns
        the (Ny, Nx) bool array of WET cells; row Ny is its own fold image by symmetry (enforced)."""
        lam, phi = self.nodes_2d(Center, Center)
        lp, pp, pm = self.first_pole_longitude, self.north_poles_latitude, self.southernmost_latitude
        dl = lambda l0: np.abs((lam - l0 + 180.0) % 360.0 - 180.0)      # noqa: E731
        land = ((dl(lp) < radius) & (np.abs(pp - phi) < radius)) | ((dl(lp + 180.0) < radius) & (np.abs(pp - phi) < radius)) | \
               (phi < pm + radius)
        wet = ~land
        wet[-1, :] &= wet[-1, ::-1]
        return wet

    def _ynode(self, j, LY):
        raise NotImplementedError("a TripolarGrid has no per-row y nodes (use nodes_2d / coriolis_planes)")

    def xnodes(self, LX):
        raise NotImplementedError("a TripolarGrid has 2-D nodes (nodes_2d)")

    ynodes = xnodes


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
