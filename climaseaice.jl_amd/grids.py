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


class Center:
    pass


class Face:
    pass


def _topo2(topology):
    t = tuple(topology)[:2]
    for x in t:
        if x not in (Periodic, Bounded):
            raise ValueError("horizontal topology must be Periodic or Bounded")
    return t


class _Grid2D:
    Nx: int
    Ny: int
    Hx: int
    Hy: int

    def field_size(self, LX, LY):
        """Parent-array extents (ni, nj) of a field at (LX, LY): Oceananigans' rule, SURVEY.md A.0."""
        ni = self.Nx + 2 * self.Hx + (1 if (LX is Face and self.topology[0] is Bounded) else 0)
        nj = self.Ny + 2 * self.Hy + (1 if (LY is Face and self.topology[1] is Bounded) else 0)
        return ni, nj

    def interior_size(self, LX, LY):
        nx = self.Nx + (1 if (LX is Face and self.topology[0] is Bounded) else 0)
        ny = self.Ny + (1 if (LY is Face and self.topology[1] is Bounded) else 0)
        return nx, ny

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
        i = np.arange(1, self.Nx + 1 + (1 if (LX is Face and self.topology[0] is Bounded) else 0))
        return self.x[0] + (i - 1) * self.dx if LX is Face else self.x[0] + (i - 0.5) * self.dx

    def ynodes(self, LY):
        j = np.arange(1, self.Ny + 1 + (1 if (LY is Face and self.topology[1] is Bounded) else 0))
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
        i = np.arange(1, self.Nx + 1 + (1 if (LX is Face and self.topology[0] is Bounded) else 0))
        return self.longitude[0] + (i - 1) * self.dlam if LX is Face else self.longitude[0] + (i - 0.5) * self.dlam

    def ynodes(self, LY):
        j = np.arange(1, self.Ny + 1 + (1 if (LY is Face and self.topology[1] is Bounded) else 0))
        return self.latitude[0] + (j - 1) * self.dphi if LY is Face else self.latitude[0] + (j - 0.5) * self.dphi

    def metrics(self):
        return dict(kind="per_j", dy=self.dy, dxc=self.dxc, dxf=self.dxf, azc=self.azc, azf=self.azf)
