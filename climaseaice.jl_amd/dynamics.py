"""Host-side mirror of the reference's dynamics containers (plain data; no arithmetic here).

ElastoViscoPlasticRheology   Rheologies/elasto_visco_plastic_rheology.jl:14-25,119-137
Auxiliaries                  :140-173
SplitExplicitSolver          SeaIceDynamics/split_explicit_momentum_equations.jl:18-46
SemiImplicitStress           SeaIceDynamics/sea_ice_external_stress.jl:84-130
SeaIceMomentumEquation       SeaIceDynamics/sea_ice_momentum_equations.jl:3-12,67-94
FPlane, BetaPlane            upstream Oceananigans.Coriolis
"""
import math

import numpy as np
from dataclasses import dataclass, field as dc_field
from types import SimpleNamespace

from .fields import CenterField, CornerField, Field, XFaceField, YFaceField
from .grids import Center, Face


class ReplacementPressure:
    pass


class IceStrength:
    pass


@dataclass
class ElastoViscoPlasticRheology:
    ice_compressive_strength: float = 27500.0
    ice_compaction_hardening: float = 20.0
    yield_curve_eccentricity: float = 2.0
    minimum_plastic_stress: float = 2e-9
    min_relaxation_parameter: float = 50.0
    max_relaxation_parameter: float = 300.0
    relaxation_strength: float = math.pi ** 2
    pressure_formulation: object = dc_field(default_factory=ReplacementPressure)


@dataclass
class SplitExplicitSolver:
    """SplitExplicitSolver(grid; substeps=120): default 120 (:31); SeaIceMomentumEquation's default is 150."""
    substeps: int = 120


_OMEGA_EARTH, _R_EARTH = 7.292115e-5, 6371.0e3      # Oceananigans defaults


class FPlane:
    """FPlane(f = ...) or FPlane(latitude = ..., rotation_rate = Omega_Earth): f = 2 Omega sin(latitude)."""

    def __init__(self, f=None, latitude=None, rotation_rate=_OMEGA_EARTH):
        if (f is None) == (latitude is None):
            if f is None:
                f = 1e-4
            else:
                raise ValueError("FPlane: give f or latitude, not both")
        self.f = float(f) if f is not None else 2.0 * rotation_rate * float(np.sin(np.deg2rad(latitude)))


class BetaPlane:
    """BetaPlane(f0 = ..., beta = ...) or BetaPlane(latitude = ..., rotation_rate, radius): f = f0 + beta * y with
    f0 = 2 Omega sin(latitude), beta = 2 Omega cos(latitude) / R (upstream Coriolis; test/test_time_stepping.jl:35).
    y is the node's y coordinate: (Face, Center) nodes for the u equation, (Center, Face) nodes for v."""

    def __init__(self, f0=None, beta=None, latitude=None, rotation_rate=_OMEGA_EARTH, radius=_R_EARTH):
        if latitude is not None:
            if f0 is not None or beta is not None:
                raise ValueError("BetaPlane: give (f0, beta) or latitude, not both")
            f0 = 2.0 * rotation_rate * float(np.sin(np.deg2rad(latitude)))
            beta = 2.0 * rotation_rate * float(np.cos(np.deg2rad(latitude))) / radius
        if f0 is None or beta is None:
            raise ValueError("BetaPlane needs f0 and beta, or latitude")
        self.f0, self.beta = float(f0), float(beta)

    def rows(self, grid):
        """(f at u points, f at v points) per row, rows 1-Hy .. Ny+Hy+1."""
        return (self.f0 + self.beta * grid.ynodes_with_halo(Center), self.f0 + self.beta * grid.ynodes_with_halo(Face))


class PointwiseCoriolis:
    """f given at every (Face, Center) and (Center, Face) node of an orthogonal curvilinear grid (e.g. 2 Omega sin(latitude)
    of a TripolarGrid, whose latitude varies along both indices): arrays of shape (Ny + 2Hy + 1, Nx + 2Hx + 1) laid out
    like the grid's metric arrays, halo entries holding the value of the point they image.  Applied with the FPlane /
    BetaPlane stencil (include/csi.h csi_coriolis_points_set).  On a tile: slices of the global arrays."""

    def __init__(self, f_u, f_v):
        self.f_u, self.f_v = np.asarray(f_u, dtype=np.float64), np.asarray(f_v, dtype=np.float64)

    def points(self, grid):
        n, ni = grid.Ny + 2 * grid.Hy + 1, grid.Nx + 2 * grid.Hx + 1
        j0, i0 = getattr(grid, "j_off", 0), getattr(grid, "i_off", 0)
        out = tuple(np.ascontiguousarray(a[j0:j0 + n, i0:i0 + ni]) for a in (self.f_u, self.f_v))
        assert out[0].shape == (n, ni), (out[0].shape, (n, ni))
        return out


@dataclass
class SemiImplicitStress:
    """tau = rho_e Cd |u_e - u| (u_e - u); u_e, v_e: None (ZeroField), a number (ConstantField) or a Field."""
    ue: object = None
    ve: object = None
    rho_e: float = 1026.0
    Cd: float = 5.5e-3


def Auxiliaries(rheology, grid, device=None):
    """The ten auxiliary fields of the EVP rheology; alpha pre-filled with alpha+ (evp:147-161)."""
    f = SimpleNamespace(
        s11=CenterField(grid, device, "sigma11"), s22=CenterField(grid, device, "sigma22"),
        s12=CornerField(grid, device, "sigma12"),
        un=XFaceField(grid, device, "un"), vn=YFaceField(grid, device, "vn"),
        P=CenterField(grid, device, "P"), alpha=CenterField(grid, device, "alpha"),
        Delta=CenterField(grid, device, "Delta"),
        zeta_f=CornerField(grid, device, "zeta_f"), zeta_c=CenterField(grid, device, "zeta_c"))
    f.alpha.fill_parent(rheology.max_relaxation_parameter)
    return SimpleNamespace(fields=f)


class StressBalanceFreeDrift:
    """Free-drift velocity of marginal ice from the balance of the top and bottom stresses
    (SeaIceDynamics/stress_balance_free_drift.jl:3-121).  Exactly one of the model's two stresses must be a
    SemiImplicitStress; the arguments are accepted for API parity and replaced by the model's stresses."""

    def __init__(self, top_momentum_stress=None, bottom_momentum_stress=None):
        self.top_momentum_stress = top_momentum_stress
        self.bottom_momentum_stress = bottom_momentum_stress


class SeaIceMomentumEquation:
    def __init__(self, grid, coriolis=None, rheology=None, top_momentum_stress=None, bottom_momentum_stress=None,
                 free_drift=None, solver=None, minimum_concentration=1e-3, minimum_mass=1.0, device=None):
        self.grid = grid
        self.coriolis = coriolis
        self.rheology = rheology if rheology is not None else ElastoViscoPlasticRheology()
        if not isinstance(self.rheology, ElastoViscoPlasticRheology):
            raise NotImplementedError("only ElastoViscoPlasticRheology is on the accelerated path (SURVEY.md 2, row 3)")
        self.solver = solver if solver is not None else SplitExplicitSolver(substeps=150)
        # free_drift: None (`nothing`: marginal ice is at rest) or StressBalanceFreeDrift(); like the reference's
        # materialize_free_drift (stress_balance_free_drift.jl:44-46) the balance uses the model's own stresses
        if free_drift is not None and not isinstance(free_drift, StressBalanceFreeDrift):
            raise NotImplementedError("free_drift: None or StressBalanceFreeDrift() (prescribed free-drift velocity fields are not on the accelerated path)")
        self.free_drift = free_drift
        self.external_momentum_stresses = SimpleNamespace(top=top_momentum_stress, bottom=bottom_momentum_stress)
        self.minimum_concentration = float(minimum_concentration)
        self.minimum_mass = float(minimum_mass)
        self.auxiliaries = Auxiliaries(self.rheology, grid, device)
