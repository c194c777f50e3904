"""climaseaice.jl_amd -- MI355X-native drop-in for the hot path of CliMA/ClimaSeaIce.jl.

The split-explicit EVP momentum sub-cycle, the WENO advection of h / aice and their launch loop
as hand-written gfx950 HIP kernels behind a C ABI (include/csi.h, libcsi_hip.so), plus this thin
host-side mirror of the reference's `SeaIceModel` / `time_step!` surface.  The directory name
contains a dot, so import it through the repo-root shim: `import climaseaice_jl_amd as csi`.
"""
from . import _lib
from ._lib import Context, CsiError, LocalGroup, plan_exchange, plan_ranges
from .dynamics import (Auxiliaries, BetaPlane, PointwiseCoriolis, ElastoViscoPlasticRheology, FPlane, IceStrength, ReplacementPressure,
                       SeaIceMomentumEquation, SemiImplicitStress, SplitExplicitSolver, StressBalanceFreeDrift)
from .fields import CenterField, CornerField, Field, XFaceField, YFaceField
from .grids import (Bounded, Center, Face, Flat, FullyConnected, LatitudeLongitudeGrid, LeftConnected,
                    OrthogonalCurvilinearGrid, Periodic, LeftConnectedRightFolded, RightFolded, fold_north,
                    RectilinearGrid, RightConnected, TileGrid, TripolarGrid)
from .model import (FieldBoundaryConditions, FluxBoundaryCondition, ImmersedBoundaryCondition, MeltingConstrainedFluxBalance, ValueBoundaryCondition, PrescribedTemperature, SeaIceModel, SlabThermodynamics, SnowSlabThermodynamics,
                    snow_slab_thermodynamics, UpwindBiased, WENO, set_, time_step, time_step_momentum, update_state,
                    prognostic_state, restore_prognostic_state)

__all__ = [n for n in dir() if not n.startswith("_")]
