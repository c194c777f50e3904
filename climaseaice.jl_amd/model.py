"""SeaIceModel / set! / time_step! / update_state! -- host-side mirror of the reference API for the
accelerated path.  All arithmetic happens in libcsi_hip.so through the C ABI (include/csi.h).

SeaIceModel       sea_ice_model.jl:22-51,140-297 (fields allocated as :182-200, timestepper :235)
set!              :301-315
time_step!        sea_ice_fe_step.jl:13-34 (ForwardEuler) / upstream SplitRungeKutta3 loop around
                  rk_substep!, sea_ice_rk_substep.jl:81-94
update_state!     sea_ice_model.jl:379-394
"""
import ctypes as C
from types import SimpleNamespace

import numpy as np
import torch

from . import _lib
from .dynamics import (ElastoViscoPlasticRheology, FPlane, IceStrength, SeaIceMomentumEquation, SemiImplicitStress)
from .fields import CenterField, Field, XFaceField, YFaceField
from .grids import (METRIC_NAMES, Bounded, FullyConnected, LeftConnected, LeftConnectedRightFolded, Periodic, RightConnected,
                    RightFolded, TileGrid)


class ValueBoundaryCondition:
    """Oceananigans ValueBoundaryCondition(value): on a tangential velocity at a wall, value 0 is no-slip."""

    def __init__(self, value=0.0):
        self.value = float(value)


class FieldBoundaryConditions:
    """FieldBoundaryConditions(north =, south =, west =, east =): sides left out keep the default (no-flux for the
    tangential velocity, impenetrable for the normal one).  Used as SeaIceModel(boundary_conditions = dict(u =, v =))."""

    def __init__(self, north=None, south=None, west=None, east=None, immersed=None):
        self.north, self.south, self.west, self.east, self.immersed = north, south, west, east, immersed


class FluxBoundaryCondition:
    """Oceananigans FluxBoundaryCondition(number)."""

    def __init__(self, value=0.0):
        self.value = float(value)


class ImmersedBoundaryCondition:
    """ImmersedBoundaryCondition(west =, east =, south =, north =) of FluxBoundaryConditions with number values: the
    stresses the ice feels on immersed faces (ice_stress_divergence.jl:65-123).  Passed as
    SeaIceModel(boundary_conditions = dict(u = FieldBoundaryConditions(immersed = ImmersedBoundaryCondition(...)), ...))."""

    def __init__(self, west=None, east=None, south=None, north=None):
        self.west, self.east, self.south, self.north = west, east, south, north

    def values(self):
        out = []
        for bc in (self.west, self.east, self.south, self.north):
            if bc is None:
                out.append(0.0)
            elif isinstance(bc, FluxBoundaryCondition):
                out.append(bc.value)
            else:
                raise NotImplementedError("immersed boundary conditions: FluxBoundaryCondition(number) or None")
        return out


class PrescribedTemperature:
    """HeatBoundaryConditions.PrescribedTemperature(T)."""

    def __init__(self, temperature):
        self.temperature = float(temperature)


class MeltingConstrainedFluxBalance:
    """HeatBoundaryConditions.MeltingConstrainedFluxBalance (top_heat_boundary_conditions.jl:5-52): the top temperature
    balances the external and conductive fluxes, capped at the melting temperature.  On the accelerated path the
    external flux is a number, for which the reference's secant solve has a closed-form root (include/csi.h)."""


class SlabThermodynamics:
    """SlabThermodynamics of the ice layer with a PrescribedTemperature (top_temperature = number) or
    MeltingConstrainedFluxBalance (top_heat_boundary_condition) top boundary condition and IceWaterThermalEquilibrium
    at the bottom (SeaIceThermodynamics/slab_sea_ice_thermodynamics.jl:82-109); PhaseTransitions defaults
    (SeaIceThermodynamics.jl:106-124).  Heat fluxes: numbers; top_heat_flux=None is the reference's default: for a
    prescribed temperature the external flux in equilibrium with the internal conductive flux, otherwise 0
    (sea_ice_model.jl:243-256); bottom_heat_flux="frazil" is the -(1 - aice) W m^-2 flux of
    examples/freezing_bucket.jl:79-81."""

    def __init__(self, top_temperature=-10.0, conductivity=2.0, top_heat_flux=None, bottom_heat_flux=0.0,
                 heat_capacity=2000.0, density=917.0, liquid_density=999.8, liquid_heat_capacity=4186.0,
                 reference_latent_heat=334e3, reference_temperature=0.0, liquidus_slope=0.054,
                 freshwater_melting_temperature=0.0, bottom_salinity=0.0, ice_consolidation_thickness=0.05,
                 top_heat_boundary_condition=None, ice_salinity=0.0):
        self.__dict__.update(locals())
        del self.__dict__["self"]
        if isinstance(top_heat_boundary_condition, PrescribedTemperature):
            self.top_temperature = top_heat_boundary_condition.temperature
        self.flux_balance = isinstance(top_heat_boundary_condition, MeltingConstrainedFluxBalance)

    def params(self, sea_ice_density, snow=False):
        frazil = self.bottom_heat_flux == "frazil"
        equilibrium = self.top_heat_flux is None and not self.flux_balance and not snow     # sea_ice_model.jl:245-256
        return _lib.SlabParams(self.conductivity, sea_ice_density, self.density, self.liquid_density, self.liquid_heat_capacity,
                               self.heat_capacity, self.reference_latent_heat, self.reference_temperature, self.liquidus_slope,
                               self.freshwater_melting_temperature, self.bottom_salinity, self.ice_consolidation_thickness,
                               self.top_temperature, 1 if equilibrium else 0, 1 if frazil else 0,
                               0.0 if self.top_heat_flux is None else float(self.top_heat_flux),
                               1.0 if frazil else float(self.bottom_heat_flux),
                               1 if self.flux_balance else 0, 0, float(self.ice_salinity))


class SnowSlabThermodynamics:
    """snow_slab_thermodynamics(grid; conductivity = 0.31) (slab_sea_ice_thermodynamics.jl:42-49): the snow layer of
    the layered step.  Top boundary condition: MeltingConstrainedFluxBalance (the default) or PrescribedTemperature."""

    def __init__(self, conductivity=0.31, top_heat_boundary_condition=None):
        self.conductivity = float(conductivity)
        self.top_heat_boundary_condition = top_heat_boundary_condition or MeltingConstrainedFluxBalance()

    def params(self, snow_density, snowfall):
        bc = self.top_heat_boundary_condition
        prescribed = isinstance(bc, PrescribedTemperature)
        return _lib.SnowParams(self.conductivity, float(snow_density), float(snowfall),
                               bc.temperature if prescribed else 0.0, 0 if prescribed else 1, 0)


def snow_slab_thermodynamics(grid=None, conductivity=0.31, **kw):
    return SnowSlabThermodynamics(conductivity=conductivity, **kw)


class WENO:
    """WENO(order = 3 | 5 | 7), upstream's upwind-biased WENO-Z scheme (called at sea_ice_advection.jl:51-58).
    weight_dtype: "f64" (default) or "f32" -- the precision of the smoothness indicators and nonlinear weights, mirroring the second
    float type parameter FT2 of newer upstream versions (include/csi.h: csi_set_weno_weight_dtype; recalled, unverified)."""

    def __init__(self, order=5, weight_dtype="f64"):
        if order not in (3, 5, 7):
            raise NotImplementedError("WENO order 3, 5 or 7")
        if weight_dtype not in ("f64", "f32"):
            raise ValueError("weight_dtype must be 'f64' or 'f32'")
        self.order = order
        self.scheme = order
        self.weight_dtype = weight_dtype


class UpwindBiased:
    def __init__(self, order=5):
        if order not in (1, 3, 5):
            raise NotImplementedError("UpwindBiased order 1, 3 or 5")
        self.order = order
        self.scheme = 1 if order == 1 else -order


_TOPO = {Periodic: _lib.PERIODIC, Bounded: _lib.BOUNDED, FullyConnected: _lib.FULLY_CONNECTED,
         LeftConnected: _lib.LEFT_CONNECTED, RightConnected: _lib.RIGHT_CONNECTED,
         RightFolded: _lib.RIGHT_FOLDED, LeftConnectedRightFolded: _lib.LEFT_CONNECTED_RIGHT_FOLDED}


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class SeaIceModel:
    def __init__(self, grid, dynamics=None, advection=None, timestepper="SplitRungeKutta3", sea_ice_density=900.0,
                 ice_thermodynamics=None, snow_thermodynamics=None, snow_density=330.0, snowfall=0.0,
                 boundary_conditions=None, forcing=None, device="cuda:0", mode="fast", stream=None):
        self.grid = grid
        self.dynamics = dynamics
        self.advection = advection
        self.ice_thermodynamics = ice_thermodynamics
        self.boundary_conditions = boundary_conditions or {}
        self.forcing = forcing          # dict(u = array-like, v = array-like): model.forcing given as arrays (m s^-2)
        self.snow_thermodynamics = snow_thermodynamics
        self.snow_density, self.snowfall = float(snow_density), float(snowfall)
        if snow_thermodynamics is not None and ice_thermodynamics is None:
            raise ValueError("a snow layer needs ice_thermodynamics")
        if timestepper not in ("SplitRungeKutta3", "ForwardEuler"):
            raise ValueError("timestepper must be 'SplitRungeKutta3' or 'ForwardEuler'")
        self.timestepper_kind = timestepper
        self.sea_ice_density = float(sea_ice_density)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("SeaIceModel needs a HIP device (torch 'cuda' device); there is no CPU path")
        dev = self.device
        # prognostic fields (sea_ice_model.jl:182-183,199-200)
        self.velocities = SimpleNamespace(u=XFaceField(grid, dev, "u"), v=YFaceField(grid, dev, "v"))
        self.ice_thickness = CenterField(grid, dev, "h")
        self.ice_concentration = CenterField(grid, dev, "aice")
        # TimeStepper(timestepper, grid, prognostic_fields): G^n for both, Psi^- for RK (:235)
        Gn = SimpleNamespace(h=CenterField(grid, dev, "Gh"), aice=CenterField(grid, dev, "Gaice"))
        self.timestepper = SimpleNamespace(Gn=Gn, Psi_minus=None)
        if timestepper == "SplitRungeKutta3":
            self.timestepper.Psi_minus = SimpleNamespace(h=CenterField(grid, dev, "h-"), aice=CenterField(grid, dev, "aice-"),
                                                         u=XFaceField(grid, dev, "u-"), v=YFaceField(grid, dev, "v-"))
        # snow layer: hs is a prognostic field with its own tendency and cache (sea_ice_model.jl:201-224)
        self.snow_thickness = None
        self.mass_fluxes = None
        if snow_thermodynamics is not None:
            self.snow_thickness = CenterField(grid, dev, "hs")
            Gn.hs = CenterField(grid, dev, "Ghs")
            if self.timestepper.Psi_minus is not None:
                self.timestepper.Psi_minus.hs = CenterField(grid, dev, "hs-")
            self.mass_fluxes = SimpleNamespace(thermodynamics=SimpleNamespace(ice=CenterField(grid, dev, "mass_flux"),
                                                                              snow=CenterField(grid, dev, "mass_flux_snow")),
                                               intercepted_snowfall=CenterField(grid, dev, "intercepted_snowfall"))
            self.snow_top_temperature = CenterField(grid, dev, "Tu_snow")
            self.ice_top_temperature = CenterField(grid, dev, "Tu")
        self.clock = SimpleNamespace(time=0.0, iteration=0)
        self._keep = []
        self._stress_fields = {}
        self.ctx = _lib.Context(dev.index or 0, stream)
        self._configure()
        self.set_mode(mode)
        self.ctx.call("csi_set_weno_weight_dtype", 1 if getattr(advection, "weight_dtype", "f64") == "f32" else 0)
        torch.cuda.synchronize(self.device)   # field initialisation ran on torch's stream

    # ---- plumbing: describe the problem to the library -----------------------------------------
    def _bind(self, name, fld):
        self.ctx.call("csi_field_bind", _lib.F[name], C.c_void_p(fld.data.data_ptr()), fld.ni, fld.ni, fld.nj)

    def _configure(self):
        g = self.grid
        m = g.metrics()
        met = _lib.Metrics()
        if m["kind"] == "uniform":
            kind = _lib.METRIC_UNIFORM
            met.dx, met.dy = m["dx"], m["dy"]
        elif m["kind"] == "full":
            kind = _lib.METRIC_FULL
            met.full_ld = g.Nx + 2 * g.Hx + 1
            for k, name in enumerate(METRIC_NAMES):
                a = np.ascontiguousarray(m[name], dtype=np.float64)
                assert a.shape == (g.Ny + 2 * g.Hy + 1, g.Nx + 2 * g.Hx + 1), (name, a.shape)
                self._keep.append(a)
                met.full[k] = _dptr(a)
        else:
            kind = _lib.METRIC_PER_J
            met.dy = m["dy"]
            for k in ("dxc", "dxf", "azc", "azf"):
                a = np.ascontiguousarray(m[k], dtype=np.float64)
                self._keep.append(a)
                setattr(met, k, _dptr(a))
        self.ctx.call("csi_grid_set", g.Nx, g.Ny, g.Hx, g.Hy, _TOPO[g.topology[0]], _TOPO[g.topology[1]], kind, C.byref(met))
        if isinstance(g, TileGrid):
            self._init_tiles(g)
        self._bind("H", self.ice_thickness)
        self._bind("A", self.ice_concentration)
        self._bind("U", self.velocities.u)
        self._bind("V", self.velocities.v)
        self._bind("GH", self.timestepper.Gn.h)
        self._bind("GA", self.timestepper.Gn.aice)
        if self.timestepper.Psi_minus is not None:
            pm = self.timestepper.Psi_minus
            self._bind("HM", pm.h); self._bind("AM", pm.aice); self._bind("UM", pm.u); self._bind("VM", pm.v)
        if self.ice_thermodynamics is not None:
            sp = self.ice_thermodynamics.params(self.sea_ice_density, snow=self.snow_thermodynamics is not None)
            self.ctx.call("csi_slab_params_set", C.byref(sp))
        if self.snow_thermodynamics is not None:
            self._bind("HS", self.snow_thickness)
            self._bind("GHS", self.timestepper.Gn.hs)
            if self.timestepper.Psi_minus is not None:
                self._bind("HSM", self.timestepper.Psi_minus.hs)
            self._bind("MASS_FLUX", self.mass_fluxes.thermodynamics.ice)
            self._bind("MASS_FLUX_SNOW", self.mass_fluxes.thermodynamics.snow)
            self._bind("SNOWFALL_INTERCEPTED", self.mass_fluxes.intercepted_snowfall)
            self._bind("TU", self.ice_top_temperature)
            self._bind("TUS", self.snow_top_temperature)
            wp = self.snow_thermodynamics.params(self.snow_density, self.snowfall)
            self.ctx.call("csi_snow_params_set", C.byref(wp))
        # boundary_conditions = (u = FieldBoundaryConditions(north = ValueBoundaryCondition(0), ...), v = ...)
        for name, sides in (("U", ("south", "north")), ("V", ("west", "east"))):
            bcs = self.boundary_conditions.get(name.lower())
            for k, side in enumerate(sides):
                bc = getattr(bcs, side, None) if bcs is not None else None
                if bc is not None and not isinstance(bc, ValueBoundaryCondition):
                    raise NotImplementedError("velocity boundary conditions: ValueBoundaryCondition or the default")
                self.ctx.call("csi_velocity_bc_set", _lib.F[name], k, 0 if bc is None else 1, 0.0 if bc is None else bc.value)
            imm = getattr(bcs, "immersed", None) if bcs is not None else None
            self.ctx.call("csi_immersed_flux_bc_set", _lib.F[name], *(imm.values() if imm is not None else (0.0, 0.0, 0.0, 0.0)))
        # model.forcing = (u = array, v = array): user forcing of the velocity tendencies (sum_of_forcing_u / _v)
        if self.forcing is not None:
            fu, fv = self.forcing["u"], self.forcing["v"]
            self.forcing_fields = SimpleNamespace(u=XFaceField(g, self.device, "forcing_u"), v=YFaceField(g, self.device, "forcing_v"))
            self.forcing_fields.u.set(fu); self.forcing_fields.v.set(fv)
            self._bind("FORCING_U", self.forcing_fields.u)
            self._bind("FORCING_V", self.forcing_fields.v)
            torch.cuda.synchronize(self.device)
            self.ctx.call("csi_fill_halo_local", _lib.F["FORCING_U"])
            self.ctx.call("csi_fill_halo_local", _lib.F["FORCING_V"])
        d = self.dynamics
        if d is None:
            return
        if d.grid is not g:
            raise ValueError("dynamics was built on a different grid")
        f = d.auxiliaries.fields
        for fld in vars(f).values():
            if fld.data.device != self.device:
                fld.data = fld.data.to(self.device)
        for name, fld in (("S11", f.s11), ("S22", f.s22), ("S12", f.s12), ("UN", f.un), ("VN", f.vn), ("P", f.P),
                          ("ALPHA", f.alpha), ("DELTA", f.Delta), ("ZETA_F", f.zeta_f), ("ZETA_C", f.zeta_c)):
            self._bind(name, fld)
        r = d.rheology
        p = _lib.EvpParams(r.ice_compressive_strength, r.ice_compaction_hardening, r.yield_curve_eccentricity,
                           r.minimum_plastic_stress, r.min_relaxation_parameter, r.max_relaxation_parameter,
                           r.relaxation_strength,
                           _lib.PRESSURE_ICE_STRENGTH if isinstance(r.pressure_formulation, IceStrength) else _lib.PRESSURE_REPLACEMENT,
                           0 if d.coriolis is None else 1, float(getattr(d.coriolis, "f", 0.0)),
                           d.minimum_concentration, d.minimum_mass, self.sea_ice_density)
        self.ctx.call("csi_evp_params_set", C.byref(p))
        if hasattr(d.coriolis, "points"):          # per-point f (curvilinear grids)
            fu, fv = d.coriolis.points(g)
            self._keep += [fu, fv]
            self.ctx.call("csi_coriolis_rows_set", None, None, 0)
            self.ctx.call("csi_coriolis_points_set", _dptr(fu), _dptr(fv), fu.shape[1])
        elif hasattr(d.coriolis, "rows"):          # BetaPlane: f per row, evaluated here like the metric vectors
            fu, fv = (np.ascontiguousarray(a, dtype=np.float64) for a in d.coriolis.rows(g))
            self.ctx.call("csi_coriolis_rows_set", _dptr(fu), _dptr(fv), fu.size)
        else:
            self.ctx.call("csi_coriolis_rows_set", None, None, 0)
        self._set_stress(_lib.STRESS_TOP, d.external_momentum_stresses.top, "TOP")
        self._set_stress(_lib.STRESS_BOTTOM, d.external_momentum_stresses.bottom, "BOT")
        self.ctx.call("csi_free_drift_set", 1 if d.free_drift is not None else 0)

    def _init_tiles(self, g):
        """csi_tile_set + RCCL communicator: rank 0 makes the unique id, the host broadcasts it
        (torch.distributed when there is more than one process; plumbing only)."""
        self.ctx.call("csi_tile_set", g.rx, g.ry, g.Rx, g.Ry, int(g.periodic[0]), int(g.periodic[1]))
        world = g.Rx * g.Ry
        if getattr(g, "local_group", None) is not None:      # several tiles in this process (one thread each): no RCCL
            if g.local_group.world_size != world:
                raise RuntimeError("the local group's size does not match the partition")
            self._keep.append(g.local_group)
            self.ctx.call("csi_comm_init_local", g.local_group.h, g.rank)
            return
        if getattr(g, "host_group", None) is not None:       # one process per tile, no RCCL: shared memory + HIP IPC
            self.ctx.call("csi_comm_init_host", str(g.host_group).encode(), world, g.rank)
            return
        idbuf = (C.c_uint8 * 128)()
        if world > 1:
            import torch.distributed as dist
            if not dist.is_initialized():
                raise RuntimeError("a tiled model with more than one tile needs torch.distributed to be initialised")
            if dist.get_world_size() != world or dist.get_rank() != g.rank:
                raise RuntimeError("tile rank / partition do not match torch.distributed")
            payload = [None]
            if g.rank == 0:
                rc = self.ctx.L.csi_comm_unique_id(idbuf)
                if rc != _lib.OK:
                    raise _lib.CsiError(rc, self.ctx.L.csi_last_error(None).decode())
                payload = [bytes(idbuf)]
            dist.broadcast_object_list(payload, src=0)
            idbuf = (C.c_uint8 * 128).from_buffer_copy(payload[0])
        else:
            rc = self.ctx.L.csi_comm_unique_id(idbuf)
            if rc != _lib.OK:
                raise _lib.CsiError(rc, self.ctx.L.csi_last_error(None).decode())
        self.ctx.call("csi_comm_init", world, g.rank, idbuf)

    def _stress_field(self, slot, comp, value):
        """materialize_stress (sea_ice_external_stress.jl:63-69,132-137): a device copy on the model grid."""
        mk = XFaceField if comp == "U" else YFaceField
        if isinstance(value, Field):
            fld = value
            if fld.data.device != self.device:
                fld.data = fld.data.to(self.device)
        else:
            fld = mk(self.grid, self.device, f"{slot}_{comp}".lower())
            fld.set(value)
        self._stress_fields[f"{slot}_{comp}"] = fld
        self._bind(f"{slot}_{comp}", fld)
        return fld

    def _set_stress(self, side, stress, slot):
        s = _lib.Stress()
        if stress is None:
            s.kind = _lib.STRESS_NONE
        elif isinstance(stress, SemiImplicitStress):
            s.kind = _lib.STRESS_SEMI_IMPLICIT
            s.rho_e, s.Cd = stress.rho_e, stress.Cd
            for comp, val in (("u", stress.ue), ("v", stress.ve)):
                if val is None:
                    setattr(s, comp + "e_kind", _lib.VEL_ZERO)
                elif np.isscalar(val):
                    setattr(s, comp + "e_kind", _lib.VEL_CONST)
                    setattr(s, comp + "e", float(val))
                else:
                    setattr(s, comp + "e_kind", _lib.VEL_FIELD)
                    self._stress_field(slot, comp.upper(), val)
        else:
            tu, tv = (stress["u"], stress["v"]) if isinstance(stress, dict) else (stress[0], stress[1])
            if np.isscalar(tu) and np.isscalar(tv):
                s.kind = _lib.STRESS_CONST
                s.tau_u, s.tau_v = float(tu), float(tv)
            else:
                s.kind = _lib.STRESS_FIELD
                self._stress_field(slot, "U", tu)
                self._stress_field(slot, "V", tv)
        self.ctx.call("csi_stress_set", side, C.byref(s))

    def external_stress_field(self, slot, comp):
        return self._stress_fields[f"{slot}_{comp}"]

    def set_mode(self, mode):
        self.mode = mode
        self.ctx.call("csi_set_mode", _lib.MODE_FAST if mode == "fast" else _lib.MODE_STRICT)

    def set_fusion(self, level):
        """FAST mode: 0 / False = three-kernel path, 1 = one fused launch per sub-step, 2 / True (default) = two
        sub-steps per launch where the configuration allows it, 3 = three per launch on fully periodic grids with halo >= 6
        (two elsewhere); bit-identical results."""
        level = 2 if level is True else int(level)
        self.ctx.call("csi_set_fusion", level)

    def set_tile_skipping(self, on):
        """Untiled grids on the two-sub-steps kernel: skip the tiles with no ice mass in or around them (exact: include/csi.h,
        csi_set_tile_skipping; default on)."""
        self.ctx.call("csi_set_tile_skipping", 1 if on else 0)

    def tile_activity(self):
        """(tiles, live, used): the newest counts that have arrived from the device (synchronize() first for the last sub-cycle's)."""
        import ctypes as C
        t, l, u = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        self.ctx.call("csi_tile_activity", C.byref(t), C.byref(l), C.byref(u))
        return t.value, l.value, u.value

    def set_row_constant(self, on=True, rtol=0.0):
        """CSI_METRIC_FULL grids: rows whose twelve metric planes hold one value per row are read from per-row vectors (bitwise-equal
        columns only unless rtol > 0: include/csi.h, csi_set_row_constant)."""
        self.ctx.call("csi_set_row_constant", 1 if on else 0, float(rtol))

    def row_constant_rows(self):
        import ctypes as C
        n = C.c_int32(0)
        self.ctx.call("csi_row_constant_rows", C.byref(n))
        return n.value

    def set_exchange_interval(self, k):
        """Tiles: exchange u, v halos of width 2k every k sub-steps (0 = automatic from the halo size)."""
        self.ctx.call("csi_set_exchange_interval", int(k))

    def set_halo_transport(self, kind):
        """Tiles: "peer" (default) = peer-direct halo writes over xGMI with flags, one RCCL exchange per sub-cycle; "rccl" = the
        k-batched ncclSend / ncclRecv exchange (include/csi.h, csi_set_halo_transport)."""
        self.ctx.call("csi_set_halo_transport", {"rccl": 0, "peer": 1}[kind])

    def set_peer_tier(self, tier):
        """Tiles on the peer transport: the memory-ordering tier of its flag protocol, the SAME on every rank (-1 automatic, the
        default: 1 across processes / devices, 0 for a tile connected to itself; 0 no fence, 1 + acquire fence behind the flags,
        2 + release fence before them; include/csi.h, csi_set_peer_tier)."""
        self.ctx.call("csi_set_peer_tier", int(tier))

    def set_mask(self, active):
        """ImmersedBoundaryGrid stand-in: `active` is a (Ny, Nx) boolean array of wet cells of the WHOLE domain (for a
        tile: of the global grid; the tile's mask, halo included, is sliced from it -- the mask is static, so no
        exchange is needed)."""
        g = self.grid
        G = g.global_grid if isinstance(g, TileGrid) else g
        act = np.ascontiguousarray(active).astype(np.uint8)
        if act.shape != (G.Ny, G.Nx):
            raise ValueError("mask must have the shape (Ny, Nx) of the (global) grid")
        # global mask with halos: periodic wrap; cells beyond a wall are inactive by definition
        full = np.zeros((G.Ny + 2 * G.Hy, G.Nx + 2 * G.Hx), dtype=np.uint8)
        full[G.Hy:G.Hy + G.Ny, G.Hx:G.Hx + G.Nx] = act
        if G.topology[0] is Periodic:
            full[:, :G.Hx] = full[:, G.Nx:G.Nx + G.Hx]
            full[:, G.Nx + G.Hx:] = full[:, G.Hx:2 * G.Hx]
        if G.topology[1] is Periodic:
            full[:G.Hy, :] = full[G.Ny:G.Ny + G.Hy, :]
            full[G.Ny + G.Hy:, :] = full[G.Hy:2 * G.Hy, :]
        if G.topology[1] is RightFolded:              # cells beyond the north fold are images of real cells
            from .grids import fold_north
            full = fold_north(full, G.Nx, G.Ny, G.Hx, G.Hy, False, False, 1).astype(np.uint8)
        if isinstance(g, TileGrid):
            full = np.ascontiguousarray(full[g.j_off:g.j_off + g.Ny + 2 * g.Hy, g.i_off:g.i_off + g.Nx + 2 * g.Hx])
        self.mask = torch.from_numpy(full).to(self.device)
        self.ctx.call("csi_mask_set", C.c_void_p(self.mask.data_ptr()), self.mask.shape[1])

    # ---- convenience ----------------------------------------------------------------------------
    @property
    def substeps(self):
        return 0 if self.dynamics is None else self.dynamics.solver.substeps

    @property
    def scheme(self):
        return 0 if self.advection is None else self.advection.scheme

    def synchronize(self):
        """Wait for the library's stream (call before reading fields with torch / numpy)."""
        self.ctx.call("csi_sync")

    def copy_to_field(self, fld, array):
        """Overwrite a field's parent array from a host array of the same shape (ordered with the library)."""
        self.synchronize()
        fld.data.copy_(torch.from_numpy(np.ascontiguousarray(array, dtype=np.float64)))
        torch.cuda.synchronize(self.device)


def set_(model, **kw):
    """set!(model; h, aice (or ℵ), u, v) then update_state! (sea_ice_model.jl:301-315)."""
    # Python NFKC-normalises identifiers, so the keyword `ℵ` (U+2135) arrives as U+05D0
    names = {"h": model.ice_thickness, "aice": model.ice_concentration, "\u2135": model.ice_concentration,
             "\u05d0": model.ice_concentration, "u": model.velocities.u, "v": model.velocities.v}
    if model.snow_thickness is not None:
        names["hs"] = model.snow_thickness
    for k, val in kw.items():
        if k not in names:
            raise KeyError(f"set!: unknown field {k}")
        names[k].set(val)
    torch.cuda.synchronize(model.device)   # torch wrote on its stream; the library uses its own
    update_state(model)
    model.synchronize()                    # ... and whoever reads the fields next (torch / numpy) sees the filled halos


def update_state(model):
    model.ctx.call("csi_update_state")


def _state_fields(model):
    """name -> Field of what Oceananigans.prognostic_state(model) saves (sea_ice_model.jl:414-426): velocities, ice
    thickness / concentration, snow thickness, the time stepper's G^n and Psi^-, the dynamics' auxiliary fields, the mass fluxes,
    the thermodynamics' top surface temperatures.  Not mirrored: `tracers` (the HIP path advects h, aice, hs only; a model with
    further tracers is not attached) and the clock's fields beyond (time, iteration)."""
    out = {"u": model.velocities.u, "v": model.velocities.v, "h": model.ice_thickness, "aice": model.ice_concentration}
    if model.snow_thickness is not None:
        out["hs"] = model.snow_thickness
    for k, f in vars(model.timestepper.Gn).items():
        out["Gn." + k] = f
    if model.timestepper.Psi_minus is not None:
        for k, f in vars(model.timestepper.Psi_minus).items():
            out["Psi_minus." + k] = f
    if model.dynamics is not None:
        for k, f in vars(model.dynamics.auxiliaries.fields).items():
            out["dynamics." + k] = f
    if model.mass_fluxes is not None:
        mf = model.mass_fluxes
        out.update({"mass_fluxes.ice": mf.thermodynamics.ice, "mass_fluxes.snow": mf.thermodynamics.snow,
                    "mass_fluxes.intercepted_snowfall": mf.intercepted_snowfall})
    # ice_thermodynamics / snow_thermodynamics (sea_ice_model.jl:422-423): their top_surface_temperature fields (the layered
    # step's outputs, bound as TU / TUS) where the model has them
    for key, name in (("ice_thermodynamics.top_surface_temperature", "ice_top_temperature"),
                      ("snow_thermodynamics.top_surface_temperature", "snow_top_temperature")):
        f = getattr(model, name, None)
        if f is not None:
            out[key] = f
    return {k: f for k, f in out.items() if isinstance(f, Field)}


def prognostic_state(model):
    """Oceananigans.prognostic_state(model) (sea_ice_model.jl:414-426): host copies of the whole parent arrays (halos included)
    plus the clock.  The library keeps no state of its own between calls -- pointers, scratch that every sub-cycle rebuilds --
    so this is all a checkpoint needs (tests/test_gpu_steps.py::test_checkpoint_round_trip_bitwise)."""
    # (a tiled model: the collective check -- csi_sync on every rank + the halo transport's status reduced over ALL ranks -- so that a
    #  checkpoint is never written from an aborted sub-cycle's fields; every rank of the decomposition saves at the same step)
    if isinstance(model.grid, TileGrid):
        model.ctx.validate_all()
    else:
        model.synchronize()
    state = {k: f.numpy().copy() for k, f in _state_fields(model).items()}
    state["clock"] = (model.clock.time, model.clock.iteration)
    return state


def restore_prognostic_state(model, state):
    """Oceananigans.restore_prognostic_state!(model, state) (sea_ice_model.jl:428-445): writes the saved parents back into the
    fields the library is bound to (same arrays: nothing to re-attach)."""
    for k, f in _state_fields(model).items():
        if k in state:
            model.copy_to_field(f, state[k])
    model.clock.time, model.clock.iteration = state["clock"]
    return model


def time_step_momentum(model, dt, rk_reset=False):
    """time_step_momentum!(model, model.dynamics, dt), split_explicit_momentum_equations.jl:103-195."""
    model.ctx.call("csi_time_step_momentum", float(dt), model.substeps, int(rk_reset))


def time_step(model, dt):
    """time_step!(model, dt)."""
    if model.dynamics is None and model.scheme in (0, None):
        # dynamics = nothing, advection = nothing: the step is the thermodynamic update alone
        # (sea_ice_fe_step.jl:13-34 with time_step_momentum!, compute_tendencies! and dynamic_time_step! no-ops)
        if model.ice_thermodynamics is None:
            raise NotImplementedError("a model without dynamics, advection and thermodynamics has nothing to step")
        snow = model.snow_thermodynamics
        sp = model.ice_thermodynamics.params(model.sea_ice_density, snow=snow is not None)
        if snow is None:
            model.ctx.call("csi_slab_thermo_step", C.byref(sp), float(dt))
        else:
            wp = snow.params(model.snow_density, model.snowfall)
            model.ctx.call("csi_layered_thermo_step", C.byref(sp), C.byref(wp), float(dt))
        model.ctx.call("csi_update_state")
    elif model.timestepper_kind == "ForwardEuler":
        model.ctx.call("csi_time_step_fe", float(dt), model.substeps, model.scheme, int(model.clock.iteration == 0))
    else:
        model.ctx.call("csi_time_step_rk3", float(dt), model.substeps, model.scheme)
    model.clock.time += float(dt)
    model.clock.iteration += 1
    # tiles: once per step, the state is checked on EVERY rank before anything between two steps (output, checkpoints, callbacks) reads
    # it -- what the Julia stub does in the update_state! that ends time_step! (julia/ClimaSeaIceHIP.jl)
    if isinstance(model.grid, TileGrid):
        model.ctx.validate_all()
