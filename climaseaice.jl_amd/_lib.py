"""ctypes binding of libcsi_hip.so (include/csi.h).

The HIP library is the product; there is no CPU fallback.  Importing this module never touches
the GPU; `load()` raises loudly when the shared library is missing, and every compute entry
point raises `CsiError` when there is no HIP device.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CSI_HIP_LIBRARY", os.path.join(_HERE, "libcsi_hip.so"))   # override: A/B runs of two builds

# ---- enums (include/csi.h) ---------------------------------------------------------------------
OK = 0
PERIODIC, BOUNDED, FULLY_CONNECTED, LEFT_CONNECTED, RIGHT_CONNECTED, RIGHT_FOLDED, LEFT_CONNECTED_RIGHT_FOLDED = 0, 1, 2, 3, 4, 5, 6
METRIC_UNIFORM, METRIC_PER_J, METRIC_FULL = 0, 1, 2
FIELD_IDS = ["U", "V", "H", "A", "S11", "S22", "S12", "UN", "VN", "P", "ALPHA", "DELTA", "ZETA_F", "ZETA_C",
             "GH", "GA", "HM", "AM", "UM", "VM", "TOP_U", "TOP_V", "BOT_U", "BOT_V", "MASS_FLUX",
             "HS", "GHS", "HSM", "MASS_FLUX_SNOW", "SNOWFALL_INTERCEPTED", "TU", "TUS", "FORCING_U", "FORCING_V"]
F = {n: k for k, n in enumerate(FIELD_IDS)}
STRESS_NONE, STRESS_CONST, STRESS_FIELD, STRESS_SEMI_IMPLICIT = 0, 1, 2, 3
VEL_ZERO, VEL_CONST, VEL_FIELD = 0, 1, 2
STRESS_TOP, STRESS_BOTTOM = 0, 1
MODE_STRICT, MODE_FAST = 0, 1
PRESSURE_REPLACEMENT, PRESSURE_ICE_STRENGTH = 0, 1

# every symbol include/csi.h declares (checked by tests/test_abi.py against the header text)
SYMBOLS = ["csi_version", "csi_context_create", "csi_context_destroy", "csi_last_error", "csi_sync", "csi_set_mode",
           "csi_grid_set", "csi_mask_set", "csi_field_bind", "csi_evp_params_set", "csi_stress_set",
           "csi_evp_initialize", "csi_evp_subcycle", "csi_evp_finalize", "csi_time_step_momentum",
           "csi_compute_tracer_tendencies", "csi_dynamic_step_tracers", "csi_cache_current_fields",
           "csi_update_state", "csi_fill_halo_local", "csi_time_step_fe", "csi_time_step_rk3",
           "csi_slab_thermo_step", "csi_slab_params_set", "csi_layered_thermo_step", "csi_snow_params_set", "csi_tile_set", "csi_comm_unique_id", "csi_comm_init", "csi_comm_count", "csi_local_group_create", "csi_local_group_destroy", "csi_comm_init_local", "csi_comm_init_host", "csi_halo_exchange",
           "csi_plan_exchange", "csi_set_fusion", "csi_set_exchange_interval", "csi_set_halo_transport", "csi_halo_transport", "csi_set_peer_tier", "csi_peer_tier", "csi_plan_ranges", "csi_profile_substeps", "csi_last_path", "csi_last_subcycle_ms", "csi_launches_per_substep", "csi_last_launches", "csi_plan_pair", "csi_plan_peer_chunks", "csi_free_drift_set", "csi_coriolis_rows_set", "csi_velocity_bc_set",
           "csi_immersed_flux_bc_set", "csi_coriolis_points_set", "csi_validate_all", "csi_debug_peer_abort", "csi_set_weno_weight_dtype", "csi_weno_weight_dtype", "csi_subcycle_stats_begin", "csi_subcycle_stats_end",
           "csi_set_tile_skipping", "csi_tile_activity", "csi_set_row_constant", "csi_row_constant_rows"]


class Metrics(C.Structure):
    _fields_ = [("dx", C.c_double), ("dy", C.c_double),
                ("dxc", C.POINTER(C.c_double)), ("dxf", C.POINTER(C.c_double)),
                ("azc", C.POINTER(C.c_double)), ("azf", C.POINTER(C.c_double)),
                ("full", C.POINTER(C.c_double) * 12), ("full_ld", C.c_int64)]


class EvpParams(C.Structure):
    _fields_ = [("ice_compressive_strength", C.c_double), ("ice_compaction_hardening", C.c_double),
                ("yield_curve_eccentricity", C.c_double), ("minimum_plastic_stress", C.c_double),
                ("min_relaxation_parameter", C.c_double), ("max_relaxation_parameter", C.c_double),
                ("relaxation_strength", C.c_double), ("pressure_formulation", C.c_int32), ("has_coriolis", C.c_int32),
                ("coriolis_f", C.c_double), ("minimum_concentration", C.c_double), ("minimum_mass", C.c_double),
                ("sea_ice_density", C.c_double)]


class Stress(C.Structure):
    _fields_ = [("kind", C.c_int32), ("ue_kind", C.c_int32), ("ve_kind", C.c_int32), ("reserved", C.c_int32),
                ("tau_u", C.c_double), ("tau_v", C.c_double), ("ue", C.c_double), ("ve", C.c_double),
                ("rho_e", C.c_double), ("Cd", C.c_double)]


class SlabParams(C.Structure):
    _fields_ = [("conductivity", C.c_double), ("sea_ice_density", C.c_double), ("density", C.c_double),
                ("liquid_density", C.c_double), ("liquid_heat_capacity", C.c_double), ("heat_capacity", C.c_double),
                ("reference_latent_heat", C.c_double), ("reference_temperature", C.c_double),
                ("liquidus_slope", C.c_double), ("freshwater_melting_temperature", C.c_double),
                ("bottom_salinity", C.c_double), ("ice_consolidation_thickness", C.c_double),
                ("top_temperature", C.c_double), ("top_flux_kind", C.c_int32), ("bottom_flux_kind", C.c_int32),
                ("top_heat_flux", C.c_double), ("bottom_heat_flux", C.c_double),
                ("top_bc_kind", C.c_int32), ("pad_", C.c_int32), ("ice_salinity", C.c_double)]


class SnowParams(C.Structure):
    _fields_ = [("conductivity", C.c_double), ("snow_density", C.c_double), ("snowfall", C.c_double),
                ("top_temperature", C.c_double), ("top_bc_kind", C.c_int32), ("pad_", C.c_int32)]


class CsiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libcsi_hip error {code}: {msg}")
        self.code = code


_lib = None


def load():
    """Load libcsi_hip.so; fail loudly if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(make -C climaseaice.jl_amd/csrc).  There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    L.csi_version.restype = i32
    L.csi_last_error.restype = C.c_char_p
    L.csi_last_error.argtypes = [vp]
    sig = {
        "csi_context_create": [i32, vp, C.POINTER(vp)],
        "csi_context_destroy": [vp], "csi_sync": [vp], "csi_set_mode": [vp, i32],
        "csi_grid_set": [vp, i32, i32, i32, i32, i32, i32, i32, C.POINTER(Metrics)],
        "csi_mask_set": [vp, vp, i64],
        "csi_field_bind": [vp, i32, vp, i64, i32, i32],
        "csi_evp_params_set": [vp, C.POINTER(EvpParams)],
        "csi_stress_set": [vp, i32, C.POINTER(Stress)],
        "csi_evp_initialize": [vp], "csi_evp_subcycle": [vp, dbl, i32, i32], "csi_evp_finalize": [vp],
        "csi_time_step_momentum": [vp, dbl, i32, i32],
        "csi_compute_tracer_tendencies": [vp, i32], "csi_dynamic_step_tracers": [vp, dbl, i32],
        "csi_cache_current_fields": [vp], "csi_update_state": [vp], "csi_fill_halo_local": [vp, i32],
        "csi_time_step_fe": [vp, dbl, i32, i32, i32], "csi_time_step_rk3": [vp, dbl, i32, i32],
        "csi_slab_thermo_step": [vp, C.POINTER(SlabParams), dbl],
        "csi_slab_params_set": [vp, C.POINTER(SlabParams)],
        "csi_layered_thermo_step": [vp, C.POINTER(SlabParams), C.POINTER(SnowParams), dbl],
        "csi_snow_params_set": [vp, C.POINTER(SnowParams)],
        "csi_tile_set": [vp, i32, i32, i32, i32, i32, i32],
        "csi_comm_unique_id": [C.POINTER(C.c_uint8)],
        "csi_comm_init": [vp, i32, i32, C.POINTER(C.c_uint8)],
        "csi_comm_count": [vp, C.POINTER(i32)],
        "csi_local_group_create": [i32, C.POINTER(vp)], "csi_comm_init_local": [vp, vp, i32],
        "csi_comm_init_host": [vp, C.c_char_p, i32, i32],
        "csi_halo_exchange": [vp, C.POINTER(i32), i32, i32],
        "csi_plan_ranges": [i32, i32, i32, i32, i32, i32, i32, C.POINTER(i32)],
        "csi_plan_pair": [i32, i32, i32, i32, i32, i32, i32, i32, C.POINTER(i32)],
        "csi_plan_peer_chunks": [i32, i32, i32, i32, i32, i32, i32, C.POINTER(i32), C.POINTER(i32), i32],
        "csi_set_exchange_interval": [vp, i32],
        "csi_set_halo_transport": [vp, i32], "csi_halo_transport": [vp, C.POINTER(i32)],
        "csi_set_peer_tier": [vp, i32], "csi_peer_tier": [vp, C.POINTER(i32)],
        "csi_set_fusion": [vp, i32], "csi_free_drift_set": [vp, i32],
        "csi_set_tile_skipping": [vp, i32], "csi_tile_activity": [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)],
        "csi_set_row_constant": [vp, i32, dbl], "csi_row_constant_rows": [vp, C.POINTER(i32)],
        "csi_coriolis_rows_set": [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), i32],
        "csi_velocity_bc_set": [vp, i32, i32, i32, dbl],
        "csi_coriolis_points_set": [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), i64],
        "csi_immersed_flux_bc_set": [vp, i32, dbl, dbl, dbl, dbl],
        "csi_plan_exchange": [i32] * 14 + [C.POINTER(i32)],
        "csi_profile_substeps": [vp, dbl, i32, C.POINTER(dbl)],
        "csi_last_path": [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)],
        "csi_last_subcycle_ms": [vp, C.POINTER(dbl)], "csi_launches_per_substep": [vp, C.POINTER(i32)],
        "csi_last_launches": [vp, C.POINTER(i32), C.POINTER(i32)],
        "csi_validate_all": [vp], "csi_debug_peer_abort": [vp], "csi_set_weno_weight_dtype": [vp, i32], "csi_weno_weight_dtype": [vp, C.POINTER(i32)], "csi_subcycle_stats_begin": [vp],
        "csi_subcycle_stats_end": [vp, C.POINTER(dbl), C.POINTER(i32), C.POINTER(i32)],
    }
    for name, args in sig.items():
        fn = getattr(L, name, None)
        if fn is None and "CSI_HIP_LIBRARY" in os.environ:
            continue                      # an older build in an A/B run (scripts/ab_libs.sh): entry points added since are absent
        if fn is None:
            raise AttributeError(f"libcsi_hip.so does not export {name}: rebuild it (python -c 'import __graft_entry__ as g; g.build()')")
        fn.restype = i32
        fn.argtypes = args
    L.csi_local_group_destroy.restype = None
    L.csi_local_group_destroy.argtypes = [vp]
    _lib = L
    return L


def plan_ranges(Nx, Ny, Hx, Hy, topo_x, topo_y, valid_width=2):
    """(stress, u-first, v-first, second-velocity) index ranges of the launch loop (pure host function)."""
    out = (C.c_int32 * 16)()
    rc = load().csi_plan_ranges(Nx, Ny, Hx, Hy, topo_x, topo_y, valid_width, out)
    if rc != OK:
        raise CsiError(rc, "csi_plan_ranges")
    v = list(out)
    return tuple(tuple(v[4 * k:4 * k + 4]) for k in range(4))


def plan_pair(Nx, Ny, Hx, Hy, topo_x, topo_y, k=1, m=0):
    """csi_plan_pair as a dict (None when the two-sub-steps-per-launch kernel does not apply)."""
    L = load()
    out = (C.c_int32 * 32)()
    rc = L.csi_plan_pair(Nx, Ny, Hx, Hy, topo_x, topo_y, k, m, out)
    if rc != OK:
        raise CsiError(rc, "csi_plan_pair")
    if not out[0]:
        return None
    r = lambda q: tuple(out[4 + 4 * q: 8 + 4 * q])
    return dict(nstrips=out[1], nchunks=out[2], rows=out[3], first_compute=r(0), second_compute=r(1), store_sigma=r(2),
                store_first_u=r(3), store_first_v=r(4), store_second=r(5), walls=bool(out[28]))


def plan_peer_chunks(Nx, Ny, Hx, Hy, peer_south=True, peer_north=True):
    """csi_plan_peer_chunks as a dict: the chunk layout of a pair launch on the peer transport (None: the pair kernel does not apply)."""
    out = (C.c_int32 * 8)()
    rows = (C.c_int32 * 4096)()
    rc = load().csi_plan_peer_chunks(Nx, Ny, Hx, Hy, int(peer_south), int(peer_north), 256, out, rows, 2048)
    if rc != OK:
        raise CsiError(rc, "csi_plan_peer_chunks")
    if not out[0]:
        return None
    n = out[2]
    return dict(nstrips=out[1], nchunks=n, rows=out[3], elo=out[4], ehi=out[5], nS=out[6], nN=out[7],
                chunks=[(rows[2 * q], rows[2 * q + 1]) for q in range(min(n, 2048))])


def plan_exchange(Nx, Ny, Hx, Hy, topo_x, topo_y, rx, ry, Rx, Ry, periodic_x, periodic_y, width, halo):
    """Eight (peer, i0, j0, ni, nj) entries of the library's exchange plan (pure host function)."""
    out = (C.c_int32 * 40)()
    rc = load().csi_plan_exchange(Nx, Ny, Hx, Hy, topo_x, topo_y, rx, ry, Rx, Ry, int(periodic_x), int(periodic_y), width, int(halo), out)
    if rc != OK:
        raise CsiError(rc, "csi_plan_exchange")
    v = list(out)
    return [tuple(v[5 * k:5 * k + 5]) for k in range(8)]


class LocalGroup:
    """csi_local_group: the tiles of one process (one thread each) exchange halos through device copies instead of RCCL
    (include/csi.h).  Pass it to TileGrid(..., local_group=...); keep it alive as long as its models."""

    def __init__(self, world_size):
        self.L = load()
        self.h = C.c_void_p()
        rc = self.L.csi_local_group_create(int(world_size), C.byref(self.h))
        if rc != OK:
            raise CsiError(rc, "csi_local_group_create")
        self.world_size = int(world_size)

    def close(self):
        if self.h:
            self.L.csi_local_group_destroy(self.h)
            self.h = C.c_void_p()


class Context:
    """One csi_context (one GPU).  Thin: every method is one ABI call that raises on failure."""

    def __init__(self, device_id=0, stream=None):
        self.L = load()
        self.h = C.c_void_p()
        rc = self.L.csi_context_create(device_id, C.c_void_p(stream) if stream else None, C.byref(self.h))
        if rc != OK:
            raise CsiError(rc, self.L.csi_last_error(None).decode())

    def _ck(self, rc):
        if rc != OK:
            raise CsiError(rc, self.L.csi_last_error(self.h).decode())

    def close(self):
        if self.h:
            self.L.csi_context_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def call(self, name, *args):
        self._ck(getattr(self.L, name)(self.h, *args))

    def last_subcycle_ms(self):
        v = C.c_double()
        self.call("csi_last_subcycle_ms", C.byref(v))
        return v.value

    def subcycle_stats_begin(self):
        self.call("csi_subcycle_stats_begin")

    def subcycle_stats_end(self):
        """(total device ms, sub-cycles, kernel launches) of every sub-cycle since subcycle_stats_begin (synchronises)."""
        t, n, l = C.c_double(), C.c_int32(), C.c_int32()
        self.call("csi_subcycle_stats_end", C.byref(t), C.byref(n), C.byref(l))
        return t.value, n.value, l.value

    def validate_all(self):
        """csi_sync + the halo transport's status over ALL ranks (collective)."""
        self.call("csi_validate_all")

    def profile_substeps(self, dt, substeps=16):
        out = (C.c_double * 4)()
        self.call("csi_profile_substeps", float(dt), int(substeps), out)
        return dict(stress=out[0], ustep=out[1], vstep=out[2], exchange=out[3])

    def last_path(self):
        f, k, n = C.c_int32(), C.c_int32(), C.c_int32()
        self.call("csi_last_path", C.byref(f), C.byref(k), C.byref(n))
        return dict(fused=bool(f.value), level=f.value, exchange_interval=k.value, exchanges=n.value)

    def comm_count(self):
        """ranks of the RCCL communicator (ncclCommCount); 0 without one"""
        v = C.c_int32()
        self.call("csi_comm_count", C.byref(v))
        return v.value

    def halo_transport(self):
        """"peer" / "rccl": what the last sub-cycle moved its halos with (csi_halo_transport)"""
        v = C.c_int32()
        self.call("csi_halo_transport", C.byref(v))
        return "peer" if v.value == 1 else "rccl"

    def peer_tier(self):
        """protocol tier of the peer halo transport (csi_peer_tier): 0 write-through + flags, 1 + acquire fence, 2 + release fence"""
        v = C.c_int32()
        self.call("csi_peer_tier", C.byref(v))
        return v.value

    def last_launches(self):
        """(kernel launches, sub-steps) of the last fused sub-cycle."""
        a, b = C.c_int32(), C.c_int32()
        self.call("csi_last_launches", C.byref(a), C.byref(b))
        return a.value, b.value

    def launches_per_substep(self):
        v = C.c_int32()
        self.call("csi_launches_per_substep", C.byref(v))
        return v.value
