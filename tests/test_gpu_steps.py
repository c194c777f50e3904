"""GPU parity of the phases around the sub-cycle and of whole model steps against the oracle:
WENO / upwind tracer tendencies, tracer update (FE and RK3 forms), immersed masks, FE and RK3
time_step!, bare-ice slab thermodynamics."""
import ctypes as C

import numpy as np
import pytest

import cases
import climaseaice_jl_amd as csi
import oracle as O
from test_gpu_evp import cmp_region

pytestmark = pytest.mark.gpu


def anticyclone_case(N, H=4, **kw):
    """Config 2 (BASELINE.json): periodic N^2, 1 km spacing, prescribed cyclonic eddy, h0 sinusoid, aice < 1."""
    c = cases.make_case(Nx=N, Ny=N, H=H, spacing=1000.0, topo=("periodic", "periodic"), patches=False, noise=0.0, **kw)
    L = N * 1000.0
    g = c["g"]
    xf, yc = g.xnodes(csi.Face), g.ynodes(csi.Center)
    xc, yf = g.xnodes(csi.Center), g.ynodes(csi.Face)
    V = 0.5
    c["u"] = np.broadcast_to(V * np.sin(2 * np.pi * yc / L)[:, None], (N, N)) * np.cos(2 * np.pi * xf / L)[None, :]
    c["v"] = -np.broadcast_to(V * np.sin(2 * np.pi * xc / L)[None, :], (N, N)) * np.cos(2 * np.pi * yf / L)[:, None]
    X, Y = np.meshgrid(xc, yc)
    c["h"] = 0.3 + 0.005 * (np.sin(60 * X / 1000e3) + np.sin(30 * Y / 1000e3)) + 0.2 * np.exp(-((X - L / 3) ** 2 + (Y - L / 2) ** 2) / (L / 20) ** 2)
    rng = np.random.default_rng(2)
    c["a"] = np.clip(1 - 0.1 * rng.random((N, N)), 0, 1)
    c["a"][N // 4:N // 3, N // 4:N // 3] = 0.0      # open water: sharp edges exercise the WENO weights
    c["h"][N // 4:N // 3, N // 4:N // 3] = 0.0
    return c


# FAST-mode advection (csrc/advect.hip): the same reconstructions with reciprocals and contraction.  Stated tolerances: h and aice
# within 1e-13 relative after an update (the quantities the north star names); the TENDENCIES within 1e-12 of max|G| -- round 5
# contracts the smoothness indicators too (they were 60 % of a reconstruction's FP64 instructions), which moves the nonlinear weights
# by ~1e-9 relative and the tendencies by ~7e-13 of max|G| (rounds 3 - 4: 1e-13 with uncontracted indicators); dt |dG| stays four
# orders inside the tolerance on h.  STRICT is bit-identical to the oracle.
ADV_TOL = 1e-13
G_TOL = 1e-12


def same_tendency(mode, got, want, what):
    assert np.all(np.isfinite(got)), what
    if mode == "strict":
        assert np.array_equal(got, want), (what, np.abs(got - want).max(), np.argwhere(got != want)[:4])
    else:
        is_g = (what if isinstance(what, str) else what[0]).startswith("G")
        tol = G_TOL if is_g else ADV_TOL
        assert np.abs(got - want).max() <= tol * np.abs(want).max(), (what, np.abs(got - want).max() / np.abs(want).max())
        assert np.array_equal(got == 0.0, want == 0.0), (what, "zero set")


@pytest.mark.parametrize("mode", ["strict", "fast"])
@pytest.mark.parametrize("scheme", [7, 5, -5, 3, -3, 1])
@pytest.mark.parametrize("N", [48, 512])
def test_tracer_tendencies_and_update_bitwise(scheme, N, mode, oracle_lib):
    c = anticyclone_case(N)
    p = cases.oracle_problem(c)
    m = cases.csi_model(c, mode=mode, timestepper="SplitRungeKutta3")
    p.compute_tracer_tendencies(scheme)
    m.ctx.call("csi_compute_tracer_tendencies", scheme)
    m.synchronize()
    for k, f in (("Gh", m.timestepper.Gn.h), ("Ga", m.timestepper.Gn.aice)):
        got, want = f.interior_numpy(), p.interior(k)
        assert np.abs(want).max() > 0
        same_tendency(mode, got, want, k)
    # tracer update, FE form (in place) then RK form (from Psi^-)
    p.L.ora_dynamic_step_tracers(p.ptr, 120.0, 0)
    m.ctx.call("csi_dynamic_step_tracers", 120.0, 0)
    m.synchronize()
    same_tendency(mode, m.ice_thickness.interior_numpy(), p.interior("h"), "h")
    same_tendency(mode, m.ice_concentration.interior_numpy(), p.interior("aice"), "aice")
    if mode == "fast":          # continue from the oracle's state: the update itself is the same code in both modes
        m.ice_thickness.set(p.interior("h").copy()); m.ice_concentration.set(p.interior("aice").copy())
    p.f["hm"][...] = p.f["h"]; p.f["am"][...] = p.f["aice"]
    m.ctx.call("csi_cache_current_fields")
    p.L.ora_dynamic_step_tracers(p.ptr, 40.0, 1)
    m.ctx.call("csi_dynamic_step_tracers", 40.0, 1)
    m.synchronize()
    same_tendency(mode, m.ice_thickness.interior_numpy(), p.interior("h"), "h (RK)")
    same_tendency(mode, m.ice_concentration.interior_numpy(), p.interior("aice"), "aice (RK)")
    # ridging / clipping happened somewhere and respected the bounds
    a = m.ice_concentration.interior_numpy()
    assert a.min() >= 0.0 and a.max() <= 1.0


@pytest.mark.parametrize("mode", ["strict", "fast"])
@pytest.mark.parametrize("scheme", [7, 5, -5])
@pytest.mark.parametrize("topo", [("bounded", "bounded"), ("periodic", "bounded"), ("bounded", "periodic")])
def test_tracer_tendencies_next_to_walls_bitwise(topo, scheme, mode, oracle_lib):
    """Boundary-order reduction of the high-order reconstructions next to walls: HIP kernel == oracle bit for bit,
    on a grid with several LDS tiles in each direction."""
    rng = np.random.default_rng(9)
    c = cases.make_case(Nx=150, Ny=21, H=4, topo=topo, spacing=1000.0, patches=False, noise=0.0)
    c["u"] = 0.4 * rng.standard_normal(c["u"].shape)
    c["v"] = 0.4 * rng.standard_normal(c["v"].shape)
    if topo[0] == "bounded":
        c["u"][:, 0] = 0.0; c["u"][:, -1] = 0.0
    if topo[1] == "bounded":
        c["v"][0, :] = 0.0; c["v"][-1, :] = 0.0
    c["h"] = 0.3 + 0.2 * rng.random(c["h"].shape)
    c["a"] = np.clip(0.5 + 0.6 * rng.random(c["a"].shape), 0, 1)
    p = cases.oracle_problem(c)
    m = cases.csi_model(c, mode=mode)
    p.compute_tracer_tendencies(scheme)
    m.ctx.call("csi_compute_tracer_tendencies", scheme)
    m.synchronize()
    for k, f in (("Gh", m.timestepper.Gn.h), ("Ga", m.timestepper.Gn.aice)):
        got, want = f.interior_numpy(), p.interior(k)
        assert np.abs(want).max() > 0
        same_tendency(mode, got, want, k)
        assert abs(got.sum()) <= 1e-9 * np.abs(got).sum()           # closed walls: flux form conserves


@pytest.mark.parametrize("mode", ["strict", "fast"])
@pytest.mark.parametrize("scheme", [7, 5, -5, 3, -3, 1])
@pytest.mark.parametrize("topo", [("periodic", "periodic"), ("periodic", "bounded"), ("bounded", "bounded")])
def test_tracer_tendencies_next_to_immersed_cells_bitwise(topo, scheme, mode, oracle_lib):
    """ImmersedBoundaryGrid: closed faces next to land and the order reduction of the reconstructions around immersed
    cells (and walls, which the immersed rule covers): HIP kernel == oracle bit for bit; land cells poisoned with 1e300
    never reach a wet cell's tendency."""
    rng = np.random.default_rng(19)
    c = cases.make_case(Nx=150, Ny=45, H=4, topo=topo, spacing=1000.0, patches=False, noise=0.0, land=0.3)
    wet = c["mask"].astype(bool)
    c["u"] = 0.4 * rng.standard_normal(c["u"].shape)
    c["v"] = 0.4 * rng.standard_normal(c["v"].shape)
    if topo[0] == "bounded":
        c["u"][:, 0] = 0.0; c["u"][:, -1] = 0.0
    if topo[1] == "bounded":
        c["v"][0, :] = 0.0; c["v"][-1, :] = 0.0
    c["h"] = np.where(wet, 0.3 + 0.2 * rng.random(wet.shape), 0.0)
    c["a"] = np.where(wet, np.clip(0.5 + 0.6 * rng.random(wet.shape), 0, 1), 0.0)
    p = cases.oracle_problem(c)
    m = cases.csi_model(c, mode=mode)
    p.compute_tracer_tendencies(scheme)
    m.ctx.call("csi_compute_tracer_tendencies", scheme)
    m.synchronize()
    first = {}
    for k, f in (("Gh", m.timestepper.Gn.h), ("Ga", m.timestepper.Gn.aice)):
        got, want = f.interior_numpy(), p.interior(k)
        assert np.abs(want).max() > 0
        same_tendency(mode, got, want, k)
        assert np.all(got[~wet] == 0.0)
        assert abs(got.sum()) <= 1e-9 * np.abs(got).sum()
        first[k] = got.copy()
    # poison the land (interior and halo images), recompute: wet cells unchanged
    for fld in (m.ice_thickness, m.ice_concentration):
        a = fld.interior_numpy().copy()
        a[~wet] = 1e300
        fld.set(a)
        m.ctx.call("csi_fill_halo_local", csi._lib.F["H" if fld is m.ice_thickness else "A"])
    m.ctx.call("csi_compute_tracer_tendencies", scheme)
    m.synchronize()
    for k, f in (("Gh", m.timestepper.Gn.h), ("Ga", m.timestepper.Gn.aice)):
        assert np.array_equal(f.interior_numpy()[wet], first[k][wet]), k


@pytest.mark.parametrize("stepper", ["ForwardEuler", "SplitRungeKutta3"])
@pytest.mark.parametrize("name", ["periodic", "latlon_channel_masked", "folded_masked"])
def test_curvilinear_grid_full_step_bitwise(stepper, name, oracle_lib):
    """Orthogonal curvilinear grid (twelve distorted 2-D metric arrays, CSI_METRIC_FULL): tracer tendencies and whole
    time steps (WENO7 + EVP sub-cycle + tracer update) equal the oracle -- tendencies bit for bit, the step to the
    rounding of exp() in the ice strength (STRICT) / to the FAST tolerance (per-point stencil coefficients,
    three-kernel path)."""
    kw = dict(periodic=dict(Nx=80, Ny=36, topo=("periodic", "periodic"), patches=True, random_uv=0.03, curvilinear=0.05),
              latlon_channel_masked=dict(Nx=70, Ny=40, topo=("periodic", "bounded"), grid="latlon", patches=True, random_uv=0.03,
                                         curvilinear=0.04, land=0.2),
              # TripolarGrid-like: north fold (Zipper) -- WENO stencils, stresses and velocities read across the fold
              folded_masked=dict(Nx=64, Ny=44, topo=("periodic", "folded"), patches=True, random_uv=0.03, curvilinear=0.04, land=0.2))[name]
    c = cases.make_case(substeps=10, **kw)
    p = cases.oracle_problem(c)
    m = cases.csi_model(c, mode="strict", timestepper=stepper, advection=csi.WENO(order=7))
    p.compute_tracer_tendencies(7)
    m.ctx.call("csi_compute_tracer_tendencies", 7)
    m.synchronize()
    for k, f in (("Gh", m.timestepper.Gn.h), ("Ga", m.timestepper.Gn.aice)):
        assert np.array_equal(f.interior_numpy(), p.interior(k)), k
    out = {}
    for mode in ("strict", "fast"):
        p = cases.oracle_problem(c)
        m = cases.csi_model(c, mode=mode, timestepper=stepper, advection=csi.WENO(order=7))
        for n in range(2):
            if stepper == "ForwardEuler":
                p.time_step_fe(c["dt"], 7, n == 0)
            else:
                p.time_step_rk3(c["dt"], 7)
            csi.time_step(m, c["dt"])
        m.synchronize()
        # FAST: the two-sub-steps kernel streams the per-point coefficient planes (a north fold runs the three kernels)
        assert m.ctx.last_path()["level"] == (2 if mode == "fast" else 0)       # (the fold: pair kernel below a three-kernel band)
        vmax = max(np.abs(p.f["u"]).max(), np.abs(p.f["v"]).max())
        tol = 1e-12 if mode == "strict" else 1e-11
        for k, f in (("u", m.velocities.u), ("v", m.velocities.v)):
            assert np.abs(f.numpy() - p.f[k]).max() <= tol * vmax, (mode, k, np.abs(f.numpy() - p.f[k]).max() / vmax)
        for k, f in (("h", m.ice_thickness), ("aice", m.ice_concentration)):
            assert np.abs(f.numpy() - p.f[k]).max() <= tol * np.abs(p.f[k]).max(), (mode, k)
        out[mode] = {k: f.numpy().copy() for k, f in (("u", m.velocities.u), ("h", m.ice_thickness))}
    assert not np.array_equal(out["strict"]["u"], out["fast"]["u"])       # FAST really ran its own kernels


@pytest.mark.parametrize("mode", ["strict", "fast"])
@pytest.mark.parametrize("stepper", ["ForwardEuler", "SplitRungeKutta3"])
def test_advection_only_time_step_bitwise(stepper, mode, oracle_lib):
    """BASELINE config 2: dynamics = nothing, prescribed velocities, time_step! = tendencies + tracer update +
    update_state! (time_step_momentum!(model, ::Nothing, dt) is a no-op, SeaIceDynamics.jl:40); FE and the RK3 stage
    loop with the Psi^- cache.  Five steps, h and aice bit for bit."""
    c = anticyclone_case(96)
    p = cases.oracle_problem(c)
    g = c["g"]
    m = csi.SeaIceModel(g, dynamics=None, advection=csi.WENO(order=7), timestepper=stepper, mode=mode)
    csi.set_(m, h=c["h"], aice=c["a"], u=c["u"], v=c["v"])
    dt = 120.0
    for n in range(5):
        if stepper == "ForwardEuler":
            if n == 0:
                p.update_state()
            p.compute_tracer_tendencies(7)
            p.dynamic_step_tracers(dt, False)
            p.update_state()
        else:
            p.f["hm"][...] = p.f["h"]; p.f["am"][...] = p.f["aice"]
            for beta in (3, 2, 1):
                p.compute_tracer_tendencies(7)
                p.dynamic_step_tracers(dt / beta, True)
                p.update_state()
        csi.time_step(m, dt)
    m.synchronize()
    if mode == "strict":
        assert np.array_equal(m.ice_thickness.numpy(), p.f["h"])
        assert np.array_equal(m.ice_concentration.numpy(), p.f["aice"])
    else:                       # five steps of FAST advection: 1e-13 relative per step (measured ~1e-16)
        assert np.abs(m.ice_thickness.numpy() - p.f["h"]).max() <= 5 * ADV_TOL * np.abs(p.f["h"]).max()
        assert np.abs(m.ice_concentration.numpy() - p.f["aice"]).max() <= 5 * ADV_TOL
        assert np.array_equal(m.ice_thickness.numpy() == 0.0, p.f["h"] == 0.0)
    assert np.abs(p.interior("h") - c["h"]).max() > 1e-4


@pytest.mark.parametrize("mode", ["strict", "fast"])
@pytest.mark.parametrize("topo", [("periodic", "periodic"), ("bounded", "bounded"), ("periodic", "bounded")])
@pytest.mark.parametrize("adv", ["WENO7", "WENO5", "WENO3", "Upwind5", "Upwind3", "Upwind1"])
def test_rk3_advection_only_one_launch_per_stage_bitwise(adv, topo, mode):
    """An RK3 step of an advection-only model runs ONE launch per stage (tendencies + tracer update into rotating copies of
    h, aice; csi_abi.hip rk3_advection_only).  Against the separate kernels (csi_set_fusion(0)): h, aice, Psi^- and the
    tendencies, whole parents with halos, bit for bit after four steps."""
    scheme = {"WENO7": csi.WENO(order=7), "WENO5": csi.WENO(order=5), "WENO3": csi.WENO(order=3), "Upwind5": csi.UpwindBiased(order=5),
              "Upwind3": csi.UpwindBiased(order=3), "Upwind1": csi.UpwindBiased(order=1)}[adv]
    c = cases.make_case(Nx=100, Ny=72, H=4, topo=topo, patches=True, random_uv=0.3)
    out = {}
    for fusion in (0, 2):
        m = csi.SeaIceModel(c["g"], dynamics=None, advection=scheme, timestepper="SplitRungeKutta3", mode=mode)
        m.set_fusion(fusion)
        csi.set_(m, h=c["h"], aice=c["a"], u=c["u"], v=c["v"])
        for _ in range(4):
            csi.time_step(m, 300.0)
        m.synchronize()
        ts = m.timestepper
        out[fusion] = {"h": m.ice_thickness.numpy().copy(), "a": m.ice_concentration.numpy().copy(),
                       "hm": ts.Psi_minus.h.numpy().copy(), "am": ts.Psi_minus.aice.numpy().copy(),
                       "Gh": ts.Gn.h.interior_numpy().copy(), "Ga": ts.Gn.aice.interior_numpy().copy(),
                       "hi": m.ice_thickness.interior_numpy().copy()}
    assert np.abs(out[0]["hi"] - c["h"]).max() > 1e-5          # the advection did something
    for k in out[0]:
        assert np.array_equal(out[0][k], out[2][k]), (k, np.abs(out[0][k] - out[2][k]).max(), np.argwhere(out[0][k] != out[2][k])[:4])


def test_advection_conserves_volume_at_full_size():
    """Config 2 at 512^2: flux-form divergence on a periodic grid conserves sum(h) and sum(aice) to rounding
    (a size-independent property, checked without the oracle)."""
    c = anticyclone_case(512)
    m = cases.csi_model(c, mode="fast")
    m.ctx.call("csi_compute_tracer_tendencies", 7)
    m.synchronize()
    Gh = m.timestepper.Gn.h.interior_numpy()
    assert abs(Gh.sum()) <= 1e-9 * np.abs(Gh).sum()


STEP_CASES = {
    "periodic": dict(Nx=48, Ny=40, topo=("periodic", "periodic"), patches=True, random_uv=0.02),
    "masked": dict(Nx=64, Ny=48, topo=("periodic", "periodic"), patches=True, random_uv=0.02, land=0.3),
    "masked_channel": dict(Nx=48, Ny=48, topo=("periodic", "bounded"), patches=False, random_uv=0.02, land=0.25),
}


@pytest.mark.parametrize("name", ["masked", "masked_channel"])
def test_masked_subcycle_strict_bitwise_and_fast(name, oracle_lib):
    """Immersed land mask: peripheral-node masking of u, v (split_explicit:226,261), masked stresses in the
    divergence (ice_stress_divergence.jl:21-24), mask_immersed_field_xy! in update_state!."""
    c = cases.make_case(substeps=8, **STEP_CASES[name])
    p = cases.oracle_problem(c)
    ms = cases.csi_model(c, mode="strict")
    mf = cases.csi_model(c, mode="fast")
    for m in (ms, mf):
        assert np.array_equal(m.velocities.u.numpy(), p.f["u"]) and np.array_equal(m.velocities.v.numpy(), p.f["v"])
    p.initialize_rheology()
    for m in (ms, mf):
        m.ctx.call("csi_evp_initialize")
        m.copy_to_field(m.dynamics.auxiliaries.fields.P, p.f["P"])
    p.L.ora_fill_halo_u(p.ptr); p.L.ora_fill_halo_v(p.ptr)
    p.subcycle(c["dt"], 1, 8)
    p.L.ora_finalize_rheology(p.ptr)            # sigma halos: the fused FAST kernels only define them after the fill
    for m in (ms, mf):
        m.ctx.call("csi_evp_subcycle", c["dt"], 8, 1)
        m.ctx.call("csi_evp_finalize")
        m.synchronize()
    vmax = max(np.abs(p.f["u"]).max(), np.abs(p.f["v"]).max())
    for k, get in (("u", lambda m: m.velocities.u), ("v", lambda m: m.velocities.v),
                   ("s11", lambda m: m.dynamics.auxiliaries.fields.s11), ("s12", lambda m: m.dynamics.auxiliaries.fields.s12)):
        assert np.array_equal(get(ms).numpy(), p.f[k]), k
        scale = vmax if k in ("u", "v") else np.abs(p.f[k]).max()
        assert np.abs(cmp_region(c, k, get(mf).numpy()) - cmp_region(c, k, p.f[k])).max() <= 1e-11 * scale, k
    # bit-exact masks: velocities vanish exactly on the peripheral nodes of the immersed grid
    wet = c["mask"]
    land_u = ~wet | ~np.roll(wet, 1, axis=1)
    assert np.all(mf.velocities.u.interior_numpy()[:, :wet.shape[1]][land_u] == 0.0)


@pytest.mark.parametrize("stepper", ["ForwardEuler", "SplitRungeKutta3"])
@pytest.mark.parametrize("name", ["periodic", "masked"])
def test_full_time_step_vs_oracle(stepper, name, oracle_lib):
    """time_step!(model, dt) with WENO(order = 7) advection: FE (sea_ice_fe_step.jl:13-34) and the RK3 stage loop."""
    c = cases.make_case(substeps=12, **STEP_CASES[name])
    p = cases.oracle_problem(c)
    for mode, tol in (("strict", 1e-12), ("fast", 1e-11)):
        p = cases.oracle_problem(c)
        m = cases.csi_model(c, mode=mode, timestepper=stepper, advection=csi.WENO(order=7))
        for n in range(2):
            if stepper == "ForwardEuler":
                p.time_step_fe(c["dt"], 7, n == 0)
            else:
                p.time_step_rk3(c["dt"], 7)
            csi.time_step(m, c["dt"])
        m.synchronize()
        vmax = max(np.abs(p.f["u"]).max(), np.abs(p.f["v"]).max())
        for k, f in (("u", m.velocities.u), ("v", m.velocities.v)):
            assert np.abs(f.numpy() - p.f[k]).max() <= tol * vmax, (mode, k)
        for k, f in (("h", m.ice_thickness), ("aice", m.ice_concentration)):
            assert np.abs(f.numpy() - p.f[k]).max() <= tol * np.abs(p.f[k]).max(), (mode, k)
            assert np.array_equal(f.numpy() == 0.0, p.f[k] == 0.0), (mode, k, "zero set")


def _slab(**kw):
    d = dict(conductivity=2.0, sea_ice_density=900.0, density=917.0, liquid_density=999.8, liquid_heat_capacity=4186.0,
             heat_capacity=2000.0, reference_latent_heat=334e3, reference_temperature=0.0, liquidus_slope=0.054,
             freshwater_melting_temperature=0.0, bottom_salinity=0.0, ice_consolidation_thickness=0.05,
             top_temperature=-10.0, top_flux_kind=1, bottom_flux_kind=0, top_heat_flux=0.0, bottom_heat_flux=0.0)
    d.update(kw)
    return d


def test_slab_thermodynamics_vs_oracle():
    """Config 1 plumbing (freezing_bucket) on a 16x8 grid with per-cell different states, 50 steps; and the
    constant-flux case of test/test_thermodynamic_mass_fluxes.jl:56."""
    g = csi.RectilinearGrid((16, 8), x=(0, 1), y=(0, 1), halo=(3, 3))
    rng = np.random.default_rng(7)
    for params, okw in ((_slab(heat_capacity=2100.0, bottom_flux_kind=1, bottom_heat_flux=1.0),
                         dict(c_ice=2100.0, Tu=-10.0, top_flux_kind=1, bot_flux_kind=1, Qb=1.0)),
                        (_slab(top_flux_kind=0, top_heat_flux=100.0, bottom_heat_flux=10.0, top_temperature=-5.0),
                         dict(Tu=-5.0, top_flux_kind=0, Qu=100.0, Qb=10.0))):
        model = csi.SeaIceModel(g, dynamics=None)
        h0 = rng.random((8, 16)) * (rng.random((8, 16)) > 0.3)
        a0 = np.where(h0 > 0, rng.random((8, 16)), 0.0)
        h0[0, 0], a0[0, 0] = 0.0, 0.0
        csi.set_(model, h=h0, aice=a0)
        mf = csi.CenterField(g, model.device, "mass_flux")
        model.ctx.call("csi_field_bind", csi._lib.F["MASS_FLUX"], C.c_void_p(mf.data.data_ptr()), mf.ni, mf.ni, mf.nj)
        sp = csi._lib.SlabParams(**params)
        h, a = h0.ravel().copy(), a0.ravel().copy()
        for n in range(50):
            h, a, flux = O.slab_step(h, a, 600.0, **okw)
            model.ctx.call("csi_slab_thermo_step", C.byref(sp), 600.0)
        model.synchronize()
        assert np.array_equal(model.ice_thickness.interior_numpy().ravel(), h)
        assert np.array_equal(model.ice_concentration.interior_numpy().ravel(), a)
        assert np.array_equal(mf.interior_numpy().ravel(), flux)


@pytest.mark.parametrize("bc", [0, 1])
def test_layered_snow_step_bitwise(bc):
    """_layered_thermodynamic_time_step! (thermodynamic_time_step.jl:131-298): every cell of a 64 x 32 grid in a
    different regime (open water, thin / consolidated ice, with / without snow), 20 steps of melting, freezing and
    snowfall; the bare-ice step with the MeltingConstrainedFluxBalance top boundary condition likewise.  HIP == oracle
    bit for bit on h, aice, hs, the three mass fluxes and the two surface temperatures."""
    g = csi.RectilinearGrid((64, 32), x=(0, 1), y=(0, 1), halo=(3, 3))
    rng = np.random.default_rng(31 + bc)
    shape = (32, 64)
    h0 = rng.random(shape) * 2.0 * (rng.random(shape) > 0.15)
    a0 = np.where(h0 > 0, rng.random(shape), 0.0)
    hs0 = rng.random(shape) * 0.5 * (rng.random(shape) > 0.4) * (h0 > 0)
    for Qu, Qb, Ps in ((-150.0, 5.0, 2e-5), (80.0, -10.0, 0.0), (0.0, "frazil", 1e-5)):
        ice = csi.SlabThermodynamics(top_temperature=-8.0, top_heat_flux=Qu, bottom_heat_flux=Qb, bottom_salinity=30.0,
                                     top_heat_boundary_condition=csi.MeltingConstrainedFluxBalance() if bc else None)
        snow = csi.snow_slab_thermodynamics(top_heat_boundary_condition=None if bc else csi.PrescribedTemperature(-8.0))
        model = csi.SeaIceModel(g, dynamics=None, ice_thermodynamics=ice, snow_thermodynamics=snow, snowfall=Ps,
                                timestepper="ForwardEuler")
        csi.set_(model, h=h0, aice=a0, hs=hs0)
        slab_o = O.make_slab(top_bc_kind=bc, Tu=-8.0, top_flux_kind=0, Qu=Qu, salinity=30.0,
                             bot_flux_kind=1 if Qb == "frazil" else 0, Qb=1.0 if Qb == "frazil" else Qb)
        snow_o = O.make_snow(top_bc_kind=bc, Tu=-8.0, snowfall=Ps)
        r = dict(h=h0.ravel(), aice=a0.ravel(), hs=hs0.ravel())
        for n in range(20):
            r = O.layered_step(r["h"], r["aice"], r["hs"], 600.0, slab_o, snow_o)
            csi.time_step(model, 600.0)
        model.synchronize()
        got = dict(h=model.ice_thickness, aice=model.ice_concentration, hs=model.snow_thickness,
                   mf_ice=model.mass_fluxes.thermodynamics.ice, mf_snow=model.mass_fluxes.thermodynamics.snow,
                   mf_int=model.mass_fluxes.intercepted_snowfall, tu_ice=model.ice_top_temperature, tu_snow=model.snow_top_temperature)
        for k, f in got.items():
            a = f.interior_numpy().ravel()
            assert np.all(np.isfinite(a)), k
            assert np.array_equal(a, r[k]), (bc, Qu, k, np.abs(a - r[k]).max())
        assert (r["hs"] > 0).any()
    # bare ice with the flux-balance boundary condition
    ice = csi.SlabThermodynamics(top_heat_flux=-60.0, bottom_heat_flux=4.0, bottom_salinity=30.0, ice_salinity=5.0,
                                 top_heat_boundary_condition=csi.MeltingConstrainedFluxBalance())
    model = csi.SeaIceModel(g, dynamics=None, ice_thermodynamics=ice, timestepper="ForwardEuler")
    csi.set_(model, h=h0, aice=a0)
    slab_o = O.make_slab(top_bc_kind=1, top_flux_kind=0, Qu=-60.0, Qb=4.0, salinity=30.0, ice_salinity=5.0)
    h, a = h0.ravel().copy(), a0.ravel().copy()
    mf = np.zeros_like(h)
    for n in range(20):
        O.lib().ora_slab_thermo_step(C.byref(slab_o), h.size, O._dptr(h), O._dptr(a), O._dptr(mf), 600.0)
        csi.time_step(model, 600.0)
    model.synchronize()
    assert np.array_equal(model.ice_thickness.interior_numpy().ravel(), h)
    assert np.array_equal(model.ice_concentration.interior_numpy().ravel(), a)


@pytest.mark.parametrize("stepper", ["ForwardEuler", "SplitRungeKutta3"])
def test_full_time_step_with_snow_vs_oracle(stepper, oracle_lib):
    """time_step! with EVP dynamics, WENO7 advection of h, aice AND hs (tracer_tendency_kernel_functions.jl:49-52,
    sea_ice_fe_step.jl:86-94), the layered thermodynamic step and update_state! on an immersed channel
    (the configuration of test/test_snow_thermodynamics.jl:171-187 with dynamics switched on)."""
    c = cases.make_case(Nx=48, Ny=40, substeps=12, topo=("periodic", "bounded"), patches=True, random_uv=0.02, land=0.2)
    rng = np.random.default_rng(41)
    hs0 = np.where(c["a"] > 0, 0.3 * rng.random(c["a"].shape), 0.0)
    slab_o = O.make_slab(top_bc_kind=1, top_flux_kind=0, Qu=-80.0, Qb=6.0, salinity=30.0)
    snow_o = O.make_snow(snowfall=3e-5)
    for mode, tol in (("strict", 1e-12), ("fast", 1e-11)):
        p = cases.oracle_problem(c)
        p.s.has_snow = 1
        p.interior("hs")[...] = hs0
        p.update_state()
        g = c["g"]
        ice = csi.SlabThermodynamics(top_heat_flux=-80.0, bottom_heat_flux=6.0, bottom_salinity=30.0,
                                     top_heat_boundary_condition=csi.MeltingConstrainedFluxBalance())
        dyn = csi.SeaIceMomentumEquation(g, coriolis=csi.FPlane(f=c["coriolis"]), top_momentum_stress=c["top"],
                                         bottom_momentum_stress=csi.SemiImplicitStress(), solver=csi.SplitExplicitSolver(substeps=12))
        m = csi.SeaIceModel(g, dynamics=dyn, advection=csi.WENO(order=7), ice_thermodynamics=ice,
                            snow_thermodynamics=csi.snow_slab_thermodynamics(), snowfall=3e-5, timestepper=stepper, mode=mode)
        m.set_mask(c["mask"])
        csi.set_(m, h=c["h"], aice=c["a"], u=c["u"], v=c["v"], hs=hs0)
        for n in range(2):
            if stepper == "ForwardEuler":
                p.time_step_fe(c["dt"], 7, n == 0, slab=slab_o, snow=snow_o)
            else:
                p.time_step_rk3(c["dt"], 7, slab=slab_o, snow=snow_o)
            csi.time_step(m, c["dt"])
        m.synchronize()
        vmax = max(np.abs(p.f["u"]).max(), np.abs(p.f["v"]).max())
        for k, f in (("u", m.velocities.u), ("v", m.velocities.v)):
            assert np.abs(f.numpy() - p.f[k]).max() <= tol * vmax, (mode, k)
        for k, f in (("h", m.ice_thickness), ("aice", m.ice_concentration), ("hs", m.snow_thickness)):
            assert np.abs(f.numpy() - p.f[k]).max() <= tol * np.abs(p.f[k]).max(), (mode, k, np.abs(f.numpy() - p.f[k]).max())
            assert np.array_equal(f.numpy() == 0.0, p.f[k] == 0.0), (mode, k, "zero set")
        assert np.abs(p.interior("hs") - hs0).max() > 1e-4            # the snow did something
        # update_state! masks the mass-flux diagnostics on land too (sea_ice_model.jl:387-390)
        land = ~c["mask"].astype(bool)
        for f in (m.mass_fluxes.thermodynamics.ice, m.mass_fluxes.thermodynamics.snow, m.mass_fluxes.intercepted_snowfall):
            assert np.all(f.interior_numpy()[land] == 0.0) and np.abs(f.interior_numpy()[~land]).max() > 0


@pytest.mark.parametrize("snow", [False, True])
@pytest.mark.parametrize("stepper", ["ForwardEuler", "SplitRungeKutta3"])
def test_full_time_step_on_a_tile_equals_untiled(stepper, snow):
    """Whole time_step! on a tile whose periodic sides go through the RCCL exchange (to itself): the halo refresh of
    update_state! (h, aice, [hs,] u, v exchanged with the full halo width), WENO7 advection reading those halos, the
    batched sub-cycle exchange and the thermodynamic step give the untiled run bit for bit on the owned cells."""
    c = cases.make_case(Nx=72, Ny=56, H=8, substeps=12, topo=("periodic", "periodic"), patches=True, random_uv=0.03)
    rng = np.random.default_rng(51)
    hs0 = np.where(c["a"] > 0, 0.2 * rng.random(c["a"].shape), 0.0)
    out = {}
    for tile in (None, (1, 1, 0, True)):
        kw = {}
        if snow:
            kw = dict(ice_thermodynamics=csi.SlabThermodynamics(top_heat_flux=-60.0, bottom_heat_flux=5.0, bottom_salinity=30.0,
                                                                top_heat_boundary_condition=csi.MeltingConstrainedFluxBalance()),
                      snow_thermodynamics=csi.snow_slab_thermodynamics(), snowfall=2e-5)
        m = cases.csi_model(c, mode="fast", timestepper=stepper, advection=csi.WENO(order=7), tile=tile, **kw)
        if snow:
            csi.set_(m, hs=hs0)
        for n in range(3):
            csi.time_step(m, c["dt"])
        m.synchronize()
        out[tile is None] = {k: f.interior_numpy().copy() for k, f in (("u", m.velocities.u), ("v", m.velocities.v), ("h", m.ice_thickness),
                                                                      ("aice", m.ice_concentration))}
        if snow:
            out[tile is None]["hs"] = m.snow_thickness.interior_numpy().copy()
        if tile is not None:
            assert m.ctx.last_path()["exchanges"] > 0
    for k in out[True]:
        assert np.all(np.isfinite(out[False][k])), k
        assert np.array_equal(out[True][k], out[False][k]), (k, np.abs(out[True][k] - out[False][k]).max())
    assert np.abs(out[True]["h"] - c["h"]).max() > 1e-5


@pytest.mark.parametrize("stepper", ["ForwardEuler", "SplitRungeKutta3"])
def test_dynamics_and_thermodynamics_without_advection(stepper, oracle_lib):
    """advection = nothing (the default of SeaIceModel) with dynamics and slab thermodynamics: zero tendencies, the
    tracer update still resets h, aice to Psi^- at every RK stage (so the stage-wise thermodynamic steps do not add up)."""
    c = cases.make_case(Nx=40, Ny=32, substeps=8, topo=("periodic", "bounded"), patches=True, random_uv=0.02)
    slab_o = O.make_slab(Tu=-5.0, top_flux_kind=0, Qu=100.0, Qb=10.0)
    p = cases.oracle_problem(c)
    m = cases.csi_model(c, mode="strict", timestepper=stepper, advection=None,
                        ice_thermodynamics=csi.SlabThermodynamics(top_temperature=-5.0, top_heat_flux=100.0, bottom_heat_flux=10.0))
    h0 = p.interior("h").copy()
    for n in range(2):
        if stepper == "ForwardEuler":
            p.time_step_fe(c["dt"], 0, n == 0, slab=slab_o)
        else:
            p.time_step_rk3(c["dt"], 0, slab=slab_o)
        csi.time_step(m, c["dt"])
    m.synchronize()
    assert np.array_equal(m.ice_thickness.numpy(), p.f["h"]) and np.array_equal(m.ice_concentration.numpy(), p.f["aice"])
    assert np.abs(p.interior("h") - h0).max() > 1e-6


def test_config4_style_latlon_evp_plus_slab_thermodynamics(oracle_lib):
    """BASELINE config 4 in miniature: lat-lon (lon 0..60, lat 20..70) channel, EVP + WENO7 (order reduced next to
    the walls) + bare-ice slab thermodynamics (top 100 W m^-2, bottom 10 W m^-2, test/test_thermodynamic_mass_fluxes.jl:56), RK3, 2 steps."""
    c = cases.make_case(Nx=48, Ny=40, substeps=12, topo=("periodic", "bounded"), grid="latlon", patches=True, random_uv=0.02)
    slab_o = O.make_slab(Tu=-5.0, top_flux_kind=0, Qu=100.0, Qb=10.0)
    for mode, tol in (("strict", 1e-12), ("fast", 1e-11)):
        p = cases.oracle_problem(c)
        thermo = csi.SlabThermodynamics(top_temperature=-5.0, top_heat_flux=100.0, bottom_heat_flux=10.0)
        g = c["g"]
        dyn = csi.SeaIceMomentumEquation(g, coriolis=csi.FPlane(f=c["coriolis"]), top_momentum_stress=c["top"],
                                         bottom_momentum_stress=csi.SemiImplicitStress(), solver=csi.SplitExplicitSolver(substeps=12))
        m = csi.SeaIceModel(g, dynamics=dyn, advection=csi.WENO(order=7), ice_thermodynamics=thermo, mode=mode)
        csi.set_(m, h=c["h"], aice=c["a"], u=c["u"], v=c["v"])
        for _ in range(2):
            p.time_step_rk3(c["dt"], 7, slab=slab_o)
            csi.time_step(m, c["dt"])
        m.synchronize()
        for k, f in (("h", m.ice_thickness), ("aice", m.ice_concentration), ("u", m.velocities.u), ("v", m.velocities.v)):
            scale = max(np.abs(p.f[k]).max(), 1e-30)
            assert np.abs(f.numpy() - p.f[k]).max() <= tol * scale, (mode, k)
        assert np.abs(p.interior("h") - c["h"]).max() > 1e-6      # the slab step really changed the ice


def test_freezing_bucket_full_run_matches_oracle(oracle_lib):
    """examples/freezing_bucket.py (the reference's examples/freezing_bucket.jl: 1440 steps of 10 minutes): thickness
    and concentration after every model day equal the oracle's slab step bit for bit; the first step is the known
    answer of SURVEY.md 8(d) config 1 (tests/test_oracle_properties.py), and the ice ends up consolidated and growing."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("freezing_bucket", os.path.join(os.path.dirname(__file__), "..", "examples", "freezing_bucket.py"))
    fb = importlib.util.module_from_spec(spec); spec.loader.exec_module(fb)
    m = fb.build()
    series = fb.run(m, steps=1440, dt=600.0, every=144)
    h = np.zeros((1, 1)); a = np.zeros((1, 1))
    want = []
    for n in range(1440):
        h, a, _ = O.slab_step(h, a, 600.0, Tu=-10.0, c_ice=2100.0, top_flux_kind=1, bot_flux_kind=1, Qb=1.0)
        if (n + 1) % 144 == 0:
            want.append((float(h[0, 0]), float(a[0, 0])))
    got = [(hh, aa) for _, hh, aa in series]
    assert got == want, (got[:2], want[:2])
    assert 0.99 < got[-1][1] <= 1.0 and got[-1][0] > got[0][0] > 0.05        # consolidated, still thickening


@pytest.mark.parametrize("name", ["periodic_rk3", "masked_channel_fe_slab", "snow_rk3"])
def test_checkpoint_round_trip_bitwise(name):
    """prognostic_state / restore_prognostic_state! (sea_ice_model.jl:414-445) through the library: two steps, checkpoint, two more
    steps -- against a FRESH model (new context, new library scratch) that restores the checkpoint and takes the same two steps.
    Every saved field comes out bit for bit, halos included: the library holds pointers and per-call scratch only, so a
    checkpoint of the Oceananigans-side fields is a complete one."""
    kw = {"periodic_rk3": dict(Nx=96, Ny=64, topo=("periodic", "periodic"), patches=True, random_uv=0.03),
          "masked_channel_fe_slab": dict(Nx=80, Ny=56, topo=("periodic", "bounded"), patches=True, random_uv=0.03, land=0.25, field_forcing=True),
          "snow_rk3": dict(Nx=72, Ny=48, topo=("bounded", "bounded"), grid="latlon", patches=True, random_uv=0.03)}[name]
    c = cases.make_case(substeps=12, **kw)
    stepper = "ForwardEuler" if "fe" in name else "SplitRungeKutta3"
    extra = {}
    if "slab" in name:
        extra["ice_thermodynamics"] = csi.SlabThermodynamics(top_temperature=-10.0, top_heat_flux=100.0, bottom_heat_flux=10.0)
    if "snow" in name:
        extra.update(ice_thermodynamics=csi.SlabThermodynamics(top_heat_boundary_condition=csi.MeltingConstrainedFluxBalance(), top_heat_flux=-70.0,
                                                              bottom_heat_flux=5.0, bottom_salinity=30.0),
                     snow_thermodynamics=csi.snow_slab_thermodynamics(), snowfall=2e-5)

    def build():
        return cases.csi_model(c, mode="fast", timestepper=stepper, advection=csi.WENO(order=7), **extra)

    a = build()
    for _ in range(2):
        csi.time_step(a, c["dt"])
    ckpt = csi.prognostic_state(a)
    for _ in range(2):
        csi.time_step(a, c["dt"])
    want = csi.prognostic_state(a)
    b = build()
    csi.time_step(b, c["dt"])                     # (a model that has already stepped: the restore must overwrite everything that matters)
    csi.restore_prognostic_state(b, ckpt)
    assert b.clock.iteration == 2
    for _ in range(2):
        csi.time_step(b, c["dt"])
    got = csi.prognostic_state(b)
    assert set(got) == set(want) and got["clock"] == want["clock"]
    assert {"u", "v", "h", "aice", "Gn.h", "dynamics.s11", "dynamics.alpha"} <= set(want)
    for k in want:
        if k != "clock":
            assert np.array_equal(want[k], got[k]), (k, np.abs(want[k] - got[k]).max())
    assert not np.array_equal(ckpt["h"], want["h"])


# ---- round 5: WENO weights in single precision (csi_set_weno_weight_dtype; upstream's second float type FT2 = Float32, recalled) ----
@pytest.mark.parametrize("mode", ["strict", "fast"])
@pytest.mark.parametrize("scheme", [7, 5, 3])
@pytest.mark.parametrize("topo", [("periodic", "periodic"), ("bounded", "bounded")])
def test_weno_f32_weights_match_the_oracle(topo, scheme, mode, oracle_lib):
    """weight_dtype f32: STRICT equals the oracle's f32-weight mode bit for bit (same float expressions, two compilers); FAST runs
    the SAME float weights and contracts only the double part: 1e-13 of max|G| like the double mode.  The walls exercise the
    buffer schemes (WENO7 -> 5 -> 3) in the f32 mode too.  A whole RK3 step with the stage launches follows."""
    c = cases.make_case(Nx=96, Ny=80, topo=topo, substeps=2, random_uv=0.3, patches=True)
    p = cases.oracle_problem(c)
    p.s.weno_weights_f32 = 1
    m = cases.csi_model(c, mode=mode, timestepper="SplitRungeKutta3", advection=csi.WENO(order=scheme, weight_dtype="f32"))
    v = C.c_int32()
    m.ctx.call("csi_weno_weight_dtype", C.byref(v))
    assert v.value == 1
    p.compute_tracer_tendencies(scheme)
    m.ctx.call("csi_compute_tracer_tendencies", scheme)
    m.synchronize()
    for k, f in (("Gh", m.timestepper.Gn.h), ("Ga", m.timestepper.Gn.aice)):
        got, want = f.interior_numpy(), p.interior(k)
        assert np.abs(want).max() > 0
        same_tendency(mode, got, want, (k, "f32 weights"))
    # the switch is live: the double mode gives other bits
    m.ctx.call("csi_set_weno_weight_dtype", 0)
    m.ctx.call("csi_compute_tracer_tendencies", scheme)
    m.synchronize()
    assert not np.array_equal(m.timestepper.Gn.h.interior_numpy(), p.interior("Gh"))
    with pytest.raises(csi.CsiError):
        m.ctx.call("csi_set_weno_weight_dtype", 2)


@pytest.mark.parametrize("mode", ["strict", "fast"])
def test_weno_f32_weights_whole_rk3_step_advection_only(mode, oracle_lib):
    """BASELINE config 2's path (advection-only RK3: one launch per stage, k_tendencies<..., STEP>) in the f32-weight mode."""
    c = anticyclone_case(96)
    p = cases.oracle_problem(c)
    p.s.weno_weights_f32 = 1
    m = csi.SeaIceModel(c["g"], dynamics=None, advection=csi.WENO(order=7, weight_dtype="f32"), timestepper="SplitRungeKutta3", mode=mode)
    csi.set_(m, h=c["h"], aice=c["a"], u=c["u"], v=c["v"])
    dt = 120.0
    for n in range(3):
        p.f["hm"][...] = p.f["h"]; p.f["am"][...] = p.f["aice"]
        for beta in (3, 2, 1):
            p.compute_tracer_tendencies(7)
            p.dynamic_step_tracers(dt / beta, True)
            p.update_state()
        csi.time_step(m, dt)
    m.synchronize()
    if mode == "strict":
        assert np.array_equal(m.ice_thickness.numpy(), p.f["h"])
        assert np.array_equal(m.ice_concentration.numpy(), p.f["aice"])
    else:
        assert np.abs(m.ice_thickness.numpy() - p.f["h"]).max() <= 3 * ADV_TOL * np.abs(p.f["h"]).max()
        assert np.abs(m.ice_concentration.numpy() - p.f["aice"]).max() <= 3 * ADV_TOL


# ---- round 5: two tracers per thread in the tendency kernel (large grids) ----------------------------------------------------------
@pytest.mark.parametrize("mode", ["strict", "fast"])
@pytest.mark.parametrize("scheme", [7, 5, -5, 3, 1])
@pytest.mark.parametrize("topo", [("periodic", "periodic"), ("bounded", "bounded"), ("periodic", "bounded")])
def test_two_tracers_per_thread_bit_identical(topo, scheme, mode, oracle_lib, monkeypatch):
    """k_tendencies<..., NT = 2> (h and aice in one thread: large grids) against NT = 1 (one thread per cell and tracer): the same
    operations per value, so the tendencies and a whole advection-only RK3 step (the one-launch-per-stage variant too) are
    bit-identical; the layout is forced with CSI_ADV_NT, which a context reads when it is created.  With land in the bounded case."""
    kw = dict(Nx=130, Ny=75, topo=topo, substeps=2, random_uv=0.3, patches=True)
    if topo == ("bounded", "bounded"):
        kw["land"] = 0.2
    c = cases.make_case(**kw)
    out = {}
    for nt in (1, 2):
        monkeypatch.setenv("CSI_ADV_NT", str(nt))
        adv = csi.WENO(order=scheme) if scheme > 1 else csi.UpwindBiased(order=abs(scheme))
        m = cases.csi_model(c, mode=mode, timestepper="SplitRungeKutta3", advection=adv)
        m.ctx.call("csi_compute_tracer_tendencies", adv.scheme)
        m.synchronize()
        G = (m.timestepper.Gn.h.numpy().copy(), m.timestepper.Gn.aice.numpy().copy())
        m2 = csi.SeaIceModel(c["g"], dynamics=None, advection=adv, timestepper="SplitRungeKutta3", mode=mode)
        if c.get("mask") is not None:
            m2.set_mask(c["mask"])
        csi.set_(m2, h=c["h"], aice=c["a"], u=c["u"], v=c["v"])
        for _ in range(2):
            csi.time_step(m2, 120.0)
        m2.synchronize()
        out[nt] = G + (m2.ice_thickness.numpy().copy(), m2.ice_concentration.numpy().copy())
    for a, b, what in zip(out[1], out[2], ("Gh", "Ga", "h after 2 RK3 steps", "aice after 2 RK3 steps")):
        assert np.array_equal(a, b), (what, np.abs(a - b).max())
    assert np.abs(out[1][0]).max() > 0
