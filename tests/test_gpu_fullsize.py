"""BASELINE.json's configurations at their STATED sizes.  The oracle does not finish
these in seconds, so parity is carried by the chain  oracle == STRICT (bit for bit, small sizes, test_gpu_evp.py)
-> STRICT vs FAST at full size (stated tolerance) -> FAST three-kernel == fused kernels (bit for bit, full size)
-> untiled == tiled (bit for bit, full size), plus size-independent properties (finite, land exactly zero, zero
sets identical).  Reference set-ups: test/distributed_tests_utils.jl:104-137 (config 3), :183-259 (configs 4 / 5
geometry), examples/arctic_basin_seasonal_cycle.jl (config 4 physics).

  config 3: 1024^2 (and the metric's 2048^2) periodic f-plane, EVP, 120 sub-steps
  config 4: 2048^2 lat-lon (lon 0..60, lat 20..70, per-row metrics), EVP + slab thermodynamics, one RK3 time_step!
            with WENO7; also on a self-connected tile with bench.py's halo 32 / k = 16 and with k = 1
  config 5: 4096^2 channel, 30 % land discs + solid caps, 500 sub-steps

Tolerances (fp64): FAST vs STRICT after a 120-sub-step cycle <= 1e-12 max|u| on u, v and <= 1e-11 max|sigma| on
sigma, or 10 x STRICT's own sensitivity to a 1e-15 relative input perturbation where the mEVP iteration is
ill-conditioned (which branch was taken is recorded in gpurun_out/fullsize_tolerance.json and printed).
"""
import json
import os

import numpy as np
import pytest

import cases
import climaseaice_jl_amd as csi
from test_gpu_evp import EVP_FIELDS, FAST_TOL_SIG, FAST_TOL_VEL

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_REPORT = {}


def _record(name, entry):
    _REPORT[name] = entry
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "fullsize_tolerance.json"), "w") as f:
            json.dump(_REPORT, f, indent=1, sort_keys=True)
    print(f"[fullsize] {name}: {entry}")


def run_cycle(c, mode, fusion=None, tile=None, k=None, transport=None):
    m = cases.csi_model(c, mode=mode, tile=tile)
    if transport is not None:
        m.set_halo_transport(transport)
    if fusion is not None:
        m.set_fusion(fusion)
    if k is not None:
        m.set_exchange_interval(k)
    csi.time_step_momentum(m, c["dt"])
    m.synchronize()
    out = {f: EVP_FIELDS[f](m).interior_numpy().copy() for f in ("u", "v", "s11", "s22", "s12")}
    path = m.ctx.last_path()
    del m
    return out, path


def perturbed(c, seed=11):
    rng = np.random.default_rng(seed)
    c2 = dict(c)
    for k in ("u", "v", "h"):
        c2[k] = c[k] * (1 + 1e-15 * rng.standard_normal(c[k].shape))
    return c2


@pytest.mark.parametrize("N", [1024, 2048])
def test_config3_fplane_120_substeps(N):
    """Config 3 (1024^2) and the metric's grid (2048^2): periodic f-plane, EVP defaults, 120 sub-steps, constant wind
    stress, semi-implicit ocean drag (test/distributed_tests_utils.jl:104-137 physics)."""
    c = cases.make_case(Nx=N, Ny=N, topo=("periodic", "periodic"), patches=True, random_uv=0.02, substeps=120)
    strict, p0 = run_cycle(c, "strict")
    three, p1 = run_cycle(c, "fast", fusion=0)
    pair, p2 = run_cycle(c, "fast", fusion=2)
    assert p0["level"] == 0 and p1["level"] == 0 and p2["level"] == 2
    # the benchmarked kernel (two sub-steps per launch) == the three-kernel FAST path, bit for bit, every field
    for f in three:
        assert np.all(np.isfinite(pair[f])), f
        assert np.array_equal(three[f], pair[f]), (f, np.abs(three[f] - pair[f]).max())
    # FAST vs STRICT: stated tolerance, or 10 x STRICT's own sensitivity where the iteration is ill-conditioned
    sens_run, _ = run_cycle(perturbed(c), "strict")
    vmax = max(np.abs(strict["u"]).max(), np.abs(strict["v"]).max())
    smax = max(np.abs(strict[f]).max() for f in ("s11", "s22", "s12"))
    entry = {"vmax": vmax, "smax": smax}
    for f in ("u", "v", "s11", "s22", "s12"):
        tol = (FAST_TOL_VEL * vmax) if f in ("u", "v") else (FAST_TOL_SIG * smax)
        sens = float(np.abs(sens_run[f] - strict[f]).max())
        d = float(np.abs(pair[f] - strict[f]).max())
        entry[f] = {"diff": d, "stated_tol": tol, "strict_self_sensitivity": sens, "branch": "stated" if d <= tol else "10x sensitivity"}
        assert d <= max(tol, 10 * sens), (f, d, tol, sens)
    _record(f"config3_{N}", entry)
    # threshold decisions: open water / marginal ice cells are the same set in every path
    for f in ("u", "v"):
        assert np.array_equal(strict[f] == 0.0, pair[f] == 0.0), f
    assert 0 < (strict["u"] == 0.0).sum() < strict["u"].size
    assert vmax < 10.0


def latlon_case(N, H, topo, substeps=120):
    c = cases.make_case(Nx=N, Ny=N, H=H, topo=topo, grid="latlon", patches=True, random_uv=0.02, substeps=substeps)
    return c


def slab():
    # test/test_thermodynamic_mass_fluxes.jl:56 (numeric top / bottom fluxes), prescribed top temperature
    return csi.SlabThermodynamics(top_temperature=-10.0, top_heat_flux=100.0, bottom_heat_flux=10.0)


def full_model(c, mode, tile=None, k=None, fusion=None):
    m = cases.csi_model(c, mode=mode, timestepper="SplitRungeKutta3", advection=csi.WENO(order=7), tile=tile,
                        ice_thermodynamics=slab())
    if k is not None:
        m.set_exchange_interval(k)
    if fusion is not None:
        m.set_fusion(fusion)
    return m


def model_state(m):
    m.synchronize()
    out = {"h": m.ice_thickness.interior_numpy().copy(), "a": m.ice_concentration.interior_numpy().copy()}
    out.update({f: EVP_FIELDS[f](m).interior_numpy().copy() for f in ("u", "v", "s11", "s22", "s12")})
    return out


def test_config4_latlon_rk3_step_fast_vs_strict():
    """Config 4 physics at 2048^2: lat-lon grid with per-row metrics, bounded, one SplitRungeKutta3 time_step! =
    3 x [WENO7 advection of h, aice; 120 EVP sub-steps; tracer update; slab thermodynamics]."""
    N = 2048
    c = latlon_case(N, 4, ("bounded", "bounded"))
    res = {}
    for mode in ("strict", "fast"):
        m = full_model(c, mode)
        csi.time_step(m, c["dt"])
        res[mode] = model_state(m)
        if mode == "fast":
            assert m.ctx.last_path()["level"] == 2
        del m
    sens_runs = []
    for seed in (11, 12):                         # STRICT's own response to 1e-15 relative input perturbations: two samples, the larger counts
        ms = full_model(perturbed(c, seed), "strict")
        csi.time_step(ms, c["dt"])
        sens_runs.append(model_state(ms))
        del ms
    sens = {f: np.where(np.abs(sens_runs[0][f] - res["strict"][f]) >= np.abs(sens_runs[1][f] - res["strict"][f]), sens_runs[0][f], sens_runs[1][f])
            for f in sens_runs[0]}
    entry = {}
    # tracers: advection, update and slab step are computed in the reference's order in both modes; they see the
    # velocities of the sub-cycle, so they inherit its rounding-level differences only
    for f, tol in (("h", 1e-11), ("a", 1e-11)):
        scale = np.abs(res["strict"][f]).max()
        d = float(np.abs(res["fast"][f] - res["strict"][f]).max())
        s = float(np.abs(sens[f] - res["strict"][f]).max())
        entry[f] = {"diff": d, "stated_tol": tol * scale, "strict_self_sensitivity": s, "branch": "stated" if d <= tol * scale else "10x sensitivity",
                    "margin": max(tol * scale, 10 * s) / max(d, 1e-300)}
        assert np.all(np.isfinite(res["fast"][f]))
        assert d <= max(tol * scale, 10 * s), (f, d, scale, s)
    vmax = max(np.abs(res["strict"]["u"]).max(), np.abs(res["strict"]["v"]).max())
    smax = max(np.abs(res["strict"][f]).max() for f in ("s11", "s22", "s12"))
    for f in ("u", "v", "s11", "s22", "s12"):
        tol = (FAST_TOL_VEL * vmax) if f in ("u", "v") else (FAST_TOL_SIG * smax)
        d = float(np.abs(res["fast"][f] - res["strict"][f]).max())
        s = float(np.abs(sens[f] - res["strict"][f]).max())
        entry[f] = {"diff": d, "stated_tol": tol, "strict_self_sensitivity": s, "branch": "stated" if d <= tol else "10x sensitivity"}
        assert np.all(np.isfinite(res["fast"][f]))
        assert d <= max(tol, 10 * s), (f, d, tol, s)
    _record("config4_latlon_rk3", entry)
    a = res["fast"]["a"]
    assert a.min() >= 0.0 and a.max() <= 1.0 and res["fast"]["h"].min() >= 0.0
    # the slab step changed the thickness (top flux 100 W m^-2 melts, bottom growth): not a no-op
    m0 = cases.csi_model(c, mode="fast", timestepper="SplitRungeKutta3", advection=csi.WENO(order=7))
    csi.time_step(m0, c["dt"])
    assert not np.array_equal(model_state(m0)["h"], res["fast"]["h"])


@pytest.mark.parametrize("nsub", [1, 2])
@pytest.mark.parametrize("topo", [("bounded", "bounded"), ("periodic", "bounded")])
def test_config4_latlon_few_substeps_tight(topo, nsub):
    """Config 4's grid at full size, TIGHT: 2048^2 lat-lon (lon 0..60, lat 20..70: the geometry of
    test/test_rheology_energy_budget.jl:102-106, per-row metrics) after ONE and TWO sub-steps, before the mEVP iteration's own
    sensitivity can hide anything: FAST (per-row stencil coefficients folded on the host, Newton reciprocals) against STRICT
    (the reference's operator calls, bit-identical to the oracle at small sizes) within 1e-13 max|u| on u, v and 1e-10
    max|sigma| on sigma -- over the whole grid and in each of eight latitude bands, so that a defect of the per-row
    coefficients at high latitudes cannot pass as chaos.  One sub-step runs the one-sub-step kernel, two the pair kernel."""
    N = 2048
    c = latlon_case(N, 4, topo, substeps=nsub)
    strict, p0 = run_cycle(c, "strict")
    fast, p1 = run_cycle(c, "fast")
    assert p0["level"] == 0 and p1["level"] == (2 if nsub == 2 else 1), (p0, p1)
    vmax = max(np.abs(strict["u"]).max(), np.abs(strict["v"]).max())
    smax = max(np.abs(strict[f]).max() for f in ("s11", "s22", "s12"))
    entry = {"vmax": float(vmax), "smax": float(smax)}
    for f in ("u", "v", "s11", "s22", "s12"):
        tol = 1e-13 * vmax if f in ("u", "v") else 1e-10 * smax
        diff = np.abs(fast[f] - strict[f])
        assert np.all(np.isfinite(fast[f])), f
        bands = [float(b.max()) for b in np.array_split(diff[:N], 8, axis=0)]       # south to north
        entry[f] = {"diff": float(diff.max()), "tol": float(tol), "diff_by_latitude_band": bands}
        assert diff.max() <= tol, (f, float(diff.max()), tol, bands)
    for f in ("u", "v"):
        assert np.array_equal(strict[f] == 0.0, fast[f] == 0.0), f        # open water / marginal ice: the same cells
    _record(f"config4_latlon_{topo[0]}_{nsub}_substeps_tight", entry)


def test_config4_latlon_two_substeps_from_states_along_the_strict_cycle():
    """Config 4's grid at full size (2048^2 lat-lon, bounded): the FAST pair kernel restarted from STRICT's state at twelve points of
    the 120-sub-step cycle (every tenth sub-step: u, v, sigma copied parent for parent; P, u^n, v^n too) and advanced by two sub-steps
    beside it, each time on the TIGHT bound of the test above (1e-13 max|u|, 1e-10 max|sigma| on the owned cells).  With this the
    "10 x STRICT's own sensitivity" branch the whole-cycle and RK3 comparisons of this grid may take is about the ITERATION, not the
    kernel: its error is rounding-level at every state the cycle passes through (round 5; the small-grid twin against the oracle:
    tests/test_gpu_evp.py::test_fast_two_substeps_from_every_state_of_the_oracle_cycle)."""
    import torch
    N = 2048
    c = latlon_case(N, 4, ("bounded", "bounded"))
    ms = cases.csi_model(c, mode="strict")
    mf = cases.csi_model(c, mode="fast")
    ms.ctx.call("csi_evp_initialize"); mf.ctx.call("csi_evp_initialize")
    ms.synchronize(); mf.synchronize()
    names = ("u", "v", "s11", "s22", "s12", "P", "un", "vn")
    entry = {}
    s = 1
    while s + 1 <= 120:
        for k in names:
            EVP_FIELDS[k](mf).data.copy_(EVP_FIELDS[k](ms).data)
        torch.cuda.synchronize()
        ms.ctx.call("csi_evp_subcycle", float(c["dt"]), 2, s)
        mf.ctx.call("csi_evp_subcycle", float(c["dt"]), 2, s)
        ms.synchronize(); mf.synchronize()
        assert mf.ctx.last_path()["level"] == 2 and ms.ctx.last_path()["level"] == 0
        a = {k: EVP_FIELDS[k](ms).interior_numpy() for k in ("u", "v", "s11", "s22", "s12")}
        b = {k: EVP_FIELDS[k](mf).interior_numpy() for k in ("u", "v", "s11", "s22", "s12")}
        vmax = max(np.abs(a["u"]).max(), np.abs(a["v"]).max())
        smax = max(np.abs(a[k]).max() for k in ("s11", "s22", "s12"))
        for k in ("u", "v", "s11", "s22", "s12"):
            d = float(np.abs(a[k] - b[k]).max())
            tol = 1e-13 * vmax if k in ("u", "v") else 1e-10 * smax
            assert np.all(np.isfinite(b[k])) and d <= tol, ("sub-steps", s, s + 1, k, d, tol)
            entry[k] = max(entry.get(k, 0.0), d / (vmax if k in ("u", "v") else smax))
        for k in ("u", "v"):
            assert np.array_equal(a[k] == 0.0, b[k] == 0.0), (s, k)
        # STRICT goes on alone to the next sample
        if s + 10 <= 119:
            ms.ctx.call("csi_evp_subcycle", float(c["dt"]), 8, s + 2)
            ms.synchronize()           # the copies above run on torch's stream, the sub-cycle on the library's
        s += 10
    _record("config4_latlon_two_substeps_along_the_strict_cycle", {"worst_relative_difference": entry, "samples": 12})


def test_config4_latlon_one_cycle_branch_recorded():
    """One 120-sub-step momentum cycle (not the three of an RK3 step) on the 2048^2 lat-lon grid: FAST against STRICT, with the
    bound each field needed -- the stated tolerance, or 10 x STRICT's own response to a 1e-15 relative input perturbation --
    recorded per field (gpurun_out/fullsize_tolerance.json); two perturbation samples, the larger response counts."""
    N = 2048
    c = latlon_case(N, 4, ("bounded", "bounded"))
    strict, _ = run_cycle(c, "strict")
    fast, p = run_cycle(c, "fast")
    assert p["level"] == 2
    sens = [run_cycle(perturbed(c, seed), "strict")[0] for seed in (11, 12)]
    vmax = max(np.abs(strict["u"]).max(), np.abs(strict["v"]).max())
    smax = max(np.abs(strict[f]).max() for f in ("s11", "s22", "s12"))
    entry = {"vmax": float(vmax), "smax": float(smax)}
    for f in ("u", "v", "s11", "s22", "s12"):
        tol = (FAST_TOL_VEL * vmax) if f in ("u", "v") else (FAST_TOL_SIG * smax)
        s_ = max(float(np.abs(q[f] - strict[f]).max()) for q in sens)
        d = float(np.abs(fast[f] - strict[f]).max())
        entry[f] = {"diff": d, "stated_tol": float(tol), "strict_self_sensitivity": s_, "branch": "stated" if d <= tol else "10x sensitivity"}
        assert np.all(np.isfinite(fast[f])), f
        assert d <= max(tol, 10 * s_), (f, d, tol, s_)
    _record("config4_latlon_one_cycle", entry)


@pytest.mark.parametrize("k", [16, 1, 0])
def test_config4_latlon_tile_bitwise(k):
    """Config 4 on a tile: 2048^2 lat-lon channel (periodic x, walls in y); the same RK3 step on a tile whose x edges are
    CONNECTED equals the untiled run bit for bit -- with the RCCL exchange (to itself: width 2k every k sub-steps, halo 32,
    k = 16 and k = 1) and with the peer transport (k = 0: halo 4, images stored straight into the neighbour's = its own arrays,
    one RCCL exchange per sub-cycle)."""
    N = 2048
    c = latlon_case(N, 32 if k else 4, ("periodic", "bounded"))
    ref = full_model(c, "fast")
    csi.time_step(ref, c["dt"])
    want = model_state(ref)
    assert ref.ctx.last_path()["level"] == 2
    del ref
    til = full_model(c, "fast", tile=(1, 1, 0, (True, False)), k=k)
    csi.time_step(til, c["dt"])
    got = model_state(til)
    path = til.ctx.last_path()
    if k == 0:
        assert til.ctx.halo_transport() == "peer" and path["level"] == 2 and path["exchanges"] == 1, path
    else:
        assert path["exchange_interval"] == k and path["level"] == (2 if k % 2 == 0 else 1), path
        assert path["exchanges"] == (120 // k + (1 if 120 % k else 0)), path
    for f in want:
        assert np.all(np.isfinite(got[f])), f
        assert np.array_equal(want[f], got[f]), (f, np.abs(want[f] - got[f]).max(), np.argwhere(want[f] != got[f])[:4])


@pytest.mark.parametrize("k", [0, -1])
def test_config4_latlon_2x4_decomposition_bitwise(k):
    """Config 4 in its own decomposition: the 2048^2 lat-lon channel as 2 x 4 DISTINCT tiles (1024 x 512 each; a wall on one y
    side, a neighbour on the other for the top and bottom rows of tiles), one RK3 step with WENO7 advection, 3 x 120 EVP
    sub-steps and slab thermodynamics per tile -- every tile its own context, stream and host thread on this one GPU, halos
    through the in-process tile group (tests/test_gpu_local_tiles.py): k = 0 the peer transport (halo 4, images stored straight
    into the neighbours' arrays, per-tile flags between distinct tiles), k = -1 the message exchange at its automatic interval.
    Owned cells of all eight tiles equal the untiled run bit for bit."""
    from test_gpu_local_tiles import run_tile_threads
    N, Rx, Ry = 2048, 2, 4
    c = latlon_case(N, 4 if k == 0 else 16, ("periodic", "bounded"))
    ref = full_model(c, "fast")
    csi.time_step(ref, c["dt"])
    want = model_state(ref)
    del ref

    def tile(rank, group):
        m = cases.csi_model(c, mode="fast", timestepper="SplitRungeKutta3", advection=csi.WENO(order=7), tile=(Rx, Ry, rank),
                            local_group=group, ice_thermodynamics=slab())
        if k < 0:
            m.set_halo_transport("rccl")
        csi.time_step(m, c["dt"])
        got = model_state(m)
        g = m.grid
        return got, (g.i_off, g.j_off, g.Nx, g.Ny), dict(m.ctx.last_path(), transport=m.ctx.halo_transport())

    for rank, (got, (i0, j0, nx, ny), path) in enumerate(run_tile_threads(Rx * Ry, tile)):
        assert path["transport"] == ("peer" if k == 0 else "rccl") and path["level"] == 2, path
        for f in want:
            w = want[f][j0:j0 + ny, i0:i0 + nx]
            g_ = got[f][:ny, :nx]
            assert np.array_equal(w, g_), (rank, f, np.abs(w - g_).max(), np.argwhere(w != g_)[:4].tolist())


def config5_case(N, substeps):
    """4096^2 uniform 2 km channel (periodic x, walls in y), land = seeded discs covering 30 % + solid caps of N / 32
    rows at both walls, h = aice = 0 on land."""
    c = cases.make_case(Nx=N, Ny=N, topo=("periodic", "bounded"), patches=True, random_uv=0.02, substeps=substeps, land=0.3)
    cap = N // 32
    wet = c["mask"].copy()
    wet[:cap, :] = False
    wet[-cap:, :] = False
    c["mask"] = wet
    c["h"] = np.where(wet, c["h"], 0.0)
    c["a"] = np.where(wet, c["a"], 0.0)
    return c


def test_config5_masked_4096_500_substeps():
    """Config 5: 4096^2 masked channel, 500 sub-steps.  The two-sub-steps-per-launch kernel with the immersed mask ==
    the three-kernel FAST path bit for bit; everything finite; land faces exactly zero."""
    N = 4096
    c = config5_case(N, 500)
    three, p1 = run_cycle(c, "fast", fusion=0)
    pair, p2 = run_cycle(c, "fast", fusion=2)
    assert p1["level"] == 0 and p2["level"] == 2
    for f in three:
        assert np.all(np.isfinite(pair[f])), f
        assert np.array_equal(three[f], pair[f]), (f, np.abs(three[f] - pair[f]).max(), np.argwhere(three[f] != pair[f])[:4])
    wet = c["mask"]
    # peripheral nodes: faces with an inactive cell on either side carry exactly zero velocity
    u_land = ~(wet & np.roll(wet, 1, axis=1))                     # u(i, j) between cells i-1 and i (periodic x)
    assert np.all(pair["u"][u_land] == 0.0)
    v = pair["v"]                                                  # (Ny + 1, Nx): faces 1 .. Ny + 1, walls at both ends
    v_land = np.ones(v.shape, dtype=bool)
    v_land[1:N, :] = ~(wet[1:, :] & wet[:-1, :])
    assert np.all(v[v_land] == 0.0)
    assert np.abs(pair["u"]).max() < 10.0 and np.abs(pair["u"][~u_land]).max() > 0.0
    frac = 1.0 - wet.mean()
    assert 0.3 <= frac <= 0.45, frac
    _record("config5_4096", {"land_fraction": float(frac), "vmax": float(max(np.abs(pair['u']).max(), np.abs(pair['v']).max()))})


@pytest.mark.parametrize("transport", ["rccl", "peer"])
def test_config5_tile_2048x1024_bitwise(transport):
    """Config 5's tile shape (4096^2 over 2 x 4 GPUs = 2048 x 1024 per tile) as a self-connected masked tile with the
    batched exchange, 120 sub-steps: owned cells equal the untiled run bit for bit."""
    c = cases.make_case(Nx=2048, Ny=1024, H=16, topo=("periodic", "bounded"), patches=True, random_uv=0.02, substeps=120, land=0.3)
    want, p1 = run_cycle(c, "fast", fusion=2)
    got, p2 = run_cycle(c, "fast", tile=(1, 1, 0, (True, False)), transport=transport)
    assert p1["level"] == 2 and p2["level"] == 2 and (p2["exchange_interval"] == 8 if transport == "rccl" else p2["exchanges"] == 1), (p1, p2)
    for f in want:
        assert np.array_equal(want[f], got[f]), (f, np.abs(want[f] - got[f]).max())


def test_config5_masked_4096_2x4_decomposition_bitwise():
    """Config 5 in its own decomposition: the 4096^2 masked channel, 500 sub-steps, as 2 x 4 DISTINCT tiles of 2048 x 1024 on
    this one GPU (in-process tile group), peer transport: images stored into the neighbours' arrays, flags between distinct
    tiles, one message exchange per sub-cycle.  Owned cells of all eight tiles equal the untiled run bit for bit."""
    from test_gpu_local_tiles import run_tile_threads
    N, Rx, Ry = 4096, 2, 4
    c = config5_case(N, 500)
    want, p1 = run_cycle(c, "fast", fusion=2)
    assert p1["level"] == 2

    def tile(rank, group):
        m = cases.csi_model(c, mode="fast", tile=(Rx, Ry, rank), local_group=group)
        csi.time_step_momentum(m, c["dt"])
        m.synchronize()
        got = {f: EVP_FIELDS[f](m).interior_numpy().copy() for f in ("u", "v", "s11", "s22", "s12")}
        g = m.grid
        return got, (g.i_off, g.j_off, g.Nx, g.Ny), dict(m.ctx.last_path(), transport=m.ctx.halo_transport())

    for rank, (got, (i0, j0, nx, ny), path) in enumerate(run_tile_threads(Rx * Ry, tile)):
        assert path["transport"] == "peer" and path["level"] == 2 and path["exchanges"] == 1, path
        for f in want:
            w = want[f][j0:j0 + ny, i0:i0 + nx]
            g_ = got[f][:ny, :nx]
            assert np.array_equal(w, g_), (rank, f, np.abs(w - g_).max(), np.argwhere(w != g_)[:4].tolist())


# ---- the realistic global configurations at the metric's size (round 4): the pair kernel's per-point-metric instantiations run
# 999 / 888 tiles of 76 / 86 rows there (plane values through the LDS ring, the fold band beside the launch) -- shapes the small
# named cases do not reach.  FAST three-kernel path == pair kernel, bit for bit, every field.
FULLSIZE_GLOBAL = {
    "curvilinear_channel": dict(topo=("periodic", "bounded"), curvilinear=0.05),
    "curvilinear_channel_land": dict(topo=("periodic", "bounded"), curvilinear=0.05, land=0.3),
    "curvilinear_coupled": dict(topo=("periodic", "bounded"), curvilinear=0.05, land=0.3, field_forcing=True, free_drift=True),
    "tripolar_like": dict(topo=("periodic", "folded"), curvilinear=0.05, land=0.3, field_forcing=True, free_drift=True),
}


@pytest.mark.parametrize("nsub", [12, 7])
@pytest.mark.parametrize("name", sorted(FULLSIZE_GLOBAL))
def test_global_configurations_2048_pair_equals_three_kernels(name, nsub):
    c = cases.make_case(Nx=2048, Ny=2048, substeps=nsub, patches=False, noise=0.05, **FULLSIZE_GLOBAL[name])
    ref, p0 = run_cycle(c, "fast", fusion=0)
    got, p2 = run_cycle(c, "fast", fusion=2)
    assert p0["level"] == 0 and p2["level"] == 2, (p0, p2)
    for f in ref:
        assert np.isfinite(got[f]).all(), f
        assert got[f].tobytes() == ref[f].tobytes(), (name, f, float(np.abs(got[f] - ref[f]).max()))
