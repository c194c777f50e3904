"""GPU tests of the two structure cuts of round 6 (both exact):

  * tile activity (csi_set_tile_skipping, csrc/csi_activity.hip): tiles with no ice mass in or around them are left out of all
    launches of a sub-cycle but the first two and the last -- every field, halo cells included, must be BIT-IDENTICAL with
    skipping off;
  * row-constant rows of a CSI_METRIC_FULL grid (csi_set_row_constant): tiles whose rows hold one value per row in all twelve
    coefficient planes read them from per-row vectors -- bit-identical with the feature off;
  * the real tripolar geometry (csi.TripolarGrid: latitude-longitude rows below a conformal bipolar cap, the reference's analytic
    land, test/distributed_tests_utils.jl:170-183): against the oracle, with both cuts at work.

Why skipping is exact: sigma += ifelse(m > 0, sigma*, 0) (src/Rheologies/elasto_visco_plastic_rheology.jl:343-347) and the
velocity select's zero branch (src/SeaIceDynamics/split_explicit_momentum_equations.jl:217-228, 251-263).
"""
import numpy as np
import pytest
import torch

import cases
import climaseaice_jl_amd as csi
from test_gpu_evp import EVP_FIELDS

pytestmark = pytest.mark.gpu

OUT = ("u", "v", "s11", "s22", "s12", "alpha", "zeta_c", "zeta_f", "Delta")


def parents(m):
    m.synchronize()
    return {k: EVP_FIELDS[k](m).numpy().copy() for k in OUT}


def run(case, steps=3, skipping=True, row_constant=True, mode="fast", **kw):
    m = cases.csi_model(case, mode=mode, **kw)
    m.set_tile_skipping(skipping)
    m.set_row_constant(row_constant)
    acts = []
    for _ in range(steps):
        csi.time_step_momentum(m, case["dt"])
        m.synchronize()
        acts.append(m.tile_activity())
    return parents(m), acts, m


def assert_bitwise(a, b, what):
    for k in OUT:
        same = (a[k].view(np.int64) == b[k].view(np.int64)) | (np.isnan(a[k]) & np.isnan(b[k]))
        assert same.all(), f"{what}: {k} differs in {np.count_nonzero(~same)} parent cells (first at {np.argwhere(~same)[0]})"


# ice-free ocean and land large enough to hold whole 56-column tiles; odd and even numbers of sub-steps; every kernel family
SKIP_CASES = {
    "periodic_half_free": dict(Nx=448, Ny=320, topo=("periodic", "periodic"), patches=False, random_uv=0.03, ice_free_rows=(0.3, 0.8), substeps=20),
    "periodic_odd": dict(Nx=448, Ny=320, topo=("periodic", "periodic"), patches=True, random_uv=0.03, ice_free_rows=(0.0, 0.6), substeps=21),
    "bounded_free": dict(Nx=400, Ny=300, topo=("bounded", "bounded"), patches=False, random_uv=0.03, ice_free_rows=(0.5, 1.0), substeps=16),
    "latlon_free": dict(Nx=336, Ny=280, topo=("bounded", "bounded"), grid="latlon", patches=True, random_uv=0.03, ice_free_rows=(0.0, 0.5), substeps=12),
    "masked_channel": dict(Nx=448, Ny=300, topo=("periodic", "bounded"), patches=True, random_uv=0.03, land=0.45, substeps=20),
    "masked_free": dict(Nx=448, Ny=300, topo=("periodic", "bounded"), patches=False, random_uv=0.03, land=0.3, ice_free_rows=(0.2, 0.7), substeps=14),
    "coupled_free": dict(Nx=392, Ny=260, topo=("periodic", "bounded"), patches=True, random_uv=0.03, land=0.3, field_forcing=True,
                         ice_free_rows=(0.4, 0.9), substeps=12),
    "omip_free": dict(Nx=392, Ny=260, topo=("periodic", "bounded"), patches=True, random_uv=0.03, land=0.3, field_forcing=True, free_drift=True,
                      ice_free_rows=(0.1, 0.6), substeps=12),
    "user_forcing_free": dict(Nx=336, Ny=240, topo=("periodic", "periodic"), patches=False, random_uv=0.03, user_forcing=True,
                              ice_free_rows=(0.25, 0.75), substeps=10),
    "immersed_bc_free": dict(Nx=336, Ny=240, topo=("periodic", "bounded"), patches=False, random_uv=0.03, land=0.35,
                             immersed_bc=((0.02, -0.01, 0.015, 0.005), (-0.01, 0.02, 0.01, -0.015)), substeps=10),
    "wind_drag_free": dict(Nx=336, Ny=240, topo=("periodic", "periodic"), patches=False, random_uv=0.03, wind_drag="arrays", field_forcing=True,
                           ice_free_rows=(0.2, 0.8), substeps=10),
    "curvilinear_free": dict(Nx=336, Ny=260, topo=("periodic", "bounded"), patches=False, random_uv=0.03, curvilinear=0.05, land=0.3,
                             ice_free_rows=(0.3, 0.8), substeps=12),
    "fold_free": dict(Nx=336, Ny=260, topo=("periodic", "folded"), patches=False, random_uv=0.03, ice_free_rows=(0.1, 0.7), substeps=12),
    "tripolar_like_free": dict(Nx=336, Ny=260, topo=("periodic", "folded"), patches=False, random_uv=0.03, curvilinear=0.05, land=0.3, field_forcing=True,
                               free_drift=True, ice_free_rows=(0.1, 0.6), substeps=12),
    "beta_free": dict(Nx=336, Ny=240, topo=("periodic", "bounded"), patches=False, random_uv=0.03, beta=2e-10, ice_free_rows=(0.0, 0.5), substeps=10),
    "noslip_free": dict(Nx=336, Ny=240, topo=("periodic", "bounded"), patches=False, random_uv=0.03, noslip=True, land=0.3,
                        ice_free_rows=(0.5, 1.0), substeps=10),
}


@pytest.mark.parametrize("name", list(SKIP_CASES))
def test_skipping_is_bit_identical(name):
    """Three sub-cycles (the launch geometry follows the live fraction from the second on) with and without tile skipping."""
    case = cases.make_case(**SKIP_CASES[name])
    on, acts, m = run(case, skipping=True)
    off, acts_off, _ = run(case, skipping=False)
    assert_bitwise(on, off, name)
    tiles, live, used = acts[0]
    assert used >= 1 and 0 < live < tiles, (name, acts)          # the first sub-cycle always tests: something was skipped, something ran
    if 10 * live < 9 * tiles:                                     # (with nine tiles in ten live the grid is only probed: csi_launch.hip)
        assert acts[-1][2] >= 1 and 0 < acts[-1][1] < acts[-1][0], (name, acts)
    assert acts_off[-1][2] == 0
    assert m.ctx.last_path()["level"] == 2


def test_tiles_quiescent_from_the_start_skip_the_first_launches_too():
    """From the second sub-cycle on the ice-free velocities are +0.0 and a sample has shown quiescent tiles: the first two launches then
    run from a list that leaves those tiles out as well (csi_tile_activity `used` = 2) -- same bits as all-tile launches."""
    case = cases.make_case(Nx=560, Ny=420, topo=("periodic", "bounded"), patches=False, random_uv=0.03, land=0.3, ice_free_rows=(0.25, 0.8), substeps=10)
    on, acts, _ = run(case, steps=4, skipping=True)
    off, _, _ = run(case, steps=4, skipping=False)
    assert_bitwise(on, off, "quiescent from the start")
    assert acts[0][2] == 1 and acts[-1][2] == 2, acts


def test_negative_zero_stress_keeps_a_tile_alive():
    """fma(x, 0, -0.0) takes the sign of x: a -0.0 among the stresses of an ice-free tile must keep it in the launches."""
    case = cases.make_case(Nx=448, Ny=320, topo=("periodic", "periodic"), patches=False, random_uv=0.03, ice_free_rows=(0.3, 0.9), substeps=12)
    res = []
    for skipping in (True, False):
        m = cases.csi_model(case, mode="fast")
        m.set_tile_skipping(skipping)
        for key in ("s11", "s22", "s12"):
            p = EVP_FIELDS[key](m).data
            p[int(0.55 * p.shape[0]):int(0.65 * p.shape[0]), :] = -0.0           # a band of rows inside the ice-free part
        torch.cuda.synchronize()
        csi.time_step_momentum(m, case["dt"])
        m.synchronize()
        res.append((parents(m), m.tile_activity()))
    assert_bitwise(res[0][0], res[1][0], "negative zeros")
    # the -0.0 rows are gone after the sub-cycle either way (x * 0 + (-0.0) = +0.0 for x >= 0), and fewer tiles were skipped
    m2 = cases.csi_model(case, mode="fast")
    csi.time_step_momentum(m2, case["dt"])
    m2.synchronize()
    assert res[0][1][1] > m2.tile_activity()[1]


def test_ice_free_velocities_are_zeroed_like_the_reference():
    """u0 = 0.1 everywhere, also where there is no ice: the first sub-step zeroes it (split_explicit...:228) -- the skipped tiles' cells
    included, because the first two launches run every tile."""
    case = cases.make_case(Nx=448, Ny=320, topo=("periodic", "periodic"), patches=False, u0=0.1, v0=-0.05, random_uv=0.03, ice_free_rows=(0.2, 0.9), substeps=10)
    on, acts, m = run(case, steps=1, skipping=True)
    H = case["H"]
    j0, j1 = int(0.3 * 320), int(0.8 * 320)
    assert np.all(on["u"][H + j0:H + j1, :] == 0.0) and np.all(on["v"][H + j0:H + j1, :] == 0.0)
    p = cases.oracle_problem(case)
    p.time_step_momentum(case["dt"])
    vmax = np.abs(p.f["u"]).max()
    for k in ("u", "v"):
        assert np.abs(m_field(m, k) - p.f[k]).max() <= 1e-12 * vmax
        assert np.array_equal(m_field(m, k) == 0.0, p.f[k] == 0.0)


def m_field(m, k):
    return EVP_FIELDS[k](m).numpy()


def test_short_subcycles_do_not_skip():
    case = cases.make_case(Nx=448, Ny=320, topo=("periodic", "periodic"), patches=False, ice_free_rows=(0.2, 0.9), substeps=6)
    _, acts, _ = run(case, steps=2)
    assert acts[-1][2] == 0


def test_whole_steps_with_skipping_match():
    """SplitRungeKutta3 steps with advection: the ice edge moves, the live set is made afresh before every sub-cycle."""
    kw = dict(Nx=392, Ny=300, topo=("periodic", "bounded"), patches=False, random_uv=0.02, land=0.3, ice_free_rows=(0.3, 0.8), substeps=12)
    case = cases.make_case(**kw)
    out = []
    for skipping in (True, False):
        m = cases.csi_model(case, mode="fast", timestepper="SplitRungeKutta3", advection=csi.WENO(order=5))
        m.set_tile_skipping(skipping)
        for _ in range(3):
            csi.time_step(m, 600.0)
        m.synchronize()
        st = parents(m)
        st["h"] = m.ice_thickness.numpy().copy()
        st["a"] = m.ice_concentration.numpy().copy()
        out.append(st)
    assert_bitwise(out[0], out[1], "RK3 steps")
    assert np.array_equal(out[0]["h"], out[1]["h"]) and np.array_equal(out[0]["a"], out[1]["a"])


# ---- tiles on the peer halo transport: the interior tiles may go quiet, the direction sets always run ---------------------------------

PEER_SKIP_CASES = {
    # (make_case keywords, which periodic directions are connected to the tile itself)
    "periodic_xy": (dict(Nx=560, Ny=420, topo=("periodic", "periodic"), patches=False, random_uv=0.03, ice_free_rows=(0.2, 0.8), substeps=16), (True, True)),
    "slab_y": (dict(Nx=448, Ny=400, topo=("periodic", "periodic"), patches=True, random_uv=0.03, ice_free_rows=(0.0, 0.55), substeps=12), (False, True)),
    "channel_land": (dict(Nx=504, Ny=360, topo=("periodic", "bounded"), patches=False, random_uv=0.03, land=0.45, substeps=14), (True, False)),
    "omip": (dict(Nx=448, Ny=320, topo=("periodic", "bounded"), patches=False, random_uv=0.03, land=0.3, field_forcing=True, free_drift=True,
                  ice_free_rows=(0.3, 0.8), substeps=12), (True, False)),
    "curvilinear": (dict(Nx=448, Ny=300, topo=("periodic", "bounded"), patches=False, random_uv=0.03, curvilinear=0.04, ice_free_rows=(0.25, 0.75),
                         substeps=12), (True, False)),
}


@pytest.mark.parametrize("name", list(PEER_SKIP_CASES))
def test_skipping_on_a_peer_connected_tile(name):
    """One tile connected to itself over the peer transport (flags, halo images through the image table): with skipping, without, and the
    untiled grid -- owned cells bit for bit; interior tiles were skipped, the transport stayed `peer`."""
    kw, fc = PEER_SKIP_CASES[name]
    case = cases.make_case(**kw)
    on, acts, m = run(case, steps=3, skipping=True, tile=(1, 1, 0, fc))
    off, _, m0 = run(case, steps=3, skipping=False, tile=(1, 1, 0, fc))
    ref, _, _ = run(case, steps=3, skipping=False)
    assert m.ctx.halo_transport() == "peer" and m0.ctx.halo_transport() == "peer"
    assert_bitwise(on, off, name)
    H = case["H"]
    for k in ("u", "v", "s11", "s22", "s12"):
        a, b = on[k][H:H + case["Ny"], H:H + case["Nx"]], ref[k][H:H + case["Ny"], H:H + case["Nx"]]
        assert np.array_equal(a.view(np.int64), b.view(np.int64)), (name, k)
    tiles, live, used = acts[0]
    assert used >= 1 and 0 < live < tiles, (name, acts)
    if 10 * live < 9 * tiles:
        assert acts[-1][2] >= 1, (name, acts)


# ---- row-constant rows ----------------------------------------------------------------------------------------------------------------

ROWC_CASES = {
    # a latitude-longitude / rectilinear grid handed over as twelve arrays: every row constant
    "latlon_as_full": dict(Nx=200, Ny=140, topo=("bounded", "bounded"), grid="latlon", patches=True, random_uv=0.03, curvilinear=0.0, substeps=10),
    "rect_as_full_masked": dict(Nx=200, Ny=140, topo=("periodic", "bounded"), patches=True, random_uv=0.03, curvilinear=0.0, land=0.25, substeps=11),
    "rect_as_full_forced": dict(Nx=200, Ny=140, topo=("periodic", "bounded"), patches=True, random_uv=0.03, curvilinear=0.0, field_forcing=True,
                                free_drift=True, land=0.2, substeps=10),
    "fold_as_full": dict(Nx=200, Ny=160, topo=("periodic", "folded"), patches=True, random_uv=0.03, curvilinear=0.0, substeps=10),
}


@pytest.mark.parametrize("name", list(ROWC_CASES))
def test_row_constant_tiles_are_bit_identical(name):
    case = cases.make_case(**ROWC_CASES[name])      # (curvilinear = 0.0: the twelve arrays, undistorted)
    on, _, m = run(case, steps=2, row_constant=True, skipping=False)
    off, _, m0 = run(case, steps=2, row_constant=False, skipping=False)
    assert_bitwise(on, off, name)
    n = case["Ny"] + 2 * case["H"] + 1
    assert m.row_constant_rows() >= n - (2 * case["H"] + 2 if case["topo"][1] == "folded" else 0)
    assert m0.row_constant_rows() == 0


def test_row_constant_needs_bitwise_equal_columns():
    """One column of one plane one ulp off in a band of rows: those rows are not marked, the others are; results unchanged."""
    case = cases.make_case(Nx=200, Ny=160, topo=("periodic", "bounded"), patches=True, random_uv=0.03, curvilinear=0.0, substeps=10)
    g = case["g"]
    a = g.metrics()["dycf"]
    a[60:70, 17] = np.nextafter(a[60:70, 17], np.inf)
    on, _, m = run(case, steps=2, row_constant=True, skipping=False)
    off, _, _ = run(case, steps=2, row_constant=False, skipping=False)
    assert_bitwise(on, off, "ulp band")
    n = case["Ny"] + 2 * case["H"] + 1
    assert m.row_constant_rows() == n - 10
    # a tolerance marks them again (the caller's decision; results then move at that level)
    m.set_row_constant(True, rtol=1e-12)
    assert m.row_constant_rows() == n


# ---- the real tripolar geometry ---------------------------------------------------------------------------------------------------

TRIPOLAR = dict(Nx=224, Ny=192, grid="tripolar", tripolar=dict(southernmost_latitude=-78.0), patches=False, random_uv=0.02,
                field_forcing=True, free_drift=True, coriolis_points=True, ice_edge=58.0, substeps=20)


def test_tripolar_against_the_oracle_and_both_cuts():
    case = cases.make_case(**TRIPOLAR)
    full, acts, m = run(case, steps=1)
    p = cases.oracle_problem(case)
    p.time_step_momentum(case["dt"])
    vmax = max(np.abs(p.f["u"]).max(), np.abs(p.f["v"]).max())
    smax = max(np.abs(p.f[k]).max() for k in ("s11", "s22", "s12"))
    for k in ("u", "v"):
        d = np.abs(full[k] - p.f[k]).max()
        assert d <= 1e-12 * vmax, (k, d, vmax)
        assert np.array_equal(full[k] == 0.0, p.f[k] == 0.0)
    for k in ("s11", "s22", "s12"):
        d = np.abs(full[k] - p.f[k]).max()
        assert d <= 1e-11 * smax, (k, d, smax)
    # both cuts off: the same bits
    plain, _, m0 = run(case, steps=1, skipping=False, row_constant=False)
    assert_bitwise(full, plain, "tripolar")
    assert m.row_constant_rows() >= m.grid.cap_first_row - 1 + case["H"] and m0.row_constant_rows() == 0
    tiles, live, used = acts[0]
    assert used >= 1 and live < tiles


def test_tripolar_strict_matches_the_oracle_bitwise():
    kw = dict(TRIPOLAR, Nx=96, Ny=80, substeps=8)
    case = cases.make_case(**kw)
    m = cases.csi_model(case, mode="strict")
    csi.time_step_momentum(m, case["dt"])
    got = parents(m)
    p = cases.oracle_problem(case)
    p.time_step_momentum(case["dt"])
    for k in ("u", "v", "s11", "s22", "s12"):
        assert np.array_equal(got[k], p.f[k]), k


# ---- fuzz (24 seeds here; scripts/fuzz_activity.py runs campaigns) ------------------------------------------------------------------

def fuzz_case(seed):
    """A random configuration with land and / or ice-free ocean large enough for whole tiles to go quiet."""
    rng = np.random.default_rng(9000 + seed)
    topo = [("periodic", "periodic"), ("periodic", "bounded"), ("bounded", "bounded"), ("periodic", "folded")][rng.integers(0, 4)]
    kw = dict(Nx=int(rng.integers(2, 9)) * 56 + int(rng.integers(-20, 21)), Ny=int(rng.integers(120, 360)), topo=topo,
              patches=bool(rng.integers(0, 2)), random_uv=0.03, substeps=int(rng.integers(8, 27)), seed=int(seed))
    fam = rng.integers(0, 8)
    if topo[1] == "folded":
        kw["Nx"] += kw["Nx"] % 2                      # (the fold pairs column i with Nx - i + 1)
        if fam % 2:
            kw["curvilinear"] = [0.0, 0.05][rng.integers(0, 2)]
    elif fam == 1 and topo[0] == "bounded":
        kw["grid"] = "latlon"
    elif fam == 2:
        kw["curvilinear"] = [0.0, 0.04][rng.integers(0, 2)]
    elif fam == 3:
        kw["beta"] = 2e-10
    if rng.random() < 0.6 and topo[0] != "bounded":
        kw["land"] = float(rng.uniform(0.2, 0.5))
    if rng.random() < 0.4:
        kw["field_forcing"] = True
        kw["free_drift"] = bool(rng.integers(0, 2))
    elif rng.random() < 0.2:
        kw["user_forcing"] = True
    if "land" not in kw or rng.random() < 0.6:
        a = float(rng.uniform(0.0, 0.5))
        kw["ice_free_rows"] = (a, min(1.0, a + float(rng.uniform(0.3, 0.6))))
    if rng.random() < 0.2:
        kw["noslip"] = topo[1] == "bounded"
    return kw


@pytest.mark.parametrize("seed", range(24))
def test_cuts_fuzz_bitwise(seed):
    kw = fuzz_case(seed)
    case = cases.make_case(**kw)
    steps = 2 + seed % 2
    on, acts, m = run(case, steps=steps, skipping=True, row_constant=True)
    off, _, _ = run(case, steps=steps, skipping=False, row_constant=False)
    assert_bitwise(on, off, f"seed {seed}: {kw}")
    assert m.ctx.last_path()["level"] == 2, kw
    LAST_FUZZ.update(activity=acts[-1], row_constant_rows=m.row_constant_rows() if case["g"].metric_kind == "full" else None)


LAST_FUZZ = {}


# ---- the fold band's fused launches (round 6b; csrc/csi_fold.hip band_substeps_fused, evp_fast.hip k_band_stress / k_band_uv) -----------
# Four launches per pair of sub-steps (stress; both velocity components in one launch, the first evaluated again at the points the
# second reads; no copy kernels) against the three kernels + two copies of rounds 4-6a (CSI_BAND_FUSED=0): every field, halo cells
# included, bit for bit -- uniform / per-row / per-point coefficients, masks, array forcing, free drift, user forcing, odd and even
# numbers of sub-steps (the trailing single sub-step; both orders of the components in either position).
BAND_CASES = {
    "rectilinear": dict(Nx=200, Ny=160, topo=("periodic", "folded"), patches=True, random_uv=0.03, substeps=10),
    "rectilinear_odd": dict(Nx=200, Ny=160, topo=("periodic", "folded"), patches=True, random_uv=0.03, substeps=11),
    "latlon": dict(Nx=200, Ny=160, topo=("periodic", "folded"), grid="latlon", patches=True, random_uv=0.03, substeps=10),
    "latlon_masked_forced": dict(Nx=200, Ny=160, topo=("periodic", "folded"), grid="latlon", patches=True, random_uv=0.03, land=0.3, field_forcing=True, substeps=9),
    "curvilinear": dict(Nx=200, Ny=160, topo=("periodic", "folded"), patches=True, random_uv=0.03, curvilinear=0.05, substeps=10),
    "curvilinear_masked": dict(Nx=224, Ny=170, topo=("periodic", "folded"), patches=True, random_uv=0.03, curvilinear=0.05, land=0.3, substeps=13),
    "tripolar_like": dict(Nx=336, Ny=260, topo=("periodic", "folded"), patches=True, random_uv=0.03, curvilinear=0.05, land=0.3, field_forcing=True, free_drift=True,
                          substeps=12),
    "tripolar_like_user_forcing": dict(Nx=224, Ny=170, topo=("periodic", "folded"), patches=True, random_uv=0.03, curvilinear=0.05, land=0.3, field_forcing=True,
                                       free_drift=True, user_forcing=True, substeps=7),
    "tripolar_like_immersed_bc": dict(Nx=224, Ny=170, topo=("periodic", "folded"), patches=True, random_uv=0.03, curvilinear=0.05, land=0.3,
                                      immersed_bc=((1e-3, -2e-3, 3e-3, 1e-3), (2e-3, 1e-3, -1e-3, 2e-3)), substeps=7),
    "rectilinear_user_forcing": dict(Nx=200, Ny=160, topo=("periodic", "folded"), patches=True, random_uv=0.03, user_forcing=True, land=0.2, substeps=8),
    "tripolar_grid": dict(Nx=240, Ny=180, grid="tripolar", tripolar=dict(southernmost_latitude=-70.0), patches=True, random_uv=0.03, field_forcing=True, free_drift=True,
                          coriolis_points=True, land=0.2, substeps=10),
    "tripolar_grid_wind": dict(Nx=240, Ny=180, grid="tripolar", tripolar=dict(southernmost_latitude=-70.0), patches=True, random_uv=0.03, wind_drag="arrays",
                               coriolis_points=True, ice_edge=40.0, substeps=15),
    "wide_halo": dict(Nx=200, Ny=160, H=6, topo=("periodic", "folded"), patches=True, random_uv=0.03, curvilinear=0.05, land=0.2, substeps=8),
}


def band_run(case, fused, monkeypatch, steps=2, **kw):
    if fused: monkeypatch.delenv("CSI_BAND_FUSED", raising=False)
    else: monkeypatch.setenv("CSI_BAND_FUSED", "0")
    out, _, m = run(case, steps=steps, **kw)          # (the knob is read when the context is created)
    return out, m


@pytest.mark.parametrize("name", list(BAND_CASES))
def test_fused_band_launches_are_bit_identical(name, monkeypatch):
    case = cases.make_case(**BAND_CASES[name])
    a, ma = band_run(case, True, monkeypatch)
    b, mb = band_run(case, False, monkeypatch)
    if not ma.ctx.last_path()["fused"]:
        pytest.skip("this configuration does not take the fold band at all (three kernels on the whole grid)")
    assert mb.ctx.last_path()["fused"]
    la, lb = ma.ctx.last_launches()[0], mb.ctx.last_launches()[0]
    assert la < lb, (la, lb)                              # four launches per band step instead of eight
    assert_bitwise(a, b, f"{name}: fused band launches vs three kernels + copies")


def test_fused_band_with_rk3_whole_steps_and_cuts(monkeypatch):
    # whole time steps (dynamics + advection, an odd and an even first sub-step index through the RK3 stages) on the tripolar grid
    case = cases.make_case(Nx=240, Ny=180, grid="tripolar", tripolar=dict(southernmost_latitude=-70.0), patches=True, random_uv=0.03, field_forcing=True,
                           free_drift=True, coriolis_points=True, land=0.2, ice_edge=45.0, substeps=9)
    outs = []
    for fused in (True, False):
        if fused: monkeypatch.delenv("CSI_BAND_FUSED", raising=False)
        else: monkeypatch.setenv("CSI_BAND_FUSED", "0")
        m = cases.csi_model(case, mode="fast", timestepper="SplitRungeKutta3", advection=csi.WENO(order=5))
        for _ in range(2): csi.time_step(m, 600.0)
        m.synchronize()
        st = parents(m)
        st["h"] = m.ice_thickness.numpy().copy(); st["a"] = m.ice_concentration.numpy().copy()
        outs.append(st)
    assert_bitwise(outs[0], outs[1], "RK3 whole steps")
    for k in ("h", "a"):
        assert np.array_equal(outs[0][k], outs[1][k], equal_nan=True), k


def band_fuzz_case(seed):
    """A random north-fold configuration (every coefficient kind, masks, forcing kinds, halo widths, sub-step counts)."""
    rng = np.random.default_rng(7000 + seed)
    kw = dict(Nx=2 * int(rng.integers(70, 200)), Ny=int(rng.integers(110, 260)), H=int([4, 4, 5, 6][rng.integers(0, 4)]), topo=("periodic", "folded"),
              patches=bool(rng.integers(0, 2)), random_uv=0.03, substeps=int(rng.integers(2, 19)), seed=int(seed))
    kind = rng.integers(0, 4)
    if kind == 1: kw["grid"] = "latlon"
    elif kind == 2: kw["curvilinear"] = float([0.0, 0.03, 0.06][rng.integers(0, 3)])
    elif kind == 3:
        kw.update(grid="tripolar", tripolar=dict(southernmost_latitude=float(rng.uniform(-75.0, -40.0))), coriolis_points=bool(rng.integers(0, 2)))
        kw["Ny"] = max(kw["Ny"], 140)
    if rng.random() < 0.6: kw["land"] = float(rng.uniform(0.1, 0.4))
    r = rng.random()
    if r < 0.45:
        kw["field_forcing"] = True
        kw["free_drift"] = bool(rng.integers(0, 2))
    elif r < 0.6: kw["wind_drag"] = ["numbers", "arrays"][rng.integers(0, 2)]
    elif r < 0.7: kw["user_forcing"] = True
    if rng.random() < 0.3:
        a = float(rng.uniform(0.0, 0.5))
        kw["ice_free_rows"] = (a, min(1.0, a + float(rng.uniform(0.3, 0.5))))
    if kind != 3 and rng.random() < 0.25: kw["beta"] = 2e-10 if kind != 2 else None
    return {k: v for k, v in kw.items() if v is not None}


@pytest.mark.parametrize("seed", range(16))
def test_band_fuzz_bitwise(seed, monkeypatch):
    kw = band_fuzz_case(seed)
    case = cases.make_case(**kw)
    a, ma = band_run(case, True, monkeypatch, steps=2 + seed % 2)
    b, mb = band_run(case, False, monkeypatch, steps=2 + seed % 2)
    assert_bitwise(a, b, f"seed {seed}: {kw}")
    LAST_FUZZ.update(band=bool(ma.ctx.last_path()["fused"]) and ma.ctx.last_launches()[0] < mb.ctx.last_launches()[0])
