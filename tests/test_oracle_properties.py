"""CPU tests that pin the oracle with what the reference's own tests pin for this path
(SURVEY.md 8c): the strain / stress-divergence adjoint identity, the drag bound, the slab
known answer; plus the bit-for-bit cross-check of the two independent restatements."""
import ctypes as C

import numpy as np
import pytest

import cases
import climaseaice_jl_amd as csi
import oracle as O
import oracle_np as ONP


def _latlon_problem(N):
    g = csi.LatitudeLongitudeGrid((N, N), longitude=(0, 60), latitude=(20, 70), topology=(csi.Bounded, csi.Bounded), halo=(4, 4))
    p = O.Problem(N, N, 4, 4, (O.BOUNDED, O.BOUNDED), per_j=g.metrics())
    return g, p


@pytest.mark.parametrize("N", [40, 80])
def test_discrete_energy_budget_of_the_stress_divergence(N, oracle_lib):
    """test/test_rheology_energy_budget.jl:50-125: sum u d_j s_1j Az + v d_j s_2j Az = - sum sigma:eps Az to 1e-10
    on a 40^2 / 80^2 lat-lon grid (lon 0..60, lat 20..70, halo 4); the old flux-form operator must fail (> 1e-3)."""
    g, p = _latlon_problem(N)
    lam = lambda l: (l - 0) / 60 * 2 * np.pi
    phi = lambda f: (f - 20) / 50 * 2 * np.pi

    def set_smooth(name, LX, LY, fn, margin=2):
        x, y = g.xnodes(LX), g.ynodes(LY)
        a = p.interior(name)
        a[...] = 0
        # 1-based i in 1+margin : Nx-margin
        sl_i = slice(margin, N - margin)
        sl_j = slice(margin, N - margin)
        a[sl_j, sl_i] = fn(x[None, sl_i], y[sl_j, None])

    set_smooth("u", csi.Face, csi.Center, lambda l, f: np.sin(2 * lam(l)) * np.cos(3 * phi(f)))
    set_smooth("v", csi.Center, csi.Face, lambda l, f: np.cos(3 * lam(l)) * np.sin(2 * phi(f)))
    set_smooth("s11", csi.Center, csi.Center, lambda l, f: np.sin(lam(l)) * np.sin(2 * phi(f)))
    set_smooth("s22", csi.Center, csi.Center, lambda l, f: np.cos(2 * lam(l)) * np.cos(phi(f)))
    set_smooth("s12", csi.Face, csi.Face, lambda l, f: np.sin(3 * lam(l)) * np.cos(2 * phi(f)))
    L, P = p.L, p.ptr
    Wn = Wo = D = 0.0
    C_, F_ = O.CENTER, O.FACE
    u, v, s11, s22, s12 = (p.interior(k) for k in ("u", "v", "s11", "s22", "s12"))
    for i in range(1, N + 1):
        for j in range(1, N + 1):
            Wn += u[j - 1, i - 1] * L.ora_div_sigma_1(P, i, j) * L.ora_az(P, F_, C_, i, j)
            Wn += v[j - 1, i - 1] * L.ora_div_sigma_2(P, i, j) * L.ora_az(P, C_, F_, i, j)
            Wo += u[j - 1, i - 1] * L.ora_old_div_sigma_1(P, i, j) * L.ora_az(P, F_, C_, i, j)
            Wo += v[j - 1, i - 1] * L.ora_old_div_sigma_2(P, i, j) * L.ora_az(P, C_, F_, i, j)
            D += s11[j - 1, i - 1] * L.ora_strain_xx(P, i, j) * L.ora_az(P, C_, C_, i, j)
            D += s22[j - 1, i - 1] * L.ora_strain_yy(P, i, j) * L.ora_az(P, C_, C_, i, j)
            D += 2 * s12[j - 1, i - 1] * L.ora_strain_xy(P, i, j) * L.ora_az(P, F_, F_, i, j)
    imb = lambda W: abs(W + D) / max(abs(W), abs(D))
    assert imb(Wn) < 1e-10
    assert imb(Wo) > 1e-3
    assert imb(Wn) < 1e-6 * imb(Wo)


def test_semi_implicit_ocean_drag_bound(oracle_lib):
    """test/test_time_stepping.jl:56-80: 8x8 periodic 10 km box, h = aice = 1, ocean u = 0.1 through
    SemiImplicitStress, 20 steps of 60 s, substeps = 10: finite, 0 < max(u) <= 0.1 (RK3 and FE)."""
    for stepper in ("rk3", "fe"):
        p = O.Problem(8, 8, 4, 4, (O.PERIODIC, O.PERIODIC), dx=1250.0, dy=1250.0, substeps=10)
        p.set_stress("bottom", O.STRESS_SEMI_IMPLICIT, ue=0.1)
        p.f["h"][...] = 1.0
        p.f["aice"][...] = 1.0
        p.update_state()
        for n in range(20):
            p.time_step_rk3(60.0, 0) if stepper == "rk3" else p.time_step_fe(60.0, 0, n == 0)
        u = p.interior("u")
        assert np.all(np.isfinite(u)) and u.max() > 0 and u.max() <= 0.1


def test_freezing_bucket_first_step_known_answer(oracle_lib):
    """Config 1 (examples/freezing_bucket.jl:43-99): h = aice = 0 at t = 0, Tu = -10 prescribed, bottom flux
    -(1 - aice), dt = 600 s.  Closed form from thermodynamic_time_step.jl:304-324,358-370:
    V = dt / (900 * 334e3), aice = V / 0.05, h = 0.05."""
    h, a, mf = O.slab_step(np.zeros(1), np.zeros(1), 600.0, c_ice=2100.0, Tu=-10.0, top_flux_kind=1,
                           bot_flux_kind=1, Qb=1.0)
    V = 600.0 / (900.0 * 334e3)
    assert np.isclose(a[0], V / 0.05, rtol=1e-14) and np.isclose(h[0], 0.05, rtol=1e-14)
    assert np.isclose(mf[0], 900.0 * V / 600.0, rtol=1e-13)


def test_slab_mass_closure_and_melt_to_extinction(oracle_lib):
    """test/test_thermodynamic_mass_fluxes.jl:51-170: the recorded ice mass flux closes the mass change; a slab that
    melts completely ends at exact zeros."""
    h0, a0 = np.array([1.0, 0.3, 0.02]), np.array([1.0, 0.5, 0.2])
    h, a, mf = O.slab_step(h0, a0, 3600.0, top_flux_kind=0, Qu=100.0, Qb=10.0, Tu=-5.0)
    assert np.allclose(mf, 900.0 * (h * a - h0 * a0) / 3600.0, rtol=0, atol=1e-12)
    h, a, mf = O.slab_step(np.array([0.01]), np.array([0.5]), 3600.0 * 24 * 30, top_flux_kind=0, Qu=-500.0, Qb=500.0, Tu=0.0)
    assert h[0] == 0.0 and a[0] == 0.0


CROSS = {
    "periodic": dict(topo=("periodic", "periodic")),
    "bounded": dict(topo=("bounded", "bounded")),
    "channel": dict(topo=("periodic", "bounded")),
    "latlon": dict(topo=("bounded", "bounded"), grid="latlon"),
    "beta_bounded": dict(topo=("bounded", "bounded"), beta=3e-10),
    "beta_channel": dict(topo=("periodic", "bounded"), beta=-2e-10),
}


@pytest.mark.parametrize("free_drift", [False, True])
@pytest.mark.parametrize("name", list(CROSS))
def test_c_oracle_equals_numpy_restatement_bitwise(name, free_drift, oracle_lib):
    c = cases.make_case(Nx=28, Ny=22, substeps=6, random_uv=0.05, ue=0.05, ve=-0.02, top=(0.01, 0.02), free_drift=free_drift,
                        **CROSS[name])
    p = cases.oracle_problem(c)
    g = c["g"]
    topo = tuple(0 if t == "periodic" else 1 for t in c["topo"])
    m = g.metrics()
    n = ONP.NP(g.Nx, g.Ny, g.Hx, g.Hy, topo, dx=m.get("dx"), dy=m.get("dy"), per_j=m if m["kind"] == "per_j" else None)
    n.f_rows = cases.coriolis_rows(c, g)
    n.f, n.top, n.bottom = c["coriolis"], ("const",) + tuple(c["top"]), ("semi", c["ue"], c["ve"], 1026.0, 5.5e-3)
    n.free_drift = free_drift
    for k, v in p.f.items():
        n.fld[k] = v.copy()
    p.initialize_rheology()
    n.initialize()
    assert np.abs(p.f["P"] - n.fld["P"]).max() <= 4e-16 * np.abs(p.f["P"]).max()   # exp(): libm vs numpy
    n.fld["P"][...] = p.f["P"]
    p.L.ora_fill_halo_u(p.ptr); p.L.ora_fill_halo_v(p.ptr)
    n.fill_halo("u", 1, 0); n.fill_halo("v", 0, 1)
    p.subcycle(c["dt"], 1, 6)
    n.subcycle(c["dt"], 1, 6)
    for k in ("u", "v", "s11", "s22", "s12", "alpha", "zeta_c", "zeta_f", "Delta"):
        assert np.array_equal(p.f[k], n.fld[k]), k


def test_beta_plane_rows_and_limits(oracle_lib):
    """BetaPlane: f = f0 + beta * y at the (Face, Center) / (Center, Face) nodes of each row (upstream Coriolis;
    reference test matrix test/test_time_stepping.jl:35).  Known values of the host mirror's rows, beta = 0 is the
    FPlane bit for bit, and the Coriolis term of one sub-step changes by exactly (f(y) - f0) * vbar * dtau."""
    bp = csi.BetaPlane(latitude=45)
    assert abs(bp.f0 - 2 * 7.292115e-5 * np.sin(np.pi / 4)) < 1e-19
    assert abs(bp.beta - 2 * 7.292115e-5 * np.cos(np.pi / 4) / 6371e3) < 1e-25
    assert abs(csi.FPlane(latitude=45).f - bp.f0) < 1e-19
    g = csi.RectilinearGrid((8, 6), x=(0, 8e3), y=(1e3, 7e3), topology=(csi.Bounded, csi.Bounded), halo=(3, 3))
    fu, fv = csi.BetaPlane(f0=1e-4, beta=2e-10).rows(g)
    j = np.arange(1 - 3, 6 + 3 + 2)
    assert fu.shape == fv.shape == (6 + 2 * 3 + 1,)
    assert np.array_equal(fu, 1e-4 + 2e-10 * (1e3 + (j - 0.5) * 1e3)) and np.array_equal(fv, 1e-4 + 2e-10 * (1e3 + (j - 1) * 1e3))
    gp = csi.RectilinearGrid((8, 6), x=(0, 8e3), y=(0, 6e3), topology=(csi.Periodic, csi.Periodic), halo=(3, 3))
    fup, _ = csi.BetaPlane(f0=1e-4, beta=2e-10).rows(gp)
    assert np.array_equal(fup[:3], fup[6:9]) and np.array_equal(fup[9:12], fup[3:6])      # halo rows image their owners
    # beta = 0 == FPlane
    out = {}
    for beta in (None, 0.0):
        c = cases.make_case(Nx=24, Ny=20, substeps=5, topo=("bounded", "bounded"), random_uv=0.05, beta=beta)
        p = cases.oracle_problem(c)
        p.time_step_momentum(c["dt"])
        out[beta] = {k: p.f[k].copy() for k in ("u", "v", "s11")}
    for k in out[None]:
        assert np.array_equal(out[None][k], out[0.0][k]), k
    # a beta plane rotates the ice differently where y is large: the u difference after one sub-step is the Coriolis
    # difference, row by row (same sigma, same drag: only the -f x U term differs)
    c0 = cases.make_case(Nx=24, Ny=20, substeps=1, topo=("periodic", "bounded"), patches=False, v0=0.2, u0=0.0, beta=None)
    c1 = dict(c0, beta=4e-10)
    u = {}
    for key, c in (("f", c0), ("b", c1)):
        p = cases.oracle_problem(c)
        p.time_step_momentum(c["dt"])
        u[key] = p.interior("u").copy()
    d = u["b"] - u["f"]
    yc = c0["g"].ynodes(csi.Center)
    rel = d[2:-2, :] / (4e-10 * yc[2:-2, None])            # proportional to y (away from the walls, where vbar = v0)
    assert np.all(d[2:-2] > 0) and rel.std() / rel.mean() < 2e-2, (rel.mean(), rel.std())


def _bounded_advection_case(topo, scheme_seed=0):
    rng = np.random.default_rng(5 + scheme_seed)
    c = cases.make_case(Nx=40, Ny=32, H=4, topo=topo, spacing=1000.0, patches=False, noise=0.0)
    g = c["g"]
    nyu, nxu = c["u"].shape
    nyv, nxv = c["v"].shape
    c["u"] = 0.4 * rng.standard_normal((nyu, nxu))
    c["v"] = 0.4 * rng.standard_normal((nyv, nxv))
    if topo[0] == "bounded":
        c["u"][:, 0] = 0.0; c["u"][:, -1] = 0.0          # impenetrable walls
    if topo[1] == "bounded":
        c["v"][0, :] = 0.0; c["v"][-1, :] = 0.0
    c["h"] = 0.3 + 0.2 * rng.random(c["h"].shape)
    c["a"] = np.clip(0.5 + 0.6 * rng.random(c["a"].shape), 0, 1)
    return c


@pytest.mark.parametrize("scheme", [7, 5, -5])
@pytest.mark.parametrize("topo", [("bounded", "bounded"), ("periodic", "bounded"), ("bounded", "periodic")])
def test_advection_next_to_walls_reduces_its_order_and_conserves(topo, scheme, oracle_lib):
    """Boundary-order reduction (upstream topologically_conditional_interpolation, recalled): next to a wall the
    reconstruction drops to the highest order whose biased stencil stays inside the domain.  Pinned here by two
    properties that do not depend on the recollection of the coefficients: (1) no stencil reads across a wall --
    1e300 planted in every cell beyond the walls never reaches a tendency (the wall faces themselves multiply their
    halo neighbour by u = 0, as upstream does); (2) with closed walls the flux-form divergence conserves the
    tracer sums to rounding."""
    c = _bounded_advection_case(topo)
    p = cases.oracle_problem(c)
    H, Nx, Ny = c["H"], c["Nx"], c["Ny"]
    for k in ("h", "aice"):
        a = p.f[k]
        if topo[0] == "bounded":
            a[:, :H] = 1e300; a[:, H + Nx:] = 1e300
        if topo[1] == "bounded":
            a[:H, :] = 1e300; a[H + Ny:, :] = 1e300
    p.compute_tracer_tendencies(scheme)
    for k in ("Gh", "Ga"):
        G = p.interior(k)
        assert np.all(np.isfinite(G)) and np.abs(G).max() < 1.0, (k, topo, scheme, np.abs(G).max())
        assert np.abs(G).max() > 0
        assert abs(G.sum()) <= 1e-9 * np.abs(G).sum(), (k, G.sum())        # uniform cells: plain sums


@pytest.mark.parametrize("scheme", [7, 5, -5])
def test_boundary_order_reduction_rule(scheme, oracle_lib):
    """Which faces fall back to first-order upwind, read off a linear profile on a bounded line of N = 12 cells: every
    scheme of order >= 3 reconstructs it exactly whatever its nonlinear weights (all candidate stencils agree), first
    order returns the upwind cell value.  Flow towards +x: only face 2 (one upwind cell before the wall) is first
    order; flow towards -x: only face N."""
    N = 12
    for sign in (+1.0, -1.0):
        c = cases.make_case(Nx=N, Ny=8, H=4, topo=("bounded", "periodic"), spacing=1.0, patches=False, noise=0.0)
        xc = np.arange(N) + 0.5
        prof = 1.0 + 0.25 * xc
        c["h"] = np.broadcast_to(prof[None, :], c["h"].shape).copy()
        c["a"] = np.ones_like(c["h"])
        c["u"] = sign * np.ones_like(c["u"]); c["u"][:, 0] = 0.0; c["u"][:, -1] = 0.0
        c["v"] = np.zeros_like(c["v"])
        p = cases.oracle_problem(c)
        got = np.array([p.L.ora_weno_flux_x(p.ptr, scheme, p.field_struct("h"), int(i), 1) for i in range(2, N + 1)]) / sign
        exact = 1.0 + 0.25 * np.arange(1, N)                    # the profile at faces 2 .. N (x = 1 .. N-1)
        upwind = prof[:-1] if sign > 0 else prof[1:]            # cell on the upwind side of each face
        first_order = np.zeros(N - 1, dtype=bool)
        first_order[0 if sign > 0 else -1] = True
        assert np.allclose(got[~first_order], exact[~first_order], rtol=0, atol=1e-13)
        assert np.array_equal(got[first_order], upwind[first_order])


def _masked_advection_case(topo, seed=0):
    rng = np.random.default_rng(11 + seed)
    c = cases.make_case(Nx=48, Ny=40, H=4, topo=topo, spacing=1000.0, patches=False, noise=0.0, land=0.3)
    wet = c["mask"].astype(bool)
    c["u"] = 0.4 * rng.standard_normal(c["u"].shape)
    c["v"] = 0.4 * rng.standard_normal(c["v"].shape)
    if topo[0] == "bounded":
        c["u"][:, 0] = 0.0; c["u"][:, -1] = 0.0
    if topo[1] == "bounded":
        c["v"][0, :] = 0.0; c["v"][-1, :] = 0.0
    c["h"] = np.where(wet, 0.3 + 0.2 * rng.random(wet.shape), 0.0)
    c["a"] = np.where(wet, np.clip(0.5 + 0.6 * rng.random(wet.shape), 0, 1), 0.0)
    return c


@pytest.mark.parametrize("scheme", [7, 5, -5, 3, -3, 1])
@pytest.mark.parametrize("topo", [("periodic", "periodic"), ("periodic", "bounded"), ("bounded", "bounded")])
def test_advection_next_to_immersed_cells(topo, scheme, oracle_lib):
    """ImmersedBoundaryGrid (the reference's horizontal_div_Uc goes through upstream's _advective_tracer_flux_x/y,
    sea_ice_advection.jl:1-5,51-54): no flux through faces next to land, and a reconstruction never reads a land
    cell (its order drops until the stencil fits: recalled upstream rule).  Land cells poisoned with
    1e300 therefore change nothing, whatever the velocities at the coast; the scheme conserves the tracer."""
    c = _masked_advection_case(topo)
    wet = c["mask"].astype(bool)
    G = {}
    for poison in (0.0, 1e300):
        p = cases.oracle_problem(c)
        for k in ("h", "aice"):
            p.interior(k)[~wet] = poison
        for k in ("h", "aice"):                         # halos again (update_state would wipe the poison)
            p.L.ora_fill_halo_center(p.ptr, p.field_struct(k))
        p.compute_tracer_tendencies(scheme)
        G[poison] = {k: p.interior(k).copy() for k in ("Gh", "Ga")}
    for k in ("Gh", "Ga"):
        a, b = G[0.0][k], G[1e300][k]
        assert np.all(np.isfinite(b[wet])), (k, topo, scheme)
        assert np.array_equal(a[wet], b[wet]), (k, topo, scheme, np.abs(a[wet] - b[wet]).max())
        assert np.abs(a[wet]).max() > 0
        assert abs(a[wet].sum()) <= 1e-9 * np.abs(a[wet]).sum(), (k, a[wet].sum())      # uniform cells: plain sums
        assert np.all(a[~wet] == 0.0)                                                    # every face of a land cell is closed


@pytest.mark.parametrize("scheme", [7, 5, -5])
def test_immersed_order_reduction_rule(scheme, oracle_lib):
    """Which order each face gets next to an immersed cell, read off a cubic profile on a periodic line with one land
    cell at i = L: buffer B (order 2B-1) needs the 2B cells i-B .. i+B-1 around face i to be wet.  A cubic is exact
    for order >= 5 linear weights only approximately under WENO, so the test uses the schemes' exactness classes on a
    LINEAR profile (orders >= 3 exact, order 1 = upwind value) plus the count of first-order faces: faces L-1 and L+2
    (one wet cell between face and land on one side) are first order, L and L+1 are closed, the rest exact."""
    N, L = 24, 12
    for sign in (+1.0, -1.0):
        c = cases.make_case(Nx=N, Ny=8, H=4, topo=("periodic", "periodic"), spacing=1.0, patches=False, noise=0.0)
        prof = 1.0 + 0.25 * (np.arange(N) + 0.5)
        c["h"] = np.broadcast_to(prof[None, :], c["h"].shape).copy()
        c["a"] = np.ones_like(c["h"])
        c["u"] = sign * np.ones_like(c["u"]); c["v"] = np.zeros_like(c["v"])
        wet = np.ones((8, N), dtype=np.uint8); wet[:, L - 1] = 0          # land cell i = L (1-based)
        c["mask"] = wet
        c["u"][:, L - 1] = 0.0; c["u"][:, L] = 0.0                        # its two faces L, L+1
        p = cases.oracle_problem(c)
        faces = np.arange(6, N - 4)                                       # away from the periodic seam
        got = np.array([p.L.ora_weno_flux_x(p.ptr, scheme, p.field_struct("h"), int(i), 3) for i in faces]) / sign
        exact = 1.0 + 0.25 * (faces - 1.0)
        upwind = prof[faces - 2] if sign > 0 else prof[faces - 1]
        closed = (faces == L) | (faces == L + 1)
        first = (faces == L - 1) | (faces == L + 2)
        assert np.all(got[closed] == 0.0)
        assert np.array_equal(got[first], upwind[first]), (got[first], upwind[first])
        rest = ~(closed | first)
        assert np.allclose(got[rest], exact[rest], rtol=0, atol=1e-13)


def test_stress_balance_free_drift_closed_form(oracle_lib):
    """StressBalanceFreeDrift (stress_balance_free_drift.jl:73-95): where the ice is marginal (present, but below
    minimum_mass / minimum_concentration) the velocity after a step is U_e - tau / sqrt(rho_e C_D |tau|), whatever
    the rheology did; with `nothing` it is zero.  Constant wind stress, constant ocean velocity."""
    tau, ue, ve = (0.03, -0.04), 0.05, -0.02
    for fd in (False, True):
        c = cases.make_case(Nx=32, Ny=24, substeps=4, random_uv=0.02, top=tau, ue=ue, ve=ve, patches=True, free_drift=fd)
        p = cases.oracle_problem(c)
        p.time_step_momentum(c["dt"])
        h, a = p.interior("h"), p.interior("aice")
        m = 900.0 * h * a
        mi = 0.5 * (m + np.roll(m, 1, axis=1)); ai = 0.5 * (a + np.roll(a, 1, axis=1))       # u points (periodic)
        marginal_u = (mi > 2.3e-16) & (ai > 2.3e-16) & ~((mi >= 1.0) & (ai >= 1e-3))
        assert marginal_u.any()
        t = np.hypot(*tau)
        want = ue - tau[0] / np.sqrt(1026.0 * 5.5e-3 * t) if fd else 0.0
        got = p.interior("u")[marginal_u]
        assert np.allclose(got, want, rtol=1e-15, atol=0), (fd, got[:3], want)
        assert p.L.ora_free_drift_u(p.ptr, 3, 3) == (want if fd else 0.0) or fd


# ---- snow layer (thermodynamic_time_step.jl:131-298) -------------------------------------------------------------------
def test_snow_layer_reference_scenarios(oracle_lib):
    """The scenarios of test/test_snow_thermodynamics.jl on the oracle's layered step: flooding (:99-122), snowfall
    accumulation (:124-144), snow melts before ice (:146-169), interface temperature (:76-97)."""
    # flooding: heavy snow on thin ice -> ice grows, snow shrinks, mass moves from snow to ice
    r = O.layered_step([0.5], [1.0], [1.0], 1.0, O.make_slab(top_bc_kind=0, Tu=-5.0, top_flux_kind=0), O.make_snow())
    assert r["h"][0] > 0.5 and r["hs"][0] < 1.0
    assert abs(900.0 * (r["h"][0] - 0.5) + 330.0 * (r["hs"][0] - 1.0)) < 1e-10
    hf = r["h"][0] * (1 - 900.0 / 999.8) - r["hs"][0] * 330.0 / 999.8
    assert abs(hf) < 1e-13                                        # freeboard back to zero
    # snowfall: hs = Ps / rho_s * dt on full ice cover, recorded as intercepted snowfall
    r = O.layered_step([1.0], [1.0], [0.0], 3600.0, O.make_slab(top_bc_kind=1, top_flux_kind=0), O.make_snow(snowfall=1e-5))
    assert r["hs"][0] == 1e-5 / 330.0 * 3600.0 and r["mf_int"][0] == 1e-5 and r["h"][0] == 1.0
    # incoming heat melts snow first: hs drops by Q dt / (rho_s L), the ice is untouched, the surface sits at 0 C
    r = O.layered_step([2.0], [1.0], [0.1], 3600.0, O.make_slab(top_bc_kind=1, top_flux_kind=0, Qu=-100.0), O.make_snow())
    assert abs((0.1 - r["hs"][0]) - 100.0 * 3600.0 / (330.0 * 334e3)) < 1e-15 and r["h"][0] == 2.0 and r["tu_snow"][0] == 0.0
    # interface temperature of a cold prescribed surface: between Tu and Tb, resistors in series
    slab = O.make_slab(top_bc_kind=0, top_flux_kind=0, salinity=1.8 / 0.054)
    r = O.layered_step([1.0], [1.0], [0.3], 1.0, slab, O.make_snow(top_bc_kind=0, Tu=-10.0))
    Tb, Ri, Rs = -0.054 * (1.8 / 0.054), 1.0 / 2.0, 0.3 / 0.31
    assert abs(r["tu_ice"][0] - (Tb + (-10.0 - Tb) * Ri / (Rs + Ri))) < 1e-14 and -10.0 < r["tu_ice"][0] < Tb


@pytest.mark.parametrize("bc", [0, 1])
def test_snow_layer_mass_fluxes_close_and_bare_limit(bc, oracle_lib):
    """Random cells in every regime (open water, thin / consolidated ice, with and without snow, melting and
    freezing): the recorded mass fluxes close the ice and snow volume changes (thermodynamic_time_step.jl:293-297);
    with no snow and no snowfall the layered step is the bare-ice step with the same boundary condition."""
    rng = np.random.default_rng(23 + bc)
    n = 4000
    h = rng.random(n) * 2.0 * (rng.random(n) > 0.15)
    a = np.where(h > 0, rng.random(n), 0.0)
    hs = rng.random(n) * 0.5 * (rng.random(n) > 0.4) * (h > 0)
    dt = 600.0
    for Qu, Qb, Ps in ((-150.0, 5.0, 2e-5), (80.0, -10.0, 0.0), (0.0, 0.0, 1e-5)):
        slab = O.make_slab(top_bc_kind=bc, Tu=-8.0, top_flux_kind=0, Qu=Qu, Qb=Qb, salinity=30.0)
        snow = O.make_snow(top_bc_kind=bc, Tu=-8.0, snowfall=Ps)
        r = O.layered_step(h, a, hs, dt, slab, snow)
        for k in r:
            assert np.all(np.isfinite(r[k])), k
        assert np.all((r["aice"] >= 0) & (r["aice"] <= 1) & (r["h"] >= 0) & (r["hs"] >= 0))
        assert np.all(r["hs"][r["aice"] == 0] == 0)
        dVi = 900.0 * (r["h"] * r["aice"] - h * a) / dt
        dVs = 330.0 * (r["hs"] * r["aice"] - hs * a) / dt
        assert np.array_equal(r["mf_ice"], dVi)
        assert np.abs(r["mf_snow"] + r["mf_int"] - dVs).max() <= 1e-12 * max(1e-30, np.abs(dVs).max())
        assert np.array_equal(r["mf_int"], 330.0 * np.where(r["aice"] > 0, Ps / 330.0, 0.0) * r["aice"])
        assert (r["hs"] < hs).any() or Qu >= 0
    # bare limit
    slab = O.make_slab(top_bc_kind=bc, Tu=-8.0, top_flux_kind=0, Qu=-40.0, Qb=3.0, salinity=30.0)
    r = O.layered_step(h, a, 0 * hs, dt, slab, O.make_snow(top_bc_kind=bc, Tu=-8.0))
    hb, ab = h.copy(), a.copy()
    mf = np.zeros(n)
    O.lib().ora_slab_thermo_step(C.byref(slab), n, O._dptr(hb), O._dptr(ab), O._dptr(mf), dt)
    assert np.all(r["hs"] == 0)
    assert np.abs(r["h"] - hb).max() <= 1e-13 and np.abs(r["aice"] - ab).max() <= 1e-13


# ---- orthogonal curvilinear grids: twelve 2-D metric arrays (CSI_METRIC_FULL) ---------------------------------------------
@pytest.mark.parametrize("grid,topo", [("rectilinear", ("periodic", "periodic")), ("latlon", ("bounded", "bounded")),
                                       ("latlon", ("periodic", "bounded"))])
def test_full_2d_metrics_reduce_to_the_regular_grids(grid, topo, oracle_lib):
    """The operators read dx, dy, Az at the location and indices the reference's Oceananigans.Operators calls name
    (oracle: ora_dx / ora_dy / ora_az(g, lx, ly, i, j)).  Fed with the regular grid's own numbers spread over 2-D
    arrays, a whole RK3 step (WENO7 + EVP sub-cycle) is bit-identical to the regular-grid path; a smooth distortion of
    the arrays changes the answer (every array is actually read)."""
    out = {}
    for curv in (None, 0.0, 0.05):
        c = cases.make_case(Nx=28, Ny=24, substeps=8, topo=topo, grid=grid, random_uv=0.03, curvilinear=curv)
        p = cases.oracle_problem(c)
        p.time_step_rk3(c["dt"], 7)
        out[curv] = {k: p.f[k].copy() for k in ("u", "v", "h", "aice", "s11", "s12")}
    for k in out[None]:
        assert np.array_equal(out[None][k], out[0.0][k]), k
        assert np.all(np.isfinite(out[0.05][k]))
    assert np.abs(out[0.05]["u"] - out[None]["u"]).max() > 1e-6 * np.abs(out[None]["u"]).max()
    assert np.abs(out[0.05]["h"] - out[None]["h"]).max() > 1e-9


def test_no_slip_value_boundary_condition(oracle_lib):
    """ValueBoundaryCondition(0) on the tangential velocity (a13 of the scope table; examples/ice_advected_on_coastline.jl
    :96-99): the halo fill writes ONE halo cell, c[0] = 2 val - c[1]; deeper cells keep what they had; the wall corners
    then see the shear 2 u / dy instead of the free-slip zero, so sigma12 on the wall differs and the flow next to the
    wall is slowed down."""
    out = {}
    for noslip in (False, True):
        c = cases.make_case(Nx=24, Ny=16, substeps=30, topo=("periodic", "bounded"), patches=False, noise=0.0, u0=0.2, noslip=noslip)
        p = cases.oracle_problem(c)
        H = c["H"]
        u = p.f["u"]
        if noslip:
            assert np.array_equal(u[H - 1, H:-H], -u[H, H:-H]) and np.array_equal(u[H + 16, H:-H], -u[H + 15, H:-H])
            assert np.all(u[:H - 1, :] == 0.0) and np.all(u[H + 17:, :] == 0.0)          # deeper halo rows untouched
        else:
            assert np.array_equal(u[H - 1, H:-H], u[H, H:-H]) and np.array_equal(u[H - 2, H:-H], u[H + 1, H:-H])
        p.time_step_momentum(c["dt"])
        out[noslip] = {k: p.interior(k).copy() for k in ("u", "s12")}
    assert np.abs(out[True]["s12"][0, :]).max() > 10 * np.abs(out[False]["s12"][0, :]).max()     # wall corners, j = 1
    assert np.abs(out[True]["u"][0, :]).mean() < np.abs(out[False]["u"][0, :]).mean()            # first row slowed down


def test_rk3_stages_without_advection_reset_the_tracers(oracle_lib):
    """advection = nothing with SplitRungeKutta3: the tendencies are zero (sea_ice_advection.jl:50) but the tracer update
    still runs at every stage (dynamic_time_step!, sea_ice_rk_substep.jl:134-152) and resets h, aice to Psi^-; the
    stage-wise thermodynamic steps (dt / 3, dt / 2, dt) are therefore not cumulative: after the step h, aice are the
    ONE slab step of length dt applied to the state at the start of the step."""
    c = cases.make_case(Nx=20, Ny=16, substeps=4, topo=("periodic", "periodic"), patches=True, random_uv=0.02)
    slab = O.make_slab(Tu=-5.0, top_flux_kind=0, Qu=100.0, Qb=10.0)
    p = cases.oracle_problem(c)
    h0, a0 = p.interior("h").copy(), p.interior("aice").copy()
    p.time_step_rk3(c["dt"], 0, slab=slab)
    h1, a1, _ = O.slab_step(h0.ravel(), a0.ravel(), c["dt"], Tu=-5.0, top_flux_kind=0, Qu=100.0, Qb=10.0)
    assert np.array_equal(p.interior("h").ravel(), h1) and np.array_equal(p.interior("aice").ravel(), a1)
    assert np.abs(h1 - h0.ravel()).max() > 1e-6


def test_immersed_flux_bc_and_user_forcing_terms(oracle_lib):
    """Rare terms of the velocity tendencies (momentum_tendencies_kernel_functions.jl:31-36): the immersed-boundary stress
    divergence with FluxBoundaryCondition numbers (ice_stress_divergence.jl:65-123) and array-valued model.forcing
    (elasto_visco_plastic_rheology.jl:391-401).  Closed forms on a uniform grid: next to ONE immersed cell the u point
    west of it sees (q_E * dy) / (dx dy) with q_E = +east flux, the u point east of it -(−west flux) ... ; zeros reproduce
    the default bit for bit; a user forcing F moves u by dtau * F / (1 + dtau * tau_i) in one sub-step."""
    import cases
    import oracle as O
    c = cases.make_case(Nx=24, Ny=20, substeps=1, topo=("periodic", "periodic"), patches=False, random_uv=0.02)
    wet = np.ones((20, 24), dtype=bool)
    wet[10, 12] = False                          # one immersed cell: (i, j) = (13, 11) in 1-based indices
    c["mask"] = wet
    c["h"] = np.where(wet, c["h"], 0.0); c["a"] = np.where(wet, c["a"], 0.0)
    p = cases.oracle_problem(c)
    L = p.L
    L.ora_immersed_div_sigma_1.restype = O.C.c_double; L.ora_immersed_div_sigma_1.argtypes = [O.C.POINTER(O.ProblemStruct), O.C.c_int, O.C.c_int]
    L.ora_immersed_div_sigma_2.restype = O.C.c_double; L.ora_immersed_div_sigma_2.argtypes = [O.C.POINTER(O.ProblemStruct), O.C.c_int, O.C.c_int]
    dx = dy = 2000.0
    assert L.ora_immersed_div_sigma_1(p.ptr, 13, 11) == 0.0          # default: zero everywhere
    p.set_immersed_flux_bc("u", west=0.3, east=0.5, south=0.0, north=0.0)
    p.set_immersed_flux_bc("v", west=0.0, east=0.0, south=0.7, north=1.1)
    # u point i = 13 has the immersed cell on its EAST side (cell iE = 13): +east flux * Ax / V
    assert np.isclose(L.ora_immersed_div_sigma_1(p.ptr, 13, 11), 0.5 * dy / (dx * dy), rtol=1e-15)
    # u point i = 14 has it on its WEST side (cell iW = 13): -(−west flux) * Ax / V
    assert np.isclose(L.ora_immersed_div_sigma_1(p.ptr, 14, 11), 0.3 * dy / (dx * dy), rtol=1e-15)
    assert L.ora_immersed_div_sigma_1(p.ptr, 10, 5) == 0.0
    # v point j = 11 has the cell on its NORTH side (jN = 11), j = 12 on its SOUTH side (jS = 11)
    assert np.isclose(L.ora_immersed_div_sigma_2(p.ptr, 13, 11), 1.1 * dx / (dx * dy), rtol=1e-15)
    assert np.isclose(L.ora_immersed_div_sigma_2(p.ptr, 13, 12), 0.7 * dx / (dx * dy), rtol=1e-15)
    # zeros == default, bit for bit, through a whole sub-cycle
    c10 = dict(c, substeps=10)
    a = cases.oracle_problem(c10); a.time_step_momentum(c["dt"])
    b = cases.oracle_problem(dict(c10, immersed_bc=((0, 0, 0, 0), (0, 0, 0, 0)))); b.time_step_momentum(c["dt"])
    for k in ("u", "v", "s11", "s12"):
        assert np.array_equal(a.f[k], b.f[k])
    d = cases.oracle_problem(dict(c10, immersed_bc=((0.02, -0.01, 0.015, 0.005), (-0.01, 0.02, 0.01, -0.015)))); d.time_step_momentum(c["dt"])
    assert not np.array_equal(a.f["u"], d.f["u"]) and np.all(np.isfinite(d.f["u"]))
    # locality: after 2 sub-steps (dependency radius 2 cells each) the difference stays within 5 cells
    c2 = dict(c, substeps=2)
    a2 = cases.oracle_problem(c2); a2.time_step_momentum(c["dt"])
    d2 = cases.oracle_problem(dict(c2, immersed_bc=((0.02, -0.01, 0.015, 0.005), (-0.01, 0.02, 0.01, -0.015)))); d2.time_step_momentum(c["dt"])
    far = np.ones_like(wet); far[4:17, 6:20] = False
    assert np.array_equal(a2.interior("u")[far], d2.interior("u")[far]) and not np.array_equal(a2.interior("u"), d2.interior("u"))
    # user forcing: one sub-step, u' - u'_0 = dtau F / (1 + dtau tau_i) with the same dtau, tau_i (they depend on the old state)
    c1 = cases.make_case(Nx=24, Ny=20, substeps=1, topo=("periodic", "periodic"), patches=False, random_uv=0.02)
    q0 = cases.oracle_problem(c1); q0.time_step_momentum(c1["dt"])
    c1f = cases.make_case(Nx=24, Ny=20, substeps=1, topo=("periodic", "periodic"), patches=False, random_uv=0.02, user_forcing=True)
    q1 = cases.oracle_problem(c1f); q1.time_step_momentum(c1f["dt"])
    du = q1.interior("u") - q0.interior("u")
    F = c1f["force_u"]
    ratio = du / F                                                      # = dtau / (1 + dtau tau_i) > 0, about dt / alpha
    assert np.all(ratio > 0) and np.all(ratio < c1["dt"] / 50.0 * 1.05)     # (u sees the forced v through the Coriolis term: a few 1e-4 relative)
    assert np.all(np.isfinite(q1.f["v"]))


@pytest.mark.parametrize("loc", [("c", "c"), ("f", "c"), ("c", "f"), ("f", "f")])
@pytest.mark.parametrize("sign", [1, -1])
def test_north_fold_fill_matches_numpy_restatement(loc, sign, oracle_lib):
    """Zipper boundary condition of a TripolarGrid (sea_ice_model.jl:57-64; upstream fold semantics as recalled in
    oracle/csi_oracle.c fold_north): the C oracle's fill equals the independent numpy restatement (grids.fold_north) bit for
    bit at all four locations and both signs; the folded rows are periodic in x; a field that is symmetric under the fold
    map keeps a continuous image (row Ny + 1 of a Center-y field equals its row Ny - 1 partner)."""
    import oracle as O
    Nx, Ny, H = 24, 18, 4
    p = O.Problem(Nx, Ny, H, H, (O.PERIODIC, O.RIGHT_FOLDED), dx=1.0, dy=1.0)
    rng = np.random.default_rng(5)
    name = {("c", "c"): "h", ("f", "c"): "u", ("c", "f"): "v", ("f", "f"): "s12"}[loc]
    a = p.f[name]
    assert a.shape == (Ny + 2 * H, Nx + 2 * H)                 # no extra row: the fold side is not a wall
    a[...] = rng.standard_normal(a.shape)
    before = a.copy()
    lx, ly = (O.FACE if loc[0] == "f" else O.CENTER), (O.FACE if loc[1] == "f" else O.CENTER)
    p.L.ora_fill_halo_loc(p.ptr, p.field_struct(name), lx, ly, sign)
    # numpy: x periodic pass over the interior rows, south wall mirror for Center-y fields, then the fold
    want = before.copy()
    want[H:H + Ny, :H] = want[H:H + Ny, Nx:Nx + H]
    want[H:H + Ny, Nx + H:] = want[H:H + Ny, H:2 * H]
    if ly == O.CENTER:
        for m in range(1, H + 1):
            want[H - m, :] = want[H + m - 1, :]
    want = csi.fold_north(want, Nx, Ny, H, H, loc[0] == "f", loc[1] == "f", sign)
    assert np.array_equal(a, want)
    top = a[H + Ny:, :]
    assert np.array_equal(top[:, :H], top[:, Nx:Nx + H]) and np.array_equal(top[:, Nx + H:], top[:, H:2 * H])
    # spot values straight from the definition
    i, m = 5, 2
    ip = Nx - i + (2 if loc[0] == "f" else 1)
    js = Ny - m + (1 if loc[1] == "f" else 0)
    assert a[Ny + m + H - 1, i + H - 1] == sign * before[js + H - 1, ip + H - 1]
    if loc[0] == "f":                                         # column 1 folds onto itself without the sign change
        assert a[Ny + m + H - 1, 1 + H - 1] == abs(sign) * before[js + H - 1, 1 + H - 1]
