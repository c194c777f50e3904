"""Pins of the recalled upstream arithmetic of the advection half against the PUBLISHED algorithms (the
arithmetic-bearing dependency Oceananigans.jl is absent from /root/reference, so its algorithm is restated from the
literature and anchored on the reference's call sites: src/sea_ice_advection.jl:51-58, src/sea_ice_rk_substep.jl:81-152).

Everything the oracle's WENO code "knows" -- candidate stencils, optimal linear weights, smoothness indicators, the WENO-Z
weights and tau -- is derived here a SECOND time, independently and in exact rational arithmetic, from the definitions:

* Jiang & Shu (1996), J. Comput. Phys. 126: candidate k of WENO(2r - 1) is the value at the face of the degree r - 1 polynomial
  whose cell averages match r cells; the optimal weights d_k make sum d_k q_k the value of the degree 2r - 2 polynomial that
  matches all 2r - 1 cells; beta_k = sum_{l = 1}^{r - 1} int_cell dx^(2l - 1) (d^l p_k / dx^l)^2 dx over the upwind cell.
* Balsara & Shu (2000), J. Comput. Phys. 160: the same integrals for r = 4 (the 547 / 3882 / ... / 240 table).
* Borges et al. (2008), J. Comput. Phys. 227; Castro et al. (2011), J. Comput. Phys. 230: WENO-Z,
  alpha_k = d_k (1 + (tau / (beta_k + eps))^p), tau_5 = |beta_0 - beta_2|, tau_7 = |beta_0 + 3 beta_1 - 3 beta_2 - beta_3|.
* the three-stage low-storage stepping of the reference (dt / 3, dt / 2, dt from Psi^-: sea_ice_rk_substep.jl:81-94 and the
  upstream stage loop): on a linear tendency it is the third-order Taylor polynomial of exp(dt L).

None of this replaces a run of the reference (the oracle stays "parity unpinned"); it removes "one
author, three restatements" from the advection half: a wrong coefficient, weight, indicator or stage factor fails here.
"""
import ctypes as C
from fractions import Fraction as Fr

import numpy as np
import pytest

import cases

R_OF = {3: 2, 5: 3, 7: 4}


def _hook(O, order, p):
    L = O.lib()
    L.ora_test_weno.restype = C.c_int
    L.ora_test_weno.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    p = np.ascontiguousarray(p, dtype=np.float64)
    out = np.zeros(16)
    n = L.ora_test_weno(order, p.ctypes.data_as(C.POINTER(C.c_double)), out.ctypes.data_as(C.POINTER(C.c_double)))
    assert n > 0
    if order < 0:
        return out[0]
    r = R_OF[order % 100]                # (100 + order: the single-precision-weights variant)
    return dict(value=out[0], q=out[1:1 + r].copy(), beta=out[1 + r:1 + 2 * r].copy(), alpha=out[1 + 2 * r:1 + 3 * r].copy(), tau=out[1 + 3 * r])


# ---- exact derivations (cells of width 1, face at x = 0, upwind cell [-1, 0]; stencil value m of 2r - 1 is the cell
#      [m - r, m - r + 1]: upwind-most first, the order the oracle's weno* functions take) --------------------------------------
def _solve(A, b):
    """Gaussian elimination over Fractions."""
    n = len(A)
    M = [list(map(Fr, A[i])) + [Fr(b[i])] for i in range(n)]
    for c in range(n):
        piv = next(i for i in range(c, n) if M[i][c] != 0)
        M[c], M[piv] = M[piv], M[c]
        M[c] = [x / M[c][c] for x in M[c]]
        for i in range(n):
            if i != c and M[i][c] != 0:
                M[i] = [x - M[i][c] * y for x, y in zip(M[i], M[c])]
    return [M[i][n] for i in range(n)]


def _avg_row(a, deg):
    """cell average over [a, a + 1] of the monomials x^0 .. x^deg"""
    return [(Fr(a + 1) ** (m + 1) - Fr(a) ** (m + 1)) / (m + 1) for m in range(deg + 1)]


def _poly_from_averages(cells):
    """matrix C with poly coefficient c_m = sum_j C[m][j] * average_j for the polynomial matching the averages on `cells`"""
    n = len(cells)
    A = [_avg_row(a, n - 1) for a in cells]
    cols = [_solve(A, [1 if i == j else 0 for i in range(n)]) for j in range(n)]       # column j: coefficients for unit average j
    return [[cols[j][m] for j in range(n)] for m in range(n)]


def _candidate_cells(r, k):
    """oracle numbering: candidate 0 is the most DOWNWIND stencil (upwind cell, dw, dw + 1 ...), candidate r - 1 the most upwind;
    returns (indices into the 2r - 1 stencil values, left edges of the cells)"""
    idx = list(range(r - 1 - k, 2 * r - 1 - k))
    return idx, [m - r for m in idx]


def _published(r):
    n = 2 * r - 1
    q, beta = [], []
    for k in range(r):
        idx, cells = _candidate_cells(r, k)
        Cm = _poly_from_averages(cells)                               # c_m = sum_j Cm[m][j] avg_j
        row = [Fr(0)] * n
        for j, m in enumerate(idx):
            row[m] = Cm[0][j]                                         # p(0) = c_0
        q.append(row)
        # Jiang-Shu indicator: sum_l int_{-1}^{0} (p^(l))^2 dx as a quadratic form in the r averages
        Q = [[Fr(0)] * n for _ in range(n)]
        for l in range(1, r):
            # p^(l)(x) = sum_m c_m m! / (m - l)! x^(m - l)
            D = [[Fr(0)] * r for _ in range(r)]                       # D[e][j]: coefficient of x^e in p^(l), per unit average j
            for m in range(l, r):
                f = Fr(1)
                for t in range(l):
                    f *= (m - t)
                for j in range(r):
                    D[m - l][j] += f * Cm[m][j]
            for e1 in range(r):
                for e2 in range(r):
                    w = -Fr(-1) ** (e1 + e2 + 1) / (e1 + e2 + 1)      # int_{-1}^{0} x^(e1 + e2) dx
                    for j1 in range(r):
                        for j2 in range(r):
                            Q[idx[j1]][idx[j2]] += w * D[e1][j1] * D[e2][j2]
        beta.append(Q)
    # optimal weights: sum_k d_k q_k = the (2r - 1)-cell reconstruction
    big = _poly_from_averages([m - r for m in range(n)])[0]
    d = _solve([[q[k][m] for k in range(r)] for m in list(range(r - 1)) + [n - 1]], [big[m] for m in list(range(r - 1)) + [n - 1]])
    for m in range(n):
        assert sum(d[k] * q[k][m] for k in range(r)) == big[m]
    return q, d, beta, big


@pytest.mark.parametrize("order", [3, 5, 7])
def test_candidate_stencils_and_optimal_weights_are_the_published_ones(order, oracle_lib):
    """q_k: the degree r - 1 reconstructions from cell averages; d_k: 1/3, 2/3 | 1/10, 3/5, 3/10 | 1/35, 12/35, 18/35, 4/35."""
    r, n = R_OF[order], order
    q, d, _, _ = _published(r)
    assert sorted(d) == sorted({2: [Fr(1, 3), Fr(2, 3)], 3: [Fr(1, 10), Fr(3, 5), Fr(3, 10)], 4: [Fr(1, 35), Fr(12, 35), Fr(18, 35), Fr(4, 35)]}[r])
    for m in range(n):
        e = np.zeros(n); e[m] = 1.0
        got = _hook(oracle_lib, order, e)["q"]
        for k in range(r):
            assert abs(got[k] - float(q[k][m])) <= 2e-16, (order, k, m, got[k], q[k][m])
    # beta = 0, tau = 0 -> the unnormalised weights ARE the optimal weights (zero data: with a constant the decimal
    # coefficients of the WENO7 indicators leave rounding-level betas)
    a = _hook(oracle_lib, order, np.zeros(n))["alpha"]
    for k in range(r):
        assert a[k] == float(d[k]), (order, k, a[k], d[k])
    a = _hook(oracle_lib, order, np.ones(n))["alpha"]
    for k in range(r):
        assert abs(a[k] - float(d[k])) <= 1e-12, (order, k, a[k], d[k])


@pytest.mark.parametrize("order", [3, 5, 7])
def test_smoothness_indicators_are_the_jiang_shu_integrals(order, oracle_lib):
    """beta_k as quadratic forms, entry by entry, against sum_l int (d^l p_k)^2 over the upwind cell.  WENO3 / WENO5: equal;
    WENO7: Balsara-Shu's table is the integral x 240, upstream stores it / 1000 -> ONE common factor 0.24 for all four
    candidates (a common factor leaves the weights' ratios alone and only moves them relative to eps)."""
    r, n = R_OF[order], order
    _, _, beta, _ = _published(r)
    scale = Fr(240, 1000) if order == 7 else Fr(1)
    rng = np.random.default_rng(order)
    for k in range(r):
        # the oracle's quadratic form, recovered from evaluations at e_i and e_i + e_j
        diag = np.array([_hook(oracle_lib, order, np.eye(n)[i])["beta"][k] for i in range(n)])
        for i in range(n):
            assert abs(diag[i] - float(scale * beta[k][i][i])) <= 1e-14 * max(1.0, abs(diag[i])), (order, k, i)
            for j in range(i + 1, n):
                e = np.zeros(n); e[i] = 1.0; e[j] = 1.0
                cross = _hook(oracle_lib, order, e)["beta"][k] - diag[i] - diag[j]
                assert abs(cross - float(scale * 2 * beta[k][i][j])) <= 2e-14 * max(1.0, abs(cross)), (order, k, i, j, cross)
        # and on random data
        p = rng.standard_normal(n)
        want = float(scale) * sum(float(beta[k][i][j]) * p[i] * p[j] for i in range(n) for j in range(n))
        assert abs(_hook(oracle_lib, order, p)["beta"][k] - want) <= 1e-12 * max(1.0, abs(want))


@pytest.mark.parametrize("order", [3, 5, 7])
def test_weno_z_weights_and_tau(order, oracle_lib):
    """alpha_k = d_k (1 + (tau / (beta_k + eps))^2) with eps = 1e-8 and tau_3 = |b0 - b1|, tau_5 = |b0 - b2|,
    tau_7 = |b0 + 3 b1 - 3 b2 - b3| (Borges et al. 2008, Castro et al. 2011); the value is the alpha-weighted mean of the q_k."""
    r, n = R_OF[order], order
    _, d, _, _ = _published(r)
    rng = np.random.default_rng(100 + order)
    for _ in range(50):
        p = rng.standard_normal(n) * 10.0 ** rng.integers(-3, 3)
        o = _hook(oracle_lib, order, p)
        b = o["beta"]
        tau = abs(b[0] - b[1]) if order == 3 else (abs(b[0] - b[2]) if order == 5 else abs(b[0] + 3 * b[1] - 3 * b[2] - b[3]))
        assert o["tau"] == tau
        alpha = np.array([float(d[k]) * (1 + (tau / (b[k] + 1e-8)) ** 2) for k in range(r)])
        assert np.allclose(o["alpha"], alpha, rtol=1e-14, atol=0)
        assert abs(o["value"] - float(np.dot(alpha, o["q"]) / alpha.sum())) <= 1e-13 * max(1.0, np.abs(o["q"]).max())
    # tau is a combination of the betas that vanishes when every candidate sees the same polynomial: degree <= r - 1 data
    for deg in range(r):
        avg = np.array([float(_avg_row(m - r, deg)[deg]) for m in range(n)])
        o = _hook(oracle_lib, order, avg)
        assert o["tau"] <= 1e-12 * max(1.0, o["beta"].max()), (order, deg, o["tau"])


@pytest.mark.parametrize("order", [3, 5, 7])
def test_polynomial_exactness(order, oracle_lib):
    """(a) the nonlinear scheme reproduces polynomials of degree <= r - 1 exactly whatever its weights (every candidate does);
    (b) with the weights frozen at the optimal ones it reproduces degree <= 2r - 2 (2 / 4 / 6): order 2r - 1 = 3 / 5 / 7;
    (c) degree 2r - 1 is NOT reproduced (the leading error term is there); UpwindBiased(3 / 5) is (b) exactly."""
    r, n = R_OF[order], order
    _, d, _, _ = _published(r)
    dd = np.array([float(x) for x in d])
    for deg in range(2 * r):
        avg = np.array([float(_avg_row(m - r, deg)[deg]) for m in range(n)])      # averages of x^deg
        exact = 1.0 if deg == 0 else 0.0                                          # x^deg at the face x = 0
        o = _hook(oracle_lib, order, avg)
        scale = np.abs(avg).max()
        if deg <= r - 1:
            assert abs(o["value"] - exact) <= 1e-13 * scale, (order, deg, o["value"])
        frozen = float(np.dot(dd, o["q"]))
        if deg <= 2 * r - 2:
            assert abs(frozen - exact) <= 1e-12 * scale, (order, deg, frozen)
            if order in (3, 5):
                assert abs(_hook(oracle_lib, -order, avg) - exact) <= 1e-12 * scale
        else:
            assert abs(frozen - exact) > 1e-3, (order, deg, frozen)


@pytest.mark.parametrize("order,expect", [(3, 3), (5, 5), (7, 7)])
def test_observed_order_of_the_linear_scheme_and_weight_convergence(order, expect, oracle_lib):
    """Grid refinement on a smooth profile (exact cell averages of sin(x + 0.3) around a face away from critical points):
    the frozen-weight scheme converges with order 2r - 1; the WENO-Z value converges at least that fast down to rounding,
    and the normalised nonlinear weights tend to the optimal ones (tau -> 0 faster than the betas)."""
    r, n = R_OF[order], order
    _, d, _, _ = _published(r)
    dd = np.array([float(x) for x in d])
    x0 = 0.3
    errs_lin, errs_z, wdev, hs = [], [], [], []
    for N in (8, 16, 32, 64):
        h = 1.0 / N
        edges = x0 + h * (np.arange(n + 1) - r)
        avg = -(np.cos(edges[1:]) - np.cos(edges[:-1])) / h          # exact cell averages of sin
        o = _hook(oracle_lib, order, avg)
        errs_lin.append(abs(float(np.dot(dd, o["q"])) - np.sin(x0)))
        errs_z.append(abs(o["value"] - np.sin(x0)))
        wdev.append(np.abs(o["alpha"] / o["alpha"].sum() - dd).max())
        hs.append(h)
    slopes = [np.log(errs_lin[i] / errs_lin[i + 1]) / np.log(2.0) for i in range(3) if errs_lin[i + 1] > 1e-14]
    assert slopes and abs(slopes[0] - expect) < 0.35, (order, errs_lin, slopes)
    for i in range(3):
        if errs_lin[i + 1] > 1e-13:
            assert errs_z[i + 1] <= 4 * errs_lin[i + 1] + 1e-14, (order, errs_z, errs_lin)
    assert wdev[-1] < 1e-3 and wdev[-1] < 0.1 * wdev[0] + 1e-12, (order, wdev)


def test_bias_mirror_symmetry_through_the_flux_function(oracle_lib):
    """The wiring of ora_weno_flux_x / _y (what horizontal_div_Uc calls, sea_ice_advection.jl:51-58): left bias for U > 0,
    right bias for U <= 0 is the mirror image; the face value is the hook's value on the stencil read off the line."""
    N = 24
    rng = np.random.default_rng(7)
    prof = 1.0 + 0.3 * rng.random(N)
    for scheme in (3, 5, 7):
        r = R_OF[scheme]
        for sign in (+1.0, -1.0):
            c = cases.make_case(Nx=N, Ny=8, H=4, topo=("periodic", "periodic"), spacing=1.0, patches=False, noise=0.0)
            c["h"] = np.broadcast_to(prof[None, :], c["h"].shape).copy()
            c["a"] = np.ones_like(c["h"])
            c["u"] = sign * 0.5 * np.ones_like(c["u"]); c["v"] = np.zeros_like(c["v"])
            p = cases.oracle_problem(c)
            for i in (1, 5, N):                                              # face i = west face of cell i (1-based), periodic
                got = p.L.ora_weno_flux_x(p.ptr, scheme, p.field_struct("h"), i, 3) / (sign * 0.5)
                cells = [(i - 1 + m - r) % N for m in range(2 * r - 1)] if sign > 0 else [(i - 1 - (m - r) - 1) % N for m in range(2 * r - 1)]
                want = _hook(oracle_lib, scheme, prof[cells])["value"]
                assert got == want, (scheme, sign, i, got, want)


@pytest.mark.parametrize("scheme", [1, -3, -5])
def test_rk3_stage_factors_reproduce_the_third_order_taylor_polynomial(scheme, oracle_lib):
    """SplitRungeKutta3 as the reference steps the tracers (sea_ice_rk_substep.jl:81-94,134-152 and the upstream stage loop:
    Psi = Psi^- + (dt / beta) G(Psi_previous stage), beta = 3, 2, 1).  With a LINEAR scheme (UpwindBiased), a uniform
    prescribed velocity and no clipping the tendency is G = L h, and one step must be
    (1 + z + z^2 / 2 + z^3 / 6) h0, z = dt L -- third order, every stage restarting from Psi^-.  L is applied by the oracle's own
    tendency routine, so only the stage factors and the restart are under test."""
    N = 32
    c = cases.make_case(Nx=N, Ny=8, H=4, topo=("periodic", "periodic"), spacing=1000.0, patches=False, noise=0.0, substeps=0,
                        top=None, bottom=None, coriolis=None)
    xc = (np.arange(N) + 0.5) / N
    c["h"] = np.broadcast_to((1.0 + 0.1 * np.sin(2 * np.pi * xc) + 0.05 * np.cos(6 * np.pi * xc))[None, :], c["h"].shape).copy()
    c["a"] = np.broadcast_to((0.5 + 0.1 * np.cos(2 * np.pi * xc))[None, :], c["a"].shape).copy()
    c["u"] = 0.7 * np.ones_like(c["u"]); c["v"] = np.zeros_like(c["v"])
    dt = 400.0                                                               # CFL 0.28

    def G(field):
        q = cases.oracle_problem(c)
        q.interior("h")[...] = field
        q.update_state()
        q.compute_tracer_tendencies(scheme)
        return q.interior("Gh").copy()

    p = cases.oracle_problem(c)
    p.s.substeps = 0                                                         # prescribed velocities: no sub-cycle
    p.time_step_rk3(dt, scheme)
    h0 = c["h"]
    g1 = G(h0); g2 = G(g1); g3 = G(g2)                                       # L h0, L^2 h0, L^3 h0 (G is linear: G(G(h)) = L^2 h)
    want = h0 + dt * g1 + dt ** 2 / 2 * g2 + dt ** 3 / 6 * g3
    got = p.interior("h")
    assert np.abs(got - want).max() <= 1e-13, np.abs(got - want).max()
    # ... and it is NOT the second-order or the fourth-order polynomial (the cubic term matters at this CFL number)
    assert np.abs(got - (h0 + dt * g1 + dt ** 2 / 2 * g2)).max() > 1e-6
    assert np.array_equal(p.interior("u"), c["u"])


# ---- round 5: the weight-precision switch (weight_dtype f64 | f32; upstream's second float type FT2, recalled) --------------------
@pytest.mark.parametrize("wd", ["f64", "f32"])
@pytest.mark.parametrize("order", [3, 5, 7])
def test_c_oracle_equals_the_numpy_restatement_bit_for_bit_in_both_weight_precisions(order, wd, oracle_lib):
    """oracle/csi_oracle.c weno*(_f32) against oracle/oracle_np.weno_value on 4000 random stencils (smooth, rough, with plateaus
    and zeros): two restatements, written separately, the same bits -- in the double mode and in the single-precision-weights mode."""
    import oracle_np
    rng = np.random.default_rng(50 + order)
    r = R_OF[order]
    n = 2 * r - 1
    P = np.concatenate([0.3 + 0.005 * rng.standard_normal((1500, n)), rng.random((1500, n)), np.round(rng.random((600, n)) * 3) / 3,
                        1e-6 * rng.standard_normal((400, n))])
    P[3000:3100] = 0.7                                   # plateaus: beta = tau = 0
    want = oracle_np.weno_value(P, order, wd)
    got = np.array([_hook(oracle_lib, order + (100 if wd == "f32" else 0), p)["value"] for p in P])
    assert np.array_equal(got, want), (order, wd, np.abs(got - want).max(), int((got != want).sum()))


@pytest.mark.parametrize("order", [3, 5, 7])
def test_single_precision_weights_keep_the_published_scheme_to_single_precision(order, oracle_lib):
    """The f32 mode is the SAME scheme with its nonlinear weights rounded to float: indicators, tau and unnormalised weights agree
    with the double mode to a few float ulps of their scale, the face value to ~1e-6 of the stencil's spread (the candidates are
    convexly combined, so a weight error of 1e-7 moves the value by 1e-7 of the candidates' spread), and a polynomial of degree
    <= r - 1 is still reproduced whatever the weights (every candidate is exact for it), to the precision of their normalisation."""
    rng = np.random.default_rng(70 + order)
    r = R_OF[order]
    n = 2 * r - 1
    for _ in range(300):
        p = 0.3 + 0.05 * rng.standard_normal(n)
        a, b = _hook(oracle_lib, order, p), _hook(oracle_lib, order + 100, p)
        assert np.array_equal(a["q"], b["q"])                                            # candidates: double in both modes
        scale = np.abs(a["beta"]).max() + 1e-8
        assert np.abs(a["beta"] - b["beta"]).max() <= 2e-5 * scale + 5e-6 * np.abs(p).max() ** 2     # (float cancellation in the quadratic forms: coefficients up to 31 x p^2 x 6e-8)
        spread = np.abs(a["q"] - a["q"].mean()).max()
        assert abs(a["value"] - b["value"]) <= 2e-3 * spread + 1e-15, (a["value"], b["value"], spread)
    # smooth data (the regime the advection runs in): the two modes agree to a few 1e-6 of the value.  (Not better: converted to
    # float, values of 0.3 carry 2e-8 of rounding, the quadratic forms of the indicators 5e-9 of cancellation noise -- the size of
    # eps and far above the true beta ~ 1e-10 of such data -- so the single-precision weights wander between the candidates, whose
    # spread is ~1e-5 here.  A property of the mode, which is why the default stays double.)
    x = np.arange(n) - (r - 1)
    worst = 0.0
    for k in range(200):
        p = 0.3 + 0.005 * np.sin(0.05 * x + 0.1 * k)
        a, b = _hook(oracle_lib, order, p), _hook(oracle_lib, order + 100, p)
        worst = max(worst, abs(a["value"] - b["value"]) / np.abs(p).max())
    assert worst <= 5e-6, (order, worst)
    # polynomial exactness for degree <= r - 1 survives any weights -- to the precision of their NORMALISATION: the assumed form
    # divides the double sum of alpha_s q_s by the FLOAT sum of the alphas, which differs from their double sum by up to a float
    # ulp, so even a constant comes back with 6e-8 relative error (in the double mode: 1e-16).  Whether upstream's FT2 arithmetic
    # has this property is exactly what a reference run has to show.
    for deg in range(r):
        coef = rng.standard_normal(deg + 1)
        cellavg = lambda lo: sum(c * ((lo + 1) ** (m + 1) - lo ** (m + 1)) / (m + 1) for m, c in enumerate(coef))      # noqa: E731
        p = np.array([cellavg(float(m - r)) for m in range(n)])
        exact = coef[0]                                                                     # the polynomial at the face x = 0
        v = _hook(oracle_lib, order + 100, p)["value"]
        assert abs(v - exact) <= 2e-7 * (1 + np.abs(p).max()), (order, deg, v, exact)
        v64 = _hook(oracle_lib, order, p)["value"]
        assert abs(v64 - exact) <= 1e-12 * (1 + np.abs(p).max()), (order, deg, v64, exact)


def test_weight_precision_reaches_the_tendencies(oracle_lib):
    """ora_problem.weno_weights_f32 switches the reconstruction inside horizontal_div_Uc: the tendencies of the two modes differ (the
    switch is live) by no more than single precision allows, and the default is the double mode."""
    c = cases.make_case(Nx=40, Ny=32, substeps=2, topo=("periodic", "periodic"), random_uv=0.3)
    p = cases.oracle_problem(c)
    assert p.s.weno_weights_f32 == 0
    p.compute_tracer_tendencies(7)
    g64 = p.f["Gh"].copy()
    p.s.weno_weights_f32 = 1
    p.compute_tracer_tendencies(7)
    g32 = p.f["Gh"].copy()
    assert not np.array_equal(g64, g32)
    assert np.abs(g64 - g32).max() <= 1e-4 * np.abs(g64).max()
