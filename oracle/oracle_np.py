"""Independent NumPy restatement of the EVP sub-cycle (TEST INFRASTRUCTURE ONLY).

Second, differently-structured restatement of the same reference arithmetic as csi_oracle.c
(whole-array slices instead of per-cell loops), used to catch indexing mistakes in the C
oracle: tests/test_oracle_crosscheck.py demands bit-for-bit agreement between the two.
"parity unpinned" (see csi_oracle.h): neither restatement can be run against Julia here.

Reference files (relative to /root/reference/src):
  Rheologies/elasto_visco_plastic_rheology.jl:211-219 (initialize), :236-273 (viscosities),
  :294-354 (stresses), :360-375 (strain rates), :384-401 (sub-step dt, forcing)
  Rheologies/ice_stress_divergence.jl:36-51
  SeaIceDynamics/momentum_tendencies_kernel_functions.jl:11-74
  SeaIceDynamics/sea_ice_external_stress.jl:176-202
  SeaIceDynamics/split_explicit_momentum_equations.jl:173-264
Arrays are (nj, ni), i fastest; element (i, j) (1-based) is a[j + Hy - 1, i + Hx - 1].
"""
import numpy as np

EPS64 = 2.220446049250313e-16


class NP:
    def __init__(self, Nx, Ny, Hx, Hy, topo, dx=None, dy=None, per_j=None):
        self.Nx, self.Ny, self.Hx, self.Hy = Nx, Ny, Hx, Hy
        self.topo = topo          # (0|1, 0|1): 0 periodic, 1 bounded
        self.dx, self.dy = dx, dy
        self.per_j = per_j
        if per_j is not None:
            self.dy = per_j["dy"]
        # EVP defaults (evp:119-127) and momentum-equation defaults
        self.P_star, self.C_star, self.ecc, self.Dmin = 27500.0, 20.0, 2.0, 2e-9
        self.amin, self.amax, self.ca = 50.0, 300.0, float(np.pi) ** 2
        self.replacement = True
        self.min_mass, self.min_conc, self.rho = 1.0, 1e-3, 900.0
        self.f = None
        self.f_rows = None        # BetaPlane: (fu, fv) per row, entry for row j at [j + Hy - 1]
        self.top = None           # None | ("const", tu, tv)
        self.bottom = None        # None | ("semi", ue, ve, rho_e, Cd) with scalar ue, ve
        self.fld = {}

    # ---- views --------------------------------------------------------------------------------
    def V(self, a, r, di=0, dj=0):
        i0, i1, j0, j1 = r
        return a[j0 + dj + self.Hy - 1:j1 + dj + self.Hy, i0 + di + self.Hx - 1:i1 + di + self.Hx]

    def rowvec(self, name, r, dj=0):
        """metric `name` in ('dxc','dxf','azc','azf') for rows j0+dj..j1+dj as a column."""
        if self.per_j is None:
            return {"dxc": self.dx, "dxf": self.dx, "azc": self.dx * self.dy, "azf": self.dx * self.dy}[name]
        j0, j1 = r[2], r[3]
        return self.per_j[name][j0 + dj + self.Hy - 1:j1 + dj + self.Hy][:, None]

    # ---- strain rates on a range, evp:360-375 --------------------------------------------------
    def strain_c(self, r):
        """(e11, e22) at the cells of range r."""
        u, v, dy = self.fld["u"], self.fld["v"], self.dy
        V, rv = self.V, self.rowvec
        az = rv("azc", r)
        a = dy * V(u, r, 1, 0) - dy * V(u, r)
        b = rv("dxf", r, 1) * V(v, r, 0, 1) - rv("dxf", r) * V(v, r)
        eD = (a + b) / az
        dxcc = rv("dxc", r)
        a = V(u, r, 1, 0) / dy - V(u, r) / dy
        b = V(v, r, 0, 1) / rv("dxf", r, 1) - V(v, r) / rv("dxf", r)
        eT = ((dy * dy) * a - (dxcc * dxcc) * b) / az
        return (eD + eT) / 2, (eD - eT) / 2

    def strain_f(self, r):
        """e12 at the corners of range r."""
        u, v, dy = self.fld["u"], self.fld["v"], self.dy
        V, rv = self.V, self.rowvec
        dxff = rv("dxf", r)
        a = V(u, r) / rv("dxc", r) - V(u, r, 0, -1) / rv("dxc", r, -1)
        b = V(v, r) / dy - V(v, r, -1, 0) / dy
        eS = ((dxff * dxff) * a + (dy * dy) * b) / rv("azf", r)
        return eS / 2

    def initialize(self):
        f = self.fld
        f["P"][...] = self.P_star * f["h"] * np.exp(-self.C_star * (1 - f["aice"]))
        ny, nx = f["P"].shape
        f["un"][:ny, :nx] = f["u"][:ny, :nx]
        f["vn"][:ny, :nx] = f["v"][:ny, :nx]

    def stress_range(self):
        return (-self.Hx + 2, self.Nx + self.Hx - 1, -self.Hy + 2, self.Ny + self.Hy - 1)

    def viscosities(self, r):
        i0, i1, j0, j1 = r
        # centre strain rates on (i0-1..i1, j0-1..j1); corner strain rate on (i0..i1+1, j0..j1+1)
        e11, e22 = self.strain_c((i0 - 1, i1, j0 - 1, j1))
        e12 = self.strain_f((i0, i1 + 1, j0, j1 + 1))
        e11c, e22c = e11[1:, 1:], e22[1:, 1:]
        e12f = e12[:-1, :-1]
        ff = lambda a: ((a[:-1, :-1] + a[:-1, 1:]) / 2 + (a[1:, :-1] + a[1:, 1:]) / 2) / 2   # Iy(Ix): rows j-1 then j
        e11f, e22f = ff(e11), ff(e22)
        e12c = ((e12[:-1, :-1] + e12[:-1, 1:]) / 2 + (e12[1:, :-1] + e12[1:, 1:]) / 2) / 2
        ie = 1.0 / self.ecc
        em2 = ie * ie
        dc, df = e11c + e22c, e11f + e22f
        sc = np.sqrt((e11c - e22c) * (e11c - e22c) + 4 * (e12c * e12c))
        sf = np.sqrt((e11f - e22f) * (e11f - e22f) + 4 * (e12f * e12f))
        Dc = np.maximum(np.sqrt(dc * dc + (sc * sc) * em2), self.Dmin)
        Df = np.maximum(np.sqrt(df * df + (sf * sf) * em2), self.Dmin)
        P = self.fld["P"]
        V = self.V
        Pf = ((V(P, r, -1, -1) + V(P, r, 0, -1)) / 2 + (V(P, r, -1, 0) + V(P, r)) / 2) / 2
        V(self.fld["zeta_f"], r)[...] = Pf / (2 * Df)
        V(self.fld["zeta_c"], r)[...] = V(P, r) / (2 * Dc)
        V(self.fld["Delta"], r)[...] = Dc

    def mass(self):
        return self.fld["h"] * self.rho * self.fld["aice"]

    def stresses(self, dt, r):
        f, V, rv = self.fld, self.V, self.rowvec
        e11, e22 = self.strain_c(r)
        e12 = self.strain_f(r)
        zc, zf = V(f["zeta_c"], r), V(f["zeta_f"], r)
        P, D = V(f["P"], r), V(f["Delta"], r)
        Pr = P * D / (D + self.Dmin) if self.replacement else P
        ie = 1.0 / self.ecc
        em2 = ie * ie
        etac, etaf = zc * em2, zf * em2
        s11n = 2 * etac * e11 + ((zc - etac) * (e11 + e22) - Pr / 2)
        s22n = 2 * etac * e22 + ((zc - etac) * (e11 + e22) - Pr / 2)
        s12n = 2 * etaf * e12
        m = self.mass()
        mc = V(m, r)
        mf = ((V(m, r, -1, -1) + V(m, r, 0, -1)) / 2 + (V(m, r, -1, 0) + V(m, r)) / 2) / 2
        with np.errstate(divide="ignore", invalid="ignore"):
            g2c = zc * self.ca * dt / mc / rv("azc", r)
            g2c = np.where(np.isnan(g2c), self.amax * self.amax, g2c)
            gc = np.clip(np.sqrt(g2c), self.amin, self.amax)
            g2f = zf * self.ca * dt / mf / rv("azf", r)
            g2f = np.where(np.isnan(g2f), self.amax * self.amax, g2f)
            gf = np.clip(np.sqrt(g2f), self.amin, self.amax)
        s11, s22, s12 = V(f["s11"], r), V(f["s22"], r), V(f["s12"], r)
        s11 += np.where(mc > 0, (s11n - s11) / gc, 0.0)
        s22 += np.where(mc > 0, (s22n - s22) / gc, 0.0)
        s12 += np.where(mf > 0, (s12n - s12) / gf, 0.0)
        V(f["alpha"], r)[...] = gc

    # ---- velocity steps ------------------------------------------------------------------------
    def _ext(self, which, comp):
        st = getattr(self, which)
        return st

    def u_step(self, dt, r=None):
        f, V, rv, dy = self.fld, self.V, self.rowvec, self.dy
        r = r or (1, self.Nx, 1, self.Ny)
        m = self.mass()
        mi = (V(m, r, -1, 0) + V(m, r)) / 2
        ai = (V(f["aice"], r, -1, 0) + V(f["aice"], r)) / 2
        abar = (V(f["alpha"], r, -1, 0) + V(f["alpha"], r)) / 2
        dtau = dt / abar
        u, v = f["u"], f["v"]
        vbar = ((V(v, r, -1, 0) + V(v, r)) / 2 + (V(v, r, -1, 1) + V(v, r, 0, 1)) / 2) / 2
        if self.f_rows is not None:
            cor = -self.f_rows[0][r[2] + self.Hy - 1:r[3] + self.Hy][:, None] * vbar
        else:
            cor = -self.f * vbar if self.f is not None else np.zeros_like(vbar)
        sD = f["s11"] + f["s22"]
        sT = f["s11"] - f["s22"]
        d = dy * (V(sD, r) - V(sD, r, -1, 0)) / 2
        T = ((dy * dy) * V(sT, r) - (dy * dy) * V(sT, r, -1, 0)) / dy / 2
        dxfn, dxf = rv("dxf", r, 1), rv("dxf", r)
        S = ((dxfn * dxfn) * V(f["s12"], r, 0, 1) - (dxf * dxf) * V(f["s12"], r)) / rv("dxc", r)
        div = (d + T + S) / rv("azc", r)
        ex_t = self.top[1] if self.top else 0.0
        if self.bottom:
            _, ue, ve, rho_e, Cd = self.bottom
            du = ue - V(u, r)
            dv = ((ve + ve) / 2 + (ve + ve) / 2) / 2 - vbar
            norm = np.sqrt(du * du + dv * dv)
            ex_b = rho_e * Cd * norm * ue
            im_b = rho_e * Cd * norm
        else:
            ex_b, im_b = 0.0, 0.0
        with np.errstate(divide="ignore", invalid="ignore"):
            forcing = 0.0 + (V(f["un"], r) - V(u, r)) / dtau / abar
            G = (-cor - ex_t / mi * ai + ex_b / mi * ai + div / mi + 0.0 / mi + forcing)
            G = np.where(mi <= 0, 0.0, G)
            tau_i = (im_b - 0.0) / mi * ai
            tau_i = np.where(mi <= 0, 0.0, tau_i)
            uD = (V(u, r) + dtau * G) / (1 + dtau * tau_i)
        active_ice = (mi >= self.min_mass) & (ai >= self.min_conc)
        active = np.ones_like(uD)
        if self.topo[0] == 1:
            i0 = r[0]
            cols = np.arange(i0, r[1] + 1)
            active[:, (cols == 1) | (cols == self.Nx + 1)] = 0.0
        uF = 0.0
        if getattr(self, "free_drift", False):          # StressBalanceFreeDrift, bottom semi-implicit, top constant
            tx, ty0 = (self.top[1], self.top[2]) if self.top else (0.0, 0.0)
            ty = ((ty0 + ty0) / 2 + (ty0 + ty0) / 2) / 2
            t = np.sqrt(tx * tx + ty * ty)
            _, ue, ve, rho_e, Cd = self.bottom
            uF = ue - (t if t == 0 else tx / np.sqrt(rho_e * Cd * t))
        marginal = (mi > np.finfo(np.float64).eps) & (ai > np.finfo(np.float64).eps)
        sel = np.where(active_ice, uD, np.where(marginal, uF, 0.0))
        V(u, r)[...] = np.where(active != 0, sel, np.copysign(0.0, sel))   # Julia Bool: strong zero

    def v_step(self, dt, r=None):
        f, V, rv, dy = self.fld, self.V, self.rowvec, self.dy
        r = r or (1, self.Nx, 1, self.Ny)
        m = self.mass()
        mi = (V(m, r, 0, -1) + V(m, r)) / 2
        ai = (V(f["aice"], r, 0, -1) + V(f["aice"], r)) / 2
        abar = (V(f["alpha"], r, 0, -1) + V(f["alpha"], r)) / 2
        dtau = dt / abar
        u, v = f["u"], f["v"]
        ubar = ((V(u, r, 0, -1) + V(u, r, 1, -1)) / 2 + (V(u, r) + V(u, r, 1, 0)) / 2) / 2
        if self.f_rows is not None:
            cor = self.f_rows[1][r[2] + self.Hy - 1:r[3] + self.Hy][:, None] * ubar
        else:
            cor = self.f * ubar if self.f is not None else np.zeros_like(ubar)
        sD = f["s11"] + f["s22"]
        sT = f["s11"] - f["s22"]
        dxcf = rv("dxf", r)
        d = dxcf * (V(sD, r) - V(sD, r, 0, -1)) / 2
        dxc, dxcm = rv("dxc", r), rv("dxc", r, -1)
        T = -((dxc * dxc) * V(sT, r) - (dxcm * dxcm) * V(sT, r, 0, -1)) / dxcf / 2
        S = ((dy * dy) * V(f["s12"], r, 1, 0) - (dy * dy) * V(f["s12"], r)) / dy
        div = (d + T + S) / rv("azf", r)
        ex_t = self.top[2] if self.top else 0.0
        if self.bottom:
            _, ue, ve, rho_e, Cd = self.bottom
            dv = ve - V(v, r)
            du = ((ue + ue) / 2 + (ue + ue) / 2) / 2 - ubar
            norm = np.sqrt(du * du + dv * dv)
            ex_b = rho_e * Cd * norm * ve
            im_b = rho_e * Cd * norm
        else:
            ex_b, im_b = 0.0, 0.0
        with np.errstate(divide="ignore", invalid="ignore"):
            forcing = 0.0 + (V(f["vn"], r) - V(v, r)) / dtau / abar
            G = (-cor - ex_t / mi * ai + ex_b / mi * ai + div / mi + 0.0 / mi + forcing)
            G = np.where(mi <= 0, 0.0, G)
            tau_i = (im_b - 0.0) / mi * ai
            tau_i = np.where(mi <= 0, 0.0, tau_i)
            vD = (V(v, r) + dtau * G) / (1 + dtau * tau_i)
        active_ice = (mi >= self.min_mass) & (ai >= self.min_conc)
        active = np.ones_like(vD)
        if self.topo[1] == 1:
            rows = np.arange(r[2], r[3] + 1)
            active[(rows == 1) | (rows == self.Ny + 1), :] = 0.0
        vF = 0.0
        if getattr(self, "free_drift", False):
            tx0, ty = (self.top[1], self.top[2]) if self.top else (0.0, 0.0)
            tx = ((tx0 + tx0) / 2 + (tx0 + tx0) / 2) / 2
            t = np.sqrt(tx * tx + ty * ty)
            _, ue, ve, rho_e, Cd = self.bottom
            vF = ve - (t if t == 0 else ty / np.sqrt(rho_e * Cd * t))
        marginal = (mi > np.finfo(np.float64).eps) & (ai > np.finfo(np.float64).eps)
        sel = np.where(active_ice, vD, np.where(marginal, vF, 0.0))
        V(v, r)[...] = np.where(active != 0, sel, np.copysign(0.0, sel))

    # ---- local halo fill (upstream) ------------------------------------------------
    def fill_halo(self, name, lx, ly):
        a = self.fld[name]
        Nx, Ny, Hx, Hy = self.Nx, self.Ny, self.Hx, self.Hy
        rows = slice(Hy, Hy + Ny)
        if self.topo[0] == 0:
            a[rows, :Hx] = a[rows, Nx:Nx + Hx]
            a[rows, Nx + Hx:Nx + 2 * Hx] = a[rows, Hx:2 * Hx]
        elif lx == 0:
            a[rows, :Hx] = a[rows, Hx:2 * Hx][:, ::-1]
            a[rows, Nx + Hx:Nx + 2 * Hx] = a[rows, Nx:Nx + Hx][:, ::-1]
        if self.topo[1] == 0:
            a[:Hy, :] = a[Ny:Ny + Hy, :]
            a[Ny + Hy:Ny + 2 * Hy, :] = a[Hy:2 * Hy, :]
        elif ly == 0:
            a[:Hy, :] = a[Hy:2 * Hy, :][::-1, :]
            a[Ny + Hy:Ny + 2 * Hy, :] = a[Ny:Ny + Hy, :][::-1, :]

    def subcycle(self, dt, first, last):
        r = self.stress_range()
        for s in range(first, last + 1):
            self.viscosities(r)
            self.stresses(dt, r)
            if s % 2 == 0:
                self.u_step(dt); self.fill_halo("u", 1, 0)
                self.v_step(dt); self.fill_halo("v", 0, 1)
            else:
                self.v_step(dt); self.fill_halo("v", 0, 1)
                self.u_step(dt); self.fill_halo("u", 1, 0)


# ---- WENO reconstructions (round 5) ---------------------------------------------------------------------------------------------
# Whole-array restatement of the upwind-biased WENO-Z face values the advection of h and aice uses (upstream Oceananigans, called at
# /root/reference/src/sea_ice_advection.jl:51-58; candidates, optimal weights, smoothness indicators and tau as derived independently in
# tests/test_weno_published.py), written without looking at csi_oracle.c's expressions beyond their documented ORDER -- the two must
# agree bit for bit (tests/test_weno_published.py), in both weight precisions.
#   P[..., k], k = 0 .. 2B-2: the stencil, upwind-most value first (B = (order + 1) / 2)
#   weight_dtype "f64": everything in double.  "f32": the smoothness indicators, tau, the ratios, the unnormalised weights and their
#   sum in float32 on float32-converted stencil values (upstream's second float type FT2 = Float32, recalled -- csi_oracle.c), the
#   candidates and the final combination in double.
_WENO = {
    3: dict(q=[[0, 1, 1], [-1, 3, 0]], qd=2.0, C=[2.0 / 3, 1.0 / 3]),
    5: dict(q=[[0, 0, 2, 5, -1], [0, -1, 5, 2, 0], [2, -7, 11, 0, 0]], qd=6.0, C=[3.0 / 10, 3.0 / 5, 1.0 / 10]),
    7: dict(q=[[0, 0, 0, 3, 13, -5, 1], [0, 0, -1, 7, 7, -1, 0], [0, 1, -5, 13, 3, 0, 0], [-3, 13, -23, 25, 0, 0, 0]], qd=12.0,
            C=[4.0 / 35, 18.0 / 35, 12.0 / 35, 1.0 / 35]),
}


def _weno_beta(p, order, T):
    """smoothness indicators of the candidates (downwind-most stencil first), dtype T, csi_oracle.c's grouping"""
    c = lambda x: T(x)      # noqa: E731
    if order == 3:
        return [p[1] * (p[1] - c(2) * p[2]) + p[2] * p[2], p[0] * (p[0] - c(2) * p[1]) + p[1] * p[1]]
    if order == 5:
        return [(p[2] * (c(10) * p[2] - c(31) * p[3] + c(11) * p[4]) + p[3] * (c(25) * p[3] - c(19) * p[4]) + p[4] * (c(4) * p[4])) / c(3),
                (p[1] * (c(4) * p[1] - c(13) * p[2] + c(5) * p[3]) + p[2] * (c(13) * p[2] - c(13) * p[3]) + p[3] * (c(4) * p[3])) / c(3),
                (p[0] * (c(4) * p[0] - c(19) * p[1] + c(11) * p[2]) + p[1] * (c(25) * p[1] - c(31) * p[2]) + p[2] * (c(10) * p[2])) / c(3)]
    tab = [[2.107, -9.402, 7.042, -1.854, 11.003, -17.246, 4.642, 7.043, -3.882, 0.547],
           [0.547, -2.522, 1.922, -0.494, 3.443, -5.966, 1.602, 2.843, -1.642, 0.267],
           [0.267, -1.642, 1.602, -0.494, 2.843, -5.966, 1.922, 3.443, -2.522, 0.547],
           [0.547, -3.882, 4.642, -1.854, 7.043, -17.246, 7.042, 11.003, -9.402, 2.107]]
    out = []
    for s, t in enumerate(tab):
        a, b, cc, d = p[3 - s], p[4 - s], p[5 - s], p[6 - s]
        out.append(a * (c(t[0]) * a + c(t[1]) * b + c(t[2]) * cc + c(t[3]) * d) + b * (c(t[4]) * b + c(t[5]) * cc + c(t[6]) * d) +
                   cc * (c(t[7]) * cc + c(t[8]) * d) + d * (c(t[9]) * d))
    return out


def weno_value(P, order, weight_dtype="f64"):
    """upwind-biased WENO-Z face value from stencils P[..., 2B-1] (upwind-most first)"""
    P = np.asarray(P, dtype=np.float64)
    W = _WENO[order]
    B = (order + 1) // 2
    p = [P[..., k] for k in range(2 * B - 1)]
    q = []
    for row in W["q"]:
        acc = None
        for k, w in enumerate(row):          # left to right over the non-zero entries, as written in the C oracle
            if w == 0:
                continue
            term = p[k] if w == 1 else (-p[k] if w == -1 else w * p[k])
            acc = term if acc is None else acc + term
        q.append(acc / W["qd"])
    # (the C oracle writes the subtractions as "- c p": a + (-c) p == a - c p bit for bit)
    T = np.float32 if weight_dtype == "f32" else np.float64
    pw = [x.astype(T) for x in p]
    beta = _weno_beta(pw, order, T)
    if order == 3:
        tau = np.abs(beta[0] - beta[1])
    elif order == 5:
        tau = np.abs(beta[0] - beta[2])
    else:
        tau = np.abs(beta[0] + T(3) * beta[1] - T(3) * beta[2] - beta[3])
    eps = T(1e-8)
    alpha = []
    for s in range(B):
        r = tau / (beta[s] + eps)
        alpha.append(T(W["C"][s]) * (T(1) + r * r))
    ssum = alpha[0]
    for s in range(1, B):
        ssum = ssum + alpha[s]
    num = alpha[0].astype(np.float64) * q[0]
    for s in range(1, B):
        num = num + alpha[s].astype(np.float64) * q[s]
    return num / ssum.astype(np.float64)
