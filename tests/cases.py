"""Synthetic test cases shared by the oracle tests, the GPU parity tests and bench.py.

A case is a plain dict (grid description, physics switches, seeded numpy initial fields).
`oracle_problem(case)` builds the CPU oracle problem; `csi_model(case, mode)` builds the product
model on the GPU through the C ABI.  Inputs follow SURVEY.md 8(d): h0 is the sinusoid of
test/distributed_tests_utils.jl:126 (+ optional seeded noise), optional open-water / thin-ice
patches exercise the active / marginal / zero branches of the velocity kernels.
"""
import numpy as np

import climaseaice_jl_amd as csi


def make_case(Nx=64, Ny=48, H=4, topo=("periodic", "periodic"), grid="rectilinear", spacing=2000.0,
              substeps=10, dt=120.0, coriolis=1e-4, top=(0.01, 0.01), bottom="semi", ue=0.0, ve=0.0,
              patches=True, noise=0.05, seed=3, u0=0.1, v0=0.0, random_uv=0.0, pressure="replacement",
              field_forcing=False, land=0.0, free_drift=False, beta=None, curvilinear=None, noslip=False,
              user_forcing=False, immersed_bc=None, coriolis_points=False, wind_drag=None,
              tripolar=None, ice_edge=None, ice_free_rows=None):
    rng = np.random.default_rng(seed)
    c = dict(Nx=Nx, Ny=Ny, H=H, topo=topo, grid=grid, spacing=spacing, substeps=substeps, dt=dt, coriolis=coriolis,
             top=top, bottom=bottom, ue=ue, ve=ve, pressure=pressure, field_forcing=field_forcing,
             free_drift=free_drift, beta=beta,     # beta: BetaPlane(f0 = coriolis, beta)
             noslip=noslip,                        # ValueBoundaryCondition(0) on the tangential velocity at every wall
             user_forcing=user_forcing,            # model.forcing.u / .v as arrays
             immersed_bc=immersed_bc,              # ((uW, uE, uS, uN), (vW, vE, vS, vN)): immersed FluxBoundaryCondition numbers
             coriolis_points=coriolis_points,      # per-point f planes on a curvilinear grid (PointwiseCoriolis)
             wind_drag=wind_drag)                  # top stress = SemiImplicitStress(air velocities; rho 1.3, Cd 1.2e-3): "numbers" / "arrays"
    T = {"periodic": csi.Periodic, "bounded": csi.Bounded, "folded": csi.RightFolded}     # "folded": y of a TripolarGrid
    tt = (T[topo[0]], T[topo[1]])
    # grid = "tripolar": csi.TripolarGrid(size, **tripolar) -- latitude-longitude rows below a bipolar cap, north fold, the reference's
    # analytic land around the poles and along the southern edge (+ `land` discs); f = 2 Omega sin(latitude) per point when
    # coriolis_points.  ice_edge (degrees): ice only poleward of that latitude (a seasonal-ice state: most of the ocean ice-free);
    # ice_free_rows = (y0, y1): on the other grids, no ice in that fraction of the rows.
    c["ice_edge"], c["ice_free_rows"] = ice_edge, ice_free_rows
    if grid == "tripolar":
        topo = ("periodic", "folded")
        c["topo"] = topo
        g = csi.TripolarGrid((Nx, Ny), halo=(H, H), **(tripolar or {}))
    elif grid == "rectilinear":
        g = csi.RectilinearGrid((Nx, Ny), x=(0.0, Nx * spacing), y=(0.0, Ny * spacing), topology=tt, halo=(H, H))
    else:
        g = csi.LatitudeLongitudeGrid((Nx, Ny), longitude=(0, 60), latitude=(20, 70), topology=tt, halo=(H, H))
    if curvilinear is not None:
        # the same grid described by twelve 2-D metric arrays (CSI_METRIC_FULL), optionally distorted by `curvilinear`
        g = csi.OrthogonalCurvilinearGrid.from_grid(g, distort=curvilinear, seed=seed)
    c["g"] = g
    xc = (np.arange(Nx) + 0.5) / Nx
    yc = (np.arange(Ny) + 0.5) / Ny
    X, Y = xc[None, :], yc[:, None]
    h = 0.3 + 0.005 * (np.sin(2 * np.pi * 3 * X) + np.sin(2 * np.pi * 2 * Y)) + 0.0 * X * Y
    if noise:
        h = h * (1.0 + noise * (rng.random((Ny, Nx)) - 0.5))
    a = np.ones((Ny, Nx))
    if patches:
        a = np.clip(0.6 + 0.6 * np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y) + 0.2 * rng.random((Ny, Nx)), 0.0, 1.0)
        i0, i1, j0, j1 = Nx // 8, Nx // 4, Ny // 6, Ny // 3
        a[j0:j1, i0:i1] = 0.0                       # open water: zero branch
        h[j0:j1, i0:i1] = 0.0
        a[j1:j1 + 2, i0:i1] = 5e-4                  # below minimum_concentration: marginal branch
        h[j1:j1 + 2, i0:i1] = 1e-3
    nxu, nyu = g.interior_size(csi.Face, csi.Center)
    nxv, nyv = g.interior_size(csi.Center, csi.Face)
    u = np.full((nyu, nxu), float(u0)) + random_uv * rng.standard_normal((nyu, nxu))
    v = np.full((nyv, nxv), float(v0)) + random_uv * rng.standard_normal((nyv, nxv))
    if topo[0] == "bounded":
        u[:, 0] = 0.0
        u[:, -1] = 0.0
    if topo[1] == "bounded":
        v[0, :] = 0.0
        v[-1, :] = 0.0
    if topo[1] == "folded":
        v[0, :] = 0.0                               # the south wall; the north side is the fold (no wall face row)
    c["mask"] = None
    if grid == "tripolar":
        wet_t = g.analytic_land()
        lam_c, phi_c = g.nodes_2d(csi.Center, csi.Center)
        if ice_edge is not None:
            icy = np.abs(phi_c) > ice_edge
            h = np.where(icy, h, 0.0)
            a = np.where(icy, a, 0.0)
        if coriolis_points:
            c["f_points"] = g.coriolis_planes()
    elif ice_free_rows is not None:
        j0, j1 = int(ice_free_rows[0] * Ny), int(ice_free_rows[1] * Ny)
        h[j0:j1, :] = 0.0
        a[j0:j1, :] = 0.0
    if land or grid == "tripolar":
        # immersed "land": union of seeded discs covering about `land` of the domain
        lr = np.random.default_rng(5)
        wet = np.ones((Ny, Nx), dtype=bool)
        II, JJ = np.meshgrid(np.arange(Nx), np.arange(Ny))
        if grid == "tripolar":
            wet &= wet_t
        while land and 1.0 - wet.mean() < land:
            cx, cy, rad = lr.integers(0, Nx), lr.integers(0, Ny), lr.integers(2, max(3, min(Nx, Ny) // 6))
            ddx = np.minimum(np.abs(II - cx), Nx - np.abs(II - cx)) if topo[0] == "periodic" else np.abs(II - cx)
            ddy = np.minimum(np.abs(JJ - cy), Ny - np.abs(JJ - cy)) if topo[1] == "periodic" else np.abs(JJ - cy)
            wet &= (ddx ** 2 + ddy ** 2) > rad ** 2
        if topo[1] == "folded":
            # the fold runs through the centres of row Ny: that row must be its own mirror image (cell i <-> Nx - i + 1)
            wet[-1, :] &= wet[-1, ::-1]
        c["mask"] = wet
        h = np.where(wet, h, 0.0)
        a = np.where(wet, a, 0.0)
    c.update(h=h, a=a, u=u, v=v)
    if coriolis_points and grid != "tripolar":
        # f(i, j) = f0 (1 + 0.3 sin cos) at the u / v nodes, metric-plane layout; halo entries image their points
        n, ni = Ny + 2 * H + 1, Nx + 2 * H + 1
        ia, ja = np.arange(ni) - (H - 1), np.arange(n) - (H - 1)
        if topo[0] == "periodic":
            ia = (ia - 1) % Nx + 1
        if topo[1] == "periodic":
            ja = (ja - 1) % Ny + 1
        planes = []
        for (ox, oy) in ((0.0, 0.5), (0.5, 0.0)):          # u nodes (Face, Center), v nodes (Center, Face)
            X2, Y2 = (ia[None, :] - 1 + ox) / Nx, (ja[:, None] - 1 + oy) / Ny
            fpl = (coriolis or 1e-4) * (1.0 + 0.3 * np.sin(2 * np.pi * X2) * np.cos(np.pi * Y2))
            if topo[1] == "folded":
                fpl = csi.fold_north(fpl, Nx, Ny, H, H, ox == 0.0, oy == 0.0, 1)     # f is a scalar: no sign change
            planes.append(np.ascontiguousarray(fpl))
        c["f_points"] = tuple(planes)
    if user_forcing:
        # an acceleration of a few 1e-6 m s^-2 (comparable to the Coriolis term), smooth + seeded noise
        c["force_u"] = 3e-6 * np.sin(2 * np.pi * np.linspace(0, 1, nyu))[:, None] * np.ones((1, nxu)) + 1e-6 * rng.standard_normal((nyu, nxu))
        c["force_v"] = -2e-6 * np.cos(2 * np.pi * np.linspace(0, 1, nxv))[None, :] * np.ones((nyv, 1)) + 1e-6 * rng.standard_normal((nyv, nxv))
    if bottom == "arrays":
        c["bot_u"] = -0.004 * np.cos(2 * np.pi * np.linspace(0, 1, nyu))[:, None] * np.ones((1, nxu)) + 5e-4 * rng.standard_normal((nyu, nxu))
        c["bot_v"] = 0.003 * np.sin(2 * np.pi * np.linspace(0, 1, nxv))[None, :] * np.ones((nyv, 1)) + 5e-4 * rng.standard_normal((nyv, nxv))
    if wind_drag == "arrays":
        c["ua_f"] = 6.0 * np.cos(2 * np.pi * np.linspace(0, 1, nyu))[:, None] * np.ones((1, nxu)) + 0.5 * rng.standard_normal((nyu, nxu))
        c["va_f"] = 4.0 * np.sin(2 * np.pi * np.linspace(0, 1, nxv))[None, :] * np.ones((nyv, 1)) + 0.5 * rng.standard_normal((nyv, nxv))
    if field_forcing:
        c["top_u"] = 0.01 * (1 + 0.5 * np.sin(2 * np.pi * X)) * np.ones((nyu, 1))[:, :1] * np.ones((1, 1))
        c["top_u"] = np.broadcast_to(0.01 * (1 + 0.5 * np.sin(2 * np.pi * np.linspace(0, 1, nxu)))[None, :], (nyu, nxu)).copy()
        c["top_v"] = np.broadcast_to(0.01 * (1 + 0.5 * np.cos(2 * np.pi * np.linspace(0, 1, nyv)))[:, None], (nyv, nxv)).copy()
        c["ue_f"] = 0.05 * np.cos(2 * np.pi * np.linspace(0, 1, nyu))[:, None] * np.ones((1, nxu))
        c["ve_f"] = 0.05 * np.sin(2 * np.pi * np.linspace(0, 1, nxv))[None, :] * np.ones((nyv, 1))
    return c


def _fill_parent_like(p, name, interior):
    """numpy parent array of oracle field `name` holding `interior` with locally filled halos."""
    import oracle as O
    arr = np.zeros_like(p.f[name])
    s = p.s
    ny, nx = interior.shape
    arr[s.Hy:s.Hy + ny, s.Hx:s.Hx + nx] = interior
    return arr


def coriolis_rows(case, grid):
    """BetaPlane cases: f0 + beta * ynode per row of `grid` (u points, v points); None for FPlane / no Coriolis."""
    if case.get("beta") is None:
        return None
    return csi.BetaPlane(f0=case["coriolis"], beta=case["beta"]).rows(grid)


def coriolis_of(case):
    if case.get("coriolis_points"):
        return csi.PointwiseCoriolis(*case["f_points"])
    if case["coriolis"] is None:
        return None
    if case.get("beta") is not None:
        return csi.BetaPlane(f0=case["coriolis"], beta=case["beta"])
    return csi.FPlane(f=case["coriolis"])


def oracle_problem(case, omp=False):
    import oracle as O
    g = case["g"]
    topo = tuple({"periodic": O.PERIODIC, "bounded": O.BOUNDED, "folded": O.RIGHT_FOLDED}[t] for t in case["topo"])
    m = g.metrics()
    if m["kind"] == "uniform":
        p = O.Problem(g.Nx, g.Ny, g.Hx, g.Hy, topo, dx=m["dx"], dy=m["dy"], substeps=case["substeps"], omp=omp)
    elif m["kind"] == "full":
        p = O.Problem(g.Nx, g.Ny, g.Hx, g.Hy, topo, full=m, substeps=case["substeps"], omp=omp)
    else:
        p = O.Problem(g.Nx, g.Ny, g.Hx, g.Hy, topo, per_j=m, substeps=case["substeps"], omp=omp)
    p.set_coriolis(case["coriolis"], rows=coriolis_rows(case, g))
    if case.get("coriolis_points"):
        p.set_coriolis_points(*case["f_points"])
    if case.get("noslip"):
        for side in (0, 1):
            p.set_value_bc("u", side, 0.0)
            p.set_value_bc("v", side, 0.0)
    if case["pressure"] != "replacement":
        p.s.pressure_kind = O.PRESSURE_ICE_STRENGTH
    if case.get("field_forcing"):
        tu = _fill_parent_like(p, "u", case["top_u"]); tv = _fill_parent_like(p, "v", case["top_v"])
        p.set_stress("top", O.STRESS_FIELD, fu=tu, fv=tv)
        ue = _fill_parent_like(p, "u", case["ue_f"]); ve = _fill_parent_like(p, "v", case["ve_f"])
        p.set_stress("bottom", O.STRESS_SEMI_IMPLICIT, ue=ue, ve=ve)
        # halos of the forcing fields: update_external_stress! (sea_ice_external_stress.jl:72-78,148-152)
        for arr, (lx, ly) in ((tu, (O.FACE, O.CENTER)), (tv, (O.CENTER, O.FACE)), (ue, (O.FACE, O.CENTER)), (ve, (O.CENTER, O.FACE))):
            fld = O.Field(arr.ctypes.data_as(O.C.POINTER(O.C.c_double)), arr.shape[1])
            p.L.ora_fill_halo_loc(p.ptr, fld, lx, ly, -1)      # vector components: they change sign across a north fold
    else:
        if case["top"] is not None:
            p.set_stress("top", O.STRESS_CONST, tau=case["top"])
        if case["bottom"] == "semi":
            p.set_stress("bottom", O.STRESS_SEMI_IMPLICIT, ue=case["ue"] or None, ve=case["ve"] or None)
    if case.get("bottom") == "arrays":              # an explicit bottom stress given as arrays (sea_ice_external_stress.jl:54-61)
        bu = _fill_parent_like(p, "u", case["bot_u"]); bv = _fill_parent_like(p, "v", case["bot_v"])
        p.set_stress("bottom", O.STRESS_FIELD, fu=bu, fv=bv)
        for arr, (lx, ly) in ((bu, (O.FACE, O.CENTER)), (bv, (O.CENTER, O.FACE))):
            p.L.ora_fill_halo_loc(p.ptr, O.Field(arr.ctypes.data_as(O.C.POINTER(O.C.c_double)), arr.shape[1]), lx, ly, -1)
    if case.get("wind_drag") == "numbers":
        p.set_stress("top", O.STRESS_SEMI_IMPLICIT, ue=5.0, ve=-3.0, rho_e=1.3, Cd=1.2e-3)
    elif case.get("wind_drag") == "arrays":
        ua = _fill_parent_like(p, "u", case["ua_f"]); va = _fill_parent_like(p, "v", case["va_f"])
        p.set_stress("top", O.STRESS_SEMI_IMPLICIT, ue=ua, ve=va, rho_e=1.3, Cd=1.2e-3)
        for arr, (lx, ly) in ((ua, (O.FACE, O.CENTER)), (va, (O.CENTER, O.FACE))):
            p.L.ora_fill_halo_loc(p.ptr, O.Field(arr.ctypes.data_as(O.C.POINTER(O.C.c_double)), arr.shape[1]), lx, ly, -1)
    if case.get("free_drift"):
        p.s.free_drift_kind = 1                     # StressBalanceFreeDrift on the model's own stresses
    if case.get("user_forcing"):
        fu = _fill_parent_like(p, "u", case["force_u"]); fv = _fill_parent_like(p, "v", case["force_v"])
        for arr, (lx, ly) in ((fu, (O.FACE, O.CENTER)), (fv, (O.CENTER, O.FACE))):
            p.L.ora_fill_halo_loc(p.ptr, O.Field(arr.ctypes.data_as(O.C.POINTER(O.C.c_double)), arr.shape[1]), lx, ly, -1)
        p.set_forcing(fu, fv)
    if case.get("immersed_bc"):
        (uw, ue_, us, un_), (vw, ve_, vs, vn_) = case["immersed_bc"]
        p.set_immersed_flux_bc("u", uw, ue_, us, un_)
        p.set_immersed_flux_bc("v", vw, ve_, vs, vn_)
    if case.get("mask") is not None:
        s_ = p.s
        full = np.zeros(p.f["h"].shape, dtype=np.uint8)
        full[s_.Hy:s_.Hy + s_.Ny, s_.Hx:s_.Hx + s_.Nx] = case["mask"]
        if topo[0] == O.PERIODIC:
            full[:, :s_.Hx] = full[:, s_.Nx:s_.Nx + s_.Hx]
            full[:, s_.Nx + s_.Hx:] = full[:, s_.Hx:2 * s_.Hx]
        if topo[1] == O.PERIODIC:
            full[:s_.Hy, :] = full[s_.Ny:s_.Ny + s_.Hy, :]
            full[s_.Ny + s_.Hy:, :] = full[s_.Hy:2 * s_.Hy, :]
        if topo[1] == O.RIGHT_FOLDED:                # cells beyond the fold are images of real cells
            full = csi.fold_north(full, s_.Nx, s_.Ny, s_.Hx, s_.Hy, False, False, 1).astype(np.uint8)
        p.set_mask(full)
    p.interior("h")[...] = case["h"]
    p.interior("aice")[...] = case["a"]
    p.interior("u")[...] = case["u"]
    p.interior("v")[...] = case["v"]
    p.update_state()
    return p


def csi_model(case, mode="fast", timestepper="ForwardEuler", advection=None, device="cuda:0", tile=None, local_group=None, host_group=None, **model_kw):
    """tile = (Rx, Ry, rank[, force_connected]): build the model of one tile of the global case (local_group: the tiles of this
    process exchange through a csi.LocalGroup instead of RCCL)."""
    g = case["g"]
    if tile is not None:
        Rx, Ry, rank = tile[:3]
        g = csi.TileGrid(g, Rx, Ry, rank % Rx, rank // Rx, force_connected=tile[3] if len(tile) > 3 else False, local_group=local_group,
                         host_group=host_group)
        case = dict(case)
        for key, loc in (("h", (csi.Center, csi.Center)), ("a", (csi.Center, csi.Center)), ("u", (csi.Face, csi.Center)),
                         ("v", (csi.Center, csi.Face))):
            case[key] = g.local_interior(case[key], *loc)
        if case.get("field_forcing"):
            for key, loc in (("top_u", (csi.Face, csi.Center)), ("ue_f", (csi.Face, csi.Center)),
                             ("top_v", (csi.Center, csi.Face)), ("ve_f", (csi.Center, csi.Face))):
                case[key] = g.local_interior(case[key], *loc)
        # the mask stays global: SeaIceModel.set_mask slices the tile (halo included) out of it
    if case.get("field_forcing"):
        top = (case["top_u"], case["top_v"])
        bottom = csi.SemiImplicitStress(ue=case["ue_f"], ve=case["ve_f"])
    else:
        top = case["top"]
        bottom = csi.SemiImplicitStress(ue=case["ue"] or None, ve=case["ve"] or None) if case["bottom"] == "semi" else None
    if case.get("bottom") == "arrays":
        bu, bv = case["bot_u"], case["bot_v"]
        if tile is not None:
            bu, bv = g.local_interior(bu, csi.Face, csi.Center), g.local_interior(bv, csi.Center, csi.Face)
        bottom = (bu, bv)
    if case.get("wind_drag") == "numbers":
        top = csi.SemiImplicitStress(ue=5.0, ve=-3.0, rho_e=1.3, Cd=1.2e-3)
    elif case.get("wind_drag") == "arrays":
        ua, va = case["ua_f"], case["va_f"]
        if tile is not None:
            ua, va = g.local_interior(ua, csi.Face, csi.Center), g.local_interior(va, csi.Center, csi.Face)
        top = csi.SemiImplicitStress(ue=ua, ve=va, rho_e=1.3, Cd=1.2e-3)
    rheo = csi.ElastoViscoPlasticRheology()
    if case["pressure"] != "replacement":
        rheo.pressure_formulation = csi.IceStrength()
    dyn = csi.SeaIceMomentumEquation(g, coriolis=coriolis_of(case),
                                     rheology=rheo, top_momentum_stress=top, bottom_momentum_stress=bottom,
                                     free_drift=csi.StressBalanceFreeDrift() if case.get("free_drift") else None,
                                     solver=csi.SplitExplicitSolver(substeps=case["substeps"]), device=device)
    ubc, vbc_ = {}, {}
    if case.get("noslip"):
        vbc = csi.ValueBoundaryCondition(0.0)
        ubc.update(north=vbc, south=vbc); vbc_.update(west=vbc, east=vbc)
    if case.get("immersed_bc"):
        F = csi.FluxBoundaryCondition
        (uw, ue_, us, un_), (vw, ve_, vs, vn_) = case["immersed_bc"]
        ubc["immersed"] = csi.ImmersedBoundaryCondition(west=F(uw), east=F(ue_), south=F(us), north=F(un_))
        vbc_["immersed"] = csi.ImmersedBoundaryCondition(west=F(vw), east=F(ve_), south=F(vs), north=F(vn_))
    if ubc or vbc_:
        model_kw = dict(model_kw, boundary_conditions=dict(u=csi.FieldBoundaryConditions(**ubc), v=csi.FieldBoundaryConditions(**vbc_)))
    if case.get("user_forcing"):
        fu, fv = case["force_u"], case["force_v"]
        if tile is not None:
            fu, fv = g.local_interior(fu, csi.Face, csi.Center), g.local_interior(fv, csi.Center, csi.Face)
        model_kw = dict(model_kw, forcing=dict(u=fu, v=fv))
    model = csi.SeaIceModel(g, dynamics=dyn, advection=advection, timestepper=timestepper, device=device, mode=mode, **model_kw)
    if case.get("field_forcing") or case.get("wind_drag") == "arrays" or case.get("bottom") == "arrays":
        slots = [sl for sl, on in (("TOP", (case.get("field_forcing") and case.get("wind_drag") != "numbers") or case.get("wind_drag") == "arrays"),
                                   ("BOT", case.get("field_forcing") or case.get("bottom") == "arrays")) if on]
        for slot in slots:
            for comp in ("U", "V"):
                model.ctx.call("csi_fill_halo_local", csi._lib.F[f"{slot}_{comp}"])
    if case.get("mask") is not None:
        model.set_mask(case["mask"])
    csi.set_(model, h=case["h"], aice=case["a"], u=case["u"], v=case["v"])
    return model
