"""Fuzz of whole model steps (FE / RK3, WENO / upwind advection with walls, masks, array forcing, free drift) against the
oracle: STRICT 1e-12, FAST 1e-11 relative on u, v, h, aice; zero sets of h, aice identical (run on the GPU box)."""
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import climaseaice_jl_amd as csi, cases
bad = 0
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, int(sys.argv[1]) if len(sys.argv) > 1 else 40)
for seed in range(lo, hi):
    rng = np.random.default_rng(9000 + seed)
    topo = (("periodic", "bounded")[rng.integers(2)], ("periodic", "bounded")[rng.integers(2)])
    H = int(rng.integers(4, 7))
    Nx = int(rng.integers(2 * H + 2, 90)); Ny = int(rng.integers(2 * H + 2, 60))
    kw = dict(Nx=Nx, Ny=Ny, H=H, topo=topo, patches=bool(rng.integers(2)), random_uv=0.03,
              grid=("rectilinear", "latlon")[rng.integers(2)] if topo[1] == "bounded" else "rectilinear",
              field_forcing=bool(rng.integers(2)), land=(0.0, 0.25)[rng.integers(2)], free_drift=bool(rng.integers(2)))
    if topo[1] == "bounded" and rng.integers(3) == 0:
        kw["beta"] = 2e-10 if kw["grid"] == "rectilinear" else 1e-6
    if rng.integers(4) == 0:
        kw["curvilinear"] = 0.04
    if "bounded" in topo and rng.integers(3) == 0:
        kw["noslip"] = True
    snow = bool(rng.integers(3) == 0)
    scheme = [7, 5, -5, 3, -3, 1][rng.integers(6)]
    stepper = ["ForwardEuler", "SplitRungeKutta3"][rng.integers(2)]
    nsub = int(rng.integers(2, 10))
    adv = {7: csi.WENO(order=7), 5: csi.WENO(order=5), 3: csi.WENO(order=3), -5: csi.UpwindBiased(order=5),
           -3: csi.UpwindBiased(order=3), 1: csi.UpwindBiased(order=1)}[scheme]
    import oracle as O
    slab_o = O.make_slab(top_bc_kind=1, top_flux_kind=0, Qu=-70.0, Qb=5.0, salinity=30.0) if snow else None
    snow_o = O.make_snow(snowfall=2e-5) if snow else None
    try:
        c = cases.make_case(substeps=nsub, **kw)
        for mode, tol in (("strict", 1e-12), ("fast", 1e-11)):
            p = cases.oracle_problem(c)
            if snow:
                hs0 = np.where(c["a"] > 0, 0.3 * np.random.default_rng(seed).random(c["a"].shape), 0.0)
                p.s.has_snow = 1
                p.interior("hs")[...] = hs0
                p.update_state()
                ice = csi.SlabThermodynamics(top_heat_flux=-70.0, bottom_heat_flux=5.0, bottom_salinity=30.0,
                                             top_heat_boundary_condition=csi.MeltingConstrainedFluxBalance())
                m = cases.csi_model(c, mode=mode, timestepper=stepper, advection=adv, ice_thermodynamics=ice,
                                    snow_thermodynamics=csi.snow_slab_thermodynamics(), snowfall=2e-5)
                csi.set_(m, hs=hs0)
            else:
                m = cases.csi_model(c, mode=mode, timestepper=stepper, advection=adv)
            for n in range(2):
                if stepper == "ForwardEuler":
                    p.time_step_fe(c["dt"], scheme, n == 0, slab=slab_o, snow=snow_o)
                else:
                    p.time_step_rk3(c["dt"], scheme, slab=slab_o, snow=snow_o)
                csi.time_step(m, c["dt"])
            m.synchronize()
            vmax = max(np.abs(p.f["u"]).max(), np.abs(p.f["v"]).max(), 1e-30)
            for k, f in (("u", m.velocities.u), ("v", m.velocities.v)):
                d = np.abs(f.interior_numpy() - p.interior(k)).max()
                assert d <= tol * vmax, (mode, k, d, vmax)
            for k, f in (("h", m.ice_thickness), ("aice", m.ice_concentration)) + ((("hs", m.snow_thickness),) if snow else ()):
                d = np.abs(f.interior_numpy() - p.interior(k)).max()
                assert d <= tol * np.abs(p.f[k]).max(), (mode, k, d)
                assert np.array_equal(f.interior_numpy() == 0.0, p.interior(k) == 0.0), (mode, k, "zero set")
    except Exception as e:
        bad += 1
        print("FAIL", seed, kw, scheme, stepper, nsub, snow, type(e).__name__, str(e)[:200])
print("done, failures:", bad)
