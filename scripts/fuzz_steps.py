"""Fuzz of whole model steps (FE / RK3, WENO / upwind advection with walls, masks, array forcing, free drift) against the
oracle: STRICT 1e-12, FAST 1e-11 relative on u, v, h, aice; zero sets of h, aice identical (run on the GPU box)."""
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import climaseaice_jl_amd as csi, cases
bad = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    rng = np.random.default_rng(9000 + seed)
    topo = (("periodic", "bounded")[rng.integers(2)], ("periodic", "bounded")[rng.integers(2)])
    H = int(rng.integers(4, 7))
    Nx = int(rng.integers(2 * H + 2, 90)); Ny = int(rng.integers(2 * H + 2, 60))
    kw = dict(Nx=Nx, Ny=Ny, H=H, topo=topo, patches=bool(rng.integers(2)), random_uv=0.03,
              grid=("rectilinear", "latlon")[rng.integers(2)] if topo[1] == "bounded" else "rectilinear",
              field_forcing=bool(rng.integers(2)), land=(0.0, 0.25)[rng.integers(2)], free_drift=bool(rng.integers(2)))
    scheme = [7, 5, -5, 1][rng.integers(4)]
    stepper = ["ForwardEuler", "SplitRungeKutta3"][rng.integers(2)]
    nsub = int(rng.integers(2, 10))
    adv = {7: csi.WENO(order=7), 5: csi.WENO(order=5), -5: csi.UpwindBiased(order=5), 1: csi.UpwindBiased(order=1)}[scheme]
    try:
        c = cases.make_case(substeps=nsub, **kw)
        for mode, tol in (("strict", 1e-12), ("fast", 1e-11)):
            p = cases.oracle_problem(c)
            m = cases.csi_model(c, mode=mode, timestepper=stepper, advection=adv)
            for n in range(2):
                if stepper == "ForwardEuler":
                    p.time_step_fe(c["dt"], scheme, n == 0)
                else:
                    p.time_step_rk3(c["dt"], scheme)
                csi.time_step(m, c["dt"])
            m.synchronize()
            vmax = max(np.abs(p.f["u"]).max(), np.abs(p.f["v"]).max(), 1e-30)
            for k, f in (("u", m.velocities.u), ("v", m.velocities.v)):
                d = np.abs(f.interior_numpy() - p.interior(k)).max()
                assert d <= tol * vmax, (mode, k, d, vmax)
            for k, f in (("h", m.ice_thickness), ("aice", m.ice_concentration)):
                d = np.abs(f.interior_numpy() - p.interior(k)).max()
                assert d <= tol * np.abs(p.f[k]).max(), (mode, k, d)
                assert np.array_equal(f.interior_numpy() == 0.0, p.interior(k) == 0.0), (mode, k, "zero set")
    except Exception as e:
        bad += 1
        print("FAIL", seed, kw, scheme, stepper, nsub, type(e).__name__, str(e)[:200])
print("done, failures:", bad)
