import sys, time; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, climaseaice_jl_amd as csi, cases
N = 2048
for name, kw, mode in (("strict uniform", dict(), "strict"), ("strict curvilinear (2-D metrics)", dict(curvilinear=0.03), "strict"),
                       ("fast curvilinear (per-point coefficients, three-kernel path)", dict(curvilinear=0.03), "fast")):
    c = cases.make_case(Nx=N, Ny=N, substeps=120, topo=("periodic", "periodic"), patches=False, noise=0.05, **kw)
    m = cases.csi_model(c, mode=mode)
    for _ in range(2): csi.time_step_momentum(m, c["dt"])
    m.synchronize(); t0 = time.perf_counter()
    for _ in range(3): csi.time_step_momentum(m, c["dt"])
    m.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(name, round(N * N * 120 / dt / 1e9, 2), "G cell-updates/s")
