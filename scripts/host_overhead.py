"""Host time of one csi_time_step_momentum call (the GPU has next to nothing to do: small grid, few sub-steps): cuts on / off."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "scripts")):
    sys.path.insert(0, p)
import climaseaice_jl_amd as csi
import cases
from structure_cases import CASES
for name, N, sub in (("headline", 256, 10), ("masked", 256, 10), ("tripolar_land", 256, 10), ("tripolar_land", 2048, 10)):
    c = cases.make_case(Nx=N, Ny=N, substeps=sub, patches=False, noise=0.05, **CASES[name])
    for on in (True, False):
        m = cases.csi_model(c, mode="fast")
        m.set_tile_skipping(on); m.set_row_constant(on)
        for _ in range(5):
            csi.time_step_momentum(m, c["dt"])
        m.synchronize()
        n = 50
        t0 = time.perf_counter()
        for _ in range(n):
            csi.time_step_momentum(m, c["dt"])
        t1 = time.perf_counter()          # host enqueue time only
        m.synchronize()
        t2 = time.perf_counter()
        print(f"{name:14s} {N:5d} sub {sub}: cuts {'on ' if on else 'off'} host {1e3 * (t1 - t0) / n:.3f} ms per call, with the GPU {1e3 * (t2 - t0) / n:.3f} ms; activity {m.tile_activity()}", flush=True)
