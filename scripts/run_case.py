"""One configuration of scripts/ab_skip.py in ONE mode, for profilers: python3 scripts/run_case.py <case> <N> <on|off> [steps] [substeps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "scripts")):
    sys.path.insert(0, p)
import climaseaice_jl_amd as csi
import cases
from structure_cases import CASES          # noqa: E402
argv = sys.argv
name, N, on = argv[1], int(argv[2]), argv[3] == "on"
steps = int(argv[4]) if len(argv) > 4 else 3
sub = int(argv[5]) if len(argv) > 5 else 120
c = cases.make_case(Nx=N, Ny=N, substeps=sub, patches=False, noise=0.05, **CASES[name])
m = cases.csi_model(c, mode="fast")
m.set_tile_skipping(on)
m.set_row_constant(on)
for _ in range(3):
    csi.time_step_momentum(m, c["dt"])
m.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    csi.time_step_momentum(m, c["dt"])
m.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"{name} {N} cuts {'on' if on else 'off'}: {N * N * sub / dt / 1e9:.2f} G all cells, {dt * 1e3:.3f} ms per step, activity {m.tile_activity()}, "
      f"row-constant rows {m.row_constant_rows()}", flush=True)
