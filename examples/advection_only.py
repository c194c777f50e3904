"""BASELINE.json config 2: ice_advected_by_anticyclone-style 512^2 periodic grid, WENO(order = 7) advection of h and aice
only -- prescribed cyclonic velocities, dynamics = nothing -- RK3 time stepping, dt = 2 minutes.

    python examples/advection_only.py [N] [steps] [fusion level]       (needs the GPU)
"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import climaseaice_jl_amd as csi

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
L = N * 1000.0
grid = csi.RectilinearGrid((N, N), x=(0.0, L), y=(0.0, L), topology=(csi.Periodic, csi.Periodic), halo=(4, 4))
xf, yc = grid.xnodes(csi.Face), grid.ynodes(csi.Center)
xc, yf = grid.xnodes(csi.Center), grid.ynodes(csi.Face)
V = 0.5
u = V * np.sin(2 * np.pi * yc / L)[:, None] * np.cos(2 * np.pi * xf / L)[None, :]
v = -V * np.sin(2 * np.pi * xc / L)[None, :] * np.cos(2 * np.pi * yf / L)[:, None]
X, Y = np.meshgrid(xc, yc)
h0 = 0.3 + 0.005 * (np.sin(60 * X / 1000e3) + np.sin(30 * Y / 1000e3))
a0 = np.clip(1 - 0.1 * np.random.default_rng(2).random((N, N)), 0, 1)
model = csi.SeaIceModel(grid, dynamics=None, advection=csi.WENO(order=7), timestepper="SplitRungeKutta3")
if len(sys.argv) > 3:
    model.set_fusion(int(sys.argv[3]))      # 0: separate tendency / update launches instead of one launch per RK stage
csi.set_(model, h=h0, aice=a0, u=u, v=v)
V0 = (model.ice_thickness.interior_numpy() * model.ice_concentration.interior_numpy()).sum()
for n in range(10):
    csi.time_step(model, 120.0)
model.synchronize()
t0 = time.perf_counter()
for n in range(steps):
    csi.time_step(model, 120.0)
model.synchronize()
wall = time.perf_counter() - t0
h, a = model.ice_thickness.interior_numpy(), model.ice_concentration.interior_numpy()
print(f"{steps} RK3 steps of 2 min on {N}^2 in {wall * 1e3:.1f} ms: {wall / steps * 1e6:.0f} us per step, "
      f"{3 * steps * N * N / wall / 1e9:.2f} G cell-stages/s, {steps * 120.0 / 86400.0 / (wall / 3600.0):.0f} model-days/hr")
print(f"h in [{h.min():.4f}, {h.max():.4f}], aice in [{a.min():.4f}, {a.max():.4f}], ice volume drift {((h * a).sum() - V0) / V0:.2e}")
