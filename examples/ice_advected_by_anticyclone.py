"""The reference's examples/ice_advected_by_anticyclone.jl scaled to the library: EVP dynamics under a prescribed
anticyclonic wind stress on a bounded 512 km box, WENO(order = 7) advection, RK3 time stepping, dt = 2 minutes.

    python examples/ice_advected_by_anticyclone.py [N] [steps]       (needs the GPU)
"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import climaseaice_jl_amd as csi

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
L = 512e3
grid = csi.RectilinearGrid((N, N), x=(0.0, L), y=(0.0, L), topology=(csi.Bounded, csi.Bounded), halo=(4, 4))
# anticyclonic wind stress (arrays at the u and v points), ocean at rest under a quadratic drag
xu, yu = grid.xnodes(csi.Face)[None, :], grid.ynodes(csi.Center)[:, None]
xv, yv = grid.xnodes(csi.Center)[None, :], grid.ynodes(csi.Face)[:, None]
tau0 = 0.1
tau_u = -tau0 * (2 * yu - L) / L + 0 * xu
tau_v = tau0 * (2 * xv - L) / L + 0 * yv
dyn = csi.SeaIceMomentumEquation(grid, coriolis=csi.FPlane(f=1e-4), rheology=csi.ElastoViscoPlasticRheology(),
                                 top_momentum_stress=(tau_u, tau_v), bottom_momentum_stress=csi.SemiImplicitStress(),
                                 solver=csi.SplitExplicitSolver(substeps=120))
model = csi.SeaIceModel(grid, dynamics=dyn, advection=csi.WENO(order=7), timestepper="SplitRungeKutta3")
xc, yc = grid.xnodes(csi.Center)[None, :], grid.ynodes(csi.Center)[:, None]
h0 = 0.3 + 0.005 * (np.sin(60 * xc / 1000e3) + np.sin(30 * yc / 1000e3))
csi.set_(model, h=h0, aice=np.ones((N, N)), u=0.0, v=0.0)
V0 = (model.ice_thickness.interior_numpy() * model.ice_concentration.interior_numpy()).sum()
t0 = time.perf_counter()
for n in range(steps):
    csi.time_step(model, 120.0)
model.synchronize()
wall = time.perf_counter() - t0
u, v = model.velocities.u.interior_numpy(), model.velocities.v.interior_numpy()
h, a = model.ice_thickness.interior_numpy(), model.ice_concentration.interior_numpy()
print(f"{steps} steps of 2 min on {N}^2 in {wall:.2f} s ({steps * 120.0 / 86400.0 / (wall / 3600.0):.0f} model-days/hr); path {model.ctx.last_path()}")
print(f"max |u| = {np.abs(u).max():.4f} m/s, max |v| = {np.abs(v).max():.4f} m/s, h in [{h.min():.4f}, {h.max():.4f}], aice in [{a.min():.4f}, {a.max():.4f}]")
print(f"ice volume drift: {((h * a).sum() - V0) / V0:.2e} (closed box: conserved up to ridging clips)")
