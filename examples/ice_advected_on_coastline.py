"""The reference's examples/ice_advected_on_coastline.jl on the library: a 512 km x 256 km channel (periodic in x, walls in
y) with a triangular coastline as an immersed boundary, no-slip walls (ValueBoundaryCondition(0) on u), a constant wind
blowing the ice onto the coast (wind-stress array in x, zero in y), quadratic ocean drag, EVP with 150 sub-steps, WENO(order = 7)
advection, RK3, dt = 5 minutes, 3 days.

    python examples/ice_advected_on_coastline.py [days]       (needs the GPU)
"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import climaseaice_jl_amd as csi

days = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
Lx, Ly, Nx, Ny = 512e3, 256e3, 256, 128
y_max = Ly / 2
grid = csi.RectilinearGrid((Nx, Ny), x=(-Lx / 2, Lx / 2), y=(0.0, Ly), topology=(csi.Periodic, csi.Bounded), halo=(4, 4))
xc, yc = grid.xnodes(csi.Center)[None, :], grid.ynodes(csi.Center)[:, None]
# bottom(x, y) = y > y_max ? 0 : (|x / Lx| Nx + y / Ly Ny > 24 ? 0 : 1): the solid triangle is where bottom == 1
wet = (yc > y_max) | (np.abs(xc / Lx) * Nx + yc / Ly * Ny > 24)
# wind stress tau_u = - rho_a C_D U_a^2 with U_a = 10 m/s everywhere, tau_v = 0
tau_u = -1.3 * 1.2e-3 * 10.0 ** 2
vbc = csi.ValueBoundaryCondition(0.0)
dyn = csi.SeaIceMomentumEquation(grid, rheology=csi.ElastoViscoPlasticRheology(),
                                 top_momentum_stress=(np.full((Ny, Nx), tau_u), 0.0), bottom_momentum_stress=csi.SemiImplicitStress(),
                                 solver=csi.SplitExplicitSolver(substeps=150))
model = csi.SeaIceModel(grid, dynamics=dyn, advection=csi.WENO(order=7), timestepper="SplitRungeKutta3",
                        boundary_conditions=dict(u=csi.FieldBoundaryConditions(north=vbc, south=vbc)))
model.set_mask(wet)
for comp in ("U", "V"):
    model.ctx.call("csi_fill_halo_local", csi._lib.F[f"TOP_{comp}"])
csi.set_(model, h=1.0, aice=1.0)
V0 = (model.ice_thickness.interior_numpy() * model.ice_concentration.interior_numpy()).sum()
steps = int(days * 86400 / 300)
t0 = time.perf_counter()
for n in range(steps):
    csi.time_step(model, 300.0)
model.synchronize()
wall = time.perf_counter() - t0
u, v = model.velocities.u.interior_numpy(), model.velocities.v.interior_numpy()
h, a = model.ice_thickness.interior_numpy(), model.ice_concentration.interior_numpy()
print(f"{steps} steps of 5 min on {Nx} x {Ny} in {wall:.2f} s ({days / (wall / 3600.0):.0f} model-days/hr); path {model.ctx.last_path()}")
print(f"max |u| = {np.abs(u).max():.4f}, max |v| = {np.abs(v).max():.4f} m/s; h in [{h[wet].min():.3f}, {h[wet].max():.3f}], "
      f"aice in [{a[wet].min():.3f}, {a[wet].max():.3f}]")
print(f"ice piles up against the coast: max h {h.max():.3f} m (started at 1 m); on land h == 0: {bool(np.all(h[~wet] == 0))}; "
      f"volume drift {((h * a).sum() - V0) / V0:.2e}")
