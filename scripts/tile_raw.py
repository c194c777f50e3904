"""Raw launch time of the sub-cycle on one tile shape, no result checks (for TIMING EXPERIMENTS with builds that compute wrong
results, e.g. -DCSI_EXP_RINGCUT): python scripts/tile_raw.py NX NY [peer_y [tier]]   (CSI_HIP_LIBRARY selects the build)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import bench
import climaseaice_jl_amd as csi

nx, ny = int(sys.argv[1]), int(sys.argv[2])
fc = (False, True) if len(sys.argv) > 3 else None
tier = int(sys.argv[4]) if len(sys.argv) > 4 else None      # protocol tier of the peer transport (default: the library's choice)
g = csi.RectilinearGrid((nx, ny), x=(0.0, nx * 2000.0), y=(0.0, ny * 2000.0), topology=(csi.Periodic, csi.Periodic), halo=(4, 4))
tg = csi.TileGrid(g, 1, 1, 0, 0, force_connected=fc) if fc else g
f = bench.tile_fields(np, nx, ny, 1, 1, 0, 0)
dyn = csi.SeaIceMomentumEquation(tg, coriolis=csi.FPlane(f=1e-4), rheology=csi.ElastoViscoPlasticRheology(), top_momentum_stress=(0.01, 0.01),
                                 bottom_momentum_stress=csi.SeaIceMomentumEquation and csi.SemiImplicitStress(), solver=csi.SplitExplicitSolver(substeps=120), device="cuda:0")
m = csi.SeaIceModel(tg, dynamics=dyn, advection=None, timestepper="ForwardEuler", device="cuda:0", mode="fast")
if tier is not None:
    m.set_peer_tier(tier)
best = 0.0
for rep in range(3):
    csi.set_(m, h=f["h"], aice=f["a"], u=f["u"], v=f["v"])
    for _ in range(2):
        csi.time_step_momentum(m, 120.0)
    m.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(6):
        csi.time_step_momentum(m, 120.0)
    m.synchronize(); torch.cuda.synchronize()
    best = max(best, nx * ny * 120 * 6 / (time.perf_counter() - t0) / 1e9)
print(os.path.basename(os.environ.get("CSI_HIP_LIBRARY", "default")), nx, ny, ("peer_y tier %s" % m.ctx.peer_tier()) if fc else "untiled", round(best, 2), round(m.ctx.last_subcycle_ms() * 1e3 / 60, 2), "us/launch")
