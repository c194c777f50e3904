"""Round 4: which N = 8 decomposition of the metric's 2048^2 grid gives the fastest tile?  One GPU, the tile connected to itself
over the peer transport in the directions its partition connects (2x4: both; 1x8: y only; 8x1: x only), 120 sub-steps.
python scripts/tile_shapes.py  ->  G cell-updates/s per tile shape (x 8 / the 2048^2 rate = projected parallel efficiency)"""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
import climaseaice_jl_amd as csi

def run(nx, ny, fc, steps=10, warm=3):
    g = csi.RectilinearGrid((nx, ny), x=(0.0, nx * 2000.0), y=(0.0, ny * 2000.0), topology=(csi.Periodic, csi.Periodic), halo=(4, 4))
    tg = csi.TileGrid(g, 1, 1, 0, 0, force_connected=fc) if fc else g
    f = bench.tile_fields(np, nx, ny, 1, 1, 0, 0)
    dyn = csi.SeaIceMomentumEquation(tg, coriolis=csi.FPlane(f=1e-4), rheology=csi.ElastoViscoPlasticRheology(),
                                     top_momentum_stress=(0.01, 0.01), bottom_momentum_stress=csi.SemiImplicitStress(),
                                     solver=csi.SplitExplicitSolver(substeps=120), device="cuda:0")
    m = csi.SeaIceModel(tg, dynamics=dyn, advection=None, timestepper="ForwardEuler", device="cuda:0", mode="fast")
    csi.set_(m, h=f["h"], aice=f["a"], u=f["u"], v=f["v"])
    for _ in range(warm):
        csi.time_step_momentum(m, 120.0)
    m.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        csi.time_step_momentum(m, 120.0)
    m.synchronize(); torch.cuda.synchronize()
    e = time.perf_counter() - t0
    return nx * ny * 120 * steps / e / 1e9, (m.ctx.halo_transport() if fc else "none")

out = {}
for name, nx, ny, fc in (("2048x2048 untiled", 2048, 2048, None),
                         ("2x4: 1024x512 untiled", 1024, 512, None), ("2x4: 1024x512 peer x+y", 1024, 512, (True, True)),
                         ("1x8: 2048x256 untiled", 2048, 256, None), ("1x8: 2048x256 peer y", 2048, 256, (False, True)),
                         ("8x1: 256x2048 peer x", 256, 2048, (True, False)), ("4x2: 512x1024 peer x+y", 512, 1024, (True, True)),
                         ("1x4: 2048x512 peer y", 2048, 512, (False, True)), ("2x2: 1024x1024 peer x+y", 1024, 1024, (True, True)),
                         ("1x2: 2048x1024 peer y", 2048, 1024, (False, True)), ("2x1: 1024x2048 peer x", 1024, 2048, (True, False))):
    for rep in range(2):
        v, tr = run(nx, ny, fc)
        out.setdefault(name, []).append(round(v, 2))
    print(name, out[name], tr, flush=True)
print(json.dumps(out))
