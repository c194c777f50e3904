"""Round 6, the N = 8 question: would overlapping consecutive launches of ONE tile buy anything?  A launch on the 2048 x 256 slab is
4.9 us of ramp + 17 row iterations of 1.1 us (profiles/r05_tile.md); launches of one tile depend on each other, so overlapping
them needs every tile on a flag protocol.  BEFORE building that: an upper bound.  Two INDEPENDENT tile models (two contexts, two
streams) advanced at the same time overlap everything that can overlap -- ramps, tails, idle SIMD slots -- with no dependency to
wait for.  If the pair's aggregate rate is not well above one model's, a flag protocol on all tiles cannot pay either.
python scripts/tile_overlap_bound.py  ->  one tile alone vs two at once, per tile shape."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
import climaseaice_jl_amd as csi


def model(nx, ny, fc):
    g = csi.RectilinearGrid((nx, ny), x=(0.0, nx * 2000.0), y=(0.0, ny * 2000.0), topology=(csi.Periodic, csi.Periodic), halo=(4, 4))
    tg = csi.TileGrid(g, 1, 1, 0, 0, force_connected=fc) if fc else g
    f = bench.tile_fields(np, nx, ny, 1, 1, 0, 0)
    dyn = csi.SeaIceMomentumEquation(tg, coriolis=csi.FPlane(f=1e-4), rheology=csi.ElastoViscoPlasticRheology(),
                                     top_momentum_stress=(0.01, 0.01), bottom_momentum_stress=csi.SemiImplicitStress(),
                                     solver=csi.SplitExplicitSolver(substeps=120), device="cuda:0")
    m = csi.SeaIceModel(tg, dynamics=dyn, advection=None, timestepper="ForwardEuler", device="cuda:0", mode="fast")
    csi.set_(m, h=f["h"], aice=f["a"], u=f["u"], v=f["v"])
    return m


def rate(models, nx, ny, steps=10, warm=3):
    for _ in range(warm):
        for m in models:
            csi.time_step_momentum(m, 120.0)
    for m in models:
        m.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for m in models:                      # the host runs far ahead of the GPU: both streams hold work all the time
            csi.time_step_momentum(m, 120.0)
    for m in models:
        m.synchronize()
    e = time.perf_counter() - t0
    return len(models) * nx * ny * 120 * steps / e / 1e9


out = {}
for name, nx, ny, fc in (("1x8 slab 2048x256, peer y", 2048, 256, (False, True)), ("2x4 tile 1024x512, peer x+y", 1024, 512, (True, True)),
                         ("1x4 slab 2048x512, peer y", 2048, 512, (False, True)), ("2048x2048 untiled", 2048, 2048, None)):
    a, b = model(nx, ny, fc), model(nx, ny, fc)
    rec = {"one": [], "two_at_once": []}
    for rep in range(3):
        rec["one"].append(round(rate([a], nx, ny), 2))
        rec["two_at_once"].append(round(rate([a, b], nx, ny), 2))
    rec["gain"] = round(max(rec["two_at_once"]) / max(rec["one"]), 3)
    out[name] = rec
    print(name, rec, flush=True)
    a = b = None
print(json.dumps(out))
