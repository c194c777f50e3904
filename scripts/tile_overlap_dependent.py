"""Round 6, the N = 8 question, second part: the DEPENDENT overlap, measured.  One N = 8 tile connected to itself
over the peer transport; three ways of ordering consecutive launches:
  base       : the product -- the two edge chunks on the flag protocol, one stream (launch n + 1 starts when launch n has ended)
  all_flags  : CSI_EXP_OVERLAP=1 -- EVERY tile publishes / waits (what the protocol itself costs), still one stream
  two_streams: CSI_EXP_OVERLAP=3 -- every tile on the protocol AND launches alternating over two streams: nothing but the flags orders
               launch n + 1 behind launch n, so its workgroups are dispatched, run their prologue and spin while launch n's last tiles finish
With the default library every tile waits for ALL tiles of the previous launch (the existing wait: the ramp of consecutive launches overlaps,
nothing else); with the `nbr9` build (scripts/build_variant.sh nbr9 "-DCSI_PEER_EXP=11", CSI_HIP_LIBRARY=.../libcsi_hip_nbr9.so) a tile waits
for the nine tiles around it only -- the true dependency.  Results of the three are compared bit for bit before anything is timed.
python scripts/tile_overlap_dependent.py [tier]"""
import os, sys, time, json, hashlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
import climaseaice_jl_amd as csi

TIER = int(sys.argv[1]) if len(sys.argv) > 1 else -1


def model(nx, ny, fc, exp):
    if exp: os.environ["CSI_EXP_OVERLAP"] = str(exp)
    else: os.environ.pop("CSI_EXP_OVERLAP", None)
    g = csi.RectilinearGrid((nx, ny), x=(0.0, nx * 2000.0), y=(0.0, ny * 2000.0), topology=(csi.Periodic, csi.Periodic), halo=(4, 4))
    tg = csi.TileGrid(g, 1, 1, 0, 0, force_connected=fc)
    f = bench.tile_fields(np, nx, ny, 1, 1, 0, 0)
    dyn = csi.SeaIceMomentumEquation(tg, coriolis=csi.FPlane(f=1e-4), rheology=csi.ElastoViscoPlasticRheology(),
                                     top_momentum_stress=(0.01, 0.01), bottom_momentum_stress=csi.SemiImplicitStress(),
                                     solver=csi.SplitExplicitSolver(substeps=120), device="cuda:0")
    m = csi.SeaIceModel(tg, dynamics=dyn, advection=None, timestepper="ForwardEuler", device="cuda:0", mode="fast")
    os.environ.pop("CSI_EXP_OVERLAP", None)
    if TIER >= 0: m.set_peer_tier(TIER)
    csi.set_(m, h=f["h"], aice=f["a"], u=f["u"], v=f["v"])
    return m


def state(m):
    m.synchronize()
    a = m.dynamics.auxiliaries.fields
    return [np.array(x.numpy()) for x in (m.velocities.u, m.velocities.v, a.s11, a.s22, a.s12)]


def rate(m, nx, ny, steps=10, warm=3):
    for _ in range(warm): csi.time_step_momentum(m, 120.0)
    m.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): csi.time_step_momentum(m, 120.0)
    m.synchronize()
    return nx * ny * 120 * steps / (time.perf_counter() - t0) / 1e9


out = {"library": os.environ.get("CSI_HIP_LIBRARY", "default"), "tier": TIER}
NBR9 = "nbr9" in out["library"]
REF_FILE = os.path.join(ROOT, "gpurun_out", "ov", "ref_tier%d.json" % TIER)
REF = json.load(open(REF_FILE)) if NBR9 and os.path.exists(REF_FILE) else {}
for name, nx, ny, fc in (("1x8 slab 2048x256, peer y", 2048, 256, (False, True)), ("2x4 tile 1024x512, peer x+y", 1024, 512, (True, True)),
                         ("1x4 slab 2048x512, peer y", 2048, 512, (False, True))):
    if NBR9 and fc[0]: continue          # (the nine-tile wait reads the S array only: y-connected tiles)
    rec = {}
    try:
        ms = {"base": model(nx, ny, fc, 0), "all_flags": model(nx, ny, fc, 1), "two_streams": model(nx, ny, fc, 3)}
        if NBR9: del ms["base"]                               # (that build's wait is valid only with every tile in the sets)
        for m in ms.values(): csi.time_step_momentum(m, 120.0); m.synchronize()
    except Exception as e:                                     # (e.g. more tiles than flag slots)
        out[name] = {"error": str(e)[:300]}
        print(name, out[name], flush=True)
        continue
    for k, m in ms.items():
        for _ in range(2): csi.time_step_momentum(m, 120.0)
    # (three steps so far, the same in every run: the default library's `base` result is the reference, by hash across processes)
    digest = {k: hashlib.sha256(b"".join(a.tobytes() for a in state(m))).hexdigest()[:16] for k, m in ms.items()}
    if not NBR9: REF[name] = {"sha": digest["base"]}
    for k in ("all_flags", "two_streams"): rec["bitwise_" + k] = digest[k] == REF.get(name, {}).get("sha")
    for k in ms: rec[k] = []
    for rep in range(3):
        for k, m in ms.items(): rec[k].append(round(rate(m, nx, ny), 2))
    if not NBR9: REF[name]["base"] = max(rec["base"])
    rec["base_rate_used"] = REF.get(name, {}).get("base")
    for k in ("all_flags", "two_streams"): rec["gain_" + k] = round(max(rec[k]) / rec["base_rate_used"], 3) if rec["base_rate_used"] else None
    out[name] = rec
    print(name, rec, flush=True)
    ms = None
if not NBR9:
    os.makedirs(os.path.dirname(REF_FILE), exist_ok=True)
    json.dump(REF, open(REF_FILE, "w"))
print(json.dumps(out))
