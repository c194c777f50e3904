"""Fuzz of real decompositions on one GPU (in-process tile group, tests/test_gpu_local_tiles.py): random partition, topology,
halo, features, sub-step count and exchange setting; the owned cells of every tile against the untiled three-kernel run, bit for
bit.  python scripts/fuzz_local_tiles.py [first last]"""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")      # (tests/conftest.py explains)
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import climaseaice_jl_amd as csi, cases
from test_gpu_evp import EVP_FIELDS
from test_gpu_local_tiles import run_tiles, check

bad = 0
stats = {"peer": 0, "rccl": 0, "level2": 0}
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 60)
for seed in range(lo, hi):
    rng = np.random.default_rng(91000 + seed)
    Rx, Ry = [(1, 2), (2, 1), (2, 2), (1, 3), (3, 1), (2, 3), (4, 1), (1, 4), (4, 2)][rng.integers(9)]
    H = int(rng.integers(4, 10))
    ty = ("periodic", "bounded", "folded")[rng.integers(3)] if Rx == 1 else ("periodic", "bounded")[rng.integers(2)]
    tx = "periodic" if ty == "folded" else ("periodic", "bounded")[rng.integers(2)]
    nx = int(rng.integers(max(2 * H + 2, 24), 160)); ny = int(rng.integers(max(2 * H + 2, 24), 80))
    if rng.integers(2):
        nx = max(nx, 128)                      # wide enough for the peer transport
    kw = dict(Nx=nx * Rx, Ny=ny * Ry, H=H, topo=(tx, ty), patches=bool(rng.integers(2)), random_uv=0.04,
              field_forcing=bool(rng.integers(2)), land=(0.0, 0.25)[rng.integers(2)], free_drift=bool(rng.integers(4) == 0),
              coriolis=(1e-4, None)[rng.integers(2)])
    if kw["free_drift"] and not kw["field_forcing"]:
        kw.update(ue=0.05, ve=-0.02, top=(0.03, -0.02))
    if ty == "folded" or (ty == "bounded" and tx == "periodic" and rng.integers(3) == 0):
        if rng.integers(2):
            kw["curvilinear"] = 0.04
    elif ty == "bounded" and rng.integers(3) == 0:
        kw["grid"] = "latlon"
    if "bounded" in (tx, ty) and rng.integers(4) == 0:
        kw["noslip"] = True
    if rng.integers(5) == 0:
        kw["user_forcing"] = True
        kw["free_drift"] = False
    nsub = int(rng.integers(1, 14))
    k = int([0, 0, -1, 1, 2][rng.integers(5)])
    if k == 2 and H < 4:
        k = 1
    try:
        c = cases.make_case(substeps=nsub, **kw)
        ref = cases.csi_model(c, mode="fast"); ref.set_fusion(0)
        csi.time_step_momentum(ref, c["dt"]); csi.time_step_momentum(ref, c["dt"])
        ref.synchronize()
        mom = {f: EVP_FIELDS[f](ref).interior_numpy().copy() for f in ("u", "v", "s11", "s22", "s12")}
        tiles = run_tiles(c, Rx, Ry, k, full_step=False)
        check(tiles, mom, {}, (seed,))
        tr = {d["path"]["transport"] for d in tiles}
        assert len(tr) == 1, tr                 # every rank or none
        stats[tr.pop()] += 1
        stats["level2"] += all(d["path"]["level"] == 2 for d in tiles)
    except BaseException as e:
        bad += 1
        print("FAIL", seed, (Rx, Ry), kw, "nsub", nsub, "k", k, type(e).__name__, str(e)[:300], flush=True)
print("done, failures:", bad, "of", hi - lo, stats)
