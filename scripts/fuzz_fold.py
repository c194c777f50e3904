"""Fuzz of north-fold configurations: fusion level 2 (pair kernel below a three-kernel band, csi_abi.hip FoldBand) against the
three-kernel run of the whole grid, bit for bit (run on the GPU box): python scripts/fuzz_fold.py [first last]"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import climaseaice_jl_amd as csi, cases
from test_gpu_evp import EVP_FIELDS, cmp_region
bad = fused = 0
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 100)
for seed in range(lo, hi):
    rng = np.random.default_rng(77000 + seed)
    H = int(rng.integers(4, 10))
    Nx = int(rng.integers(2 * H + 2, 300)); Ny = int(rng.integers(2 * H + 2, 120))
    kw = dict(Nx=Nx, Ny=Ny, H=H, topo=("periodic", "folded"), patches=bool(rng.integers(2)), random_uv=0.04,
              field_forcing=bool(rng.integers(2)), land=(0.0, 0.25)[rng.integers(2)], free_drift=bool(rng.integers(4) == 0),
              coriolis=(1e-4, None)[rng.integers(2)], pressure=("replacement", "ice_strength")[rng.integers(2)])
    if kw["free_drift"] and not kw["field_forcing"]:
        kw.update(ue=0.05, ve=-0.02, top=(0.03, -0.02))
    if rng.integers(2):
        kw["curvilinear"] = 0.04
        if kw["coriolis"] is not None and rng.integers(2):
            kw["coriolis_points"] = True
    if rng.integers(4) == 0:
        kw["noslip"] = True
    if rng.integers(4) == 0:
        kw["user_forcing"] = True
    if kw["land"] and rng.integers(4) == 0:
        kw["immersed_bc"] = ((0.02, -0.01, 0.015, 0.005), (-0.01, 0.02, 0.01, -0.015))
    if kw.get("user_forcing") or kw.get("immersed_bc"):
        kw["free_drift"] = False
    nsub = int(rng.integers(1, 14))
    try:
        c = cases.make_case(substeps=nsub, **kw)
        ref = cases.csi_model(c, mode="fast"); ref.set_fusion(0)
        new = cases.csi_model(c, mode="fast")
        for _ in range(2):
            csi.time_step_momentum(ref, c["dt"]); csi.time_step_momentum(new, c["dt"])
        ref.synchronize(); new.synchronize()
        fused += new.ctx.last_path()["level"] == 2
        for f in ("u", "v", "s11", "s22", "s12"):
            a, b = cmp_region(c, f, EVP_FIELDS[f](ref).numpy()), cmp_region(c, f, EVP_FIELDS[f](new).numpy())
            assert np.array_equal(a, b), (f, np.abs(a - b).max(), np.argwhere(a != b)[:3].tolist())
        for f in ("alpha", "zeta_c", "Delta"):
            a, b = EVP_FIELDS[f](ref).interior_numpy(), EVP_FIELDS[f](new).interior_numpy()
            assert np.array_equal(a, b), (f, np.abs(a - b).max(), np.argwhere(a != b)[:3].tolist())
    except Exception as e:
        bad += 1
        print("FAIL", seed, kw, "nsub", nsub, type(e).__name__, str(e)[:300])
print("done, failures:", bad, "of", hi - lo, "; on the band path:", fused)
