"""Fuzz of tiled (self-exchange) configurations against the untiled three-kernel run, owned cells bit for bit (run on the GPU box)."""
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import climaseaice_jl_amd as csi, cases
from test_gpu_evp import EVP_FIELDS
bad = 0
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 80)
for seed in range(lo, hi):
    rng = np.random.default_rng(5000 + seed)
    H = int(rng.integers(4, 11))
    k = int(rng.integers(1, H // 2 + 1))
    peer = bool(rng.integers(2))            # round 3: half of the configurations on the peer transport (k = 0)
    fc = [(True, True), (True, False), (False, True)][rng.integers(3)]
    topo = ("periodic" if fc[0] or rng.integers(2) else "bounded", "periodic" if fc[1] or rng.integers(2) else "bounded")
    Nx = int(rng.integers(2 * H + 2, 260)); Ny = int(rng.integers(2 * H + 2, 80))
    kw = dict(Nx=Nx, Ny=Ny, H=H, topo=topo, patches=bool(rng.integers(2)), random_uv=0.04,
              field_forcing=bool(rng.integers(2)), land=(0.0, 0.25)[rng.integers(2)], free_drift=bool(rng.integers(4) == 0))
    if kw["free_drift"] and not kw["field_forcing"]:
        kw.update(ue=0.05, ve=-0.02, top=(0.03, -0.02))
    if topo[1] == "bounded" and rng.integers(3) == 0:
        kw["beta"] = 2e-10
    if "bounded" in topo and rng.integers(3) == 0:
        kw["noslip"] = True
    if rng.integers(5) == 0:
        kw["curvilinear"] = 0.04
    nsub = int(rng.integers(2, 14))
    if rng.integers(4) == 0:
        kw["user_forcing"] = True
    if kw["land"] and rng.integers(4) == 0:
        kw["immersed_bc"] = ((0.02, -0.01, 0.015, 0.005), (-0.01, 0.02, 0.01, -0.015))
    if kw.get("free_drift") and (kw.get("user_forcing") or kw.get("immersed_bc")):
        kw["free_drift"] = False
    if peer:
        k = 0
    try:
        c = cases.make_case(substeps=nsub, **kw)
        ref = cases.csi_model(c, mode="fast"); ref.set_fusion(0)
        csi.time_step_momentum(ref, c["dt"])
        til = cases.csi_model(c, mode="fast", tile=(1, 1, 0, fc)); til.set_exchange_interval(k)
        csi.time_step_momentum(til, c["dt"])
        ref.synchronize(); til.synchronize()
        if peer and Nx >= 128:
            assert til.ctx.halo_transport() == "peer", til.ctx.last_path()
        for f in ("u", "v", "s11", "s22", "s12", "alpha"):
            a, b = EVP_FIELDS[f](ref).interior_numpy(), EVP_FIELDS[f](til).interior_numpy()
            assert np.array_equal(a, b), (f, np.abs(a - b).max(), np.argwhere(a != b)[:3].tolist())
    except Exception as e:
        bad += 1
        print("FAIL", seed, kw, "k", k, "fc", fc, "nsub", nsub, type(e).__name__, str(e)[:200])
print("done, failures:", bad)
