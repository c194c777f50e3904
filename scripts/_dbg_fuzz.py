import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import climaseaice_jl_amd as csi, cases
from test_gpu_evp import EVP_FIELDS, cmp_region
H = 4
for kwx in (dict(), dict(field_forcing=True), dict(land=0.25), dict(field_forcing=True, land=0.25)):
  for nsub in (2, 4):
    kw = dict(Nx=157, Ny=20, H=H, topo=("bounded", "periodic"), patches=False, random_uv=0.04, pressure="ice_strength", **kwx)
    c = cases.make_case(substeps=nsub, **kw)
    out = {}
    for fusion in (0, 2):
        m = cases.csi_model(c, mode="fast"); m.set_fusion(fusion)
        csi.time_step_momentum(m, c["dt"]); m.synchronize()
        out[fusion] = {k: cmp_region(c, k, EVP_FIELDS[k](m).numpy()).copy() for k in ("u", "v", "s11", "s22", "s12")}
    print(kwx, nsub)
    for k in out[0]:
        d = np.argwhere(out[0][k] != out[2][k])
        if len(d):
            off = 0 if k == "s12" else (H - 1)
            j = d[:, 0] - off; i = d[:, 1] - off
            print("  ", k, len(d), "rows j:", sorted(set(j.tolist()))[:16], "cols i:", sorted(set(i.tolist()))[:16], " max", np.abs(out[0][k] - out[2][k]).max())
