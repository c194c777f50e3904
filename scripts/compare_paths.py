"""Debug aid: run one named case of tests/test_gpu_evp.py::CASES with the three-kernel path and with the fused paths and
print where they differ (they must not): python scripts/compare_paths.py <case> [substeps]"""
import sys
sys.path[:0] = [".", "tests", "oracle"]
import faulthandler
import numpy as np
import cases
import climaseaice_jl_amd as csi
from test_gpu_evp import CASES, EVP_FIELDS
name = sys.argv[1] if len(sys.argv) > 1 else "curvilinear_bounded"
nsub = int(sys.argv[2]) if len(sys.argv) > 2 else 2
c = cases.make_case(substeps=nsub, **CASES[name])
out = {}
for fusion in (0, 2):
    m = cases.csi_model(c, mode="fast")
    m.set_fusion(fusion)
    print("running fusion", fusion, flush=True)
    csi.time_step_momentum(m, c["dt"])
    m.synchronize()
    print("level", m.ctx.last_path(), flush=True)
    out[fusion] = {k: EVP_FIELDS[k](m).interior_numpy().copy() for k in ("u", "v", "s11", "s22", "s12")}
for k in out[0]:
    d = np.abs(out[0][k] - out[2][k])
    print(k, "max diff", d.max(), "n diff", (d > 0).sum(), np.argwhere(d > 0)[:5].tolist())
