"""Bitwise comparison of two builds of the library on the named cases of tests/test_gpu_evp.py (every fusion level, FAST mode):
python scripts/compare_libs.py dump <out.npz> [substeps]     (run once per build, CSI_HIP_LIBRARY selects it)
python scripts/compare_libs.py diff <a.npz> <b.npz>"""
import sys
sys.path[:0] = [".", "tests", "oracle"]
import numpy as np
if sys.argv[1] == "diff":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    bad = 0
    for k in a.files:
        same = a[k].tobytes() == b[k].tobytes()
        if not same:
            bad += 1
            d = np.abs(a[k] - b[k])
            print("DIFF", k, "max", np.nanmax(d), "count", int((d > 0).sum()), "of", d.size)
    print("arrays", len(a.files), "different", bad)
    sys.exit(1 if bad else 0)
import cases
import climaseaice_jl_amd as csi
from test_gpu_evp import CASES, EVP_FIELDS
nsub = int(sys.argv[3]) if len(sys.argv) > 3 else 24
out = {}
for name in sorted(CASES):
    c = cases.make_case(substeps=nsub, **CASES[name])
    for fusion in (0, 2):
        try:
            m = cases.csi_model(c, mode="fast")
        except Exception as e:
            print("skip", name, e); break
        m.set_fusion(fusion)
        for _ in range(2):
            csi.time_step_momentum(m, c["dt"])
        m.synchronize()
        for k in ("u", "v", "s11", "s22", "s12", "alpha", "zeta_c", "zeta_f", "Delta"):
            if k in EVP_FIELDS:
                out[f"{name}/{fusion}/{k}"] = EVP_FIELDS[k](m).interior_numpy().copy()
        print(name, fusion, m.ctx.last_path()["level"], flush=True)
np.savez(sys.argv[2], **out)
print("saved", len(out))
