"""Generate the golden fixtures of tests/golden/*.npz with the C oracle (oracle/csi_oracle.c).

The reference is Julia and cannot run in this image ("parity unpinned", oracle/csi_oracle.h), so
these vectors are outputs of the strict-order restatement, cross-validated bit for bit by the
independent NumPy restatement (tests/test_oracle_properties.py).  They pin the oracle against
regressions and give the GPU tests a fixed target that does not depend on the build machine.

Run from the repo root:  python tests/golden/generate_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import cases  # noqa: E402

GOLDEN = {
    "evp_periodic_32x32": dict(Nx=32, Ny=32, topo=("periodic", "periodic"), patches=True, random_uv=0.05, ue=0.03, ve=-0.01),
    "evp_bounded_latlon_48x32": dict(Nx=48, Ny=32, topo=("bounded", "bounded"), grid="latlon", patches=True, random_uv=0.05),
    "evp_channel_64x48": dict(Nx=64, Ny=48, topo=("periodic", "bounded"), patches=False, random_uv=0.02),
}
OUT_FIELDS = ("u", "v", "s11", "s22", "s12", "alpha", "zeta_c", "zeta_f", "Delta")


def run(kw, nsub):
    c = cases.make_case(substeps=nsub, **kw)
    p = cases.oracle_problem(c)
    p.initialize_rheology()
    P = p.f["P"].copy()
    p.L.ora_fill_halo_u(p.ptr); p.L.ora_fill_halo_v(p.ptr)
    p.subcycle(c["dt"], 1, nsub)
    p.L.ora_finalize_rheology(p.ptr)
    return c, P, {k: p.f[k].copy() for k in OUT_FIELDS}


if __name__ == "__main__":
    for name, kw in GOLDEN.items():
        data = {"case_json": json.dumps(kw)}
        for nsub in (1, 10):
            c, P, out = run(kw, nsub)
            data["P"] = P
            for k, v in out.items():
                data[f"{k}_after{nsub}"] = v
        for k in ("h", "a", "u", "v"):
            data[f"in_{k}"] = c[k]
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **data)
        print(name, {k: v.shape for k, v in data.items() if hasattr(v, "shape")})
