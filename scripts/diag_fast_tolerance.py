"""Diagnostic: FAST-vs-oracle differences per case next to the oracle's own sensitivity to a
1e-15 relative perturbation of the inputs (conditioning of the EVP iteration)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import cases, climaseaice_jl_amd as csi
from test_gpu_evp import CASES

for sub in (1, 2, 10, 40, 120):
    print(f"== substeps {sub}")
    for name, kw in CASES.items():
        c = cases.make_case(substeps=sub, **kw)
        p = cases.oracle_problem(c); p.time_step_momentum(c["dt"])
        c2 = dict(c); rng = np.random.default_rng(11)
        c2["u"] = c["u"] * (1 + 1e-15 * rng.standard_normal(c["u"].shape)); c2["v"] = c["v"] * (1 + 1e-15 * rng.standard_normal(c["v"].shape))
        c2["h"] = c["h"] * (1 + 1e-15 * rng.standard_normal(c["h"].shape))
        p2 = cases.oracle_problem(c2); p2.time_step_momentum(c["dt"])
        m = cases.csi_model(c, mode="fast"); csi.time_step_momentum(m, c["dt"]); m.synchronize()
        vmax = max(np.abs(p.f["u"]).max(), np.abs(p.f["v"]).max())
        smax = max(np.abs(p.f["s11"]).max(), np.abs(p.f["s22"]).max(), np.abs(p.f["s12"]).max())
        du = max(np.abs(m.velocities.u.numpy() - p.f["u"]).max(), np.abs(m.velocities.v.numpy() - p.f["v"]).max()) / vmax
        ds = np.abs(m.dynamics.auxiliaries.fields.s11.numpy() - p.f["s11"]).max() / smax
        su = max(np.abs(p2.f["u"] - p.f["u"]).max(), np.abs(p2.f["v"] - p.f["v"]).max()) / vmax
        ss = np.abs(p2.f["s11"] - p.f["s11"]).max() / smax
        print(f"  {name:26s} fast-vs-oracle du {du:.2e} ds11 {ds:.2e} | oracle self-sensitivity du {su:.2e} ds11 {ss:.2e}")
