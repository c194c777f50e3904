"""What each feature of the tripolar-like configuration costs the pair kernel at 2048^2: curvilinear metrics, + arrays, + land, + free drift,
+ north fold (profiles/r05_full_metric.md, addendum).  Run on the GPU box: python scripts/feature_costs.py"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import climaseaice_jl_amd as csi
import cases
N = 2048
C = {
 "curv": dict(topo=("periodic", "bounded"), curvilinear=0.05),
 "curv+land": dict(topo=("periodic", "bounded"), curvilinear=0.05, land=0.3),
 "curv+arrays": dict(topo=("periodic", "bounded"), curvilinear=0.05, field_forcing=True),
 "curv+land+arrays": dict(topo=("periodic", "bounded"), curvilinear=0.05, land=0.3, field_forcing=True),
 "curv+land+arrays+fd": dict(topo=("periodic", "bounded"), curvilinear=0.05, land=0.3, field_forcing=True, free_drift=True),
 "curv+fold": dict(topo=("periodic", "folded"), curvilinear=0.05),
 "curv+land+arrays+fd+fold (tripolar-like)": dict(topo=("periodic", "folded"), curvilinear=0.05, land=0.3, field_forcing=True, free_drift=True),
}
for name, kw in C.items():
    c = cases.make_case(Nx=N, Ny=N, substeps=120, patches=False, noise=0.05, **kw)
    m = cases.csi_model(c, mode="fast")
    for _ in range(2): csi.time_step_momentum(m, c["dt"])
    m.synchronize(); t0 = time.perf_counter()
    for _ in range(3): csi.time_step_momentum(m, c["dt"])
    m.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(name, round(N * N * 120 / dt / 1e9, 2), "G", round(dt * 1e3 / 60, 4), "ms per launch pair", flush=True)
    del m
