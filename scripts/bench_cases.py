"""Throughput of the EVP sub-cycle on the other SURVEY.md 8(d) style configurations (not the bench.py headline):
2048^2, 120 sub-steps, FAST mode, fusion levels 0 / 1 / 2.  Run on the GPU box: python scripts/bench_cases.py"""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import climaseaice_jl_amd as csi
import cases

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
CONFIGS = {
    "periodic f-plane": dict(topo=("periodic", "periodic")),
    "channel (periodic x, walls y)": dict(topo=("periodic", "bounded")),
    "bounded": dict(topo=("bounded", "bounded")),
    "lat-lon bounded (per-row metrics)": dict(topo=("bounded", "bounded"), grid="latlon"),
    "lat-lon channel": dict(topo=("periodic", "bounded"), grid="latlon"),
    "masked channel (30 % land, config 5 style)": dict(topo=("periodic", "bounded"), land=0.3),
    "coupled channel (wind-stress arrays, ocean-velocity arrays, 30 % land)": dict(topo=("periodic", "bounded"), land=0.3, field_forcing=True),
    "coupled periodic (arrays, no land)": dict(topo=("periodic", "periodic"), field_forcing=True),
    "OMIP style (arrays, 30 % land, StressBalanceFreeDrift; test/distributed_tests_utils.jl:190-212)":
        dict(topo=("periodic", "bounded"), land=0.3, field_forcing=True, free_drift=True),
    "model.forcing arrays (periodic)": dict(topo=("periodic", "periodic"), user_forcing=True),
    "wind drag: SemiImplicitStress on top with air-velocity arrays + ocean-velocity arrays (periodic)": dict(topo=("periodic", "periodic"), wind_drag="arrays", field_forcing=True),
    "immersed flux boundary conditions (channel, 30 % land)": dict(topo=("periodic", "bounded"), land=0.3,
                                                                     immersed_bc=((0.02, -0.01, 0.015, 0.005), (-0.01, 0.02, 0.01, -0.015))),
    "beta-plane channel (per-row f on uniform metrics)": dict(topo=("periodic", "bounded"), beta=1.6e-11),
    "no-slip channel with 30 % land (coastline-example style)": dict(topo=("periodic", "bounded"), land=0.3, noslip=True),
    "curvilinear channel (twelve 2-D metric arrays, CSI_METRIC_FULL)": dict(topo=("periodic", "bounded"), curvilinear=0.05),
    "curvilinear channel with 30 % land": dict(topo=("periodic", "bounded"), curvilinear=0.05, land=0.3),
    "north fold on uniform metrics (periodic x, RightFolded y)": dict(topo=("periodic", "folded")),
    "tripolar-like (north fold, curvilinear, 30 % land, arrays, free drift)": dict(topo=("periodic", "folded"), curvilinear=0.05, land=0.3, field_forcing=True, free_drift=True),
}
if len(sys.argv) > 2:
    CONFIGS = {k: v for k, v in CONFIGS.items() if sys.argv[2] in k}
LEVELS = (2,) if (len(sys.argv) > 3 and sys.argv[3] == "level2") else (0, 1, 2)
out = {}
for name, kw in CONFIGS.items():
    c = cases.make_case(Nx=N, Ny=N, substeps=120, patches=False, noise=0.05, **kw)
    row = {}
    for level in LEVELS:
        m = cases.csi_model(c, mode="fast")
        m.set_fusion(level)
        for _ in range(2):
            csi.time_step_momentum(m, c["dt"])
        m.synchronize()
        t0 = time.perf_counter()
        n = 3
        for _ in range(n):
            csi.time_step_momentum(m, c["dt"])
        m.synchronize()
        dt = (time.perf_counter() - t0) / n
        row[f"level{level}"] = round(N * N * 120 / dt / 1e9, 2)
        row[f"path{level}"] = m.ctx.last_path()["level"]
        del m
    out[name] = row
    print(name, row, flush=True)
print(json.dumps(out))
