"""Same-box A/B of the round-6 structure cuts: cell-updates/s (ALL cells counted) of one configuration with tile skipping / row-constant
rows on and off, alternating.  python scripts/ab_skip.py <case> [N] [reps]   (cases: see CASES)"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "scripts")):
    sys.path.insert(0, p)
import climaseaice_jl_amd as csi
import cases

from structure_cases import CASES      # noqa: E402
name = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
sub = int(os.environ.get("SUBSTEPS", "120"))
c = cases.make_case(Nx=N, Ny=N, substeps=sub, patches=False, noise=0.05, **CASES[name])
wet = 1.0 if c["mask"] is None else float(c["mask"].mean())
icy = float(((c["h"] > 0) & (c["a"] > 0)).mean())
out = {"case": name, "N": N, "substeps": sub, "wet_fraction": round(wet, 4), "icy_fraction": round(icy, 4), "runs": []}
models = {}
for key, (skip, rowc) in {"both": (1, 1), "neither": (0, 0), "skip_only": (1, 0), "rowc_only": (0, 1)}.items():
    if key in ("skip_only", "rowc_only") and c["g"].metric_kind != "full":
        continue
    m = cases.csi_model(c, mode="fast")
    m.set_tile_skipping(skip)
    m.set_row_constant(rowc)
    for _ in range(3):
        csi.time_step_momentum(m, c["dt"])
    m.synchronize()
    models[key] = m
for r in range(reps):
    for key, m in models.items():
        m.synchronize()
        t0 = time.perf_counter()
        n = 3
        for _ in range(n):
            csi.time_step_momentum(m, c["dt"])
        m.synchronize()
        dt = (time.perf_counter() - t0) / n
        out["runs"].append({"mode": key, "G_all_cells": round(N * N * sub / dt / 1e9, 2), "G_icy_cells": round(icy * N * N * sub / dt / 1e9, 2),
                            "ms": round(dt * 1e3, 3), "activity": m.tile_activity(), "rowc_rows": m.row_constant_rows(),
                            "path": m.ctx.last_path()["level"]})
        print(out["runs"][-1], flush=True)
print(json.dumps(out))
