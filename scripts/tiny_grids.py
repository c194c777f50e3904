"""Degenerate grid sizes through every path (run on the GPU box): FAST levels 0 / 1 / 2 and STRICT against the oracle."""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import climaseaice_jl_amd as csi
import cases
import test_gpu_evp as T

fails = 0
for (Nx, Ny, H) in [(8, 8, 4), (9, 7, 3), (5, 5, 4), (4, 9, 4), (3, 3, 2), (2, 2, 1), (1, 6, 2), (6, 1, 2), (64, 8, 4), (8, 64, 4), (57, 9, 4), (113, 10, 5)]:
    for topo in (("periodic", "periodic"), ("bounded", "bounded"), ("periodic", "bounded")):
        if topo[0] == "periodic" and Nx < H or topo[1] == "periodic" and Ny < H:
            continue                                     # a periodic direction needs N >= H (Oceananigans' own rule)
        try:
            c = cases.make_case(Nx=Nx, Ny=Ny, H=H, topo=topo, substeps=6, patches=False, random_uv=0.02)
            p = cases.oracle_problem(c)
            p.time_step_momentum(c["dt"])
            vmax = max(np.abs(p.f["u"]).max(), np.abs(p.f["v"]).max(), 1e-30)
            for mode, fusion in (("strict", 0), ("fast", 0), ("fast", 1), ("fast", 2)):
                m = cases.csi_model(c, mode=mode)
                m.set_fusion(fusion)
                csi.time_step_momentum(m, c["dt"])
                g = T.gpu_fields(m)
                d = max(np.abs(g[k] - p.f[k]).max() for k in ("u", "v"))
                ok = np.all(np.isfinite(g["u"])) and (d == 0.0 if mode == "strict" else d <= 1e-11 * vmax)
                if not ok:
                    fails += 1
                    print("FAIL", Nx, Ny, H, topo, mode, fusion, "level", m.ctx.last_path()["level"], "diff", d, "vmax", vmax)
        except Exception as e:
            fails += 1
            print("ERR", Nx, Ny, H, topo, type(e).__name__, str(e)[:200])
print("done, failures:", fails)
