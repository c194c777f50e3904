"""Advection-only RK3 steps (BASELINE config 2: periodic grid, WENO7, prescribed velocities) and the tendency launch alone, by size:
python scripts/adv_bench.py [512 1024 2048]  (CSI_ADV_NT=1 | 2 forces the tendency kernel's layout: tracers per thread)
-> us per RK3 step, us per tendency launch (HIP events), cell-stages/s."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import climaseaice_jl_amd as csi
import test_gpu_steps as T

out = {}
for N in [int(a) for a in sys.argv[1:]] or [512, 1024, 2048]:
    c = T.anticyclone_case(N)
    m = csi.SeaIceModel(c["g"], dynamics=None, advection=csi.WENO(order=7), timestepper="SplitRungeKutta3", mode="fast")
    csi.set_(m, h=c["h"], aice=c["a"], u=c["u"], v=c["v"])
    for _ in range(5):
        csi.time_step(m, 120.0)
    m.synchronize(); torch.cuda.synchronize()
    n = 200 if N <= 1024 else 50
    t0 = time.perf_counter()
    for _ in range(n):
        csi.time_step(m, 120.0)
    m.synchronize(); torch.cuda.synchronize()
    step_us = (time.perf_counter() - t0) / n * 1e6
    # the tendency launch alone
    for _ in range(3):
        m.ctx.call("csi_compute_tracer_tendencies", 7)
    m.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        m.ctx.call("csi_compute_tracer_tendencies", 7)
    m.synchronize()
    tend_us = (time.perf_counter() - t0) / n * 1e6
    out[N] = dict(rk3_step_us=round(step_us, 1), tendency_launch_us=round(tend_us, 1), cell_stages_per_s=round(3 * N * N / step_us * 1e6 / 1e9, 2))
    print(N, out[N], flush=True)
print(json.dumps({"CSI_ADV_NT": os.environ.get("CSI_ADV_NT", "auto"), "sizes": out}))
