import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "scripts")):
    sys.path.insert(0, p)
import climaseaice_jl_amd as csi
import cases
c = cases.make_case(Nx=2048, Ny=2048, substeps=120, patches=False, noise=0.05, topo=("periodic", "periodic"))
for on in (1, 0, 1, 0):
    m = cases.csi_model(c, mode="fast")
    m.set_tile_skipping(on)
    acts = []
    for k in range(12):
        csi.time_step_momentum(m, c["dt"]); m.synchronize(); acts.append(m.tile_activity()[2])
    t0 = time.perf_counter()
    for k in range(40):
        csi.time_step_momentum(m, c["dt"])
    m.synchronize()
    print("skipping", on, "used per step", acts, "ms/step %.3f" % ((time.perf_counter() - t0) / 40 * 1e3), flush=True)
