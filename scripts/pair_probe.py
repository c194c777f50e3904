"""Debug aid: per-phase cycle counts of csi::fused::k_pair (library built with -DCSI_PAIR_PROBE)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import bench  # noqa
import climaseaice_jl_amd as csi
g = csi.RectilinearGrid((2048, 2048), x=(0, 2048 * 2000.0), y=(0, 2048 * 2000.0), topology=(csi.Periodic, csi.Periodic), halo=(4, 4))
tg, f = bench.local_case(csi, np, 2048, 2048, 1, 1, 0, force_connected=False, halo=4)
dyn = csi.SeaIceMomentumEquation(tg, coriolis=csi.FPlane(f=1e-4), rheology=csi.ElastoViscoPlasticRheology(),
                                 top_momentum_stress=(0.01, 0.01), bottom_momentum_stress=csi.SemiImplicitStress(),
                                 solver=csi.SplitExplicitSolver(substeps=4), device="cuda:0")
model = csi.SeaIceModel(tg, dynamics=dyn, advection=None, timestepper="ForwardEuler", device="cuda:0", mode="fast")
csi.set_(model, h=f["h"], aice=f["a"], u=f["u"], v=f["v"])
csi.time_step_momentum(model, 120.0)
model.synchronize()
import ctypes as C
L = C.CDLL(csi._lib.LIB_PATH)
buf = np.zeros(4096 * 16, dtype=np.uint64)
rc = L.csi_debug_probe(buf.ctypes.data_as(C.c_void_p))
p = buf.reshape(4096, 16)
p = p[p[:, 6] > 0]
it = p[:, 6].astype(float)
names = ["wait vmcnt", "flush", "prefetch issue", "consts + stage A", "stage B", "diag + shifts"]
print("waves", len(p), "iterations/wave", it.mean())
tot = 0
for k, n in enumerate(names):
    per = (p[:, k] / it)
    tot += per.mean()
    print(f"{n:20s} {per.mean():9.0f} ticks/iteration  (min {per.min():.0f} max {per.max():.0f})")
print("sum", tot, "ticks per iteration; s_memtime ticks: 100 MHz => x10 ns")
life = p[:, :6].sum(axis=1).astype(float)
w0, w1 = p[:, 8].astype(float), p[:, 9].astype(float)
t0 = w0.min()
print("wall clock (100 MHz): kernel span %.1f us; wave lifetime mean %.1f us (min %.1f max %.1f); cycle-counter lifetime mean %.0f ticks => %.2f GHz"
      % ((w1.max() - t0) / 100, (w1 - w0).mean() / 100, (w1 - w0).min() / 100, (w1 - w0).max() / 100, life.mean(), life.mean() / ((w1 - w0).mean() * 10) ))
print("start deciles (us):", (np.percentile(w0 - t0, [0, 10, 25, 50, 75, 90, 100]) / 100).round(1))
print("end   deciles (us):", (np.percentile(w1 - t0, [0, 10, 25, 50, 75, 90, 100]) / 100).round(1))
nstr = 37
lifeus = (w1 - w0) / 100
nw = len(lifeus)
chunk = np.arange(nw) // nstr; strip = np.arange(nw) % nstr
print("mean lifetime by strip:", np.array([lifeus[strip == s_].mean() for s_ in range(nstr)]).round(0))
print("mean lifetime by chunk:", np.array([lifeus[chunk == c_].mean() for c_ in range(chunk.max() + 1)]).round(0))
fl = p[:, 1] / it
print("flush ticks/iter by strip:", np.array([fl[strip == s_].mean() for s_ in range(nstr)]).round(0))
