"""Debug aid: where the producer / consumer waves of k_pair spend their time on a bench_cases configuration (walls variant:
the per-point-metric instantiations).  Library built with -DCSI_PAIR_PROBE; run with CSI_HIP_LIBRARY=.../libcsi_hip_probe.so
python scripts/pair_probe_case.py <case substring> [N]"""
import ctypes as C
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import climaseaice_jl_amd as csi
import cases
# case -> (make_case arguments, translation unit of its pair-kernel instantiation: CSI_PAIR_VARIANT)
KW = {"twelve": (dict(topo=("periodic", "bounded"), curvilinear=0.05), 1),
      "channel": (dict(topo=("periodic", "bounded")), 1),
      "bounded": (dict(topo=("bounded", "bounded")), 1),
      "coupled": (dict(topo=("periodic", "periodic"), field_forcing=True), 3),
      "omip": (dict(topo=("periodic", "bounded"), land=0.3, field_forcing=True, free_drift=True), 6),
      "tripolar": (dict(topo=("periodic", "folded"), curvilinear=0.05, land=0.3, field_forcing=True, free_drift=True), 6)}
name = sys.argv[1] if len(sys.argv) > 1 else "twelve"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
c = cases.make_case(Nx=N, Ny=N, substeps=12, patches=False, noise=0.05, **KW[name][0])
m = cases.csi_model(c, mode="fast")
m.set_fusion(2)
csi.time_step_momentum(m, c["dt"])
m.synchronize()
L = C.CDLL(csi._lib.LIB_PATH)
buf = np.zeros(8192 * 16, dtype=np.uint64)
getattr(L, "csi_debug_probe_v%d" % KW[name][1])(buf.ctypes.data_as(C.c_void_p))
p = buf.reshape(8192, 16)
names = {0: ["wait for rows (vmcnt)", "loads issue + stage A + LDS writes", "barrier (wait for consumer)"],
         1: ["barrier (wait for producer)", "LDS reads (+ delayed stores)", "stage B + stores"]}
for role in (0, 1):
    q = p[role::2]
    q = q[q[:, 6] > 0]
    it = q[:, 6].astype(float)
    print(("PRODUCER" if role == 0 else "CONSUMER"), "waves", len(q), "iterations/wave %.1f" % it.mean())
    tot = 0.0
    for k, n in enumerate(names[role]):
        per = q[:, k] / it
        tot += per.mean()
        print(f"   {n:45s} {per.mean():8.0f} cycles/iteration  (min {per.min():.0f} max {per.max():.0f})")
    for k, n in enumerate(["step: strain rates", "step: stress phase", "step: prefetch issue", "step: velocity phase"]):
        per = q[:, 12 + k] / it
        print(f"      {n:42s} {per.mean():8.0f} cycles/iteration")
    w0, w1 = q[:, 8].astype(float), q[:, 9].astype(float)
    print("   sum %.0f cycles per iteration; wave lifetime mean %.1f us (min %.1f max %.1f); kernel span %.1f us"
          % (tot, (w1 - w0).mean() / 100, (w1 - w0).min() / 100, (w1 - w0).max() / 100, (w1.max() - w0.min()) / 100))
    print("   lifetime percentiles (us):", np.percentile((w1 - w0) / 100, [0, 5, 25, 50, 75, 90, 95, 99, 100]).round(1))
    nstr = -(-(N + 2) // 56)
    life = (w1 - w0) / 100
    strip = np.arange(len(life)) % nstr
    print("   mean lifetime by strip (us):", np.array([life[strip == s_].mean() for s_ in range(nstr)]).round(0))
    print("   end time by strip (us):", np.array([(w1[strip == s_].max() - w0.min()) / 100 for s_ in range(nstr)]).round(0))
