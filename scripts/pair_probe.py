"""Debug aid: where the producer and consumer waves of csi::fused::k_pair spend their time
(library built with -DCSI_PAIR_PROBE: scripts/build_variant.sh probe "-DCSI_PAIR_PROBE"; run with
CSI_HIP_LIBRARY=.../libcsi_hip_probe.so python scripts/pair_probe.py [NX NY])."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa
import climaseaice_jl_amd as csi

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ny = int(sys.argv[2]) if len(sys.argv) > 2 else nx
FC = (False, True) if (len(sys.argv) > 4 and sys.argv[4] == "peer-y") else False
tg, f = bench.local_case(csi, np, nx, ny, 1, 1, 0, force_connected=FC, halo=4)
dyn = csi.SeaIceMomentumEquation(tg, coriolis=csi.FPlane(f=1e-4), rheology=csi.ElastoViscoPlasticRheology(),
                                 top_momentum_stress=(0.01, 0.01), bottom_momentum_stress=csi.SemiImplicitStress(),
                                 solver=csi.SplitExplicitSolver(substeps=int(sys.argv[3]) if len(sys.argv) > 3 else 12), device="cuda:0")
model = csi.SeaIceModel(tg, dynamics=dyn, advection=None, timestepper="ForwardEuler", device="cuda:0", mode="fast")
csi.set_(model, h=f["h"], aice=f["a"], u=f["u"], v=f["v"])
csi.time_step_momentum(model, 120.0)
model.synchronize()
L = C.CDLL(csi._lib.LIB_PATH)
buf = np.zeros(8192 * 16, dtype=np.uint64)
rc = L.csi_debug_probe(buf.ctypes.data_as(C.c_void_p))
p = buf.reshape(8192, 16)
names = {0: ["wait for rows (vmcnt)", "loads issue + stage A + LDS writes", "barrier (wait for consumer)"],
         1: ["barrier (wait for producer)", "wait vmcnt(0) (inputs + stores)", "stores + prefetch + LDS reads + stage B"]}
for role in (0, 1):
    q = p[role::2]
    q = q[q[:, 6] > 0]
    it = q[:, 6].astype(float)
    print(("PRODUCER" if role == 0 else "CONSUMER"), "waves", len(q), "iterations/wave %.1f" % it.mean())
    tot = 0.0
    for k, n in enumerate(names[role]):
        per = q[:, k] / it
        tot += per.mean()
        print(f"   {n:45s} {per.mean() * 10:8.0f} ns/iteration  (min {per.min() * 10:.0f} max {per.max() * 10:.0f})")
    w0, w1 = q[:, 8].astype(float), q[:, 9].astype(float)
    print("   sum %.0f ns per iteration; wave lifetime mean %.1f us (min %.1f max %.1f); kernel span %.1f us"
          % (tot * 10, (w1 - w0).mean() / 100, (w1 - w0).min() / 100, (w1 - w0).max() / 100, (w1.max() - w0.min()) / 100))
    t0 = w0.min()
    print("   start deciles (us):", (np.percentile(w0 - t0, [0, 10, 50, 90, 100]) / 100).round(1),
          " end deciles (us):", (np.percentile(w1 - t0, [0, 10, 50, 90, 100]) / 100).round(1))
    nstr = -(-(nx + 2) // 56)
    life = (w1 - w0) / 100
    n = len(life)
    chunk, strip = np.arange(n) // nstr, np.arange(n) % nstr
    print("   lifetime percentiles (us):", np.percentile(life, [0, 5, 25, 50, 75, 90, 95, 99, 100]).round(1))
    print("   mean lifetime by strip:", np.array([life[strip == s_].mean() for s_ in range(nstr)]).round(0))
    print("   mean lifetime by chunk:", np.array([life[chunk == c_].mean() for c_ in range(chunk.max() + 1)]).round(0))
    work = q[:, 2 if role else 1] / it
    print("   work ticks/iteration by strip:", np.array([work[strip == s_].mean() for s_ in range(nstr)]).round(0))
    print("   work ticks/iteration by chunk:", np.array([work[chunk == c_].mean() for c_ in range(chunk.max() + 1)]).round(0))

# ---- where the hardware placed the waves (HW_REG_HW_ID: wave_id [3:0], simd_id [5:4], cu_id [11:8], sh_id [12], se_id [15:13]) ----
hw = p[:, 10].astype(np.int64)
valid = p[:, 6] > 0
role = np.arange(8192) % 2
blk = p[:, 11].astype(np.int64)
simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
xcd = blk & 7
rank = (blk >> 3) >> 5
life = (p[:, 9].astype(float) - p[:, 8].astype(float)) / 100
key = (((xcd * 8 + se) * 2 + sh) * 16 + cu)
print("distinct (xcd, se, sh, cu):", len(set(key[valid])))
import collections
per_simd = collections.defaultdict(list)
for idx in np.nonzero(valid)[0]:
    per_simd[(key[idx], simd[idx])].append((int(rank[idx]), int(role[idx]), float(life[idx])))
hist = collections.Counter()
for k, v in per_simd.items():
    hist[tuple(sorted((r, ro) for r, ro, _ in v))] += 1
print("most common SIMD populations ((age rank, role) ...):")
for comp, n in hist.most_common(8):
    print("   ", n, comp)
same = sum(1 for idx in range(0, 8192, 2) if valid[idx] and valid[idx + 1] and key[idx] == key[idx + 1] and simd[idx] == simd[idx + 1])
print("workgroups whose two waves share a SIMD:", same, "of", int(valid.sum() // 2))
for rk in range(8):
    sel = valid & (rank == rk)
    if sel.any():
        print(f"   age rank {rk}: {sel.sum()} waves, mean lifetime {life[sel].mean():.1f} us")
wid = hw & 15
print("HW wave_id histogram:", collections.Counter(wid[valid].tolist()).most_common())
