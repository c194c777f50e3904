"""Thread sweep of the CPU oracle's EVP sub-step loop on this box (for bench.py's cpu_baseline): python scripts/cpu_sweep.py"""
import ctypes, os, sys, time
sys.path[:0] = [".", "tests", "oracle"]
import cases
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(f, open(f).read().strip())
    except Exception as e:
        print(f, "n/a")
gomp = ctypes.CDLL("libgomp.so.1")
n, sub = 1024, 6
c = cases.make_case(Nx=n, Ny=n, substeps=sub, topo=("periodic", "periodic"), patches=True, random_uv=0.02)
for nt in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    if nt > len(os.sched_getaffinity(0)):
        break
    gomp.omp_set_num_threads(nt)
    p = cases.oracle_problem(c, omp=True)
    p.initialize_rheology()
    p.L.ora_fill_halo_u(p.ptr); p.L.ora_fill_halo_v(p.ptr)
    p.subcycle(c["dt"], 1, 1)
    t0 = time.perf_counter(); t0c = time.process_time()
    p.subcycle(c["dt"], 1, sub)
    t, tc = time.perf_counter() - t0, time.process_time() - t0c
    print(f"{nt:4d} threads: {n * n * sub / t / 1e6:8.2f} M cell-updates/s   wall {t:.3f} s  cpu {tc:.3f} s  (cpu/wall {tc / t:.1f})", flush=True)
