#!/usr/bin/env python3
"""bench.py -- EVP sub-cycle throughput on MI355X (the metric of BASELINE.json).

One "step" = one time_step_momentum! call (initialize_rheology! + `substeps` EVP sub-steps +
finalize_rheology!, SeaIceDynamics/split_explicit_momentum_equations.jl:103-195) over the whole grid,
inputs resident in HBM.  value = cell-updates/s = owned cells * substeps * steps / wall time, whole job.

N = 1: the 2048 x 2048 periodic f-plane grid the metric is quoted on.
N > 1 (torch.distributed.run, one rank per GPU): STRONG scaling by default -- the SAME 2048 x 2048 grid split into
y slabs (1x2, 1x4, 1x8: 2048 x 256 per GPU at N = 8, the fastest tile shape measured; `--partition 2x4`: BASELINE config 4's), advanced by the same
kernels with the RCCL halo exchange of u, v, sigma (width 2k every k sub-steps; halo 32 -> k = 16; `--exchange-interval 1`
is the north star's one exchange per sub-step and is timed as well, outside the headline region).  `--scaling weak`
gives every GPU its own 2048 x 2048 tile instead.

Launching.  `python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment is the PARENT: before anything touches
HIP it counts the visible GPUs (fewer than N: exit 2, never a fall-back to one rank), starts N ranks as
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...` (the way
the reference's distributed tests start themselves: test/test_distributed_sea_ice.jl:41-54, `mpiexec -n 4`), relays rank 0's
single JSON line and exits with the children's code.  Under torch.distributed.run (WORLD_SIZE set) it is a rank, and
`--gpus` must equal WORLD_SIZE.  `--print-launch` prints the parent's plan as JSON and exits (host test, no GPU needed).
N > 1 lines carry `rccl_ranks` (ncclCommCount of the library's communicator), the one-GPU rate of the same global grid
measured in the same run (`single_gpu`), `parallel_efficiency` = value / (N x that rate) and
`tiled_equals_untiled_bitwise` (every rank advances the whole grid alone from the same state and compares its tile).
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import math
import os
import socket
import subprocess
import sys
import time

# multi-process GPU work on this pool: the host driver only supports dmabuf IPC (RCCL fails with hipIpcGetMemHandle otherwise);
# exported by the environment already, kept here for launches that build their own
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

ALGO_BYTES = {"stress": 96.0, "ustep": 80.0, "vstep": 80.0}   # SURVEY.md 8(d); 256 B per cell-update
HBM_PEAK_GBS = 8000.0                                          # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_ISSUE_NS = 2.29                                           # measured sustained issue interval of one FP64 vector instruction per SIMD (v_fma_f64 /
                                                               # v_mul_f64, 2-4 waves per SIMD: scripts/microbench, profiles/r01_microbenchmarks.md `valu_rate`;
                                                               # the 78.6 TFLOP/s data-sheet rate would be 1.67 ns)
NORTH_STAR_1GPU_RATE = 0.40 * 8000.0e9 / 256.0                 # 12.5 G cell-updates/s
HBM_ACHIEVABLE_GBS = 6300.0                                    # ... and the measured copy ceiling (same guide)
# Compulsory HBM bytes per cell PER LAUNCH of each kernel: what `roofline.frac` is priced on.
#   k_pair / k_substep: read u, v, P, h, aice, sigma x 3, u^n, v^n + write sigma x 3, u, v = 15 x 8 B (k_pair does two
#   sub-steps on them); three-kernel path: the per-phase figures of SURVEY.md 8(d).
KERNEL_BYTES = {"pair": 120.0, "substep": 120.0, "stress": 96.0, "ustep": 80.0, "vstep": 80.0}
class ClockSampler:
    """Shader clock and board power of THIS GPU while the timed region runs (round 5: at 2048^2 the sub-cycle runs at the board's
    power cap -- ~2.08 GHz at ~1380 W where small tiles get 2.4 GHz -- so a launch time only means something beside the clock it
    was measured at; profiles/r05_power_clock.md).  A thread reads the hwmon files of the card whose PCI address is the device's
    (the boxes show every card of their host); nothing here touches the GPU.  Unreadable sysfs -> every field None."""

    def __init__(self, device_index):
        import glob
        import threading
        self.freq = self.power = None
        self.samples = []
        self._stop = threading.Event()
        self._thread = None
        try:
            import torch
            pr = torch.cuda.get_device_properties(device_index)
            bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            for d in glob.glob("/sys/class/drm/card*/device"):
                if os.path.basename(os.path.realpath(d)).lower() == bdf:
                    f = glob.glob(os.path.join(d, "hwmon", "hwmon*", "freq1_input"))
                    w = glob.glob(os.path.join(d, "hwmon", "hwmon*", "power1_average")) or glob.glob(os.path.join(d, "hwmon", "hwmon*", "power1_input"))
                    self.freq = f[0] if f else None
                    self.power = w[0] if w else None
        except Exception:
            pass

    def _read(self, path):
        try:
            with open(path) as fh:
                return int(fh.read().strip())
        except Exception:
            return None

    def _run(self):
        while not self._stop.is_set():
            self.samples.append((self._read(self.freq) if self.freq else None, self._read(self.power) if self.power else None))
            self._stop.wait(0.004)

    def __enter__(self):
        import threading
        if self.freq or self.power:
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        if self._thread:
            self._thread.join()

    def summary(self):
        fr = sorted(x[0] for x in self.samples if x[0]); pw = sorted(x[1] for x in self.samples if x[1])
        return {"sclk_mhz": fr[len(fr) // 2] / 1e6 if fr else None, "sclk_mhz_min": fr[0] / 1e6 if fr else None,
                "sclk_mhz_max": fr[-1] / 1e6 if fr else None, "power_w": pw[-1] / 1e6 if pw else None,      # (the largest sample: hwmon's figure is a moving average that lags a 0.15 s region)
                "sclk_mhz_last": (lambda v: v[-1] / 1e6 if v else None)([x[0] for x in self.samples if x[0]]),
                "samples": len(self.samples),
                "note": "the firmware's figures are moving averages: over a region of a few tenths of a second they lag (idle time before it pulls them down); "
                        "scripts/clock_watch.sh over a 10 s run is the reference (profiles/r05_power_clock.md)",
                "source": "sysfs hwmon freq1_input / power1_average of this device, 4 ms period, over a repetition of the timed loop right behind the timed region" if self.samples else None}


# Default decomposition of the metric's grid on N GPUs: slabs in y (Rx = 1).  Measured per tile shape on one MI355X, the tile connected
# to itself over the peer transport in the directions its partition connects (scripts/tile_shapes.py, profiles/r04_tile_1024x512.md; G
# cell-updates/s): N = 8: 1x8 (2048 x 256) 39.9, 2x4 (1024 x 512) 36.8, 8x1 35.9, 4x2 35.5; N = 4: 1x4 54.4, 2x2 51.4; N = 2: 1x2 65.9,
# 2x1 63.5.  A slab keeps x periodic inside the tile (no x images into another GPU's memory: those are 8-byte write-through stores),
# has two neighbours instead of eight and loses 1 % of its lanes to the strip width instead of 4 %.  `--partition 2x4` selects
# BASELINE config 4's decomposition (also what tests/test_gpu_fullsize.py runs at full size, in process).
PARTITION = {1: (1, 1), 2: (1, 2), 4: (1, 4), 8: (1, 8)}
KERNEL_NAMES = {"substep": "csi::fused::k_substep (stress + u + v in one launch)",
                "pair": "csi::fused::k_pair (two sub-steps: 2 x [stress + u + v] in one launch)",
                "stress": "csi::fast::k_stress", "ustep": "csi::fast::k_ustep", "vstep": "csi::fast::k_vstep"}


def visible_gpus():
    """GPUs this process would see, WITHOUT initialising HIP (a parent that has touched the GPU must not start ranks):
    the KFD topology's GPU nodes, narrowed by the *_VISIBLE_DEVICES lists; torch.cuda.device_count() (which does not
    initialise HIP on this image) when the topology is not readable."""
    n = None
    try:
        base = "/sys/class/kfd/kfd/topology/nodes"
        n = 0
        for d in os.listdir(base):
            for line in open(os.path.join(base, d, "properties")):
                if line.startswith("simd_count") and int(line.split()[1]) > 0:
                    n += 1
    except Exception:
        n = None
    if n is None:
        try:
            import torch
            return int(torch.cuda.device_count())
        except Exception:
            return 0
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_plan(n, argv):
    """The command the parent runs for N ranks (one process per GPU, rendezvous on 127.0.0.1)."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def run_parent(args, argv):
    """--gpus N > 1 without WORLD_SIZE: start the ranks, relay rank 0's JSON line, exit with their code."""
    have = visible_gpus()
    cmd = launch_plan(args.gpus, [a for a in argv if a != "--print-launch"])
    if args.print_launch:
        print(json.dumps({"launch": cmd, "ranks": args.gpus, "visible_gpus": have, "would_run": have >= args.gpus}), flush=True)
        return 0
    if have < args.gpus and not args.self_test_launch and not (args.rehearse_on_one_gpu and have >= 1):
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but {have} GPU(s) visible; refusing to fall back to fewer ranks\n")
        return 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CSI_BENCH_LAUNCHED_BY_PARENT="1")
    if args.rehearse_on_one_gpu:
        env["CSI_BENCH_HOST_GROUP"] = f"/csi-bench-{os.getpid()}"        # the shared-memory segment every rank joins
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)          # stderr passes through
    line = None
    for ln in p.stdout.splitlines():
        t = ln.strip()
        if t.startswith("{") and '"metric"' in t:
            line = t
        elif t:
            sys.stderr.write(ln + "\n")
    if p.returncode != 0 or line is None:
        sys.stderr.write(f"bench.py: the {args.gpus}-rank run failed (exit code {p.returncode}, JSON line {'found' if line else 'missing'})\n")
        return p.returncode or 1
    try:
        n = json.loads(line).get("n_gpus")
    except Exception:
        n = None
    if n != args.gpus:
        sys.stderr.write(f"bench.py: asked for {args.gpus} GPUs, the ranks report n_gpus = {n}\n")
        return 1
    print(line, flush=True)
    return 0


def usable_cores():
    """Host cores this process may actually use: the scheduler affinity capped by the cgroup CPU quota (the GPU boxes
    show 256 hardware threads but grant 16 CPUs: cpu.max = 1600000 100000; more OpenMP threads than that only spin)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(math.ceil(int(quota) / int(period)))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = max(1, min(n, int(math.ceil(q / per))))
        except Exception:
            pass
    return n


def cpu_baseline(seconds_budget=12.0):
    """The oracle (strict-order C restatement of the reference's kernel split, oracle/csi_oracle.c, OpenMP over rows)
    timed on this box's host cores: the EVP sub-step loop alone (ora_subcycle: viscosities, stresses, u / v steps, halo
    fills -- no initialize_rheology!, no Python in the loop), one thread and all threads this process may use.  A
    reported baseline on a bounded sample of the same workload; never the target, and no speed-up is derived from it."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cases
    threads = usable_cores()
    try:
        gomp = ctypes.CDLL("libgomp.so.1")
    except OSError:
        gomp = None

    def run(n, sub, nt, build="omp"):
        if gomp is not None:
            gomp.omp_set_num_threads(nt)
        c = cases.make_case(Nx=n, Ny=n, substeps=sub, topo=("periodic", "periodic"), patches=True, random_uv=0.02)
        p = cases.oracle_problem(c, omp=True if build == "omp" else build)
        p.initialize_rheology()
        p.L.ora_fill_halo_u(p.ptr); p.L.ora_fill_halo_v(p.ptr)
        p.subcycle(c["dt"], 1, 2)                      # warm-up: thread team, page faults
        t0 = time.perf_counter()
        p.subcycle(c["dt"], 1, sub)
        return n * n * sub / (time.perf_counter() - t0)

    # one thread: ~1.3 M cell-updates/s on a 2 GHz core -> 512^2 x 12 sub-steps is ~2.5 s
    v1 = run(512, 12, 1)
    # all threads: size the sample from the one-thread rate, assuming it may scale perfectly
    budget_updates = v1 * threads * seconds_budget
    n = 2048
    sub = int(max(2, min(120, budget_updates / (n * n))))
    vall = run(n, sub, threads) if (gomp is not None and threads > 1) else v1
    # the same source as an optimising compiler sees it (-O3, AVX2 + FMA, contraction allowed; oracle/Makefile): what the
    # reference's kernel split costs on these cores without the strict-IEEE handicap of the checker
    try:
        vtuned = run(n, min(120, 3 * sub), threads, build="tuned")
    except Exception:
        vtuned = None
    return {"value": vall, "unit": "cell-updates/s", "cores": threads if gomp is not None else 1, "kind": "port",
            "one_thread_value": v1, "thread_scaling": vall / v1, "tuned_build_value": vtuned,
            "tuned_build_note": "same C source, gcc -O3 -march=x86-64-v3 -ffp-contract=fast, same threads: not the parity checker",
            "sample": f"EVP sub-step loop only (ora_subcycle): {sub} sub-steps of the 2048x2048 periodic f-plane workload on "
                      f"{threads} OpenMP threads (= scheduler affinity capped by the cgroup CPU quota); one-thread figure from 12 sub-steps of a 512x512 "
                      f"grid of the same workload; oracle/csi_oracle.c: the reference's four-kernel split in strict IEEE "
                      f"order (gcc -O2 -ffp-contract=off), strain rates recomputed per stencil point as the reference does -- "
                      f"a lower bound on what a tuned CPU code would reach"}


def tile_fields(np, nx, ny, Rx, Ry, rx, ry):
    """Seeded synthetic inputs of tile (rx, ry), functions of the GLOBAL coordinates plus noise seeded by the TILE (so that any
    rank can rebuild any tile, and the untiled check below assembles the same global state):
    h0 sinusoid + 5 % noise, aice patches (open water / marginal ice), u = 0.1 + 2 % noise."""
    Nx, Ny = nx * Rx, ny * Ry
    rng = np.random.default_rng(1000 + ry * Rx + rx)
    xc = ((np.arange(nx) + rx * nx) + 0.5) / Nx
    yc = ((np.arange(ny) + ry * ny) + 0.5) / Ny
    X, Y = xc[None, :], yc[:, None]
    h = (0.3 + 0.005 * (np.sin(2 * np.pi * 3 * X) + np.sin(2 * np.pi * 2 * Y))) * (1.0 + 0.05 * (rng.random((ny, nx)) - 0.5))
    a = np.clip(0.6 + 0.6 * np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y) + 0.2 * rng.random((ny, nx)), 0.0, 1.0)
    # open water and marginal ice patches at fixed GLOBAL positions (every branch of the velocity kernels is exercised)
    gi0, gi1, gj0, gj1 = Nx // 8, Nx // 4, Ny // 6, Ny // 3
    I = (np.arange(nx) + rx * nx)[None, :] + 0 * np.arange(ny)[:, None]
    J = (np.arange(ny) + ry * ny)[:, None] + 0 * np.arange(nx)[None, :]
    water = (I >= gi0) & (I < gi1) & (J >= gj0) & (J < gj1)
    thin = (I >= gi0) & (I < gi1) & (J >= gj1) & (J < gj1 + 2)
    a[water] = 0.0; h[water] = 0.0
    a[thin] = 5e-4; h[thin] = 1e-3
    u = 0.1 + 0.02 * rng.standard_normal((ny, nx))
    v = 0.02 * rng.standard_normal((ny, nx))
    return dict(h=h, a=a, u=u, v=v)


def global_fields(np, nx, ny, Rx, Ry):
    """The whole grid's inputs, assembled from the tiles' (what a one-GPU run of the same job starts from)."""
    out = {k: np.empty((ny * Ry, nx * Rx)) for k in ("h", "a", "u", "v")}
    for ry in range(Ry):
        for rx in range(Rx):
            t = tile_fields(np, nx, ny, Rx, Ry, rx, ry)
            for k in out:
                out[k][ry * ny:(ry + 1) * ny, rx * nx:(rx + 1) * nx] = t[k]
    return out


def local_case(csi, np, nx, ny, Rx, Ry, rank, force_connected=False, halo=4):
    """This rank's tile grid and inputs."""
    rx, ry = rank % Rx, rank // Rx
    Nx, Ny = nx * Rx, ny * Ry
    g = csi.RectilinearGrid((Nx, Ny), x=(0.0, Nx * 2000.0), y=(0.0, Ny * 2000.0), topology=(csi.Periodic, csi.Periodic), halo=(halo, halo))
    tg = csi.TileGrid(g, Rx, Ry, rx, ry, force_connected=force_connected) if (Rx * Ry > 1 or force_connected) else g
    return tg, tile_fields(np, nx, ny, Rx, Ry, rx, ry)


def counters():
    """PMC evidence of the dominant kernel from the newest committed rocprofv3 counter passes (profiles/counters_latest.json,
    written by scripts/pmc.sh + scripts/summarize_db.py): HBM bytes per launch, VALU instruction counts.  Static data of
    an EARLIER run of the same kernel build -- labelled as such in the output, never mixed into `achieved`."""
    f = os.path.join(ROOT, "profiles", "counters_latest.json")
    if not os.path.exists(f):
        return None
    try:
        return json.load(open(f))
    except Exception:
        return None


def advection_record(csi, np, torch, device):
    """Advection-only RK3 steps of BASELINE config 2 (512^2) and of the metric's grid size (2048^2): us per step, cell-stages/s, the
    tendency launch alone, and -- from the committed counter run of these kernels (profiles/counters_advection.json, scripts/
    adv_profile.sh) -- the share of the SIMDs' FP64 issue time the 2048^2 tendency launch uses (its bound: VALU busy 0.68, HBM 0.1)."""
    out = {"config": "advection only: periodic grid, dx = 1 km, WENO(order = 7) of h and aice, prescribed cyclonic eddy, SplitRungeKutta3, dt = 120 s"}
    for N in (512, 2048):
        L = N * 1000.0
        g = csi.RectilinearGrid((N, N), x=(0.0, L), y=(0.0, L), topology=(csi.Periodic, csi.Periodic), halo=(4, 4))
        xc = (np.arange(N) + 0.5) * L / N
        xf = np.arange(N) * L / N
        V = 0.5
        u = np.broadcast_to(V * np.sin(2 * np.pi * xc / L)[:, None], (N, N)) * np.cos(2 * np.pi * xf / L)[None, :]
        v = -np.broadcast_to(V * np.sin(2 * np.pi * xc / L)[None, :], (N, N)) * np.cos(2 * np.pi * xf / L)[:, None]
        X, Y = np.meshgrid(xc, xc)
        h = 0.3 + 0.005 * (np.sin(60 * X / 1000e3) + np.sin(30 * Y / 1000e3))
        a = np.clip(1 - 0.1 * np.random.default_rng(2).random((N, N)), 0, 1)
        m = csi.SeaIceModel(g, dynamics=None, advection=csi.WENO(order=7), timestepper="SplitRungeKutta3", mode="fast", device=device)
        csi.set_(m, h=h, aice=a, u=u, v=v)
        for _ in range(5):
            csi.time_step(m, 120.0)
        m.synchronize(); torch.cuda.synchronize()
        n = 200 if N == 512 else 40
        t0 = time.perf_counter()
        for _ in range(n):
            csi.time_step(m, 120.0)
        m.synchronize(); torch.cuda.synchronize()
        step_us = (time.perf_counter() - t0) / n * 1e6
        for _ in range(3):
            m.ctx.call("csi_compute_tracer_tendencies", 7)
        m.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            m.ctx.call("csi_compute_tracer_tendencies", 7)
        m.synchronize()
        tend_us = (time.perf_counter() - t0) / n * 1e6
        rec = {"rk3_step_us": step_us, "cell_stages_per_s": 3.0 * N * N / (step_us * 1e-6), "tendency_launch_us": tend_us}
        try:
            ctr = json.load(open(os.path.join(ROOT, "profiles", "counters_advection.json")))[str(N)]
            rec["valu_insts_per_tendency_launch"] = ctr["valu_insts_per_launch"]
            rec["lane_insts_per_cell_and_tracer"] = ctr["valu_insts_per_launch"] * 64.0 / (2.0 * N * N)
            rec["fp64_issue_frac"] = ctr["issue_ns_per_launch"] * 1e-9 / 1024.0 / (tend_us * 1e-6)
            rec["hbm_bytes_per_tendency_launch"] = ctr.get("hbm_bytes_per_launch")
            rec["hbm_frac"] = ctr["hbm_bytes_per_launch"] / (tend_us * 1e-6) / 1e9 / HBM_PEAK_GBS if ctr.get("hbm_bytes_per_launch") else None
            rec["counters_source"] = ctr.get("source")
        except Exception:
            pass
        out["config2_512" if N == 512 else "grid_2048"] = rec
        m = None
    return out


def structure_record(csi, np, torch, device, substeps, quick=False):
    """The reference's flagship workload beside the headline (round 6): grids with land and ice-free ocean, where the two exact
    structure cuts work -- tile activity (tiles with no ice mass are left out of the inner launches) and row-constant rows of a
    tripolar grid's latitude-longitude part (read from per-row vectors).  Every rate is given for ALL cells of the grid (the metric's
    counting), for the WET cells and for the cells that hold ICE, with the cuts on and off, so that nothing is oversold:
      config5_masked : BASELINE config 5's grid -- 4096^2, 38.6 % land in seeded discs (tests/cases.py), ice on every wet cell;
      tripolar       : csi.TripolarGrid 2048^2 (78 S .. 90 N, latitude-longitude rows below a conformal bipolar cap at 55 N, north fold),
                       the reference's analytic land + 30 % land discs, wind-stress and ocean-velocity arrays, StressBalanceFreeDrift,
                       f = 2 Omega sin(latitude) per point, ice poleward of 58 degrees -- test/distributed_tests_utils.jl:186-224 at scale;
      tripolar_like  : round 5's stand-in (rectilinear grid distorted everywhere, 30 % land discs, ice everywhere): no row-constant rows."""
    import cases
    todo = {
        "config5_masked": (4096 if not quick else 1024, dict(topo=("periodic", "bounded"), land=0.386)),
        "tripolar": (2048 if not quick else 512, dict(grid="tripolar", tripolar=dict(southernmost_latitude=-78.0), land=0.3, field_forcing=True, free_drift=True,
                                                       coriolis_points=True, ice_edge=58.0)),
        "tripolar_like": (2048 if not quick else 512, dict(topo=("periodic", "folded"), curvilinear=0.05, land=0.3, field_forcing=True, free_drift=True)),
    }
    out = {"substeps": substeps, "note": "cell-updates/s = cells x sub-steps / wall time of time_step_momentum!; all = every cell of the grid (the metric's counting), "
                                         "wet = cells that are not land, icy = cells with h > 0 and aice > 0; cuts = tile activity + row-constant rows "
                                         "(bit-identical results: tests/test_gpu_activity.py)"}
    for name, (N, kw) in todo.items():
        c = cases.make_case(Nx=N, Ny=N, substeps=substeps, patches=False, noise=0.05, **kw)
        wet = 1.0 if c["mask"] is None else float(c["mask"].mean())
        icy = float(((c["h"] > 0) & (c["a"] > 0)).mean())
        rec = {"grid": [N, N], "wet_fraction": wet, "icy_fraction": icy}
        for key, on in (("cuts_on", True), ("cuts_off", False)):
            m = cases.csi_model(c, mode="fast", device=device)
            m.set_tile_skipping(on)
            m.set_row_constant(on)
            for _ in range(3):                            # (the launch geometry follows the live fraction from the second sub-cycle on)
                csi.time_step_momentum(m, c["dt"])
            m.synchronize(); torch.cuda.synchronize()
            n = 3
            t0 = time.perf_counter()
            for _ in range(n):
                csi.time_step_momentum(m, c["dt"])
            m.synchronize(); torch.cuda.synchronize()
            e = (time.perf_counter() - t0) / n
            tiles, live, used = m.tile_activity()
            rec[key] = {"ms_per_step": 1e3 * e, "all_cells_per_s": N * N * substeps / e, "wet_cells_per_s": wet * N * N * substeps / e,
                        "icy_cells_per_s": icy * N * N * substeps / e, "tiles": tiles, "live_tiles": live, "live_launches_used": bool(used),
                        "row_constant_rows": m.row_constant_rows() if c["g"].metric_kind == "full" else None, "level": m.ctx.last_path()["level"]}
            m = None
        rec["speedup"] = rec["cuts_off"]["ms_per_step"] / rec["cuts_on"]["ms_per_step"]
        out[name] = rec
    return out


def config3_record(csi, np, torch, make_model, substeps, dt):
    """BASELINE config 3 as written -- 1024^2 f-plane, EVP, 120 sub-cycles, one GPU -- beside the headline (which is the same workload on the
    2048^2 grid the metric is quoted on): shorter tiles carry the same seven ring rows, so its rate is lower."""
    grid, fld = local_case(csi, np, 1024, 1024, 1, 1, 0, halo=4)
    m = make_model(grid)
    csi.set_(m, h=fld["h"], aice=fld["a"], u=fld["u"], v=fld["v"])
    for _ in range(3):
        csi.time_step_momentum(m, dt)
    m.synchronize(); torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        csi.time_step_momentum(m, dt)
    m.synchronize(); torch.cuda.synchronize()
    e = (time.perf_counter() - t0) / n
    return {"workload": f"evp_subcycle_fplane_periodic_1024x1024_{substeps}substeps", "value": 1024 * 1024 * substeps / e, "unit": "cell-updates/s",
            "ms_per_step": 1e3 * e, "steps": n, "level": m.ctx.last_path()["level"]}


def isa_mix():
    """Instruction mix of the dominant kernel's row loops from its ISA listing (scripts/isa_mix.py -> profiles/isa_mix_k_pair.json)."""
    f = os.path.join(ROOT, "profiles", "isa_mix_k_pair.json")
    try:
        return json.load(open(f))
    except Exception:
        return None


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=2048, help="global edge (strong) or tile edge (weak)")
    ap.add_argument("--tile", type=str, default="", help="debug, one GPU: NXxNY tile instead of --size (e.g. 1024x512)")
    ap.add_argument("--substeps", type=int, default=120)
    ap.add_argument("--mode", default="fast", choices=["fast", "strict"])
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-full-step", action="store_true")
    ap.add_argument("--exchange-interval", type=int, default=0, help="k: exchange width 2k every k sub-steps (0 = auto)")
    ap.add_argument("--halo", type=int, default=0, help="halo width (default: 4 on one GPU, 32 on tiles so that k = 16)")
    ap.add_argument("--transport", default="peer", choices=["peer", "rccl"],
                    help="tiles: peer-direct halo writes over xGMI with flags (default; halo 4) or the k-batched RCCL exchange (halo 32)")
    ap.add_argument("--no-fusion", action="store_true", help="three-kernel FAST path instead of the fused sub-step kernels")
    ap.add_argument("--fusion-level", type=int, default=2, choices=[0, 1, 2],
                    help="0: three kernels per sub-step, 1: one fused launch per sub-step, 2: two sub-steps per launch (default)")
    ap.add_argument("--force-connected", action="store_true",
                    help="debug: on one GPU, route the periodic halos through the RCCL exchange (to self)")
    ap.add_argument("--print-launch", action="store_true", help="--gpus N > 1: print the launch command of the N ranks as JSON and exit")
    ap.add_argument("--self-test-launch", action="store_true",
                    help="host test of the launcher, no GPU: the ranks rendezvous over gloo, rank 0 prints a stub line (self_test: true)")
    ap.add_argument("--no-compare", action="store_true", help="tiles: do not time the other halo transports (RCCL k = 16, k = 1) after the headline")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="REHEARSAL of the N > 1 path on a one-GPU box: the N ranks are processes that all use GPU 0, joined by the library's "
                         "host-channel group (shared memory + HIP IPC; RCCL refuses two ranks on one device) and gloo; every step of the N > 1 "
                         "path runs -- launcher, peer set-up over IPC, tier ladder, tiled == untiled check, timing, the JSON line -- but the "
                         "numbers are those of N processes SHARING one GPU: the line says `rehearsal_on_one_gpu` and is not a scaling result")
    ap.add_argument("--peer-tier", type=int, default=-1, choices=[-1, 0, 1, 2],
                    help="tiles on the peer transport: the memory-ordering tier the ladder STARTS at (-1: the library's automatic choice -- tier 1 "
                         "across processes / devices; 0 is the explicit opt-in to the fence-free protocol, timed as `peer_tier0` otherwise)")
    ap.add_argument("--no-second-partition", action="store_true", help="N > 1: do not time the other decomposition (2x4-style vs y slabs) after the headline")
    ap.add_argument("--partition", type=str, default="", help="RxxRy tiles instead of the default y slabs (e.g. 2x4: BASELINE config 4's decomposition); Rx * Ry = --gpus")
    ap.add_argument("--no-unfused", action="store_true", help="one GPU: do not time the unfused three-kernel path (roofline.unfused) after the headline")
    ap.add_argument("--no-structure", action="store_true", help="one GPU: do not time the masked / tripolar configurations (`structure`) and config 3's 1024^2 after the headline")
    ap.add_argument("--no-verify", action="store_true", help="N > 1: skip the tiled == untiled bitwise check and the one-GPU rate of the same grid")
    return ap.parse_args(argv)


def self_test_launch():
    """Host test of the launcher (no GPU): the ranks rendezvous over gloo, rank 0 prints a stub line."""
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo")
    t = torch.ones(1)
    dist.all_reduce(t)
    if dist.get_rank() == 0:
        print(json.dumps({"metric": "self-test of the bench.py launcher", "self_test": True, "n_gpus": int(t.item()),
                          "launched_by_parent": os.environ.get("CSI_BENCH_LAUNCHED_BY_PARENT") == "1"}), flush=True)
    dist.destroy_process_group()


class Rank:
    """One rank of the benchmark: its process-group plumbing, its tile of the metric's grid, the model being timed (`model`: the
    helpers below act on whichever model is current) and the one-GPU copy of the whole grid it is checked against (`whole`)."""
    SIG = ("s11", "s22", "s12")

    def __init__(self, args):
        import numpy as np
        import torch
        import climaseaice_jl_amd as csi
        self.args, self.np, self.torch, self.csi = args, np, torch, csi
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.rehearsal = bool(args.rehearse_on_one_gpu and self.world > 1)
        if self.rehearsal:
            local_rank = 0                                     # every rank on GPU 0
        if torch.cuda.device_count() <= local_rank:
            raise SystemExit(f"bench.py: rank {self.rank} needs GPU {local_rank} but {torch.cuda.device_count()} are visible")
        torch.cuda.set_device(local_rank)
        self.local_rank = local_rank
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            if self.rehearsal:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
            self.dist = dist
        self.coll_device = "cpu" if self.rehearsal else None             # gloo reduces host tensors
        self.Rx, self.Ry = PARTITION[self.world]
        if args.partition:
            self.Rx, self.Ry = (int(t) for t in args.partition.lower().split("x"))
            if self.Rx * self.Ry != self.world:
                raise SystemExit(f"--partition {args.partition}: {self.Rx} x {self.Ry} tiles but {self.world} rank(s)")
        if args.tile:
            self.nx_l, self.ny_l = (int(s) for s in args.tile.lower().split("x"))
        elif args.scaling == "weak":
            self.nx_l = self.ny_l = args.size
        else:
            if args.size % self.Rx or args.size % self.Ry:
                raise SystemExit("--size must be divisible by the partition")
            self.nx_l, self.ny_l = args.size // self.Rx, args.size // self.Ry
        self.tiled = self.world > 1 or args.force_connected
        self.user_halo = args.halo
        self.device = f"cuda:{local_rank}"
        self.dt = 120.0
        self.builds = 0
        self.region = {}
        self.model = self.whole = self.gf = None
        self.gN = (self.nx_l * self.Rx, self.ny_l * self.Ry)
        self.owned = self.nx_l * self.ny_l * self.world

    # ---- models ------------------------------------------------------------------------------------------------------------------
    def make_model(self, grid):
        csi, args = self.csi, self.args
        dyn = csi.SeaIceMomentumEquation(grid, coriolis=csi.FPlane(f=1e-4), rheology=csi.ElastoViscoPlasticRheology(),
                                         top_momentum_stress=(0.01, 0.01), bottom_momentum_stress=csi.SemiImplicitStress(),
                                         solver=csi.SplitExplicitSolver(substeps=args.substeps), device=self.device)
        return csi.SeaIceModel(grid, dynamics=dyn, advection=csi.WENO(order=7), timestepper="SplitRungeKutta3", device=self.device, mode=args.mode)

    def build(self, transport, part=None):
        """This rank's tile model.  Tiles: the peer transport needs the halo 4 of an untiled run; the RCCL exchange amortises its
        pack / send / unpack over k = 16 sub-steps with halo 32.  part = (Rx, Ry, nx, ny): another decomposition of the same grid."""
        args, csi = self.args, self.csi
        halo = self.user_halo or (4 if (not self.tiled or transport == "peer") else 32)
        pRx, pRy, pnx, pny = part or (self.Rx, self.Ry, self.nx_l, self.ny_l)
        grid, fld = local_case(csi, self.np, pnx, pny, pRx, pRy, self.rank, force_connected=args.force_connected, halo=halo)
        if self.rehearsal:
            self.builds += 1
            # (a fresh segment per model -- every rank builds its models in the same order; without the parent the name comes from
            #  the rendezvous port, which is the same on every rank of a job)
            base = os.environ.get("CSI_BENCH_HOST_GROUP") or f"/csi-bench-{os.environ.get('MASTER_PORT', '0')}-{os.getppid()}"      # (the ranks of one launcher share their parent: no name of an earlier job)
            grid.host_group = f"{base}-{self.builds}"
        m = self.make_model(grid)
        m.set_exchange_interval(args.exchange_interval)
        m.set_halo_transport(transport)
        if args.peer_tier >= 0:
            m.set_peer_tier(args.peer_tier)
        m.set_fusion(0 if args.no_fusion else args.fusion_level)
        csi.set_(m, h=fld["h"], aice=fld["a"], u=fld["u"], v=fld["v"])
        return grid, fld, m, halo

    # ---- collectives and timing (they act on self.model) ------------------------------------------------------------------------
    def barrier(self):
        self.model.synchronize()
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, x):
        if self.dist is None:
            return x
        t = self.torch.tensor([x], device=self.coll_device or self.device, dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def all_ranks(self, flag):
        """logical AND over the ranks"""
        if self.dist is None:
            return bool(flag)
        t = self.torch.tensor([1.0 if flag else 0.0], device=self.coll_device or self.device, dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)

    def timed(self, nsteps, stats=False):
        """K steps between two barriers, max over the ranks.  stats: HIP events around every sub-step loop INSIDE this region
        (csi_subcycle_stats_*): the dominant kernel's average launch time then comes from the timed launches themselves."""
        self.barrier()
        if stats:
            self.model.ctx.subcycle_stats_begin()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            self.csi.time_step_momentum(self.model, self.dt)
        self.barrier()
        e = self.max_over_ranks(time.perf_counter() - t0)
        if stats:
            tot, ncyc, nl = self.model.ctx.subcycle_stats_end()
            self.region.update(total_ms=tot, cycles=ncyc, launches=nl)
        return e

    def rate(self, e):
        return self.owned * self.args.substeps * self.args.steps / e

    # ---- N > 1: the same job on ONE GPU, in the same run: its answer (every rank advances the whole grid alone from the same
    # state; its own tile must come out bit for bit) and, after the timed region, its rate (parallel_efficiency) -----------------
    def restart(self, m, fld):
        self.csi.set_(m, h=fld["h"], aice=fld["a"], u=fld["u"], v=fld["v"])
        for name in self.SIG:
            getattr(m.dynamics.auxiliaries.fields, name).data.zero_()
        # torch zeroes on ITS stream, the library packs / launches on its own: without this a rank's halo message could carry the sigma of
        # the previous run (seen once, round 5: N processes sharing one GPU in the rehearsal reorder the two streams more readily)
        self.torch.cuda.synchronize()

    def tile_matches_whole(self, m=None, grid=None, fld=None, gfld=None):
        """One sub-cycle from the same state on the tiles and on the whole grid: owned cells bitwise equal (all ranks agree).
        (gfld: the global state assembled from THAT decomposition's tiles -- the noise of the inputs is seeded per tile.)"""
        csi, np, whole = self.csi, self.np, self.whole
        m = self.model if m is None else m
        grid = self.tg if grid is None else grid
        fld = self.f if fld is None else fld
        self.restart(whole, self.gf if gfld is None else gfld)
        self.restart(m, fld)
        csi.time_step_momentum(whole, self.dt)
        csi.time_step_momentum(m, self.dt)
        m.synchronize(); whole.synchronize()
        same = True
        for name, tf, wf in [("u", m.velocities.u, whole.velocities.u), ("v", m.velocities.v, whole.velocities.v)] + \
                            [(n, getattr(m.dynamics.auxiliaries.fields, n), getattr(whole.dynamics.auxiliaries.fields, n)) for n in self.SIG]:
            mine = tf.interior_numpy()[:grid.Ny, :grid.Nx]
            ref = wf.interior_numpy()[grid.j_off:grid.j_off + grid.Ny, grid.i_off:grid.i_off + grid.Nx]
            same = same and bool(np.array_equal(mine, ref))
        return same


def warm_up_and_verify(R):
    """Warm-up, then (N > 1) the bitwise check -- before anything is timed -- with the run-time ladder of the peer protocol
    (include/csi.h, csi_set_peer_tier): it starts at the tier the library chooses by itself (1 as soon as a neighbour lives in another
    process or on another device; --peer-tier overrides) and goes up to 2, then to the RCCL exchange.  A BITWISE mismatch raises the
    tier; a library error (a wait that timed out: sticky, the flags cannot recover) goes straight to the RCCL exchange on freshly
    built models.  Returns (verify, bitwise, ladder, transport_note)."""
    args, csi = R.args, R.csi
    verify = R.world > 1 and not args.no_verify
    if verify:
        gg = csi.RectilinearGrid(R.gN, x=(0.0, R.gN[0] * 2000.0), y=(0.0, R.gN[1] * 2000.0), topology=(csi.Periodic, csi.Periodic), halo=(4, 4))
        R.whole = R.make_model(gg)
        R.whole.set_fusion(0 if args.no_fusion else args.fusion_level)
        R.gf = global_fields(R.np, R.nx_l, R.ny_l, R.Rx, R.Ry)
    transport_note = None
    bitwise = None
    peer_tier = None
    ladder = []
    peer_expected = R.tiled and args.transport == "peer" and args.exchange_interval == 0 and args.substeps % 2 == 0 and args.mode == "fast" \
        and not args.no_fusion and args.fusion_level >= 2 and R.nx_l >= 128
    while True:
        # A peer transport that cannot be set up (the library then runs RCCL by itself), times out or gives another answer than one
        # GPU is replaced by the RCCL exchange on ALL ranks.
        problem = None
        lib_error = False
        if peer_tier is None and R.tiled:
            peer_tier = R.model.ctx.peer_tier()
        try:
            for _ in range(args.warmup):
                csi.time_step_momentum(R.model, R.dt)
            R.model.synchronize()
            if peer_expected and R.model.ctx.halo_transport() != "peer":
                problem = "the peer transport could not be set up on this node"
            elif verify:
                bitwise = R.tile_matches_whole()
                if not bitwise:
                    problem = "the tiled run did not reproduce the one-GPU run bit for bit"
        except csi.CsiError as e:
            problem = f"library error: {e}"
            lib_error = True
        on_peer = R.tiled and R.model.ctx.halo_transport() == "peer"
        if on_peer:
            ladder.append({"tier": peer_tier, "passed": problem is None, "problem": problem})
        if R.all_ranks(problem is None):
            break
        if R.all_ranks(on_peer) and R.all_ranks(not lib_error) and peer_tier < 2:
            # every rank is on the peer transport and some rank's check failed: the next tier of its memory-ordering protocol, on ALL ranks
            peer_tier += 1
            sys.stderr.write(f"bench.py[rank {R.rank}]: peer transport, tier {peer_tier - 1}: {problem or 'another rank reported a problem'}; trying tier {peer_tier}\n")
            R.model.set_peer_tier(peer_tier)
            continue
        if R.model.ctx.halo_transport() == "peer" or (peer_expected and transport_note is None):
            transport_note = f"peer transport given up ({problem or 'another rank reported a problem'}): RCCL exchange (halo 32, k = 16) timed instead"
            sys.stderr.write(f"bench.py[rank {R.rank}]: {transport_note}\n")
            peer_expected = False
            R.model = None
            R.tg, R.f, R.model, args.halo = R.build("rccl")
            continue
        raise SystemExit(f"bench.py[rank {R.rank}]: invalid run: {problem or 'another rank reported a problem'}")
    if verify:
        R.restart(R.model, R.f)                                  # the timed steps start from the seeded state again
        for _ in range(max(args.warmup, 1)):
            csi.time_step_momentum(R.model, R.dt)
    return verify, bitwise, ladder, transport_note


def result_check(R):
    """Outside the timed region: the state the timed steps produced is finite and non-trivial (on every rank)."""
    torch, model = R.torch, R.model
    model.synchronize()
    chk = {}
    for name, fld in (("u", model.velocities.u), ("v", model.velocities.v), ("s11", model.dynamics.auxiliaries.fields.s11)):
        t = fld.data
        chk[name] = {"finite": bool(torch.isfinite(t).all().item()), "max_abs": float(t.abs().max().item()),
                     "nonzero_frac": float((t != 0).double().mean().item())}
    ok = all(c["finite"] for c in chk.values()) and 0 < chk["u"]["max_abs"] < 10.0 and chk["u"]["nonzero_frac"] > 0.5 \
        and chk["s11"]["max_abs"] > 0
    if not R.all_ranks(ok):
        raise SystemExit(f"bench.py: the timed steps left a non-finite or trivial state: {chk}")
    return chk


def unfused_record(R):
    """The SURVEY.md 8(d)-literal figure, measured in the SAME run (outside the headline region): the unfused three-kernel FAST path
    (k_stress, k_ustep, k_vstep: every array crosses HBM once per phase) moves the contract's 256 B per cell-update, so
    cell-updates/s x 256 B is a roofline fraction in the contract's own terms (< 1); the fused kernels do the same work on fewer bytes
    (their `frac` is priced on their own compulsory bytes)."""
    args, csi, torch = R.args, R.csi, R.torch
    nx_l, ny_l = R.nx_l, R.ny_l
    grid_u, fld_u = local_case(csi, R.np, nx_l, ny_l, R.Rx, R.Ry, R.rank, halo=args.halo)
    mu = R.make_model(grid_u)
    mu.set_fusion(0)
    csi.set_(mu, h=fld_u["h"], aice=fld_u["a"], u=fld_u["u"], v=fld_u["v"])
    csi.time_step_momentum(mu, R.dt)
    mu.synchronize(); torch.cuda.synchronize()
    nun = max(2, min(args.steps, 3))
    t0u = time.perf_counter()
    for _ in range(nun):
        csi.time_step_momentum(mu, R.dt)
    mu.synchronize(); torch.cuda.synchronize()
    eu = time.perf_counter() - t0u
    vu = nx_l * ny_l * args.substeps * nun / eu
    phu = mu.ctx.profile_substeps(R.dt, 16)
    return {"value": vu, "unit": "cell-updates/s", "ms_per_step": 1e3 * eu / nun, "steps": nun,
            "kernels": "csi::fast::k_stress + k_ustep + k_vstep (three launches per sub-step)",
            "algorithmic_bytes_per_cell_update": 256.0, "achieved": vu * 256.0 / 1e9, "unit_achieved": "GB/s",
            "frac": vu * 256.0 / 1e9 / HBM_PEAK_GBS, "frac_of_achievable": vu * 256.0 / 1e9 / HBM_ACHIEVABLE_GBS,
            "phases_ms": {k: phu[k] for k in ("stress", "ustep", "vstep")},
            "phase_fracs": {k: nx_l * ny_l * ALGO_BYTES[k] / (phu[k] * 1e-3) / 1e9 / HBM_PEAK_GBS for k in ("stress", "ustep", "vstep") if phu[k] > 0},
            "note": "SURVEY.md 8(d) formula on the path that moves those bytes; same grid, same run, outside the timed region"}


def counters_into_roofline(R, roof, dom, launch_s, kernel_bytes):
    """PMC evidence of a committed rocprofv3 run of this kernel (scripts/same_lease_profile.sh): bytes and instruction counts per launch
    are properties of the kernel; FRACTIONS of a roof need a time, and only this run's own clock is used for that -- and only when
    this box runs the kernel as fast as the box the counters were taken on (within 5 %), otherwise the fractions are withheld rather
    than mixed across machines."""
    ctr = counters()
    if not (ctr and R.world == 1 and not R.tiled and (R.nx_l, R.ny_l) == (2048, 2048) and R.args.mode == "fast" and ctr.get("kernel") == dom):
        return
    roof["traffic"] = ctr.get("hbm_bytes_per_launch")
    roof["traffic_source"] = ctr.get("source", "profiles/counters_latest.json") + " (rocprofv3 --pmc passes; bytes per launch of this kernel, not re-measured by bench.py)"
    ref_us = (ctr.get("same_lease_bench") or {}).get("avg_launch_us") or ctr.get("avg_launch_us")
    same_speed = ref_us is not None and abs(launch_s * 1e6 - ref_us) <= 0.05 * ref_us
    if ctr.get("valu_insts_per_launch"):
        # third view: the kernel's vector instructions per launch (a property of the kernel, from the counters) at the FP64 issue
        # interval this chip sustains, over THIS run's launch time.  A lower bound of the true share: the ~5 % of them that
        # are v_rcp_f64 / v_rsq_f64 take 3 x as long (profiles/r04b_full_metric_kernel.md section 4)
        roof["fp64_issue_frac"] = ctr["valu_insts_per_launch"] * FP64_ISSUE_NS * 1e-9 / 1024.0 / launch_s
        roof["fp64_issue_note"] = (f"{ctr['valu_insts_per_launch'] / 1e6:.1f} M vector instructions per launch x {FP64_ISSUE_NS} ns (measured FP64 issue interval per SIMD, "
                                   "profiles/r01_microbenchmarks.md) / 1024 SIMDs / this run's launch time")
        dyn = ctr.get("valu_mix_per_launch")
        if dyn:
            # the same from the DYNAMIC mix (PMC pass SQ_INSTS_VALU_{FMA,MUL,ADD,TRANS}_F64, same committed counter run): every
            # class at its measured issue interval (profiles/r01_microbenchmarks.md): fma / mul 2.29 ns, add 2.00, transcendental
            # (v_rcp_f64 / v_rsq_f64) 7.03, integer 1.31, the rest (DPP shifts, compares, selects, 64-bit moves) 2.0
            w = {"fma_f64": 2.29, "mul_f64": 2.29, "add_f64": 2.00, "trans_f64": 7.03, "int32": 1.31, "int64": 1.79, "other": 2.0}
            busy_ns = sum(dyn.get(k, 0.0) * w[k] for k in w)
            roof["fp64_issue_frac_dynamic"] = busy_ns * 1e-9 / 1024.0 / launch_s
            roof["fp64_issue_dynamic_mix"] = {"per_launch": dyn, "issue_ns": w}
        mix = isa_mix()
        if mix and mix.get("kernel") == dom:
            # the same with every class of vector instruction at ITS measured issue interval (profiles/r01_microbenchmarks.md
            # `valu_rate` at >= 2 waves per SIMD): v_rcp_f64 / v_rsq_f64 take 7.03 ns, FP64 fma / mul 2.29, add 2.00, max / min /
            # compares 1.9-2.2, DPP shifts 2.18, 64-bit moves 1.79, 32-bit 1.31.  Shares from the ISA listing of this
            # instantiation (scripts/isa_mix.py, static counts of the two row loops).
            ns = sum(mix["share"][k] * mix["issue_ns"][k] for k in mix["share"])
            roof["fp64_issue_frac_weighted"] = ctr["valu_insts_per_launch"] * ns * 1e-9 / 1024.0 / launch_s
            roof["fp64_issue_mix"] = {"share": mix["share"], "issue_ns": mix["issue_ns"], "mean_issue_ns": ns,
                                      "valu_per_stage_row": mix.get("valu_per_stage_row"), "source": mix.get("source")}
    # the same fraction at the launch time the kernel has in the committed rocprofv3 kernel trace of the counters' lease: a reader who
    # divides the compulsory bytes by the trace's average duration gets this number (round 5's trace was a cold 3-step run, 12 % slow:
    # profiles/r06_profiler_vs_bench.md; a warm trace agrees with `frac`)
    if ctr.get("avg_launch_us"):
        roof["frac_under_profiler"] = kernel_bytes / (ctr["avg_launch_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS
        roof["frac_under_profiler_note"] = (f"{KERNEL_BYTES[dom]:.0f} B x cells / {ctr['avg_launch_us']:.1f} us (rocprofv3 --kernel-trace average of the timed launches, "
                                            f"{ctr.get('source', 'profiles/counters_latest.json')}) / 8 TB/s")
    roof["counters_run_launch_us"] = ref_us
    roof["counters_run_matches_this_box"] = bool(same_speed)
    if same_speed:
        if ctr.get("hbm_bytes_per_launch"):
            roof["traffic_frac"] = ctr["hbm_bytes_per_launch"] / launch_s / 1e9 / HBM_PEAK_GBS
        # second roof: FP64 vector issue.  valu_frac = share of the SIMDs' VALU issue time the launch used, from the
        # same counter passes: SQ_ACTIVE_INST_VALU (quad-cycles, summed over SIMDs) x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)
        if ctr.get("valu_busy_frac") is not None:
            roof["valu_frac"] = ctr["valu_busy_frac"]
            roof["valu_insts_per_launch"] = ctr.get("valu_insts_per_launch")
            roof["valu_note"] = "SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), PMC pass of the counters' run"
    else:
        roof["traffic_note"] = (f"this box runs the kernel in {launch_s * 1e6:.1f} us per launch, the counters' run took {ref_us:.1f} us: "
                                "traffic_frac / valu_frac withheld (not the same speed)")


def roofline_record(R, path, clock):
    """`roofline` of the dominant kernel: per-kernel HIP-event times outside the timed region (csi_profile_substeps) for the three-kernel
    paths; for the fused kernels the average launch time of the TIMED region itself (round 5): HIP events around every sub-step loop of the
    K timed steps, divided by the launches inside them -- so that launches x average <= ms_per_step holds by construction.  The figure of the
    separate profiling pass (a short run behind the timed region, 10-15 % colder) stays in the line as `profile_pass_launch_ms`."""
    args, model, region = R.args, R.model, R.region
    phases = model.ctx.profile_substeps(R.dt, 32)
    cells_launch = R.nx_l * R.ny_l
    spl = 1.0                                             # sub-steps per launch of the dominant kernel
    if path["fused"]:
        launches, nsub = model.ctx.last_launches()
        spl = nsub / max(launches, 1)
        dom = {2: "pair"}.get(path["level"], "substep")
        phases = {dom: phases["stress"], "exchange": phases["exchange"]}
        sub_ms = phases[dom] / spl
    else:
        dom = max(("stress", "ustep", "vstep"), key=lambda k: phases[k])
        sub_ms = phases["stress"] + phases["ustep"] + phases["vstep"]
    profile_pass_ms = phases[dom]
    launch_src = "csi_profile_substeps: a separate pass of 32 sub-steps behind the timed region"
    if path["fused"] and region.get("launches"):
        phases[dom] = region["total_ms"] / region["launches"]
        sub_ms = phases[dom] / spl
        launch_src = (f"HIP events around the sub-step loops of the {region['cycles']} TIMED steps: {region['total_ms']:.3f} ms / "
                      f"{region['launches']} launches")
    launch_s = phases[dom] * 1e-3
    # roofline.achieved: the kernel's compulsory HBM bytes per launch (owned cells; ring re-reads and the extra cells of a
    # tile's valid-halo ring are overhead, not counted) / its average launch duration from HIP events on the launch stream
    kernel_bytes = cells_launch * KERNEL_BYTES[dom]
    achieved = kernel_bytes / launch_s / 1e9
    # the SURVEY.md 8(d) figure: 256 B per cell-update of the unfused three-phase split, for comparison only (a fused kernel
    # does not move these bytes, so this can exceed 1: it measures the traffic fusion removed, not a roofline)
    algorithmic_rate = cells_launch * 256.0 * spl / launch_s / 1e9 if path["fused"] else cells_launch * ALGO_BYTES[dom] / launch_s / 1e9
    roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": None, "traffic_source": None,
            "kernel": KERNEL_NAMES[dom] if args.mode == "fast" else dom,
            "kernel_bytes_per_launch": kernel_bytes, "kernel_bytes_per_cell": KERNEL_BYTES[dom],
            "avg_launch_ms": phases[dom], "avg_launch_source": launch_src, "profile_pass_launch_ms": profile_pass_ms,
            "launches_per_step": (region["launches"] / region["cycles"]) if region.get("cycles") else None,
            "launches_x_avg_ms": (region["total_ms"] / region["cycles"]) if region.get("cycles") else None,
            "substeps_per_launch": spl, "all_phases_ms": phases,
            "algorithmic_bytes_per_launch": cells_launch * (256.0 * spl if path["fused"] else ALGO_BYTES[dom]),
            "algorithmic_frac": algorithmic_rate / HBM_PEAK_GBS,
            "algorithmic_note": "SURVEY.md 8(d): 256 B per cell-update of the unfused stress / u / v split; above 1 = traffic removed by fusion",
            "substep_ms": sub_ms}
    roof["frac_of_achievable"] = achieved / HBM_ACHIEVABLE_GBS
    # the clock and the power the launches ran at (rank 0's GPU, sampled over a repetition of the timed loop): 2048^2 sits at the board's
    # power cap, below the 2.4 GHz the issue intervals of FP64_ISSUE_NS were measured near; small tiles do not (profiles/r05_power_clock.md)
    roof["clock"] = clock.summary()
    if R.world == 1 and not R.tiled and args.mode == "fast" and path["fused"] and not args.no_unfused:
        roof["unfused"] = unfused_record(R)
    counters_into_roofline(R, roof, dom, launch_s, kernel_bytes)
    return roof


def other_transports(R, path):
    """Tiles: the other ways to move the halos, timed outside the headline region: the RCCL exchange batched over k = 16 sub-steps
    (halo 32) and once per sub-step (k = 1).  Returns (rccl16, k1)."""
    args, csi = R.args, R.csi
    k1 = rccl16 = None
    headline_model = R.model
    if R.tiled and args.exchange_interval == 0 and not args.no_compare:
        if path["halo_transport"] == "peer" and not R.user_halo:
            _, _, R.model, _ = R.build("rccl")                       # (timed() and barrier() act on R.model)
            for _ in range(max(args.warmup, 1)):
                csi.time_step_momentum(R.model, R.dt)
            e16 = R.timed(args.steps)
            p16 = R.model.ctx.last_path()
            rccl16 = {"value": R.rate(e16), "ms_per_step": 1e3 * e16 / args.steps, "halo": 32,
                      "exchange_interval": p16["exchange_interval"], "exchanges_per_step": p16["exchanges"], "level": p16["level"]}
        if R.model.ctx.halo_transport() == "rccl" and R.model.ctx.last_path()["exchange_interval"] != 1:
            R.model.set_exchange_interval(1)
            csi.time_step_momentum(R.model, R.dt)
            e1 = R.timed(args.steps)
            p1 = R.model.ctx.last_path()
            k1 = {"value": R.rate(e1), "ms_per_step": 1e3 * e1 / args.steps,
                  "exchanges_per_step": p1["exchanges"], "level": p1["level"]}
            R.model.set_exchange_interval(args.exchange_interval)
        R.model = headline_model
    return rccl16, k1


def second_partition(R, verify, transport_note):
    """N > 1: the OTHER decomposition of the same grid, same run (round 5): BASELINE config 4 names 2 x 4 tiles, the headline runs y slabs
    (the faster shape on every one-GPU stand-in) -- both are checked bit for bit and timed, so that the first run on
    a real node answers the contract's layout as written and tells whether the slab choice holds over xGMI."""
    args, csi = R.args, R.csi
    if not (R.world > 1 and not args.no_second_partition and not args.tile and args.scaling == "strong" and verify):
        return None
    alt = None
    if R.Rx == 1 and R.world >= 2:
        alt = (2, R.world // 2)
    elif R.Ry != R.world or R.Rx != 1:
        alt = (1, R.world)
    if not (alt and args.size % alt[0] == 0 and args.size % alt[1] == 0 and alt != (R.Rx, R.Ry)):
        return None
    headline_model = R.model
    second = None
    part = (alt[0], alt[1], args.size // alt[0], args.size // alt[1])
    # (every collective of this block -- the all_ranks() reductions, the barriers inside timed() -- is reached by EVERY rank whatever
    #  a rank's own library calls did: a rank that caught an error must not leave the others waiting in an all-reduce)
    err2, same2_local = None, False
    try:
        g2, f2, R.model, _ = R.build(args.transport if not transport_note else "rccl", part)
        for _ in range(max(args.warmup, 1)):
            csi.time_step_momentum(R.model, R.dt)
        same2_local = R.tile_matches_whole(R.model, g2, f2, global_fields(R.np, part[2], part[3], part[0], part[1]))
    except csi.CsiError as e:
        err2 = str(e)
    if R.all_ranks(err2 is None):
        same2 = R.all_ranks(same2_local)
        try:
            R.restart(R.model, f2)
            for _ in range(max(args.warmup, 1)):
                csi.time_step_momentum(R.model, R.dt)
        except csi.CsiError as e:
            err2 = str(e)
        if R.all_ranks(err2 is None):
            e2 = R.timed(args.steps)
            second = {"partition": list(alt), "tile": [part[2], part[3]], "value": R.rate(e2),
                      "ms_per_step": 1e3 * e2 / args.steps, "bitwise": bool(same2),
                      "halo_transport": R.model.ctx.halo_transport(), "peer_tier": R.model.ctx.peer_tier() if R.model.ctx.halo_transport() == "peer" else None}
    if second is None:
        second = {"partition": list(alt), "error": err2 or "another rank reported a library error"}
    R.model = headline_model
    return second


def tier0_record(R, path, verify):
    """The fence-free tier 0 of the peer protocol as an explicit opt-in, timed next to the default (never the default across devices:
    include/csi.h)."""
    args, csi, model = R.args, R.csi, R.model
    if not (R.tiled and path["halo_transport"] == "peer" and args.peer_tier < 0 and path.get("peer_tier", 0) >= 1 and verify and not args.no_compare):
        return None
    err0, same0_local = None, False
    try:
        model.set_peer_tier(0)
        same0_local = R.tile_matches_whole()
        R.restart(model, R.f)
        csi.time_step_momentum(model, R.dt)
    except csi.CsiError as e:
        err0 = str(e)
    if R.all_ranks(err0 is None):                      # (collectives outside the try: see second_partition)
        same0 = R.all_ranks(same0_local)
        e0 = R.timed(args.steps)
        tier0 = {"value": R.rate(e0), "ms_per_step": 1e3 * e0 / args.steps, "bitwise": bool(same0),
                 "note": "opt-in (--peer-tier 0): no acquire fence behind the flags; a passing check does not prove the protocol"}
    else:
        tier0 = {"error": err0 or "another rank reported a library error"}
    try:
        model.set_peer_tier(path["peer_tier"])
    except csi.CsiError:
        pass
    return tier0


def single_gpu_record(R, verify):
    """N > 1: the one-GPU rate of the same global grid: every rank times its own copy (no shared resource), rank 0's is reported."""
    if not verify:
        return None
    args, csi, whole, torch = R.args, R.csi, R.whole, R.torch
    R.restart(whole, R.gf)
    for _ in range(max(args.warmup, 1)):
        csi.time_step_momentum(whole, R.dt)
    whole.synchronize(); torch.cuda.synchronize()
    R.dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        csi.time_step_momentum(whole, R.dt)
    whole.synchronize(); torch.cuda.synchronize()
    e_single = time.perf_counter() - t0
    R.whole = None
    return {"value": R.gN[0] * R.gN[1] * args.substeps * args.steps / e_single, "ms_per_step": 1e3 * e_single / args.steps,
            "grid": list(R.gN), "note": "the same global grid advanced by ONE GPU (rank 0) in this run, same kernels, halo 4 "
                                        "(every rank runs its own copy at the same time, each on its own GPU)"}


def main():
    args = parse_args()
    if args.gpus not in PARTITION:
        raise SystemExit(f"--gpus must be one of {sorted(PARTITION)}")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1 or args.print_launch:
            # the parent: nothing below this line has touched HIP (no torch.cuda call, no library load)
            sys.exit(run_parent(args, sys.argv[1:]))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE = {os.environ['WORLD_SIZE']}: start N ranks with --gpus N "
                         f"(python -m torch.distributed.run --nproc-per-node N bench.py --gpus N), or run `python bench.py --gpus N` alone")
    if args.self_test_launch:
        return self_test_launch()

    R = Rank(args)
    csi, np, torch = R.csi, R.np, R.torch
    world, rank, tiled, dt = R.world, R.rank, R.tiled, R.dt
    Rx, Ry, nx_l, ny_l = R.Rx, R.Ry, R.nx_l, R.ny_l
    R.tg, R.f, R.model, args.halo = R.build(args.transport)

    # ---- warm-up, the N > 1 bitwise check and the peer tier ladder; then the headline region --------------------------------------------
    verify, bitwise, ladder, transport_note = warm_up_and_verify(R)
    R.barrier()
    # RCCL prints its version banner through C stdio at communicator creation; flush it now so that the JSON line
    # below is the last thing this process writes
    ctypes.CDLL(None).fflush(None)
    elapsed = R.timed(args.steps, stats=True)
    # the clock / power sampler reads sysfs from a Python thread: it runs over a REPETITION of the timed loop right behind it, so that
    # the headline region holds the launch loop alone, as in rounds 1-4
    with ClockSampler(R.local_rank) as clock:
        R.timed(args.steps)

    model = R.model
    value = R.rate(elapsed)
    subcycle_ms = model.ctx.last_subcycle_ms()            # HIP events on the launch stream, last step
    path = model.ctx.last_path()
    path["halo_transport"] = model.ctx.halo_transport() if tiled else "none"
    if transport_note:
        path["halo_transport_note"] = transport_note
    if tiled and path["halo_transport"] == "peer":
        path["peer_tier"] = model.ctx.peer_tier()
    if ladder:
        path["peer_tier_ladder"] = ladder          # which tiers were tried before the timed region, and what each check said
    chk = result_check(R)

    # ---- outside the timed region: the roofline of the dominant kernel, whole model steps, the other configurations ----------------------
    roof = roofline_record(R, path, clock)
    model_days_per_hr = None
    if not args.no_full_step:
        nfull = 2
        csi.time_step(model, dt)                           # warm-up (WENO kernels, RK3 copies)
        R.barrier()
        t1 = time.perf_counter()
        for _ in range(nfull):
            csi.time_step(model, dt)                       # RK3: 3 x [WENO7 tendencies + sub-cycle + tracer update + halos]
        R.barrier()
        full = R.max_over_ranks((time.perf_counter() - t1) / nfull)
        model_days_per_hr = 3600.0 / (full * 86400.0 / dt)
    one_gpu_extras = world == 1 and not tiled and args.mode == "fast"
    # BASELINE config 2 beside the headline (round 5): advection only -- 512^2 periodic, WENO(order = 7), prescribed cyclonic eddy,
    # SplitRungeKutta3 (examples/ice_advected_by_anticyclone.jl:65-66,117,126 scaled as SURVEY.md 8d) -- and the tendency launch at 2048^2
    advection = advection_record(csi, np, torch, R.device) if (one_gpu_extras and not args.no_full_step) else None
    config3 = config3_record(csi, np, torch, R.make_model, args.substeps, dt) \
        if (one_gpu_extras and not args.tile and (nx_l, ny_l) != (1024, 1024) and not args.no_structure) else None
    structure = structure_record(csi, np, torch, R.device, args.substeps, quick=args.size < 2048) \
        if (one_gpu_extras and not args.no_structure and not args.tile and not args.no_fusion and args.fusion_level >= 2) else None
    rccl16, k1 = other_transports(R, path)
    second = second_partition(R, verify, transport_note)
    tier0 = tier0_record(R, path, verify)
    rccl_ranks = R.model.ctx.comm_count()
    if world > 1 and rccl_ranks != world:
        raise SystemExit(f"bench.py: {world} ranks but the library's RCCL communicator has {rccl_ranks}")
    single = single_gpu_record(R, verify)

    out = {
        "metric": "EVP sub-cycle cell-updates/s", "value": value, "unit": "cell-updates/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"evp_subcycle_fplane_periodic_{nx_l * Rx}x{ny_l * Ry}_as_{Rx}x{Ry}_tiles_of_{nx_l}x{ny_l}"
                               f"_{args.substeps}substeps",
                   "global_grid": [nx_l * Rx, ny_l * Ry], "tile": [nx_l, ny_l], "partition": [Rx, Ry],
                   "substeps": args.substeps, "mode": args.mode,
                   "halo": args.halo,
                   "halo_exchange": "none (one tile)" if not tiled
                   else ("peer-direct halo writes over xGMI (IPC-mapped neighbour arrays, per-tile flags) inside every launch; "
                         "one RCCL exchange of u, v, sigma per sub-cycle" if path["halo_transport"] == "peer"
                         else f"RCCL send/recv of u, v, sigma: width {2 * path['exchange_interval']} every {path['exchange_interval']} sub-steps "
                              f"({path['exchanges']} exchanges per step)")},
        "model_days_per_hr": model_days_per_hr,
        "model_days_per_hr_config": "full RK3 time_step! (3 stages x [WENO7 advection of h, aice + sub-cycle + tracer update]), dt = 120 s",
        "subcycle_ms_hip_events": subcycle_ms,
        "path": path,
        "result_check": chk,
        "roofline": roof,
    }
    if advection is not None:
        out["advection"] = advection
    if config3 is not None:
        out["config3_1024"] = config3
    if structure is not None:
        out["structure"] = structure
    if k1 is not None:
        out["exchange_every_substep"] = k1
    if rccl16 is not None:
        out["rccl_exchange"] = rccl16
    if R.rehearsal:
        out["rehearsal_on_one_gpu"] = True
        out["rehearsal_note"] = (f"{world} ranks are PROCESSES sharing GPU 0 (host-channel group: shared memory + HIP IPC, gloo): a rehearsal of the "
                                 "N > 1 path, NOT a scaling measurement -- value, ms_per_step and parallel_efficiency are those of a shared device")
    if world > 1:
        out["rccl_ranks"] = rccl_ranks
        out["single_gpu"] = single
        out["tiled_equals_untiled_bitwise"] = bitwise
        if single is not None:
            # strong scaling: same grid on N GPUs vs on one; weak: N tiles of the one-GPU size vs ... the N-times larger grid
            # on one GPU (its rate, not its time, is what N GPUs are compared with)
            out["parallel_efficiency"] = value / (world * single["value"])
        # ... and against the north star's own rates: one GPU at 40 % of the HBM roofline on 256 B per cell-update
        # = 12.5 G cell-updates/s, N GPUs at 75 % parallel efficiency of THAT = N x 9.375 G (75 G at N = 8).  parallel_efficiency above is
        # measured against this run's own single GPU, which is several times faster than the target rate: the two ratios answer
        # different questions and both are printed.
        out["parallel_efficiency_vs_target_rate"] = value / (world * NORTH_STAR_1GPU_RATE)
        out["target_rate"] = {"one_gpu_cell_updates_per_s": NORTH_STAR_1GPU_RATE, "n_gpu_cell_updates_per_s": world * NORTH_STAR_1GPU_RATE * 0.75,
                              "value_over_n_gpu_target": value / (world * NORTH_STAR_1GPU_RATE * 0.75),
                              "source": "BASELINE.md section 2: >= 40 % of 8 TB/s on 256 B per cell-update on one GPU, >= 75 % parallel efficiency at N = 8"}
        if second is not None:
            if single is not None and "value" in second:
                second["parallel_efficiency"] = second["value"] / (world * single["value"])
            out[f"partition_{second['partition'][0]}x{second['partition'][1]}"] = second
        if tier0 is not None:
            out["peer_tier0"] = tier0
        out["partition_note"] = ("headline = y slabs (1 x N): the faster tile shape on every one-GPU stand-in (a tile connected to itself: "
                                 "scripts/tile_shapes.py), NOT yet measured over xGMI; BASELINE config 4's 2 x 4 layout is timed in the same run "
                                 "(partition_2x4), `--partition 2x4` makes it the headline") if Rx == 1 and world > 1 else None
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline()
    model = None                                   # contexts (and their RCCL communicators) go before the line is printed
    R.model = R.whole = None
    import gc
    gc.collect()
    if R.dist is not None:
        R.dist.barrier()
        R.dist.destroy_process_group()
    ctypes.CDLL(None).fflush(None)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
