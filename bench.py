#!/usr/bin/env python3
"""bench.py -- EVP sub-cycle throughput on MI355X (the metric of BASELINE.json).

One "step" = one time_step_momentum! call (initialize_rheology! + `substeps` EVP sub-steps +
finalize_rheology!, SeaIceDynamics/split_explicit_momentum_equations.jl:103-195) over the whole grid,
inputs resident in HBM.  value = cell-updates/s = owned cells * substeps * steps / wall time, whole job.

N = 1: the 2048 x 2048 periodic f-plane grid the metric is quoted on.
N > 1 (torch.distributed.run, one rank per GPU): the grid is an Rx x Ry arrangement of 2048 x 2048 tiles
(weak scaling: per-GPU work fixed) advanced by the SAME kernels with the RCCL halo exchange of u, v, sigma
(width 2k every k sub-steps; halo 32 -> k = 16); `--scaling strong` instead splits ONE 2048 x 2048 grid over the ranks.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

ALGO_BYTES = {"stress": 96.0, "ustep": 80.0, "vstep": 80.0}   # SURVEY.md 8(d); 256 B per cell-update
HBM_PEAK_GBS = 8000.0                                          # MI355X_MICROARCH.md: 8.0 TB/s spec
PARTITION = {1: (1, 1), 2: (2, 1), 4: (2, 2), 8: (2, 4)}


def cpu_baseline(seconds_budget=15.0):
    """The oracle (strict-order C restatement, OpenMP over rows) timed on the host cores of this box:
    a reported baseline on a bounded sample of the same workload, never the target."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cases
    n, sub = 2048, 2
    c = cases.make_case(Nx=n, Ny=n, substeps=sub, topo=("periodic", "periodic"), patches=True, random_uv=0.02)
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    p = cases.oracle_problem(c, omp=True)
    p.time_step_momentum(c["dt"])                       # warm-up
    t0 = time.perf_counter()
    reps = 0
    while True:
        p.time_step_momentum(c["dt"])
        reps += 1
        if time.perf_counter() - t0 > seconds_budget or reps >= 64:
            break
    dt = time.perf_counter() - t0
    return {"value": n * n * sub * reps / dt, "unit": "cell-updates/s", "cores": cores, "kind": "port",
            "sample": f"{reps} x time_step_momentum! of {n}x{n} periodic f-plane with {sub} sub-steps; "
                      f"oracle/csi_oracle.c (reference kernel split, strict order) with OpenMP over rows, {cores} threads"}


def local_case(csi, np, nx, ny, Rx, Ry, rank, force_connected=False, halo=4):
    """Seeded synthetic inputs of one tile, functions of the GLOBAL coordinates:
    h0 sinusoid + 5 % noise, aice patches (open water / marginal ice), u = 0.1 + 2 % noise."""
    import cases
    rx, ry = rank % Rx, rank // Rx
    Nx, Ny = nx * Rx, ny * Ry
    g = csi.RectilinearGrid((Nx, Ny), x=(0.0, Nx * 2000.0), y=(0.0, Ny * 2000.0), topology=(csi.Periodic, csi.Periodic), halo=(halo, halo))
    tg = csi.TileGrid(g, Rx, Ry, rx, ry, force_connected=force_connected) if (Rx * Ry > 1 or force_connected) else g
    rng = np.random.default_rng(1000 + rank)
    xc = ((np.arange(nx) + rx * nx) + 0.5) / Nx
    yc = ((np.arange(ny) + ry * ny) + 0.5) / Ny
    X, Y = xc[None, :], yc[:, None]
    h = (0.3 + 0.005 * (np.sin(2 * np.pi * 3 * Rx * X) + np.sin(2 * np.pi * 2 * Ry * Y))) * (1.0 + 0.05 * (rng.random((ny, nx)) - 0.5))
    a = np.clip(0.6 + 0.6 * np.sin(2 * np.pi * Rx * X) * np.cos(2 * np.pi * Ry * Y) + 0.2 * rng.random((ny, nx)), 0.0, 1.0)
    i0, i1, j0, j1 = nx // 8, nx // 4, ny // 6, ny // 3
    a[j0:j1, i0:i1] = 0.0
    h[j0:j1, i0:i1] = 0.0
    a[j1:j1 + 2, i0:i1] = 5e-4
    h[j1:j1 + 2, i0:i1] = 1e-3
    u = 0.1 + 0.02 * rng.standard_normal((ny, nx))
    v = 0.02 * rng.standard_normal((ny, nx))
    return tg, dict(h=h, a=a, u=u, v=v)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=2048, help="tile edge (weak) or global edge (strong)")
    ap.add_argument("--substeps", type=int, default=120)
    ap.add_argument("--mode", default="fast", choices=["fast", "strict"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-full-step", action="store_true")
    ap.add_argument("--exchange-interval", type=int, default=0, help="k: exchange width 2k every k sub-steps (0 = auto)")
    ap.add_argument("--halo", type=int, default=0, help="halo width (default: 4 on one GPU, 32 on tiles so that k = 16)")
    ap.add_argument("--no-fusion", action="store_true", help="three-kernel FAST path instead of the fused sub-step kernels")
    ap.add_argument("--fusion-level", type=int, default=2, choices=[0, 1, 2],
                    help="0: three kernels per sub-step, 1: one fused launch per sub-step, 2: two sub-steps per launch (default)")
    ap.add_argument("--force-connected", action="store_true",
                    help="debug: on one GPU, route the periodic halos through the RCCL exchange (to self)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import climaseaice_jl_amd as csi

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    if world not in PARTITION:
        raise SystemExit(f"--gpus must be one of {sorted(PARTITION)}")
    Rx, Ry = PARTITION[world]
    if args.scaling == "weak":
        nx_l = ny_l = args.size
    else:
        if args.size % Rx or args.size % Ry:
            raise SystemExit("--size must be divisible by the partition")
        nx_l, ny_l = args.size // Rx, args.size // Ry
    if args.halo == 0:
        args.halo = 32 if (world > 1 or args.force_connected) else 4
    device = f"cuda:{local_rank}"
    tg, f = local_case(csi, np, nx_l, ny_l, Rx, Ry, rank, force_connected=args.force_connected, halo=args.halo)
    dyn = csi.SeaIceMomentumEquation(tg, coriolis=csi.FPlane(f=1e-4), rheology=csi.ElastoViscoPlasticRheology(),
                                     top_momentum_stress=(0.01, 0.01), bottom_momentum_stress=csi.SemiImplicitStress(),
                                     solver=csi.SplitExplicitSolver(substeps=args.substeps), device=device)
    model = csi.SeaIceModel(tg, dynamics=dyn, advection=csi.WENO(order=7), timestepper="SplitRungeKutta3", device=device, mode=args.mode)
    model.set_exchange_interval(args.exchange_interval)
    model.set_fusion(0 if args.no_fusion else args.fusion_level)
    csi.set_(model, h=f["h"], aice=f["a"], u=f["u"], v=f["v"])
    dt = 120.0

    def barrier():
        model.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        csi.time_step_momentum(model, dt)
    barrier()
    # RCCL prints its version banner through C stdio at communicator creation; flush it now so that the JSON line
    # below is the last thing this process writes
    import ctypes
    ctypes.CDLL(None).fflush(None)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        csi.time_step_momentum(model, dt)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    owned = nx_l * ny_l * world
    value = owned * args.substeps * args.steps / elapsed
    subcycle_ms = model.ctx.last_subcycle_ms()            # HIP events on the launch stream, last step

    # ---- outside the timed region: per-kernel HIP-event times (roofline) and whole model steps ----
    path = model.ctx.last_path()
    phases = model.ctx.profile_substeps(dt, 32)
    cells_launch = nx_l * ny_l
    spl = 1.0                                             # sub-steps per launch of the dominant kernel
    if path["fused"]:
        # one launch performs one whole sub-step (level 1) or two (level 2): its algorithmic bytes are the 256 B per
        # cell-update of SURVEY.md 8(d) times the sub-steps it performs
        launches, nsub = model.ctx.last_launches()
        spl = nsub / max(launches, 1)
        dom = "pair" if path["level"] == 2 else "substep"
        algo = 256.0 * spl
        phases = {dom: phases["stress"], "exchange": phases["exchange"]}
        sub_ms = phases[dom] / spl
    else:
        dom = max(("stress", "ustep", "vstep"), key=lambda k: phases[k])
        algo = ALGO_BYTES[dom]
        sub_ms = phases["stress"] + phases["ustep"] + phases["vstep"]
    achieved = cells_launch * algo / (phases[dom] * 1e-3) / 1e9
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tfile) and world == 1 and args.size == 2048 and args.mode == "fast":
        tj = json.load(open(tfile))
        traffic = tj.get("bytes_per_launch", {}).get(dom)   # PMC passes of profiles/ (FETCH_SIZE doubled, gfx950 rule)
    model_days_per_hr = None
    if not args.no_full_step:
        nfull = 2
        csi.time_step(model, dt)                           # warm-up (WENO kernels, RK3 copies)
        barrier()
        t1 = time.perf_counter()
        for _ in range(nfull):
            csi.time_step(model, dt)                       # RK3: 3 x [WENO7 tendencies + sub-cycle + tracer update + halos]
        barrier()
        full = (time.perf_counter() - t1) / nfull
        if dist is not None:
            t = torch.tensor([full], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            full = float(t.item())
        model_days_per_hr = 3600.0 / (full * 86400.0 / dt)

    out = {
        "metric": "EVP sub-cycle cell-updates/s", "value": value, "unit": "cell-updates/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"evp_subcycle_fplane_periodic_{nx_l * Rx}x{ny_l * Ry}_as_{Rx}x{Ry}_tiles_of_{nx_l}x{ny_l}"
                               f"_{args.substeps}substeps",
                   "global_grid": [nx_l * Rx, ny_l * Ry], "tile": [nx_l, ny_l], "partition": [Rx, Ry],
                   "substeps": args.substeps, "mode": args.mode,
                   "halo": args.halo,
                   "halo_exchange": "none (one tile)" if (world == 1 and not args.force_connected)
                   else f"RCCL send/recv, exchange interval k={args.exchange_interval or 'auto(min(halo/2,16))'}: width 2k every k sub-steps"},
        "model_days_per_hr": model_days_per_hr,
        "model_days_per_hr_config": "full RK3 time_step! (3 stages x [WENO7 advection of h, aice + sub-cycle + tracer update]), dt = 120 s",
        "subcycle_ms_hip_events": subcycle_ms,
        "path": path,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": {"substep": "csi::fused::k_substep (stress + u + v in one launch)",
                                "pair": "csi::fused::k_pair (two sub-steps: 2 x [stress + u + v] in one launch)", "stress": "csi::fast::k_stress",
                                "ustep": "csi::fast::k_ustep", "vstep": "csi::fast::k_vstep"}[dom] if args.mode == "fast" else dom,
                     "algorithmic_bytes_per_launch": cells_launch * algo,
                     "kernel_minimum_bytes_per_launch": cells_launch * (120.0 if path["fused"] else algo),
                     "frac_of_kernel_minimum": cells_launch * (120.0 if path["fused"] else algo) / (phases[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "avg_launch_ms": phases[dom], "substeps_per_launch": spl,
                     "note": "fused kernels are co-limited by FP64 VALU issue (DESIGN.md section 3); frac is the HBM figure the contract asks for",
                     "all_phases_ms": phases,
                     "substep_frac": cells_launch * 256.0 / (sub_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
    }
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline()
    model = None                                   # contexts (and their RCCL communicators) go before the line is printed
    import gc
    gc.collect()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctypes.CDLL(None).fflush(None)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
