#!/usr/bin/env python3
"""bench.py -- EVP sub-cycle throughput on MI355X (BASELINE.json metric).

One "step" = one time_step_momentum! call (initialize_rheology! + `substeps` EVP sub-steps +
finalize_rheology!) over the whole grid, inputs resident in HBM.  value = cell-updates/s =
Nx * Ny * substeps * steps / time (whole job, all ranks).  Prints ONE JSON line (rank 0).
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

ALGO_BYTES_PER_CELL_UPDATE = 256.0     # stress 96 B + u-step 80 B + v-step 80 B
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_baseline(case_kw, seconds_budget=20.0):
    """The oracle (C restatement, OpenMP over rows) timed on the host cores: a reported baseline."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cases
    n = 2048
    kw = dict(case_kw)
    kw.update(Nx=n, Ny=n, substeps=2)
    c = cases.make_case(**kw)
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
    return {"value": n * n * c["substeps"] * reps / dt, "unit": "cell-updates/s", "cores": cores, "kind": "port",
            "sample": f"{reps} x time_step_momentum! of {n}x{n} periodic f-plane, {c['substeps']} sub-steps, "
                      f"C oracle with OpenMP rows ({cores} threads)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=2048)
    ap.add_argument("--substeps", type=int, default=120)
    ap.add_argument("--mode", default="fast")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import cases
    import climaseaice_jl_amd as csi

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    torch.cuda.set_device(local_rank)

    # Weak scaling: every rank advances its own N x N periodic f-plane tile (no data-path
    # collective yet: the tiled RCCL halo exchange is the next step; stated in DESIGN.md).
    N = args.size
    case_kw = dict(topo=("periodic", "periodic"), patches=True, random_uv=0.02, noise=0.05, seed=3 + rank)
    c = cases.make_case(Nx=N, Ny=N, substeps=args.substeps, **case_kw)
    model = cases.csi_model(c, mode=args.mode, device=f"cuda:{local_rank}")

    def step():
        csi.time_step_momentum(model, c["dt"])

    def barrier():
        model.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    kern_ms = 0.0
    for _ in range(args.steps):
        step()
        # HIP-event time of the sub-step loop on the library's stream (no host sync inside the loop:
        # the events are read after the barrier below)
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms = model.ctx.last_subcycle_ms()             # last step's sub-cycle, HIP events on the launch stream
    if world > 1:
        t = torch.tensor([elapsed], device=f"cuda:{local_rank}", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    cells = N * N * world
    updates = cells * args.substeps * args.steps
    value = updates / elapsed
    launches = model.ctx.launches_per_substep()
    # dominant kernel family = the three sub-step phases; per-launch algorithmic bytes:
    per_substep_ms = kern_ms / args.substeps
    achieved = N * N * ALGO_BYTES_PER_CELL_UPDATE / (per_substep_ms * 1e-3) / 1e9
    out = {
        "metric": "EVP sub-cycle cell-updates/s", "value": value, "unit": "cell-updates/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"evp_subcycle_{N}x{N}_fplane_periodic_{args.substeps}substeps_per_gpu",
                   "grid": [N, N], "substeps": args.substeps, "mode": args.mode, "parallelism": f"replicated_tiles_x{world}"},
        "model_days_per_hr_momentum_only": 3600.0 / (elapsed / args.steps * 3 * 720) if elapsed > 0 else None,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": "sub-step = k_stress + k_ustep + k_vstep", "launches_per_substep": launches,
                     "ms_per_substep": per_substep_ms, "algorithmic_bytes_per_cell_update": ALGO_BYTES_PER_CELL_UPDATE},
    }
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(case_kw)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
