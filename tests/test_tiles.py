"""Multi-GPU tiling logic on CPU: the library's index ranges and exchange plan (pure host functions of
libcsi_hip.so) drive a tiled run of the oracle; owned cells must be BITWISE equal to the single-domain run
(stronger than the reference's `isapprox`, test/distributed_tests_utils.jl:83-86).  world_size-2 runs use
torch.distributed (gloo) between two processes; single-process self-exchange covers the FIFO matching."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases
import climaseaice_jl_amd as csi
import tiled

FIELDS = ("u", "v", "s11", "s22", "s12")
LOC = tiled._LOC


def _global_reference(c):
    p = cases.oracle_problem(c)
    p.time_step_momentum(c["dt"])
    return p


def _check_tile(tg, p, ref, c):
    for k in FIELDS:
        LX, LY = LOC[k]
        nx, ny = tg.Nx, tg.Ny
        got = p.interior(k)[:ny, :nx]
        want = ref.interior(k)[tg.j_off:tg.j_off + ny, tg.i_off:tg.i_off + nx]
        assert np.array_equal(got, want), (k, tg.rank, np.abs(got - want).max())


def test_plan_ranges_match_reference_kernel_parameters():
    L = csi._lib
    # serial grid: stress range -H+2 : N+H-1 (evp:145), velocity kernels :xy
    assert csi.plan_ranges(48, 8, 4, 4, L.PERIODIC, L.BOUNDED) == ((-2, 51, -2, 11), (1, 48, 1, 8), (1, 48, 1, 8), (1, 48, 1, 8))
    # tile: ring 1 on connected sides, first velocity extended by the ring the second one reads
    rs, ru, rv, own = csi.plan_ranges(32, 16, 4, 4, L.FULLY_CONNECTED, L.RIGHT_CONNECTED)
    assert rs == (0, 33, -2, 17) and ru == (1, 33, 1, 16) and rv == (0, 32, 1, 17) and own == (1, 32, 1, 16)
    # exchange every 2 sub-steps (valid width 4 at the start of the batch): one ring more everywhere
    rs, ru, rv, sec = csi.plan_ranges(32, 16, 4, 4, L.FULLY_CONNECTED, L.FULLY_CONNECTED, 4)
    assert rs == (-2, 35, -2, 19) and ru == (-1, 35, -2, 18) and rv == (-2, 34, -1, 19) and sec == (-1, 34, -1, 18)


@pytest.mark.parametrize("k", [1, 2])
def test_self_connected_single_process_bitwise(k):
    """One tile, both periodic directions forced to exchange with itself (every neighbour is this rank, two
    messages per peer pair in each direction: exercises the FIFO ordering of the plan)."""
    c = cases.make_case(Nx=40, Ny=32, substeps=9, topo=("periodic", "periodic"), random_uv=0.05)
    ref = _global_reference(c)
    tg, p = tiled.tile_problem(c, 1, 1, 0, force_connected=True)
    tiled.tiled_time_step_momentum(tg, p, c["dt"], tiled.Exchanger(tg, None), k=k)
    _check_tile(tg, p, ref, c)


def _worker(rank, world, Rx, Ry, case_kw, port, q, k=1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        c = cases.make_case(**case_kw)
        tg, p = tiled.tile_problem(c, Rx, Ry, rank)
        tiled.tiled_time_step_momentum(tg, p, c["dt"], tiled.Exchanger(tg, dist), k=k)
        out = {k: p.interior(k)[:tg.Ny, :tg.Nx].copy() for k in FIELDS}
        q.put((rank, tg.i_off, tg.j_off, out))
        dist.barrier()
    finally:
        dist.destroy_process_group()


PARTITIONS = [
    ("x2_periodic", 2, 1, dict(Nx=48, Ny=32, topo=("periodic", "periodic"))),
    ("y2_periodic", 1, 2, dict(Nx=40, Ny=48, topo=("periodic", "periodic"))),
    ("y2_bounded", 1, 2, dict(Nx=40, Ny=48, topo=("bounded", "bounded"))),
    ("x2_channel_latlon", 2, 1, dict(Nx=48, Ny=32, topo=("periodic", "bounded"), grid="latlon")),
    # BetaPlane: each tile evaluates f = f0 + beta * y on its own rows (halo rows: the neighbour's / the wrapped row's)
    ("y2_beta_bounded", 1, 2, dict(Nx=40, Ny=48, topo=("bounded", "bounded"), beta=2e-10)),
    ("y2_beta_periodic", 1, 2, dict(Nx=40, Ny=48, topo=("periodic", "periodic"), beta=2e-10)),
    # orthogonal curvilinear grid: each tile gets its slice of the twelve 2-D metric arrays
    ("x2_curvilinear", 2, 1, dict(Nx=48, Ny=32, topo=("periodic", "bounded"), curvilinear=0.04)),
    # TripolarGrid-like: north fold (Zipper), partitioned in y only as the reference's own distributed tripolar test
    # (test/distributed_tests_utils.jl:239: Partition(1, 4)); the fold lives on the northernmost tile
    # model.forcing arrays: their halos beyond connected sides come from the neighbour (read on the extended ranges of k > 1)
    ("x2_user_forcing", 2, 1, dict(Nx=48, Ny=32, topo=("periodic", "bounded"), user_forcing=True)),
    ("y2_user_forcing", 1, 2, dict(Nx=40, Ny=48, topo=("periodic", "periodic"), user_forcing=True)),
    ("y2_folded", 1, 2, dict(Nx=40, Ny=48, topo=("periodic", "folded"))),
    ("y2_folded_curvilinear", 1, 2, dict(Nx=40, Ny=48, topo=("periodic", "folded"), curvilinear=0.04)),
    # round 6: the real tripolar geometry (csi.TripolarGrid: lat-lon rows + conformal bipolar cap), its analytic land and per-point f
    ("y2_tripolar_grid", 1, 2, dict(Nx=48, Ny=56, grid="tripolar", tripolar=dict(southernmost_latitude=-70.0), coriolis_points=True)),
]


@pytest.mark.parametrize("k", [1, 2])
@pytest.mark.parametrize("name,Rx,Ry,kw", PARTITIONS, ids=[p[0] for p in PARTITIONS])
def test_two_process_gloo_tiles_bitwise(name, Rx, Ry, kw, k):
    """world_size = 2 (gloo): tiled == single domain, bit for bit, on the owned cells of both tiles."""
    case_kw = dict(substeps=9, random_uv=0.05, patches=True, **kw)
    c = cases.make_case(**case_kw)
    ref = _global_reference(c)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500) + 2 * PARTITIONS.index((name, Rx, Ry, kw)) + k
    procs = [ctx.Process(target=_worker, args=(r, 2, Rx, Ry, case_kw, port, q, k)) for r in range(2)]
    for pr in procs:
        pr.start()
    results = [q.get(timeout=180) for _ in range(2)]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    for rank, i_off, j_off, out in results:
        for k in FIELDS:
            ny, nx = out[k].shape
            want = ref.interior(k)[j_off:j_off + ny, i_off:i_off + nx]
            assert np.array_equal(out[k], want), (name, rank, k, np.abs(out[k] - want).max())
