"""The peer halo transport across a real PROCESS boundary, on the one GPU a development box has: two (or four) processes, one tile
each, joined by a host-channel group (csi_comm_init_host: shared memory + HIP IPC; RCCL refuses two ranks on one device).  What
runs here is the multi-process path of csrc/csi_abi.hip peer_setup that one process per GPU takes on a real node -- IPC handles of
caller-owned (torch) and library-owned arrays and of the fine-grained flag words, exchanged and opened with hipIpcOpenMemHandle,
halo images and flags written into ANOTHER process's memory by the pair kernel -- none of which the in-process tile group
(tests/test_gpu_local_tiles.py: one address space, plain pointers) or a tile connected to itself can exercise.
The reference's distributed tests start ranks and compare them with the serial run the same way
(/root/reference/test/distributed_tests_utils.jl:40-88; test/test_distributed_sea_ice.jl:41-54).
Still NOT covered: two DEVICES (peer access over xGMI) -- tests/test_gpu_multirank.py, skipped on one-GPU boxes."""
import multiprocessing as mp
import os
import uuid

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIELDS = ("u", "v", "s11", "s22", "s12")


def _get(m, f):
    import climaseaice_jl_amd as csi  # noqa: F401
    return (getattr(m.velocities, f) if f in ("u", "v") else getattr(m.dynamics.auxiliaries.fields, f)).interior_numpy().copy()


def _rank(conn, name, kw, Rx, Ry, rank, transport, cycles, tier, device="cuda:0"):
    """tier: the protocol tier asked for (-1: the library's automatic choice, which must come out as 1 -- the neighbour lives in
    another process).  device: tests/test_gpu_multirank.py runs the same ranks on one DEVICE each."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        import cases
        import climaseaice_jl_amd as csi
        if device != "cuda:0":
            import torch
            torch.cuda.set_device(int(device.split(":")[1]))
        c = cases.make_case(**kw)
        m = cases.csi_model(c, mode="fast", tile=(Rx, Ry, rank), host_group=name, device=device)
        m.set_halo_transport(transport)
        m.set_peer_tier(tier)
        for _ in range(cycles):
            csi.time_step_momentum(m, c["dt"])
        m.synchronize()
        out = {f: _get(m, f) for f in FIELDS}
        out["transport"] = m.ctx.halo_transport()
        out["tier"] = m.ctx.peer_tier()
        out["path"] = m.ctx.last_path()
        out["ranks"] = m.ctx.comm_count()
        g = m.grid
        out["offsets"] = (g.i_off, g.j_off, g.Nx, g.Ny)
        conn.send(out)
    except Exception as e:      # noqa: BLE001  (reported to the parent, which fails the test)
        conn.send({"error": repr(e)})


HOST_CASES = {
    "2x1_periodic": (dict(Nx=280, Ny=96, topo=("periodic", "periodic")), 2, 1),
    "1x2_periodic": (dict(Nx=160, Ny=144, topo=("periodic", "periodic")), 1, 2),
    "2x1_channel_land_arrays": (dict(Nx=300, Ny=100, topo=("periodic", "bounded"), land=0.2, field_forcing=True), 2, 1),
    "1x2_bounded_noslip": (dict(Nx=150, Ny=128, topo=("bounded", "bounded"), noslip=True), 1, 2),
    "2x2_periodic": (dict(Nx=272, Ny=128, topo=("periodic", "periodic")), 2, 2),
}


@pytest.mark.parametrize("transport,nsub,tier", [("peer", 12, 0), ("peer", 7, -1), ("peer", 12, 1), ("peer", 12, 2), ("rccl", 12, -1)])
@pytest.mark.parametrize("name", sorted(HOST_CASES))
def test_processes_on_one_gpu_tiled_equals_untiled_bitwise(name, transport, nsub, tier):
    import cases
    import climaseaice_jl_amd as csi
    kw, Rx, Ry = HOST_CASES[name]
    kw = dict(kw, substeps=nsub, patches=True, random_uv=0.05, H=8 if transport == "rccl" else 4)
    cycles = 2
    c = cases.make_case(**kw)
    ref = cases.csi_model(c, mode="fast")
    for _ in range(cycles):
        csi.time_step_momentum(ref, c["dt"])
    ref.synchronize()
    want = {f: _get(ref, f) for f in FIELDS}
    world = Rx * Ry
    shm = f"/csi-test-{uuid.uuid4().hex[:12]}"
    ctx = mp.get_context("spawn")
    procs, pipes = [], []
    for r in range(world):
        a, b = ctx.Pipe()
        p = ctx.Process(target=_rank, args=(b, shm, kw, Rx, Ry, r, transport, cycles, tier))
        p.start()
        procs.append(p); pipes.append(a)
    got = []
    for r in range(world):
        assert pipes[r].poll(300), f"rank {r} did not answer"
        got.append(pipes[r].recv())
    for p in procs:
        p.join(timeout=60)
    for r in range(world):
        assert "error" not in got[r], (r, got[r].get("error"))
        assert got[r]["ranks"] == world
        # "peer": every rank mapped its neighbours' arrays over HIP IPC and ran the flag protocol; one host exchange per sub-cycle
        assert got[r]["transport"] == transport, (r, got[r]["transport"], got[r]["path"])
        # the automatic tier: 1 as soon as a neighbour lives in another process (the fence-free tier 0 is an explicit opt-in there)
        assert got[r]["tier"] == (1 if tier < 0 else tier), (r, got[r]["tier"])
        i0, j0, nx, ny = got[r]["offsets"]
        for f in FIELDS:
            mine, w = got[r][f][:ny, :nx], want[f][j0:j0 + ny, i0:i0 + nx]
            assert np.array_equal(mine, w), (name, "rank", r, f, np.abs(mine - w).max(), np.argwhere(mine != w)[:4].tolist())


@pytest.mark.parametrize("n,partition", [(2, ""), (4, ""), (4, "2x2")])
def test_bench_rehearsal_of_the_multi_rank_path_on_one_gpu(n, partition):
    """bench.py --gpus N --rehearse-on-one-gpu: the WHOLE N > 1 path of the benchmark the driver runs on an 8-GPU node -- parent,
    torch.distributed.run launch, one rank per process, peer set-up over HIP IPC, the tier ladder, the tiled == untiled check before
    anything is timed, the timed region with its barriers, the one-GPU rate of the same grid, the single JSON line -- with the
    ranks sharing GPU 0 over the host-channel group.  Its numbers are not scaling results (the line says so); what is asserted is
    that every step of that path runs and that the bitwise check passed on the peer transport at the library's automatic tier."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--rehearse-on-one-gpu", "--size", "512", "--substeps", "24",
           "--steps", "2", "--warmup", "1", "--no-full-step"] + (["--partition", partition] if partition else [])
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=root)
    assert p.returncode == 0, (p.returncode, p.stderr[-3000:])
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln][-1]
    d = json.loads(line)
    assert d["n_gpus"] == n and d["rehearsal_on_one_gpu"] is True and d["rccl_ranks"] == n
    assert d["tiled_equals_untiled_bitwise"] is True
    # the ladder starts at the library's automatic tier: 1 across processes; the fence-free tier 0 is timed beside it as an opt-in
    assert d["path"]["halo_transport"] == "peer" and d["path"]["peer_tier"] == 1, d["path"]
    assert d["path"]["peer_tier_ladder"] == [{"tier": 1, "passed": True, "problem": None}]
    assert d["peer_tier0"]["bitwise"] is True and d["peer_tier0"]["value"] > 0, d.get("peer_tier0")
    # both decompositions of the grid in one run: the slabs of the headline and the 2 x (N / 2) layout BASELINE config 4 names
    other = [k for k in d if k.startswith("partition_") and k != "partition_note"]
    assert len(other) == 1 and d[other[0]]["bitwise"] is True and d[other[0]]["value"] > 0, {k: d[k] for k in other}
    # the dominant kernel's launch time comes from the timed region: launches x average <= ms_per_step
    assert d["roofline"]["launches_x_avg_ms"] <= d["ms_per_step"] * 1.0001, (d["roofline"]["launches_x_avg_ms"], d["ms_per_step"])
    assert d["config"]["partition"] == ([int(t) for t in partition.split("x")] if partition else [1, n])
    assert d["single_gpu"]["value"] > 0 and d["parallel_efficiency"] > 0 and d["value"] > 0
