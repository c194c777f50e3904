"""Real multi-rank RCCL runs (one process per GPU): tiled == untiled, bit for bit, on the owned cells of every tile.
k = 1, 2: RCCL exchange every k sub-steps; k = -1: its automatic interval; k = 0: the default, peer-direct halo writes.
Needs >= 2 visible GPUs; the gpurun boxes have one, so on those this module skips and the N > 1 logic is covered by the
self-connected tests (test_gpu_evp.py: RCCL and peer transport with the tile as its own neighbour) and the two-process gloo
tests (test_tiles.py)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import cases
import climaseaice_jl_amd as csi
from test_gpu_evp import EVP_FIELDS

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
NGPU = torch.cuda.device_count()          # (counting devices does not initialise HIP)


def _run(world, Rx, Ry, kw, k, tmp_path):
    port = str(29600 + os.getpid() % 300 + world)
    out = str(tmp_path / "tiles")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "multirank_worker.py"), str(r), str(world), port, str(Rx), str(Ry),
                               out, json.dumps(kw), str(k)], env=env) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    return [np.load(f"{out}.rank{r}.npz") for r in range(world)]


@pytest.mark.skipif(NGPU < 2, reason="needs at least 2 GPUs (one rank per GPU)")
@pytest.mark.parametrize("k", [1, 2, -1, 0])
@pytest.mark.parametrize("Rx,Ry", [(2, 1), (1, 2)] + ([(2, 2)] if NGPU >= 4 else []) + ([(2, 4)] if NGPU >= 8 else []))
def test_multirank_rccl_tiles_bitwise(Rx, Ry, k, tmp_path):
    kw = dict(Nx=256, Ny=192, H=8, substeps=14, topo=("periodic", "bounded"), patches=True, random_uv=0.05)
    if Rx == 1:      # pure y partitions also run the TripolarGrid-like case: the north fold lives on the last rank
        kw = dict(kw, topo=("periodic", "folded"))
    c = cases.make_case(**kw)
    ref = cases.csi_model(c, mode="fast", timestepper="SplitRungeKutta3", advection=csi.WENO(order=7))
    csi.time_step_momentum(ref, c["dt"])
    ref.synchronize()
    mom = {f: EVP_FIELDS[f](ref).interior_numpy().copy() for f in ("u", "v", "s11", "s22", "s12")}
    csi.time_step(ref, c["dt"])
    ref.synchronize()
    step = {"u": ref.velocities.u.interior_numpy(), "v": ref.velocities.v.interior_numpy(),
            "h": ref.ice_thickness.interior_numpy(), "a": ref.ice_concentration.interior_numpy()}
    for d in _run(Rx * Ry, Rx, Ry, kw, k, tmp_path):
        i0, j0, nx, ny = (int(x) for x in d["offsets"])
        path = json.loads(str(d["path"]))
        # k = 0: peer-direct halo writes over xGMI (IPC-mapped neighbours, flags) -- the fold tile of a y partition included
        assert path["transport"] == ("peer" if k == 0 else "rccl"), path
        for f, want in mom.items():
            got = d[f"mom_{f}"][:ny, :nx]
            assert np.array_equal(got, want[j0:j0 + ny, i0:i0 + nx]), (f, Rx, Ry, k)
        for f, want in step.items():
            got = d[f"step_{f}"][:ny, :nx]
            assert np.array_equal(got, want[j0:j0 + ny, i0:i0 + nx]), ("step", f, Rx, Ry, k)


# ---- first contact with two DEVICES, localised (round 5): when this module first runs on a node with >= 2 GPUs, the tests above say
# WHETHER tiles reproduce the untiled run; these say WHERE it breaks if they do not -- IPC mapping between devices (host-channel
# group: no RCCL anywhere), memory ordering of the flag protocol (each tier forced), RCCL itself (the k >= 1 cases above) ----------

@pytest.mark.skipif(NGPU < 2, reason="needs at least 2 GPUs (one rank per GPU)")
@pytest.mark.parametrize("tier", [-1, 0, 1, 2])
@pytest.mark.parametrize("Rx,Ry", [(2, 1), (1, 2)])
def test_multirank_peer_tier_ladder_bitwise(Rx, Ry, tier, tmp_path):
    """One rank per GPU over RCCL, peer transport, every protocol tier forced in turn (-1: the library's choice, which must be 1
    across devices).  A failure of tier 0 alone is the memory-ordering argument of evp_fused2.hip not holding between two L2
    domains; a failure of every tier is the IPC mapping or the flag addressing."""
    kw = dict(Nx=256, Ny=192, H=4, substeps=14, topo=("periodic", "bounded"), patches=True, random_uv=0.05)
    c = cases.make_case(**kw)
    ref = cases.csi_model(c, mode="fast")
    for _ in range(3):
        csi.time_step_momentum(ref, c["dt"])
    ref.synchronize()
    want = {f: EVP_FIELDS[f](ref).interior_numpy().copy() for f in ("u", "v", "s11", "s22", "s12")}
    port = str(29900 + os.getpid() % 90 + 2 * (tier + 1) + Rx)
    out = str(tmp_path / "tier")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    world = Rx * Ry
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "multirank_worker.py"), str(r), str(world), port, str(Rx), str(Ry),
                               out, json.dumps(kw), "0", "--tier", str(tier), "--cycles", "3"], env=env) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    for r in range(world):
        d = np.load(f"{out}.rank{r}.npz")
        path = json.loads(str(d["path"]))
        assert path["transport"] == "peer" and path["tier"] == (1 if tier < 0 else tier), path
        i0, j0, nx, ny = (int(x) for x in d["offsets"])
        for f, w in want.items():
            assert np.array_equal(d[f"mom_{f}"][:ny, :nx], w[j0:j0 + ny, i0:i0 + nx]), (f, Rx, Ry, tier)


@pytest.mark.skipif(NGPU < 2, reason="needs at least 2 GPUs (one rank per GPU)")
@pytest.mark.parametrize("transport,tier", [("peer", -1), ("peer", 0), ("rccl", -1)])
@pytest.mark.parametrize("name", ["2x1_periodic", "1x2_periodic", "1x2_bounded_noslip"])
def test_host_channel_group_across_two_devices_bitwise(name, transport, tier):
    """The host-channel group (shared memory + HIP IPC, NO RCCL communicator) with one rank per DEVICE: hipIpcOpenMemHandle of
    another device's allocations, write-through stores and flags over xGMI -- the peer transport's own machinery with RCCL taken
    out of the picture ("rccl" here = the group's device-to-device copies out of the IPC-mapped send buffers)."""
    import multiprocessing as mp
    import uuid
    import test_gpu_hostgroup as hg
    kw, Rx, Ry = hg.HOST_CASES[name]
    kw = dict(kw, substeps=12, patches=True, random_uv=0.05, H=8 if transport == "rccl" else 4)
    c = cases.make_case(**kw)
    ref = cases.csi_model(c, mode="fast")
    for _ in range(2):
        csi.time_step_momentum(ref, c["dt"])
    ref.synchronize()
    want = {f: hg._get(ref, f) for f in hg.FIELDS}
    world = Rx * Ry
    shm = f"/csi-test-{uuid.uuid4().hex[:12]}"
    ctx = mp.get_context("spawn")
    procs, pipes = [], []
    for r in range(world):
        a, b = ctx.Pipe()
        p = ctx.Process(target=hg._rank, args=(b, shm, kw, Rx, Ry, r, transport, 2, tier, f"cuda:{r % NGPU}"))
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
        assert got[r]["transport"] == transport and got[r]["tier"] == (1 if tier < 0 else tier), (r, got[r]["transport"], got[r]["tier"])
        i0, j0, nx, ny = got[r]["offsets"]
        for f in hg.FIELDS:
            assert np.array_equal(got[r][f][:ny, :nx], want[f][j0:j0 + ny, i0:i0 + nx]), (name, "rank", r, f)
