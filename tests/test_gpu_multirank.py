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
