"""One rank of a real multi-GPU tiled run (started by tests/test_gpu_multirank.py, one process per GPU, RCCL):
time_step_momentum! and one whole RK3 time_step! on this rank's tile; the owned cells go to <out>.rank<r>.npz."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    rank, world, port, Rx, Ry, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    kw = json.loads(sys.argv[7])
    kw["topo"] = tuple(kw["topo"])
    k = int(sys.argv[8])
    extra = sys.argv[9:]
    tier = int(extra[extra.index("--tier") + 1]) if "--tier" in extra else None
    cycles = int(extra[extra.index("--cycles") + 1]) if "--cycles" in extra else 1
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import numpy as np
    import torch
    import torch.distributed as dist
    import cases
    import climaseaice_jl_amd as csi
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{rank}"))
    try:
        c = cases.make_case(**kw)
        m = cases.csi_model(c, mode="fast", timestepper="SplitRungeKutta3", advection=csi.WENO(order=7),
                            device=f"cuda:{rank}", tile=(Rx, Ry, rank))
        m.set_exchange_interval(max(k, 0))        # k = -1: automatic interval on the RCCL exchange; k = 0: the peer transport
        if k < 0:
            m.set_halo_transport("rccl")
        if tier is not None:
            m.set_peer_tier(tier)                 # the protocol tier, forced (tests/test_gpu_multirank.py: the ladder across devices)
        for _ in range(cycles):
            csi.time_step_momentum(m, c["dt"])
        m.synchronize()
        from test_gpu_evp import EVP_FIELDS
        res = {f"mom_{f}": EVP_FIELDS[f](m).interior_numpy().copy() for f in ("u", "v", "s11", "s22", "s12")}
        res["path"] = np.array(json.dumps(dict(m.ctx.last_path(), transport=m.ctx.halo_transport(), tier=m.ctx.peer_tier())))
        g = m.grid
        res["offsets"] = np.array([g.i_off, g.j_off, g.Nx, g.Ny])
        if tier is not None:                      # the ladder test compares the sub-cycles only
            np.savez(f"{out}.rank{rank}.npz", **res)
            dist.barrier()
            return
        csi.time_step(m, c["dt"])
        m.synchronize()
        res.update({f"step_{f}": EVP_FIELDS[f](m).interior_numpy().copy() for f in ("u", "v")})
        res["step_h"] = m.ice_thickness.interior_numpy().copy()
        res["step_a"] = m.ice_concentration.interior_numpy().copy()
        g = m.grid
        res["offsets"] = np.array([g.i_off, g.j_off, g.Nx, g.Ny])
        np.savez(f"{out}.rank{rank}.npz", **res)
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
