"""Tiled (multi-process) runs of the EVP sub-cycle with the CPU oracle as the per-tile compute and an
explicit halo exchange driven by THE LIBRARY'S OWN exchange plan and index ranges (csi_plan_exchange,
csi_plan_ranges: pure host functions of libcsi_hip.so).  The GPU path runs the same plan over RCCL;
here the transport is torch.distributed (gloo) so the N > 1 logic is covered on CPU."""
import numpy as np

import cases
import climaseaice_jl_amd as csi
import oracle as O

_TOPO_CODE = {csi.Periodic: O.PERIODIC, csi.Bounded: O.BOUNDED, csi.FullyConnected: O.FULLY_CONNECTED,
              csi.LeftConnected: O.LEFT_CONNECTED, csi.RightConnected: O.RIGHT_CONNECTED,
              csi.RightFolded: O.RIGHT_FOLDED, csi.LeftConnectedRightFolded: O.LEFT_CONNECTED_RIGHT_FOLDED}
_LOC = {"u": (csi.Face, csi.Center), "v": (csi.Center, csi.Face), "h": (csi.Center, csi.Center),
        "aice": (csi.Center, csi.Center), "s11": (csi.Center, csi.Center), "s22": (csi.Center, csi.Center),
        "s12": (csi.Face, csi.Face)}


def tile_problem(case, Rx, Ry, rank, force_connected=False):
    """Oracle problem of tile `rank` of the global case (fields sliced from the global arrays)."""
    g = case["g"]
    tg = csi.TileGrid(g, Rx, Ry, rank % Rx, rank // Rx, force_connected=force_connected)
    topo = tuple(_TOPO_CODE[t] for t in tg.topology)
    m = tg.metrics()
    if m["kind"] == "uniform":
        p = O.Problem(tg.Nx, tg.Ny, tg.Hx, tg.Hy, topo, dx=m["dx"], dy=m["dy"], substeps=case["substeps"])
    elif m["kind"] == "full":
        p = O.Problem(tg.Nx, tg.Ny, tg.Hx, tg.Hy, topo, full=m, substeps=case["substeps"])
    else:
        p = O.Problem(tg.Nx, tg.Ny, tg.Hx, tg.Hy, topo, per_j=m, substeps=case["substeps"])
    from cases import coriolis_rows
    p.set_coriolis(case["coriolis"], rows=coriolis_rows(case, tg))
    if case["top"] is not None:
        p.set_stress("top", O.STRESS_CONST, tau=case["top"])
    if case["bottom"] == "semi":
        p.set_stress("bottom", O.STRESS_SEMI_IMPLICIT, ue=case["ue"] or None, ve=case["ve"] or None)
    if case.get("coriolis_points"):
        # per-point f (a TripolarGrid's 2 Omega sin(latitude)): the tile's slice of the global planes, halo entries included
        n, ni = tg.Ny + 2 * tg.Hy + 1, tg.Nx + 2 * tg.Hx + 1
        p.set_coriolis_points(*(np.ascontiguousarray(a[tg.j_off:tg.j_off + n, tg.i_off:tg.i_off + ni]) for a in case["f_points"]))
    if case.get("mask") is not None:
        # the tile's slice (halo included) of the global activity mask as cases.oracle_problem builds it
        G = g
        full = np.zeros((G.Ny + 2 * G.Hy, G.Nx + 2 * G.Hx), dtype=np.uint8)
        full[G.Hy:G.Hy + G.Ny, G.Hx:G.Hx + G.Nx] = case["mask"]
        if G.topology[0] is csi.Periodic:
            full[:, :G.Hx] = full[:, G.Nx:G.Nx + G.Hx]
            full[:, G.Nx + G.Hx:] = full[:, G.Hx:2 * G.Hx]
        if G.topology[1] is csi.Periodic:
            full[:G.Hy, :] = full[G.Ny:G.Ny + G.Hy, :]
            full[G.Ny + G.Hy:, :] = full[G.Hy:2 * G.Hy, :]
        if G.topology[1] is csi.RightFolded:
            full = csi.fold_north(full, G.Nx, G.Ny, G.Hx, G.Hy, False, False, 1).astype(np.uint8)
        p.set_mask(np.ascontiguousarray(full[tg.j_off:tg.j_off + tg.Ny + 2 * tg.Hy, tg.i_off:tg.i_off + tg.Nx + 2 * tg.Hx]))
    for name, key in (("h", "h"), ("aice", "a"), ("u", "u"), ("v", "v")):
        p.interior(name)[...] = tg.local_interior(case[key], *_LOC[name])
    if case.get("user_forcing"):
        # model.forcing.u / .v as arrays: the tile's slice; halos by the local boundary conditions here, beyond connected
        # sides by the exchange in tiled_time_step_momentum (what csi_abi.hip do_time_step_momentum does)
        fu = cases._fill_parent_like(p, "u", tg.local_interior(case["force_u"], csi.Face, csi.Center))
        fv = cases._fill_parent_like(p, "v", tg.local_interior(case["force_v"], csi.Center, csi.Face))
        for arr, (lx, ly) in ((fu, (O.FACE, O.CENTER)), (fv, (O.CENTER, O.FACE))):
            p.L.ora_fill_halo_loc(p.ptr, O.Field(arr.ctypes.data_as(O.C.POINTER(O.C.c_double)), arr.shape[1]), lx, ly, -1)
        p.set_forcing(fu, fv)
        p.f["forcing_u"], p.f["forcing_v"] = fu, fv
    return tg, p


class Exchanger:
    """Halo exchange of oracle fields through torch.distributed, following csi_plan_exchange."""

    def __init__(self, tg, dist=None):
        self.tg, self.dist = tg, dist
        L = csi._lib
        code = {csi.Periodic: L.PERIODIC, csi.Bounded: L.BOUNDED, csi.FullyConnected: L.FULLY_CONNECTED,
                csi.LeftConnected: L.LEFT_CONNECTED, csi.RightConnected: L.RIGHT_CONNECTED,
                csi.RightFolded: L.RIGHT_FOLDED, csi.LeftConnectedRightFolded: L.LEFT_CONNECTED_RIGHT_FOLDED}
        self.topo = (code[tg.topology[0]], code[tg.topology[1]])

    def plan(self, W, halo):
        t = self.tg
        return csi.plan_exchange(t.Nx, t.Ny, t.Hx, t.Hy, self.topo[0], self.topo[1], t.rx, t.ry, t.Rx, t.Ry,
                                 t.periodic[0], t.periodic[1], W, halo)

    def __call__(self, p, names, W):
        import torch
        t = self.tg
        sp, rp = self.plan(W, 0), self.plan(W, 1)

        def view(a, i0, j0, ni, nj):
            return a[j0 + t.Hy - 1:j0 + t.Hy - 1 + nj, i0 + t.Hx - 1:i0 + t.Hx - 1 + ni]

        sends = []
        for (peer, i0, j0, ni, nj) in sp:
            if peer >= 0:
                sends.append((peer, torch.from_numpy(np.concatenate([view(p.f[n], i0, j0, ni, nj).ravel() for n in names]))))
        recvs = [(peer, i0, j0, ni, nj, torch.empty(ni * nj * len(names), dtype=torch.float64))
                 for (peer, i0, j0, ni, nj) in rp if peer >= 0]
        me = t.rank
        if self.dist is None:
            # one process: every neighbour is this tile itself; FIFO matching per peer
            assert all(peer == me for peer, _ in sends)
            for (peer, buf), r in zip(sends, recvs):
                r[5].copy_(buf)
        else:
            reqs = []
            # messages between one pair of ranks match in FIFO order: post them in plan order
            for peer, buf in sends:
                if peer == me:
                    continue
                reqs.append(self.dist.isend(buf, dst=peer))
            self_msgs = [buf for peer, buf in sends if peer == me]
            k = 0
            for r in recvs:
                if r[0] == me:
                    r[5].copy_(self_msgs[k]); k += 1
                else:
                    reqs.append(self.dist.irecv(r[5], src=r[0]))
            for q in reqs:
                q.wait()
        for (peer, i0, j0, ni, nj, buf) in recvs:
            b = buf.numpy()
            for k, n in enumerate(names):
                view(p.f[n], i0, j0, ni, nj)[...] = b[k * ni * nj:(k + 1) * ni * nj].reshape(nj, ni)


def tiled_time_step_momentum(tg, p, dt, ex, k=1):
    """time_step_momentum! on one tile: the launch loop of csi_abi.hip (do_time_step_momentum / do_subcycle)
    restated with the oracle's kernels, the library's ranges and an explicit exchange of width 2k every k
    sub-steps."""
    Hmin = min(tg.Hx, tg.Hy)
    W = 2 * k
    assert W <= Hmin
    p.update_state()
    ex(p, ["h", "aice", "u", "v"], Hmin)
    p.initialize_rheology()
    if "forcing_u" in p.f:
        ex(p, ["forcing_u", "forcing_v"], Hmin)
    p.L.ora_fill_halo_u(p.ptr); p.L.ora_fill_halo_v(p.ptr)
    xf = ["u", "v"] if k == 1 else ["u", "v", "s11", "s22", "s12"]   # sigma is history dependent (csi_abi.hip)
    ex(p, xf, W)
    m = 0
    nsub = p.s.substeps
    for s in range(1, nsub + 1):
        V = W - 2 * m
        rs, ru1, rv1, r2 = csi.plan_ranges(tg.Nx, tg.Ny, tg.Hx, tg.Hy, ex.topo[0], ex.topo[1], V)
        p.L.ora_compute_viscosities(p.ptr, *rs)
        p.L.ora_compute_stresses(p.ptr, dt, *rs)
        if s % 2 == 0:
            p.L.ora_u_velocity_step(p.ptr, dt, *ru1); p.L.ora_fill_halo_u(p.ptr)
            p.L.ora_v_velocity_step(p.ptr, dt, *r2); p.L.ora_fill_halo_v(p.ptr)
        else:
            p.L.ora_v_velocity_step(p.ptr, dt, *rv1); p.L.ora_fill_halo_v(p.ptr)
            p.L.ora_u_velocity_step(p.ptr, dt, *r2); p.L.ora_fill_halo_u(p.ptr)
        m += 1
        if m == k or s == nsub:
            ex(p, xf, W)
            m = 0
    p.L.ora_finalize_rheology(p.ptr)
    ex(p, ["s11", "s12", "s22"], Hmin)
