"""Real decompositions on ONE GPU: every tile is its own context and host thread, the halos travel through the in-process tile
group (csi_local_group: device-to-device copies with RCCL's matching rule instead of ncclSend / ncclRecv, include/csi.h) or --
k = 0 -- through the peer transport, whose flags and image stores then connect DISTINCT tiles.  Same plans, same pack / unpack
kernels, same launch loops as a multi-process RCCL run (tests/test_gpu_multirank.py, which needs one GPU per rank); tiled ==
untiled bit for bit on the owned cells of every tile, for the sub-cycle and for one whole RK3 time step."""
import threading

import numpy as np
import pytest
import torch

import cases
import climaseaice_jl_amd as csi
from test_gpu_evp import EVP_FIELDS

pytestmark = pytest.mark.gpu


def run_tile_threads(world, tile_fn):
    """tile_fn(rank, group) -> result, one host thread per tile of a csi.LocalGroup; the results in rank order"""
    group = csi.LocalGroup(world)
    out, errors = [None] * world, []

    def work(rank):
        try:
            out[rank] = tile_fn(rank, group)
        except BaseException as e:       # noqa: BLE001 -- reported by the main thread
            errors.append((rank, e))

    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=900)
    assert not any(t.is_alive() for t in threads), "a tile thread hangs"
    if errors:
        raise errors[0][1]
    group.close()
    return out


def run_tiles(c, Rx, Ry, k, full_step=True, mode="fast"):
    def tile(rank, group):
        m = cases.csi_model(c, mode=mode, timestepper="SplitRungeKutta3", advection=csi.WENO(order=7), tile=(Rx, Ry, rank),
                            local_group=group)
        m.set_exchange_interval(max(k, 0))        # k = -1: automatic interval of the message exchange; k = 0: the peer transport
        if k < 0:
            m.set_halo_transport("rccl")
        csi.time_step_momentum(m, c["dt"])
        csi.time_step_momentum(m, c["dt"])
        m.synchronize()
        res = {f"mom_{f}": EVP_FIELDS[f](m).interior_numpy().copy() for f in ("u", "v", "s11", "s22", "s12")}
        res["path"] = dict(m.ctx.last_path(), transport=m.ctx.halo_transport(), ranks=m.ctx.comm_count(), activity=m.tile_activity())
        if full_step:
            csi.time_step(m, c["dt"])
            m.synchronize()
            res.update({f"step_{f}": EVP_FIELDS[f](m).interior_numpy().copy() for f in ("u", "v")})
            res["step_h"] = m.ice_thickness.interior_numpy().copy()
            res["step_a"] = m.ice_concentration.interior_numpy().copy()
        g = m.grid
        res["offsets"] = (g.i_off, g.j_off, g.Nx, g.Ny)
        del m
        return res

    return run_tile_threads(Rx * Ry, tile)


def reference(c, full_step=True, mode="fast"):
    ref = cases.csi_model(c, mode=mode, timestepper="SplitRungeKutta3", advection=csi.WENO(order=7))
    csi.time_step_momentum(ref, c["dt"])
    csi.time_step_momentum(ref, c["dt"])
    ref.synchronize()
    mom = {f: EVP_FIELDS[f](ref).interior_numpy().copy() for f in ("u", "v", "s11", "s22", "s12")}
    step = {}
    if full_step:
        csi.time_step(ref, c["dt"])
        ref.synchronize()
        step = {"u": ref.velocities.u.interior_numpy().copy(), "v": ref.velocities.v.interior_numpy().copy(),
                "h": ref.ice_thickness.interior_numpy().copy(), "a": ref.ice_concentration.interior_numpy().copy()}
    return mom, step


def check(tiles, mom, step, what):
    for rank, d in enumerate(tiles):
        i0, j0, nx, ny = d["offsets"]
        for f, want in mom.items():
            got = d[f"mom_{f}"][:ny, :nx]
            w = want[j0:j0 + ny, i0:i0 + nx]
            assert np.array_equal(got, w), (what, "rank", rank, f, np.abs(got - w).max(), np.argwhere(got != w)[:4].tolist())
        for f, want in step.items():
            got = d[f"step_{f}"][:ny, :nx]
            w = want[j0:j0 + ny, i0:i0 + nx]
            assert np.array_equal(got, w), (what, "rank", rank, "step", f, np.abs(got - w).max(), np.argwhere(got != w)[:4].tolist())


DECOMPOSITIONS = {
    # name: (Rx, Ry, make_case keywords, does the peer transport apply (k = 0)?)
    "2x2_periodic": (2, 2, dict(Nx=256, Ny=192, topo=("periodic", "periodic")), True),
    "2x2_channel_land_arrays": (2, 2, dict(Nx=256, Ny=192, topo=("periodic", "bounded"), land=0.2, field_forcing=True), True),
    "4x1_periodic_x": (4, 1, dict(Nx=512, Ny=96, topo=("periodic", "bounded")), True),
    "1x4_periodic_y": (1, 4, dict(Nx=160, Ny=256, topo=("periodic", "periodic")), True),
    # a Bounded x direction partitioned in x: the easternmost tile's Face fields are one column wider -- unequal row strides
    "2x1_bounded_x": (2, 1, dict(Nx=256, Ny=96, topo=("bounded", "periodic")), True),
    "3x2_bounded_land": (3, 2, dict(Nx=384, Ny=128, topo=("bounded", "bounded"), land=0.2, field_forcing=True), True),
    "2x2_bounded_noslip": (2, 2, dict(Nx=256, Ny=128, topo=("bounded", "bounded"), noslip=True), True),
    "2x2_wind_drag_arrays": (2, 2, dict(Nx=256, Ny=192, topo=("periodic", "bounded"), wind_drag="arrays", field_forcing=True, land=0.2), True),
    "2x2_latlon": (2, 2, dict(Nx=256, Ny=192, topo=("periodic", "bounded"), grid="latlon"), True),
    # the y-partitioned north fold (the reference's distributed tripolar layout): the fold tile runs the pair kernel below its
    # three-kernel band, on either transport
    "1x2_fold": (1, 2, dict(Nx=192, Ny=192, topo=("periodic", "folded")), True),
    "1x4_fold_tripolar": (1, 4, dict(Nx=128, Ny=256, topo=("periodic", "folded"), curvilinear=0.04, land=0.2, field_forcing=True), True),
    # round 6: the REAL tripolar geometry (csi.TripolarGrid: latitude-longitude rows + conformal bipolar cap, the reference's analytic
    # land, per-point f, arrays, free drift) in the reference's own distributed layout, Partition(1, 4)
    # (test/distributed_tests_utils.jl:226-259: serial == distributed), with ice everywhere and with seasonal ice
    "1x4_tripolar_grid": (1, 4, dict(Nx=224, Ny=256, grid="tripolar", tripolar=dict(southernmost_latitude=-78.0), field_forcing=True,
                                   free_drift=True, coriolis_points=True), True),
    "1x2_tripolar_grid_seasonal": (1, 2, dict(Nx=448, Ny=512, grid="tripolar", tripolar=dict(southernmost_latitude=-78.0), field_forcing=True,
                                            free_drift=True, coriolis_points=True, ice_edge=60.0, land=0.2), True),
    # round 6: ice-free ocean / land wide enough for whole interior tiles of the peer-connected launches to go quiet (tile activity)
    "2x2_seasonal": (2, 2, dict(Nx=896, Ny=640, topo=("periodic", "periodic"), ice_free_rows=(0.15, 0.85)), True),
    "1x2_seasonal_land_arrays": (1, 2, dict(Nx=672, Ny=560, topo=("periodic", "bounded"), land=0.4, field_forcing=True, free_drift=True,
                                          ice_free_rows=(0.3, 0.7)), True),
}


@pytest.mark.parametrize("k", [0, -1, 1, 2])
@pytest.mark.parametrize("name", sorted(DECOMPOSITIONS))
def test_local_tiles_bitwise(name, k):
    Rx, Ry, kw, peer_ok = DECOMPOSITIONS[name]
    c = cases.make_case(H=8, substeps=14, patches=True, random_uv=0.05, **kw)
    mom, step = reference(c)
    tiles = run_tiles(c, Rx, Ry, k)
    for d in tiles:
        assert d["path"]["ranks"] == Rx * Ry
        assert d["path"]["transport"] == ("peer" if (k == 0 and peer_ok) else "rccl"), d["path"]
        # the fold tile too runs the two-sub-steps kernel (below its three-kernel band) whenever the exchange interval is even
        assert (d["path"]["level"] in (0, 1)) if k == 1 else (d["path"]["level"] == 2), d["path"]
    check(tiles, mom, step, (name, k))
    if "seasonal" in name and k == 0:
        acts = [d["path"]["activity"] for d in tiles]
        assert any(a[2] >= 1 and 0 < a[1] < a[0] for a in acts), acts      # interior tiles went quiet on the peer transport


@pytest.mark.parametrize("name", ["2x2_periodic", "1x2_fold", "2x1_bounded_x", "2x2_channel_land_arrays"])
def test_local_tiles_strict_mode_bitwise(name):
    """STRICT mode (the reference's operation order, three kernels, an exchange every sub-step): tiled == untiled too."""
    Rx, Ry, kw, _ = DECOMPOSITIONS[name]
    c = cases.make_case(H=4, substeps=6, patches=True, random_uv=0.05, **kw)
    mom, step = reference(c, mode="strict")
    tiles = run_tiles(c, Rx, Ry, 1, mode="strict")
    for d in tiles:
        assert d["path"]["transport"] == "rccl" and d["path"]["level"] == 0 and d["path"]["exchange_interval"] == 1, d["path"]
    check(tiles, mom, step, (name, "strict"))


@pytest.mark.parametrize("transport", ["peer", "rccl"])
@pytest.mark.parametrize("Rx,Ry", [(1, 2), (1, 4), (1, 8), (2, 1), (2, 2), (2, 4)])
def test_bench_decompositions_in_process(Rx, Ry, transport):
    """bench.py's own N = 2 / 4 / 8 jobs, scaled down: the headline configuration (periodic f-plane, bench.py's seeded inputs
    built per tile by bench.tile_fields / local_case) as y slabs (1 x 2, 1 x 4, 1 x 8: bench.py's default since round 4) and as
    2 x 1, 2 x 2, 2 x 4 (`--partition`) distinct tiles on this one GPU, on the peer transport
    (halo 4) and on the message exchange bench.py falls back to (halo 32, every 16 sub-steps): each tile equals the one-GPU run
    of the assembled global state bit for bit -- the check bench.py itself makes before it times anything, here on hardware."""
    import bench
    size, substeps = (512 if Ry < 8 else 1024), 120          # (a slab must be taller than 2 halos + the pair kernel's rings)
    nx, ny = size // Rx, size // Ry
    halo = 4 if transport == "peer" else 32

    def make_model(grid):
        dyn = csi.SeaIceMomentumEquation(grid, coriolis=csi.FPlane(f=1e-4), rheology=csi.ElastoViscoPlasticRheology(),
                                         top_momentum_stress=(0.01, 0.01), bottom_momentum_stress=csi.SemiImplicitStress(),
                                         solver=csi.SplitExplicitSolver(substeps=substeps))
        return csi.SeaIceModel(grid, dynamics=dyn, advection=csi.WENO(order=7), timestepper="SplitRungeKutta3", mode="fast")

    whole = csi.RectilinearGrid((size, size), x=(0.0, size * 2000.0), y=(0.0, size * 2000.0), topology=(csi.Periodic, csi.Periodic),
                                halo=(halo, halo))
    ref = make_model(whole)
    gf = bench.global_fields(np, nx, ny, Rx, Ry)
    csi.set_(ref, h=gf["h"], aice=gf["a"], u=gf["u"], v=gf["v"])
    for _ in range(2):
        csi.time_step_momentum(ref, 120.0)
    ref.synchronize()
    want = {f: EVP_FIELDS[f](ref).interior_numpy().copy() for f in ("u", "v", "s11", "s22", "s12")}

    def tile(rank, group):
        tg, fld = bench.local_case(csi, np, nx, ny, Rx, Ry, rank, halo=halo)
        tg.local_group = group
        m = make_model(tg)
        m.set_halo_transport(transport)
        csi.set_(m, h=fld["h"], aice=fld["a"], u=fld["u"], v=fld["v"])
        for _ in range(2):
            csi.time_step_momentum(m, 120.0)
        m.synchronize()
        got = {f: EVP_FIELDS[f](m).interior_numpy().copy() for f in want}
        return got, (tg.i_off, tg.j_off), dict(m.ctx.last_path(), transport=m.ctx.halo_transport())

    for rank, (got, (i0, j0), path) in enumerate(run_tile_threads(Rx * Ry, tile)):
        assert path["transport"] == transport and path["level"] == 2, path
        assert path["exchanges"] == (1 if transport == "peer" else 8), path
        for f in want:
            w = want[f][j0:j0 + ny, i0:i0 + nx]
            assert np.array_equal(w, got[f][:ny, :nx]), (rank, f, np.abs(w - got[f][:ny, :nx]).max())


@pytest.mark.parametrize("k", [0, -1])
@pytest.mark.parametrize("stepper", ["ForwardEuler", "SplitRungeKutta3"])
def test_whole_time_step_with_snow_on_2x2_tiles(stepper, k):
    """Whole time_step! on a real 2 x 2 decomposition of an immersed channel: EVP sub-cycle, WENO7 advection of h, aice AND the
    snow thickness, the layered (ice + snow) thermodynamic step, update_state!'s halo refresh of every prognostic field -- three
    steps in a row; h, aice, hs, u, v and the masked mass-flux diagnostics of every tile equal the untiled run bit for bit."""
    Rx, Ry = 2, 2
    c = cases.make_case(Nx=256, Ny=96, H=8, substeps=12, topo=("periodic", "bounded"), patches=True, random_uv=0.02, land=0.2)
    rng = np.random.default_rng(41)
    hs0 = np.where(c["a"] > 0, 0.3 * rng.random(c["a"].shape), 0.0)

    def build(tile=None, group=None):
        ice = csi.SlabThermodynamics(top_heat_flux=-80.0, bottom_heat_flux=6.0, bottom_salinity=30.0,
                                     top_heat_boundary_condition=csi.MeltingConstrainedFluxBalance())
        m = cases.csi_model(c, mode="fast", timestepper=stepper, advection=csi.WENO(order=7), tile=tile, local_group=group,
                            ice_thermodynamics=ice, snow_thermodynamics=csi.snow_slab_thermodynamics(), snowfall=3e-5)
        g = m.grid
        hs = hs0 if tile is None else g.local_interior(hs0, csi.Center, csi.Center)
        csi.set_(m, hs=hs)
        return m

    def state(m):
        m.synchronize()
        out = {"h": m.ice_thickness, "a": m.ice_concentration, "hs": m.snow_thickness, "u": m.velocities.u, "v": m.velocities.v,
               "mf_ice": m.mass_fluxes.thermodynamics.ice, "mf_snow": m.mass_fluxes.thermodynamics.snow}
        return {k_: f.interior_numpy().copy() for k_, f in out.items()}

    ref = build()
    for _ in range(3):
        csi.time_step(ref, c["dt"])
    want = state(ref)
    assert np.abs(want["hs"] - hs0).max() > 1e-4

    def tile(rank, group):
        m = build((Rx, Ry, rank), group)
        if k < 0:
            m.set_halo_transport("rccl")
        for _ in range(3):
            csi.time_step(m, c["dt"])
        g = m.grid
        return state(m), (g.i_off, g.j_off, g.Nx, g.Ny), m.ctx.halo_transport()

    for rank, (got, (i0, j0, nx, ny), transport) in enumerate(run_tile_threads(Rx * Ry, tile)):
        assert transport == ("peer" if k == 0 else "rccl")
        for f in want:
            w = want[f][j0:j0 + ny, i0:i0 + nx]
            g_ = got[f][:ny, :nx]
            assert np.array_equal(w, g_), (rank, f, np.abs(w - g_).max(), np.argwhere(w != g_)[:4].tolist())


def test_peer_transport_follows_a_change_of_the_forcing_kinds():
    """The pair kernel's tile count depends on the forcing kinds (array forcing: 1024 tiles at two waves per SIMD), and the neighbours
    wait for as many flags as a rank's tile sets had when the peer transport was set up: binding array forcing AFTER the first
    sub-cycle must lead to a new (collective) set-up, not to waits that time out.  2 x 1 tiles, peer transport, number-valued forcing
    for one sub-cycle, then wind-stress and ocean-velocity arrays: tiled == untiled bit for bit after each."""
    from climaseaice_jl_amd import _lib
    # (512 x 2560 tiles: 10 strips x 153 chunks of 17 rows without array forcing, x 102 chunks of 26 rows with it -- small grids have
    #  the same geometry either way)
    Rx, Ry = 2, 1
    kw = dict(Nx=1024, Ny=2560, topo=("periodic", "bounded"), land=0.2)
    c0 = cases.make_case(H=8, substeps=6, patches=True, random_uv=0.05, **kw)
    c1 = cases.make_case(H=8, substeps=6, patches=True, random_uv=0.05, field_forcing=True, **kw)

    def switch(m, tile_grid=None):
        loc = (lambda a, lx, ly: a) if tile_grid is None else tile_grid.local_interior
        m._set_stress(_lib.STRESS_TOP, (loc(c1["top_u"], csi.Face, csi.Center), loc(c1["top_v"], csi.Center, csi.Face)), "TOP")
        m._set_stress(_lib.STRESS_BOTTOM, csi.SemiImplicitStress(ue=loc(c1["ue_f"], csi.Face, csi.Center), ve=loc(c1["ve_f"], csi.Center, csi.Face)), "BOT")

    def fields(m):
        return {f: EVP_FIELDS[f](m).interior_numpy().copy() for f in ("u", "v", "s11", "s22", "s12")}

    ref = cases.csi_model(c0, mode="fast")
    csi.time_step_momentum(ref, c0["dt"]); ref.synchronize()
    want0 = fields(ref)
    switch(ref)
    csi.time_step_momentum(ref, c0["dt"]); ref.synchronize()
    want1 = fields(ref)
    assert not np.array_equal(want0["u"], want1["u"])

    def tile(rank, group):
        m = cases.csi_model(c0, mode="fast", tile=(Rx, Ry, rank), local_group=group)
        m.set_exchange_interval(0)
        csi.time_step_momentum(m, c0["dt"]); m.synchronize()
        a = fields(m)
        switch(m, m.grid)
        csi.time_step_momentum(m, c0["dt"]); m.synchronize()
        b = fields(m)
        g = m.grid
        res = dict(a=a, b=b, offsets=(g.i_off, g.j_off, g.Nx, g.Ny), transport=m.ctx.halo_transport())
        del m
        return res

    for rank, d in enumerate(run_tile_threads(Rx * Ry, tile)):
        assert d["transport"] == "peer", d["transport"]
        i0, j0, nx, ny = d["offsets"]
        for got, want, what in ((d["a"], want0, "numbers"), (d["b"], want1, "arrays")):
            for f in want:
                assert np.array_equal(got[f][:ny, :nx], want[f][j0:j0 + ny, i0:i0 + nx]), (what, rank, f)


def test_validate_all_spreads_a_transport_abort_to_non_neighbours():
    """csi_validate_all (round 5): a wait of the peer transport that gave up reaches only the direct neighbours' abort words; the
    all-reduce of the status makes EVERY rank of the decomposition see it -- here four slabs in a row, the error raised on rank 0
    (csi_debug_peer_abort leaves what the kernel's time-out leaves), rank 2 is not its neighbour -- and after every rank has re-armed
    the transport the tiles reproduce the untiled run bit for bit again."""
    c = cases.make_case(Nx=160, Ny=256, substeps=8, topo=("periodic", "bounded"), patches=True, random_uv=0.05)
    ref = cases.csi_model(c, mode="fast")
    for _ in range(3):
        csi.time_step_momentum(ref, c["dt"])
    ref.synchronize()
    want = {f: EVP_FIELDS[f](ref).interior_numpy().copy() for f in ("u", "v", "s11")}
    barrier = threading.Barrier(4)

    def tile(rank, group):
        m = cases.csi_model(c, mode="fast", tile=(1, 4, rank), local_group=group)
        csi.time_step_momentum(m, c["dt"])
        m.synchronize()
        assert m.ctx.halo_transport() == "peer"
        m.ctx.validate_all()                                   # clean: nothing raised anywhere
        barrier.wait()
        if rank == 0:
            m.ctx.call("csi_debug_peer_abort")
        barrier.wait()
        seen = False
        try:
            m.ctx.validate_all()
        except csi.CsiError as e:
            seen = "peer halo transport" in str(e)
        with pytest.raises(csi.CsiError):                      # sticky on every rank, not only on rank 0
            csi.time_step_momentum(m, c["dt"])
        barrier.wait()
        m.set_halo_transport("peer")                           # every rank re-arms: the next sub-cycle sets the transport up again
        csi.time_step_momentum(m, c["dt"])
        csi.time_step_momentum(m, c["dt"])
        m.synchronize()
        g = m.grid
        return dict(seen=seen, transport=m.ctx.halo_transport(), off=(g.i_off, g.j_off, g.Nx, g.Ny),
                    **{f: EVP_FIELDS[f](m).interior_numpy().copy() for f in ("u", "v", "s11")})

    for r, res in enumerate(run_tile_threads(4, tile)):
        assert res["seen"], f"rank {r} did not see the abort of rank 0"
        assert res["transport"] == "peer"
        i0, j0, nx, ny = res["off"]
        for f, w in want.items():
            assert np.array_equal(res[f][:ny, :nx], w[j0:j0 + ny, i0:i0 + nx]), (r, f)
