"""CPU tests of the host-side mirror: grid metrics, Oceananigans parent-array extents, kernel ranges."""
import numpy as np

import climaseaice_jl_amd as csi


def test_field_extents_follow_oceananigans_rule():
    g = csi.RectilinearGrid((10, 6), x=(0, 1), y=(0, 1), topology=(csi.Bounded, csi.Periodic), halo=(4, 3))
    assert g.field_size(csi.Center, csi.Center) == (18, 12)
    assert g.field_size(csi.Face, csi.Center) == (19, 12)     # Face & Bounded: +1
    assert g.field_size(csi.Center, csi.Face) == (18, 12)     # Face & Periodic: no extra point
    assert g.field_size(csi.Face, csi.Face) == (19, 12)
    assert g.stress_kernel_range() == (-2, 13, -1, 8)          # -H+2 : N+H-1 (evp:145)


def test_latlon_metrics():
    g = csi.LatitudeLongitudeGrid((40, 40), longitude=(0, 60), latitude=(20, 70), topology=(csi.Bounded, csi.Bounded), halo=(4, 4))
    R = 6371e3
    j = 1                                     # first interior row: entry j + Hy - 1
    t = j + g.Hy - 1
    dphi, dlam = 50 / 40, 60 / 40
    assert np.isclose(g.dy, R * np.deg2rad(dphi))
    assert np.isclose(g.dxf[t], R * np.cos(np.deg2rad(20.0)) * np.deg2rad(dlam))
    assert np.isclose(g.dxc[t], R * np.cos(np.deg2rad(20.0 + dphi / 2)) * np.deg2rad(dlam))
    assert np.isclose(g.azc[t], R * R * np.deg2rad(dlam) * (np.sin(np.deg2rad(20 + dphi)) - np.sin(np.deg2rad(20))))
    # the sum of cell areas is the area of the spherical patch
    total = g.azc[g.Hy:g.Hy + 40].sum() * 40
    exact = R * R * np.deg2rad(60) * (np.sin(np.deg2rad(70)) - np.sin(np.deg2rad(20)))
    assert np.isclose(total, exact, rtol=1e-12)


def test_rheology_and_solver_defaults_match_reference():
    r = csi.ElastoViscoPlasticRheology()
    assert (r.ice_compressive_strength, r.ice_compaction_hardening, r.yield_curve_eccentricity) == (27500.0, 20.0, 2.0)
    assert (r.minimum_plastic_stress, r.min_relaxation_parameter, r.max_relaxation_parameter) == (2e-9, 50.0, 300.0)
    assert np.isclose(r.relaxation_strength, np.pi ** 2)
    assert csi.SplitExplicitSolver().substeps == 120
    g = csi.RectilinearGrid((8, 8), x=(0, 1), y=(0, 1))
    d = csi.SeaIceMomentumEquation(g, device="cpu")
    assert d.solver.substeps == 150 and d.minimum_mass == 1.0 and d.minimum_concentration == 1e-3
    assert float(d.auxiliaries.fields.alpha.data.min()) == 300.0   # fill!(alpha, alpha+)


def test_pair_plan_geometry_and_ranges():
    """csi_plan_pair: how the two-sub-steps-per-launch kernel tiles a grid and which cells it stores (pure host logic,
    csrc/csi_abi.hip).  Properties: one round of at most 2048 wave tiles that covers the second sub-step's compute range;
    the first sub-step computes 2 more layers on every side and stays inside the parent arrays; periodic / wall sides store
    the interior only (sigma also the wall corners), connected sides the ring of the next pair."""
    from climaseaice_jl_amd import _lib
    P, B, FC = _lib.PERIODIC, _lib.BOUNDED, _lib.FULLY_CONNECTED
    for (Nx, Ny, H, tx, ty, k, m) in [(2048, 2048, 4, P, P, 1, 0), (2048, 2048, 4, P, B, 1, 0), (2048, 2048, 4, B, B, 1, 0),
                                      (4096, 4096, 4, P, P, 1, 0), (1024, 512, 6, P, B, 1, 0), (2048, 2048, 16, FC, FC, 8, 0),
                                      (2048, 2048, 16, FC, FC, 8, 6), (2048, 2048, 8, FC, P, 4, 2), (300, 200, 4, B, P, 1, 0)]:
        p = _lib.plan_pair(Nx, Ny, H, H, tx, ty, k, m)
        assert p is not None
        i0, i1, j0, j1 = p["second_compute"]
        assert p["nstrips"] == -(-(i1 - i0 + 1) // 56) and p["nchunks"] == -(-(j1 - j0 + 1) // p["rows"])
        assert p["nstrips"] * p["nchunks"] <= 2048 or p["rows"] == 6           # one round (tiny grids: at least 6 rows per tile)
        a = p["first_compute"]
        assert a == (i0 - 2, i1 + 2, j0 - 2, j1 + 2)
        assert a[0] - 1 >= 1 - H and a[1] + 1 <= Nx + H and a[2] - 1 >= 1 - H and a[3] + 1 <= Ny + H   # loads stay in the parent
        V = 2 * k - 2 * m - 2                                               # valid layers the second sub-step starts from
        for axis, topo, N in ((0, tx, Nx), (1, ty, Ny)):
            lo, hi = p["store_second"][2 * axis], p["store_second"][2 * axis + 1]
            slo, shi = p["store_sigma"][2 * axis], p["store_sigma"][2 * axis + 1]
            if topo == FC:
                assert (lo, hi) == (3 - V, N + V - 2) and (slo, shi) == (2 - V, N + V - 1)
            else:
                assert (lo, hi) == (1, N) and (slo, shi) == (1, N + (1 if topo == B else 0))
        assert p["walls"] == (B in (tx, ty))
    # not applicable: halo < 4, domain smaller than two halos, odd exchange interval on tiles
    assert _lib.plan_pair(64, 64, 3, 3, P, P) is None
    assert _lib.plan_pair(7, 64, 4, 4, P, P) is None
    assert _lib.plan_pair(256, 256, 8, 8, FC, FC, 3, 0) is None


def test_peer_launch_chunks_tile_the_rows_and_keep_edge_tiles_short():
    """csi_plan_peer_chunks (pure host logic, csrc/csi_core.hip pair_geom + csi_peer.hip peer_wait_counts; the kernel applies the same
    formula, evp_fused2.hip): the chunks of a pair launch on the peer transport cover the decomposed rows exactly once, in order; the
    chunk next to a peer-connected y side is shorter than the others but not shorter than the halo, so that exactly ONE chunk per
    side waits for / signals the neighbour (the side's tile set), and the launch stays within one round of 1024 two-wave tiles on
    small grids (the third wave per SIMD beyond that makes its tiles 1.5x slower: profiles/r04_tile_1024x512.md)."""
    from climaseaice_jl_amd import _lib
    for (Nx, Ny, H) in [(2048, 256, 4), (2048, 512, 4), (1024, 512, 4), (2048, 1024, 4), (2048, 2048, 4), (512, 512, 4), (1024, 1024, 6),
                        (300, 200, 4), (2048, 256, 8), (4096, 512, 4), (130, 96, 6), (128, 48, 6), (2048, 260, 5), (4096, 2048, 4), (1024, 2048, 4), (3072, 1536, 4)]:
        for south, north in ((True, True), (True, False), (False, True), (False, False)):
            p = _lib.plan_peer_chunks(Nx, Ny, H, H, south, north)
            assert p is not None, (Nx, Ny, H)
            ch = p["chunks"]
            assert len(ch) == p["nchunks"]
            assert ch[0][0] == 0 and ch[-1][1] == Ny + 1                      # the second sub-step's stress rows 2 - V .. N + V - 1, V = 2
            for (a0, b0), (a1, b1) in zip(ch, ch[1:]):
                assert a1 == b0 + 1 and b0 >= a0, (Nx, Ny, H, south, north, ch)
            assert all(b - a + 1 <= p["rows"] for a, b in ch)
            reach = max(H, 4)
            short = p["elo"] not in (0, p["rows"]) or p["ehi"] != 0
            if short:
                if south:
                    assert p["elo"] >= reach and p["elo"] < p["rows"] and ch[0][1] - ch[0][0] + 1 == p["elo"]
                    assert ch[1][0] > reach and p["nS"] == 1               # the second chunk reads no halo row and owns no imaged row
                if north:
                    assert p["ehi"] >= reach and ch[-1][1] - ch[-1][0] + 1 == p["ehi"]
                    assert ch[-2][1] + reach <= Ny and p["nN"] == 1
            # (round 5: at EVERY size on uniform coefficients -- the plan's case --, short edge chunks included: one tile over the
            #  resident set put a third wave on some SIMDs for a whole launch, 2048 x 1024 connected in y 68 -> 52.6 G)
            assert p["nstrips"] * p["nchunks"] <= 1024 or p["rows"] == 6, (Nx, Ny, p["nstrips"], p["nchunks"])
    # tall enough tiles get the short edge chunks (the metric's slabs: 2048 x 256 at N = 8, 2048 x 512 at N = 4)
    for Ny in (256, 512):
        p = _lib.plan_peer_chunks(2048, Ny, 4, 4, True, True)
        assert p["elo"] == p["rows"] - 4 == p["ehi"], p


def test_fused_paths_refuse_parents_beyond_32bit_offsets():
    """The fused kernels address fields with 32-bit byte offsets: a parent array of 4 GiB or more (about 23k x 23k cells)
    must fall back to the three-kernel path (64-bit indexing) instead of wrapping (csi_abi.hip: offsets_fit_32bit)."""
    from climaseaice_jl_amd import _lib
    P = _lib.PERIODIC
    assert _lib.plan_pair(16384, 16384, 4, 4, P, P) is not None           # 2.1 GiB per parent
    assert _lib.plan_pair(23160, 23160, 4, 4, P, P) is not None           # just below 4 GiB
    assert _lib.plan_pair(23170, 23170, 4, 4, P, P) is None               # (23179 x 23179 x 8 B >= 2^32)
    assert _lib.plan_pair(40000, 16384, 4, 4, P, P) is None


def test_julia_shim_matches_reference_types():
    """scripts/check_julia_shim.py: the Julia binding dispatches on the reference's real type-parameter positions, extends
    functions the reference defines, and ccalls symbols include/csi.h declares (static check; no Julia in this image).
    Skips where /root/reference is absent (the GPU box)."""
    import os
    import subprocess
    import sys
    import pytest
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "scripts", "check_julia_shim.py")], capture_output=True, text=True)
    if p.returncode == 77:
        pytest.skip("no reference tree here")
    assert p.returncode == 0, p.stdout


def test_exchange_plans_of_all_ranks_match_in_posting_order():
    """RCCL (and the in-process tile group) match the messages between a pair of ranks in the order they were posted.  For every
    decomposition up to 4 x 4 of periodic / bounded / north-folded grids: the sizes of the messages rank a sends to rank b, in a's
    plan order, are the sizes b expects from a, in b's plan order -- and what b receives is the strip a sent: its global
    coordinates (periodic wrap applied) equal those of the halo cells it fills."""
    import itertools
    L = csi._lib
    Nx, Ny, H, W = 12, 10, 4, 3
    matched = 0
    for (Rx, Ry), (tx, ty) in itertools.product(itertools.product(range(1, 5), range(1, 5)),
                                                itertools.product(("periodic", "bounded"), ("periodic", "bounded", "folded"))):
        if Rx * Ry == 1 or (ty == "folded" and (Rx > 1 or tx != "periodic")):
            continue
        glob = {"periodic": csi.Periodic, "bounded": csi.Bounded, "folded": csi.RightFolded}
        G = csi.RectilinearGrid((Nx * Rx, Ny * Ry), x=(0.0, 1.0), y=(0.0, 1.0), topology=(glob[tx], glob[ty]), halo=(H, H))
        code = {csi.Periodic: L.PERIODIC, csi.Bounded: L.BOUNDED, csi.FullyConnected: L.FULLY_CONNECTED, csi.LeftConnected: L.LEFT_CONNECTED,
                csi.RightConnected: L.RIGHT_CONNECTED, csi.RightFolded: L.RIGHT_FOLDED, csi.LeftConnectedRightFolded: L.LEFT_CONNECTED_RIGHT_FOLDED}
        plans = {}
        for rank in range(Rx * Ry):
            t = csi.TileGrid(G, Rx, Ry, rank % Rx, rank // Rx)
            args = (t.Nx, t.Ny, t.Hx, t.Hy, code[t.topology[0]], code[t.topology[1]], t.rx, t.ry, Rx, Ry, t.periodic[0], t.periodic[1], W)
            plans[rank] = (csi.plan_exchange(*args, 0), csi.plan_exchange(*args, 1), t)
        for a, b in itertools.product(range(Rx * Ry), repeat=2):
            sent = [(i0, j0, ni, nj) for (peer, i0, j0, ni, nj) in plans[a][0] if peer == b]
            want = [(i0, j0, ni, nj) for (peer, i0, j0, ni, nj) in plans[b][1] if peer == a]
            assert [(s[2], s[3]) for s in sent] == [(w[2], w[3]) for w in want], ((Rx, Ry), (tx, ty), a, b, sent, want)
            ta, tb = plans[a][2], plans[b][2]
            for (si, sj, ni, nj), (wi, wj, _, _) in zip(sent, want):
                gx = (ta.i_off + si - 1) % (Nx * Rx), (tb.i_off + wi - 1) % (Nx * Rx)
                gy = (ta.j_off + sj - 1) % (Ny * Ry), (tb.j_off + wj - 1) % (Ny * Ry)
                assert gx[0] == gx[1] and gy[0] == gy[1], ((Rx, Ry), (tx, ty), a, b, (si, sj), (wi, wj))
                matched += 1
    assert matched > 2000, matched
