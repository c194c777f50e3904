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
