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
