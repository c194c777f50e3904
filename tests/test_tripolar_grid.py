"""CPU tests of csi.TripolarGrid (climaseaice.jl_amd/grids.py): the structure the reference's flagship grid has
(test/distributed_tests_utils.jl:170-183, 226-259) -- latitude-longitude rows with one metric value per row, a conformal (orthogonal)
bipolar cap, a fold through the centres of row Ny, the analytic land -- and the oracle on it."""
import numpy as np

import cases
import climaseaice_jl_amd as csi


def test_tripolar_grid_structure():
    g = csi.TripolarGrid((224, 192), southernmost_latitude=-78.0)
    m = g.metrics()
    H = 4
    names = list(m)
    names.remove("kind")
    for k in names:
        assert np.isfinite(m[k]).all() and (m[k] > 0).all()
    # latitude-longitude rows: one value per row in every plane; the cap: not
    lat_rows = g.cap_first_row - 1 + H
    for k in names:
        assert (m[k][:lat_rows, :] == m[k][:lat_rows, :1]).all()
    assert not (m["dxcc"][lat_rows + 3, :] == m["dxcc"][lat_rows + 3, 0]).all()
    # the net is orthogonal: the angle between the two families of lines at the cell centres of the cap
    lam_e, phi_e = g.nodes_2d(csi.Face, csi.Center)
    lam_n, phi_n = g.nodes_2d(csi.Center, csi.Face)

    def xyz(lam, phi):
        l, p = np.deg2rad(lam), np.deg2rad(phi)
        return np.stack([np.cos(p) * np.cos(l), np.cos(p) * np.sin(l), np.sin(p)], -1)
    E, N = xyz(lam_e, phi_e), xyz(lam_n, phi_n)
    tx = E[:-1, 1:, :] - E[:-1, :-1, :]                   # across cell (i, j): east face - west face; i = 1 .. Nx - 1, j = 1 .. Ny - 1
    ty = N[1:, :-1, :] - N[:-1, :-1, :]                   # north face - south face
    cosang = np.abs((tx * ty).sum(-1)) / (np.linalg.norm(tx, axis=-1) * np.linalg.norm(ty, axis=-1))
    cap = slice(g.cap_first_row, g.Ny - 2)
    inner = np.ones(cosang.shape[1], bool)
    for ip in (1, g.Nx // 2 + 1):                         # columns next to the pole axis: the cells there wrap around the pole
        inner[max(ip - 4, 0):ip + 2] = False
    assert cosang[cap][:, inner].max() < 2e-2             # second-order in the spacing
    # the fold: row Ny is its own image
    lam_c, phi_c = g.nodes_2d(csi.Center, csi.Center)
    assert np.abs(phi_c[-1] - phi_c[-1, ::-1]).max() < 1e-10
    wet = g.analytic_land()
    assert not wet[:5].any() and 0.85 < wet.mean() < 0.99




def test_metric_halos_are_fold_and_wrap_images():
    g = csi.TripolarGrid((96, 80), southernmost_latitude=-75.0)
    m = g.metrics()
    H = 4
    for name in csi.grids.METRIC_NAMES:
        a = m[name]
        img = csi.fold_north(a, g.Nx, g.Ny, H, H, name[2] == "f", name[3] == "f", 1)
        assert np.array_equal(a, img), name                                            # rows beyond the fold: images of the interior
        assert np.array_equal(a[:, :H], a[:, g.Nx:g.Nx + H]), name                     # periodic in x
        assert np.array_equal(a[:, g.Nx + H:g.Nx + 2 * H], a[:, H:2 * H]), name
    fu, fv = g.coriolis_planes()
    lat_rows = g.cap_first_row - 1 + H
    assert (fu[:lat_rows] == fu[:lat_rows, :1]).all() and (fv[:lat_rows] == fv[:lat_rows, :1]).all()
    assert np.abs(fu).max() <= 2 * 7.292115e-5 * 1.0000001


def test_oracle_runs_on_the_tripolar_grid_and_keeps_land_and_ice_free_ocean_at_rest():
    c = cases.make_case(Nx=64, Ny=56, grid="tripolar", tripolar=dict(southernmost_latitude=-78.0), substeps=12, patches=False, random_uv=0.02,
                        field_forcing=True, free_drift=True, coriolis_points=True, ice_edge=60.0)
    p = cases.oracle_problem(c)
    s0 = {k: p.f[k].copy() for k in ("s11", "s22", "s12")}
    p.time_step_momentum(c["dt"])
    for k in ("u", "v", "s11", "s22", "s12"):
        assert np.isfinite(p.f[k]).all(), k
    assert np.abs(p.f["u"]).max() > 1e-3
    # where there is no ice mass (land, ice-free ocean) the sub-cycle is a no-op: velocities zero, stresses untouched -- the exact fixed
    # point the library's tile activity relies on (elasto_visco_plastic_rheology.jl:343-347, split_explicit_momentum_equations.jl:217-228)
    H = c["H"]
    m = (c["h"] * c["a"]) > 0
    far = ~m
    far[1:, :] &= ~m[:-1, :]; far[:-1, :] &= ~m[1:, :]; far[:, 1:] &= ~m[:, :-1]; far[:, :-1] &= ~m[:, 1:]
    far[1:, 1:] &= ~m[:-1, :-1]; far[:-1, :-1] &= ~m[1:, 1:]; far[1:, :-1] &= ~m[:-1, 1:]; far[:-1, 1:] &= ~m[1:, :-1]
    ui, vi = p.interior("u")[:c["Ny"], :c["Nx"]], p.interior("v")[:c["Ny"], :c["Nx"]]
    assert far.sum() > 500 and (ui[far] == 0).all() and (vi[far] == 0).all()
    for k in ("s11", "s22"):
        assert np.array_equal(p.f[k][H:H + c["Ny"], H:H + c["Nx"]][far], s0[k][H:H + c["Ny"], H:H + c["Nx"]][far]), k


def test_no_ice_mass_is_a_fixed_point_of_the_oracle_subcycle_on_other_grids():
    """The same fixed point on a masked channel and a lat-lon box with free drift and array forcing (sigma given non-zero values first)."""
    for kw in (dict(Nx=72, Ny=60, topo=("periodic", "bounded"), land=0.3, ice_free_rows=(0.2, 0.7)),
               dict(Nx=64, Ny=64, topo=("bounded", "bounded"), grid="latlon", field_forcing=True, free_drift=True, ice_free_rows=(0.4, 1.0))):
        c = cases.make_case(substeps=9, patches=False, random_uv=0.05, **kw)
        p = cases.oracle_problem(c)
        rng = np.random.default_rng(1)
        for k in ("s11", "s22", "s12"):
            p.f[k][...] = rng.standard_normal(p.f[k].shape)
        s0 = {k: p.f[k].copy() for k in ("s11", "s22")}
        p.time_step_momentum(c["dt"])
        H, Ny, Nx = c["H"], c["Ny"], c["Nx"]
        m = (c["h"] * c["a"]) > 0
        far = ~m
        far[1:, :] &= ~m[:-1, :]; far[:, 1:] &= ~m[:, :-1]; far[1:, 1:] &= ~m[:-1, :-1]
        far[0, :] = False; far[:, 0] = False                      # (the neighbours beyond the edge are halo cells: not examined here)
        ui, vi = p.interior("u")[:Ny, :Nx], p.interior("v")[:Ny, :Nx]
        assert far.sum() > 200 and (ui[far] == 0).all() and (vi[far] == 0).all(), kw
        for k in ("s11", "s22"):
            assert np.array_equal(p.f[k][H:H + Ny, H:H + Nx][far], s0[k][H:H + Ny, H:H + Nx][far]), (kw, k)
