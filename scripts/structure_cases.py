"""make_case arguments of the configurations scripts/ab_skip.py, run_case.py and prof_structure.sh run (2048^2 unless said otherwise)."""
CASES = {
    "headline": dict(topo=("periodic", "periodic")),
    "masked": dict(topo=("periodic", "bounded"), land=0.386),                     # config 5's mask (38.6 % land in discs)
    "masked_seasonal": dict(topo=("periodic", "bounded"), land=0.386, ice_free_rows=(0.25, 0.75)),
    "tripolar_like": dict(topo=("periodic", "folded"), curvilinear=0.05, land=0.3, field_forcing=True, free_drift=True),
    "tripolar": dict(grid="tripolar", tripolar=dict(southernmost_latitude=-78.0), field_forcing=True, free_drift=True, coriolis_points=True),
    "tripolar_seasonal": dict(grid="tripolar", tripolar=dict(southernmost_latitude=-78.0), field_forcing=True, free_drift=True, coriolis_points=True,
                              ice_edge=58.0),
    "tripolar_land": dict(grid="tripolar", tripolar=dict(southernmost_latitude=-78.0), land=0.3, field_forcing=True, free_drift=True, coriolis_points=True,
                          ice_edge=58.0),
    "arctic_cap": dict(grid="tripolar", tripolar=dict(southernmost_latitude=60.0, north_poles_latitude=65.0), field_forcing=True, free_drift=True),
    "curvilinear": dict(topo=("periodic", "bounded"), curvilinear=0.05),
    "latlon_as_full": dict(topo=("periodic", "bounded"), grid="latlon", curvilinear=0.0),
}
