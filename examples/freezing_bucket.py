"""The reference's examples/freezing_bucket.jl with the MI355X library: one grid cell of water under a -10 degC lid,
frazil formation until the concentration reaches 1, then conductive growth; dt = 10 minutes, 10 days.

    python examples/freezing_bucket.py            (needs the GPU; prints thickness and concentration once per day)
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import climaseaice_jl_amd as csi


def build(device="cuda:0"):
    grid = csi.RectilinearGrid((1, 1), x=(0.0, 1.0), y=(0.0, 1.0), topology=(csi.Periodic, csi.Periodic), halo=(1, 1))
    thermo = csi.SlabThermodynamics(top_temperature=-10.0, conductivity=2.0, heat_capacity=2100.0, bottom_heat_flux="frazil")
    return csi.SeaIceModel(grid, dynamics=None, advection=None, ice_thermodynamics=thermo, sea_ice_density=900.0,
                           timestepper="ForwardEuler", device=device)


def run(model, steps=1440, dt=600.0, every=144):
    series = []
    for n in range(steps):
        csi.time_step(model, dt)
        if (n + 1) % every == 0:
            model.synchronize()
            series.append(((n + 1) * dt / 86400.0, float(model.ice_thickness.interior_numpy()[0, 0]),
                           float(model.ice_concentration.interior_numpy()[0, 0])))
    return series


if __name__ == "__main__":
    m = build()
    for day, h, a in run(m):
        print(f"day {day:5.1f}   h = {h:.6f} m   aice = {a:.6f}")
