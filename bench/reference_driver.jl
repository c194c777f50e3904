# reference_driver.jl -- runs the REFERENCE (CliMA/ClimaSeaIce.jl + Oceananigans.jl, CPU) on the seeded inputs of this
# repository's golden cases and dumps its results, so that the oracle (oracle/csi_oracle.c) can be pinned against
# reference-produced vectors (SURVEY.md section 8c).  It uses the reference's PUBLIC API only.
#
# This image has no Julia: the script has never been executed here ("parity unpinned" until someone runs it).  Usage, on
# any machine with Julia >= 1.10 and the reference checked out:
#
#   python tests/golden/reference_io.py export /tmp/csi_cases          # inputs of every golden case -> raw .f64 files
#   julia --project=/path/to/ClimaSeaIce.jl bench/reference_driver.jl /tmp/csi_cases
#   python tests/golden/reference_io.py import /tmp/csi_cases          # -> tests/golden/ref_<case>.npz (commit them)
#
# after which tests/test_golden.py::test_oracle_matches_reference_fixture stops skipping.
#
# File format (no Julia / Python packages beyond the standard libraries): <dir>/<case>/case.txt with `key = value`
# lines, inputs in_h.f64, in_a.f64, in_u.f64, in_v.f64 [, top_u, top_v, ue, ve, mask.u8] as raw little-endian
# arrays, i fastest (Julia's column-major interior(field)[:, :, 1] == numpy's C-order (Ny, Nx)); outputs
# out_<field>_<tag>.f64 in the same layout.
#
# Entry points driven (reference file:line):
#   SeaIceModel, set!                      src/sea_ice_model.jl:140-315
#   time_step_momentum!                    src/SeaIceDynamics/split_explicit_momentum_equations.jl:103-195
#   time_step! (ForwardEuler / RK3)        src/sea_ice_fe_step.jl:13-34, src/sea_ice_rk_substep.jl:81-94
# and the set-up of test/distributed_tests_utils.jl:104-137.

using Oceananigans
using Oceananigans.Units
using Oceananigans.Grids: Periodic, Bounded, Flat
using ClimaSeaIce
using ClimaSeaIce.SeaIceDynamics: time_step_momentum!
using ClimaSeaIce.Rheologies: ElastoViscoPlasticRheology

function read_manifest(path)
    d = Dict{String, String}()
    for line in eachline(path)
        s = strip(line)
        (isempty(s) || startswith(s, "#")) && continue
        k, v = split(s, "="; limit = 2)
        d[strip(k)] = strip(v)
    end
    return d
end

geti(d, k) = parse(Int, d[k])
getf(d, k) = parse(Float64, d[k])
getb(d, k) = get(d, k, "0") == "1"

function read_array(dir, name, nx, ny)
    a = Array{Float64}(undef, nx, ny)
    read!(joinpath(dir, name * ".f64"), a)
    return a
end

function write_array(dir, name, field)
    a = Array(interior(field))[:, :, 1]
    open(joinpath(dir, name * ".f64"), "w") do io
        write(io, Float64.(a))
    end
    return size(a)
end

topo_of(s) = s == "periodic" ? Periodic : Bounded

function build_grid(d)
    Nx, Ny, H = geti(d, "Nx"), geti(d, "Ny"), geti(d, "H")
    topology = (topo_of(d["topo_x"]), topo_of(d["topo_y"]), Flat)
    if d["grid"] == "rectilinear"
        dx = getf(d, "spacing")
        grid = RectilinearGrid(CPU(); size = (Nx, Ny), x = (0, Nx * dx), y = (0, Ny * dx), halo = (H, H), topology)
    else
        grid = LatitudeLongitudeGrid(CPU(); size = (Nx, Ny), longitude = (0, 60), latitude = (20, 70), halo = (H, H), topology)
    end
    if haskey(d, "mask")
        wet = Array{UInt8}(undef, Nx, Ny)
        read!(joinpath(d["dir"], "mask.u8"), wet)
        # GridFittedBoundary takes the IMMERSED (dry) cells
        grid = ImmersedBoundaryGrid(grid, GridFittedBoundary(reshape(wet .== 0, Nx, Ny, 1)))
    end
    return grid
end

function build_model(d, grid; substeps, timestepper, advection)
    dir = d["dir"]
    Nx, Ny = geti(d, "Nx"), geti(d, "Ny")
    coriolis = haskey(d, "coriolis") ? (haskey(d, "beta") ? BetaPlane(f₀ = getf(d, "coriolis"), β = getf(d, "beta")) :
                                                            FPlane(f = getf(d, "coriolis"))) : nothing
    rheology = d["pressure"] == "replacement" ? ElastoViscoPlasticRheology() :
               ElastoViscoPlasticRheology(pressure_formulation = ClimaSeaIce.Rheologies.IceStrength())
    if getb(d, "field_forcing")
        τu = XFaceField(grid); τv = YFaceField(grid); uₑ = XFaceField(grid); vₑ = YFaceField(grid)
        for (f, name) in ((τu, "top_u"), (τv, "top_v"), (uₑ, "ue"), (vₑ, "ve"))
            nx, ny, _ = size(f)
            set!(f, read_array(dir, name, nx, ny))
        end
        top = (u = τu, v = τv)
        bottom = SemiImplicitStress(; uₑ, vₑ)
    else
        top = haskey(d, "top_u") ? (u = getf(d, "top_u"), v = getf(d, "top_v")) : nothing
        if d["bottom"] == "semi"
            ue, ve = getf(d, "ue"), getf(d, "ve")
            kw = NamedTuple()
            ue != 0 && (kw = merge(kw, (; uₑ = Oceananigans.Fields.ConstantField(ue))))
            ve != 0 && (kw = merge(kw, (; vₑ = Oceananigans.Fields.ConstantField(ve))))
            bottom = SemiImplicitStress(; kw...)
        else
            bottom = nothing
        end
    end
    free_drift = getb(d, "free_drift") ? StressBalanceFreeDrift() : nothing
    dynamics = SeaIceMomentumEquation(grid; coriolis, rheology, top_momentum_stress = top, bottom_momentum_stress = bottom,
                                      free_drift, solver = SplitExplicitSolver(grid; substeps))
    bcs = NamedTuple()
    if getb(d, "noslip")      # examples/ice_advected_on_coastline.jl:96-99
        u_bcs = FieldBoundaryConditions(north = ValueBoundaryCondition(0), south = ValueBoundaryCondition(0))
        v_bcs = FieldBoundaryConditions(west = ValueBoundaryCondition(0), east = ValueBoundaryCondition(0))
        bcs = (u = u_bcs, v = v_bcs)
    end
    model = SeaIceModel(grid; dynamics, advection, timestepper, ice_thermodynamics = nothing, boundary_conditions = bcs)
    h = read_array(dir, "in_h", Nx, Ny); a = read_array(dir, "in_a", Nx, Ny)
    nxu, nyu, _ = size(model.velocities.u); nxv, nyv, _ = size(model.velocities.v)
    u = read_array(dir, "in_u", nxu, nyu); v = read_array(dir, "in_v", nxv, nyv)
    set!(model, h = h, ℵ = a, u = u, v = v)
    return model
end

function dump_state(dir, model, tag)
    aux = model.dynamics.auxiliaries.fields
    write_array(dir, "out_u_" * tag, model.velocities.u)
    write_array(dir, "out_v_" * tag, model.velocities.v)
    write_array(dir, "out_h_" * tag, model.ice_thickness)
    write_array(dir, "out_a_" * tag, model.ice_concentration)
    for (name, f) in (("s11", aux.σ₁₁), ("s22", aux.σ₂₂), ("s12", aux.σ₁₂), ("alpha", aux.α), ("P", aux.P),
                      ("zeta_c", aux.ζᶜᶜᶜ), ("zeta_f", aux.ζᶠᶠᶜ), ("Delta", aux.Δ))
        write_array(dir, "out_" * name * "_" * tag, f)
    end
end

function run_case(dir)
    d = read_manifest(joinpath(dir, "case.txt"))
    d["dir"] = dir
    dt = getf(d, "dt")
    grid = build_grid(d)
    # (1) the EVP sub-cycle alone: time_step_momentum! with 1 and 10 sub-steps from the same initial state
    for nsub in (1, 10)
        model = build_model(d, grid; substeps = nsub, timestepper = :ForwardEuler, advection = nothing)
        time_step_momentum!(model, model.dynamics, dt)
        dump_state(dir, model, "momentum$(nsub)")
    end
    # (2) whole model steps: 3 x time_step! with WENO(order = 7), both time steppers, 10 sub-steps
    for (ts, tag) in ((:ForwardEuler, "fe"), (:SplitRungeKutta3, "rk3"))
        model = build_model(d, grid; substeps = 10, timestepper = ts, advection = WENO(order = 7))
        for _ in 1:3
            time_step!(model, dt)
        end
        dump_state(dir, model, "step3_" * tag)
    end
    open(joinpath(dir, "DONE"), "w") do io
        println(io, "ClimaSeaIce ", pkgversion(ClimaSeaIce), " Oceananigans ", pkgversion(Oceananigans), " julia ", VERSION)
        # which arithmetic the advection scheme ran in (round 5): the full type -- a second float type parameter, where this upstream
        # version has one, is the precision of the smoothness / weight arithmetic (the library's weight_dtype f64 | f32) -- and every
        # field's type, so that tests/golden/reference_io.py can pick the matching oracle mode when it imports these outputs
        scheme = WENO(order = 7)
        println(io, "advection_type ", typeof(scheme))
        floats = [p for p in typeof(scheme).parameters if p isa Type && p <: AbstractFloat]
        println(io, "advection_float_parameters ", join(string.(floats), " "))
        println(io, "weight_dtype ", (length(floats) >= 2 && floats[2] === Float32) ? "f32" : "f64")
        for name in fieldnames(typeof(scheme))
            println(io, "advection_field ", name, " :: ", typeof(getfield(scheme, name)))
        end
    end
end

function main()
    root = length(ARGS) >= 1 ? ARGS[1] : "csi_cases"
    for name in sort(readdir(root))
        dir = joinpath(root, name)
        isfile(joinpath(dir, "case.txt")) || continue
        @info "reference run" name
        run_case(dir)
    end
end

main()
