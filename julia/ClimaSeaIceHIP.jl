# ClimaSeaIceHIP.jl -- the Julia-side binding a ClimaSeaIce.jl maintainer adds to route the hot path
# through libcsi_hip.so (include/csi.h).  NOT executed in this repository: the build image has no Julia
# and no Oceananigans (SURVEY.md section 0); it is kept thin and mechanical so it can be checked by eye
# against include/csi.h.  Everything outside these methods (SeaIceModel construction, set!, Simulation,
# output writers, checkpointing) is untouched: the fields stay Oceananigans Fields whose parents are
# AMDGPU.jl ROCArrays, and the library only receives their device pointers.
module ClimaSeaIceHIP

using ClimaSeaIce
using ClimaSeaIce: SeaIceModel
using ClimaSeaIce.SeaIceDynamics: SeaIceMomentumEquation, SplitExplicitSolver, SemiImplicitStress, StressBalanceFreeDrift
using ClimaSeaIce.Rheologies: ElastoViscoPlasticRheology, ReplacementPressure
using Oceananigans
using Oceananigans: CPU
using Oceananigans.Architectures: architecture
using Oceananigans.BoundaryConditions: FluxBoundaryCondition, getbc
using Oceananigans.Coriolis: FPlane, BetaPlane
using Oceananigans.DistributedComputations: Distributed
using Oceananigans.Fields: Field, ZeroField, ConstantField
using Oceananigans.Grids: topology, halo_size, Periodic, Bounded, Center, Face, RectilinearGrid, LatitudeLongitudeGrid,
                          OrthogonalSphericalShellGrid, ynode, inactive_cell
using Oceananigans.ImmersedBoundaries: ImmersedBoundaryGrid, ImmersedBoundaryCondition
using Oceananigans.OrthogonalSphericalShellGrids: TripolarGrid
using Oceananigans.TimeSteppers: SplitRungeKuttaTimeStepper
using AMDGPU
using MPI

const libcsi = get(ENV, "LIBCSI_HIP", "libcsi_hip.so")

# ---- plain-C structs of include/csi.h --------------------------------------------------------------
struct CsiMetrics
    dx::Cdouble; dy::Cdouble
    dxc::Ptr{Cdouble}; dxf::Ptr{Cdouble}; azc::Ptr{Cdouble}; azf::Ptr{Cdouble}
    full::NTuple{12, Ptr{Cdouble}}; full_ld::Int64          # CSI_METRIC_FULL: twelve host arrays (include/csi.h)
end
CsiMetrics(dx, dy, dxc, dxf, azc, azf) = CsiMetrics(dx, dy, dxc, dxf, azc, azf, ntuple(_ -> Ptr{Cdouble}(C_NULL), 12), 0)
struct CsiEvpParams
    ice_compressive_strength::Cdouble; ice_compaction_hardening::Cdouble; yield_curve_eccentricity::Cdouble
    minimum_plastic_stress::Cdouble; min_relaxation_parameter::Cdouble; max_relaxation_parameter::Cdouble
    relaxation_strength::Cdouble; pressure_formulation::Int32; has_coriolis::Int32; coriolis_f::Cdouble
    minimum_concentration::Cdouble; minimum_mass::Cdouble; sea_ice_density::Cdouble
end
struct CsiStress
    kind::Int32; ue_kind::Int32; ve_kind::Int32; reserved::Int32
    tau_u::Cdouble; tau_v::Cdouble; ue::Cdouble; ve::Cdouble; rho_e::Cdouble; Cd::Cdouble
end

# field slots, in the order of csi_field_id
const F = (U=0, V=1, H=2, A=3, S11=4, S22=5, S12=6, UN=7, VN=8, P=9, ALPHA=10, DELTA=11, ZETA_F=12, ZETA_C=13,
           GH=14, GA=15, HM=16, AM=17, UM=18, VM=19, TOP_U=20, TOP_V=21, BOT_U=22, BOT_V=23, MASS_FLUX=24,
           HS=25, GHS=26, HSM=27, MASS_FLUX_SNOW=28, SNOWFALL_INTERCEPTED=29, TU=30, TUS=31, FORCING_U=32, FORCING_V=33)

mutable struct Context
    handle::Ptr{Cvoid}
    mask::Any            # UInt8 activity mask of an immersed grid (owned here so that it outlives the library's pointer)
    validated_iteration::Int      # Distributed grids: the clock iteration whose state csi_validate_all has last checked on every rank
    Context(handle) = new(handle, nothing, -1)
end

function check(ctx, rc)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:csi_last_error, libcsi), Cstring, (Ptr{Cvoid},), ctx === nothing ? C_NULL : ctx.handle))
    error("libcsi_hip error $rc: $msg")
end

function Context(device_id = AMDGPU.device_id(AMDGPU.device()) - 1; stream = AMDGPU.stream())
    h = Ref{Ptr{Cvoid}}(C_NULL)
    # pass AMDGPU.jl's HIP stream so library work is ordered with the rest of the Julia program
    rc = ccall((:csi_context_create, libcsi), Int32, (Int32, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), device_id, stream.stream, h)
    check(nothing, rc)
    ctx = Context(h[])
    finalizer(c -> ccall((:csi_context_destroy, libcsi), Int32, (Ptr{Cvoid},), c.handle), ctx)
    return ctx
end

# csi_topology: 0 Periodic, 1 Bounded, 2 FullyConnected, 3 LeftConnected, 4 RightConnected, 5 RightFolded (the y direction
# of a TripolarGrid: south wall + north fold / Zipper), 6 LeftConnected + RightFolded (northernmost rank of a y partition).
# The fold the library implements has its pivot on the cell CENTRES of row Ny (Center-in-y fields store that row twice:
# include/csi.h): Oceananigans' RightCenterFolded / LeftConnectedRightCenterFolded, the topologies the reference lists at
# SeaIceDynamics/split_explicit_momentum_equations.jl:7-16.  The *FaceFolded variants (pivot on the faces) and the
# LeftConnectedRight*Connected ones (a fold that crosses ranks: x-partitioned tripolar grids) are refused, not approximated.
const OG = Oceananigans.Grids
topo_code(::Type{Periodic}) = Int32(0)
topo_code(::Type{Bounded}) = Int32(1)
topo_code(::Type{OG.FullyConnected}) = Int32(2)
topo_code(::Type{OG.LeftConnected}) = Int32(3)
topo_code(::Type{OG.RightConnected}) = Int32(4)
if isdefined(OG, :RightCenterFolded)
    topo_code(::Type{OG.RightCenterFolded}) = Int32(5)
    topo_code(::Type{OG.LeftConnectedRightCenterFolded}) = Int32(6)
    for T in (:RightFaceFolded, :LeftConnectedRightFaceFolded, :LeftConnectedRightCenterConnected, :LeftConnectedRightFaceConnected)
        isdefined(OG, T) && @eval topo_code(::Type{OG.$T}) =
            error("ClimaSeaIceHIP: y topology " * $(string(T)) * " is not supported (only the Center-pivot fold on a y-partitioned or unpartitioned grid)")
    end
end
topo_code(T) = error("ClimaSeaIceHIP: unknown topology $T")
# Older Oceananigans (no *Folded topologies): a TripolarGrid reports Bounded / RightConnected / FullyConnected / LeftConnected in
# y and the fold is implied by the grid type; the rank that owns the fold is the last one of the y partition.
y_topo_code(grid, arch, TY) = topo_code(TY)
function y_topo_code(grid::TripolarGrid, arch, TY)
    isdefined(OG, :RightCenterFolded) && return topo_code(TY)
    Ry = arch isa Distributed ? arch.ranks[2] : 1
    ry = arch isa Distributed ? arch.local_index[2] : 1
    ry == Ry && return Ry == 1 ? Int32(5) : Int32(6)           # the northernmost rank holds the fold
    return ry == 1 ? Int32(4) : Int32(2)                        # southern rank: wall + connected; middle ranks: connected
end
y_topo_code(grid::ImmersedBoundaryGrid, arch, TY) = y_topo_code(grid.underlying_grid, arch, TY)

# Oceananigans parent array: column-major (ni, nj, 1); ld = ni
function bind!(ctx, slot, field)
    p = parent(field)
    ni, nj = size(p, 1), size(p, 2)
    rc = ccall((:csi_field_bind, libcsi), Int32, (Ptr{Cvoid}, Int32, Ptr{Cvoid}, Int64, Int32, Int32),
               ctx.handle, slot, pointer(p), ni, ni, nj)
    check(ctx, rc)
end

function set_grid!(ctx, grid::RectilinearGrid)
    Nx, Ny, _ = size(grid); Hx, Hy, _ = halo_size(grid); TX, TY, _ = topology(grid)
    m = Ref(CsiMetrics(grid.Δxᶜᵃᵃ, grid.Δyᵃᶜᵃ, C_NULL, C_NULL, C_NULL, C_NULL))
    check(ctx, ccall((:csi_grid_set, libcsi), Int32, (Ptr{Cvoid}, Int32, Int32, Int32, Int32, Int32, Int32, Int32, Ref{CsiMetrics}),
                     ctx.handle, Nx, Ny, Hx, Hy, topo_code(TX), topo_code(TY), 0, m))
end

function set_grid!(ctx, grid::LatitudeLongitudeGrid)
    Nx, Ny, _ = size(grid); Hx, Hy, _ = halo_size(grid); TX, TY, _ = topology(grid)
    rows = (1 - Hy):(Ny + Hy + 1)
    dxc = Array(grid.Δxᶜᶜᵃ[rows]); dxf = Array(grid.Δxᶜᶠᵃ[rows]); azc = Array(grid.Azᶜᶜᵃ[rows]); azf = Array(grid.Azᶜᶠᵃ[rows])
    GC.@preserve dxc dxf azc azf begin
        m = Ref(CsiMetrics(0.0, grid.Δyᶜᶠᵃ, pointer(dxc), pointer(dxf), pointer(azc), pointer(azf)))
        check(ctx, ccall((:csi_grid_set, libcsi), Int32, (Ptr{Cvoid}, Int32, Int32, Int32, Int32, Int32, Int32, Int32, Ref{CsiMetrics}),
                         ctx.handle, Nx, Ny, Hx, Hy, topo_code(TX), topo_code(TY), 1, m))
    end
end

# Orthogonal curvilinear grids (OrthogonalSphericalShellGrid: 2-D metric arrays): the twelve metrics at the four
# horizontal locations, evaluated with the public operators over the halo-extended index range and copied to the host once.
set_grid!(ctx, grid::ImmersedBoundaryGrid) = set_grid!(ctx, grid.underlying_grid)       # the mask goes through csi_mask_set

function set_grid!(ctx, grid::OrthogonalSphericalShellGrid)
    Nx, Ny, _ = size(grid); Hx, Hy, _ = halo_size(grid); TX, TY, _ = topology(grid)
    is, js = (1 - Hx):(Nx + Hx + 1), (1 - Hy):(Ny + Hy + 1)
    ops = (Oceananigans.Operators.Δxᶜᶜᶜ, Oceananigans.Operators.Δxᶠᶜᶜ, Oceananigans.Operators.Δxᶜᶠᶜ, Oceananigans.Operators.Δxᶠᶠᶜ,
           Oceananigans.Operators.Δyᶜᶜᶜ, Oceananigans.Operators.Δyᶠᶜᶜ, Oceananigans.Operators.Δyᶜᶠᶜ, Oceananigans.Operators.Δyᶠᶠᶜ,
           Oceananigans.Operators.Azᶜᶜᶜ, Oceananigans.Operators.Azᶠᶜᶜ, Oceananigans.Operators.Azᶜᶠᶜ, Oceananigans.Operators.Azᶠᶠᶜ)
    cpu_grid = Oceananigans.on_architecture(CPU(), grid)
    inside(i, j) = i <= Nx + Hx && j <= Ny + Hy                       # the +1 column / row exists for Face points on a wall only
    arrays = [Float64[inside(i, j) ? op(i, j, 1, cpu_grid) : 1.0 for i in is, j in js] for op in ops]
    GC.@preserve arrays begin
        m = Ref(CsiMetrics(0.0, 0.0, C_NULL, C_NULL, C_NULL, C_NULL, ntuple(k -> pointer(arrays[k]), 12), length(is)))
        check(ctx, ccall((:csi_grid_set, libcsi), Int32, (Ptr{Cvoid}, Int32, Int32, Int32, Int32, Int32, Int32, Int32, Ref{CsiMetrics}),
                         ctx.handle, Nx, Ny, Hx, Hy, topo_code(TX), y_topo_code(grid, architecture(grid), TY), 2, m))
    end
end

# activity mask of an immersed grid: 1 = active cell, parent shape of a (Center, Center) field, halos included
function active_cells_mask(grid::ImmersedBoundaryGrid)
    Nx, Ny, Nz = size(grid); Hx, Hy, _ = halo_size(grid)
    cpu_grid = Oceananigans.on_architecture(CPU(), grid)
    mask = UInt8[inactive_cell(i, j, Nz, cpu_grid) ? 0x00 : 0x01 for i in (1 - Hx):(Nx + Hx), j in (1 - Hy):(Ny + Hy)]
    return ROCArray(mask)
end

stress_struct(::Nothing) = CsiStress(0, 0, 0, 0, 0, 0, 0, 0, 0, 0)
stress_struct(τ::NamedTuple{(:u, :v), <:Tuple{Number, Number}}) = CsiStress(1, 0, 0, 0, τ.u, τ.v, 0, 0, 0, 0)
# stress given as Fields at the (Face, Center) / (Center, Face) points (the coupled-model case, sea_ice_external_stress.jl:19-20):
# CSI_STRESS_FIELD; the arrays are bound to CSI_F_TOP_U / _V or CSI_F_BOT_U / _V in attach!
stress_struct(τ::NamedTuple{(:u, :v), <:Tuple{Field, Field}}) = CsiStress(2, 0, 0, 0, 0, 0, 0, 0, 0, 0)
function stress_struct(τ::SemiImplicitStress)
    kind(x) = x isa Oceananigans.Fields.ZeroField ? Int32(0) : x isa Oceananigans.Fields.ConstantField ? Int32(1) : Int32(2)
    val(x) = x isa Oceananigans.Fields.ConstantField ? Float64(x.constant) : 0.0
    return CsiStress(3, kind(τ.uₑ), kind(τ.vₑ), 0, 0, 0, val(τ.uₑ), val(τ.vₑ), τ.ρₑ, τ.Cᴰ)
end

"""
    attach!(model) -> Context

Describe `model` to the library once (grid, fields, parameters).  Field memory stays owned by Julia.
"""
function attach!(model::SeaIceModel)
    ctx = Context()
    dyn = model.dynamics
    grid = model.velocities.u.grid
    set_grid!(ctx, grid)
    a = dyn.auxiliaries.fields
    for (slot, f) in ((F.U, model.velocities.u), (F.V, model.velocities.v), (F.H, model.ice_thickness),
                      (F.A, model.ice_concentration), (F.S11, a.σ₁₁), (F.S22, a.σ₂₂), (F.S12, a.σ₁₂), (F.UN, a.uⁿ),
                      (F.VN, a.vⁿ), (F.P, a.P), (F.ALPHA, a.α), (F.DELTA, a.Δ), (F.ZETA_F, a.ζᶠᶠᶜ), (F.ZETA_C, a.ζᶜᶜᶜ),
                      (F.GH, model.timestepper.Gⁿ.h), (F.GA, model.timestepper.Gⁿ.ℵ))
        bind!(ctx, slot, f)
    end
    if model.timestepper isa SplitRungeKuttaTimeStepper
        Ψ = model.timestepper.Ψ⁻
        bind!(ctx, F.HM, Ψ.h); bind!(ctx, F.AM, Ψ.ℵ); bind!(ctx, F.UM, Ψ.u); bind!(ctx, F.VM, Ψ.v)
    end
    # snow layer: hs is advected and updated with h and ℵ (tracer_tendency_kernel_functions.jl:49-52,
    # sea_ice_fe_step.jl:86-94); the thermodynamic step itself stays with the Julia kernel unless csi_snow_params_set is used
    if !isnothing(model.snow_thickness)
        bind!(ctx, F.HS, model.snow_thickness); bind!(ctx, F.GHS, model.timestepper.Gⁿ.hs)
        model.timestepper isa SplitRungeKuttaTimeStepper && bind!(ctx, F.HSM, model.timestepper.Ψ⁻.hs)
    end
    # diagnostics update_state! masks on immersed grids (sea_ice_model.jl:386-389)
    mf = model.mass_fluxes
    bind!(ctx, F.MASS_FLUX, mf.thermodynamics.ice); bind!(ctx, F.MASS_FLUX_SNOW, mf.thermodynamics.snow)
    bind!(ctx, F.SNOWFALL_INTERCEPTED, mf.intercepted_snowfall)
    r = dyn.rheology
    cor = dyn.coriolis
    p = Ref(CsiEvpParams(r.ice_compressive_strength, r.ice_compaction_hardening, r.yield_curve_eccentricity,
                         r.minimum_plastic_stress, r.min_relaxation_parameter, r.max_relaxation_parameter,
                         r.relaxation_strength, r.pressure_formulation isa ReplacementPressure ? 0 : 1,
                         isnothing(cor) ? 0 : 1, cor isa FPlane ? cor.f : 0.0,
                         dyn.minimum_concentration, dyn.minimum_mass, model.sea_ice_density[1, 1, 1]))
    check(ctx, ccall((:csi_evp_params_set, libcsi), Int32, (Ptr{Cvoid}, Ref{CsiEvpParams}), ctx.handle, p))
    if cor isa BetaPlane
        # f = f₀ + β y at the (Face, Center) / (Center, Face) nodes of every row, halo rows included (ynode follows the
        # halo of a distributed or periodic grid, so ring rows see their owner's value); host vectors like the metrics
        Hy, Ny = grid.Hy, grid.Ny
        rows = (1 - Hy):(Ny + Hy + 1)
        fu = Float64[cor.f₀ + cor.β * ynode(1, wrap_row(grid, j), 1, grid, Face(), Center(), Center()) for j in rows]
        fv = Float64[cor.f₀ + cor.β * ynode(1, wrap_row(grid, j), 1, grid, Center(), Face(), Center()) for j in rows]
        check(ctx, ccall((:csi_coriolis_rows_set, libcsi), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int32),
                         ctx.handle, fu, fv, length(fu)))
    elseif !(isnothing(cor) || cor isa FPlane)
        error("ClimaSeaIceHIP: coriolis must be nothing, FPlane or BetaPlane")
    end
    for (side, τ) in ((0, dyn.external_momentum_stresses.top), (1, dyn.external_momentum_stresses.bottom))
        s = Ref(stress_struct(τ))
        check(ctx, ccall((:csi_stress_set, libcsi), Int32, (Ptr{Cvoid}, Int32, Ref{CsiStress}), ctx.handle, side, s))
        if τ isa SemiImplicitStress    # field-valued external velocities
            τ.uₑ isa Field && bind!(ctx, side == 0 ? F.TOP_U : F.BOT_U, τ.uₑ)
            τ.vₑ isa Field && bind!(ctx, side == 0 ? F.TOP_V : F.BOT_V, τ.vₑ)
        elseif τ isa NamedTuple && τ.u isa Field      # stress arrays
            bind!(ctx, side == 0 ? F.TOP_U : F.BOT_U, τ.u)
            bind!(ctx, side == 0 ? F.TOP_V : F.BOT_V, τ.v)
        end
    end
    # free_drift = StressBalanceFreeDrift(...): the library rebuilds the balance on the model's own stresses, like
    # materialize_free_drift (stress_balance_free_drift.jl:44-46)
    check(ctx, ccall((:csi_free_drift_set, libcsi), Int32, (Ptr{Cvoid}, Int32), ctx.handle,
                     dyn.free_drift isa StressBalanceFreeDrift ? 1 : 0))
    # model.forcing.u / .v given as Fields (arrays): the user forcing of sum_of_forcing_u / _v (elasto_visco_plastic_rheology.jl:391-401);
    # closures cannot cross a C ABI and keep the model on the Julia kernels
    if model.forcing.u isa Field && model.forcing.v isa Field
        bind!(ctx, F.FORCING_U, model.forcing.u); bind!(ctx, F.FORCING_V, model.forcing.v)
    end
    # immersed FluxBoundaryConditions of u and v with number values (ice_stress_divergence.jl:65-123)
    for (slot, f) in ((F.U, model.velocities.u), (F.V, model.velocities.v))
        ibc = f.boundary_conditions.immersed
        if ibc isa ImmersedBoundaryCondition
            val(bc) = bc isa FluxBoundaryCondition && bc.condition isa Number ? Float64(bc.condition) : 0.0
            check(ctx, ccall((:csi_immersed_flux_bc_set, libcsi), Int32, (Ptr{Cvoid}, Int32, Cdouble, Cdouble, Cdouble, Cdouble),
                             ctx.handle, slot, val(ibc.west), val(ibc.east), val(ibc.south), val(ibc.north)))
        end
    end
    # tiles of a Distributed grid: one rank per GPU, RCCL point-to-point halos (csi_halo_exchange)
    arch = architecture(grid)
    arch isa Distributed && attach_tiles!(ctx, arch, grid)
    # immersed boundary: the activity mask (1 = active) as a UInt8 ROCArray with the parent shape of a Center field
    if grid isa ImmersedBoundaryGrid
        ctx.mask = active_cells_mask(grid)                       # kept alive by the context
        check(ctx, ccall((:csi_mask_set, libcsi), Int32, (Ptr{Cvoid}, Ptr{UInt8}, Int64), ctx.handle, pointer(ctx.mask), size(ctx.mask, 1)))
    end
    check(ctx, ccall((:csi_set_mode, libcsi), Int32, (Ptr{Cvoid}, Int32), ctx.handle, 1))   # CSI_MODE_FAST
    check(ctx, ccall((:csi_set_fusion, libcsi), Int32, (Ptr{Cvoid}, Int32), ctx.handle, 2)) # two sub-steps per launch (default)
    return ctx
end

# Distributed(arch; partition = Partition(Rx, Ry)): this rank's tile, the RCCL communicator (unique id made on rank 0 and
# broadcast over MPI) -- after which csi_time_step_momentum exchanges u, v (, sigma) itself and the 2 * substeps + 3 halo of
# split_explicit_momentum_equations.jl:51-64 is not needed (halo >= 2 k with csi_set_exchange_interval(k))
function attach_tiles!(ctx, arch::Distributed, grid)
    Rx, Ry, _ = arch.ranks
    rx, ry, _ = arch.local_index .- 1
    TX, TY, _ = topology(grid isa ImmersedBoundaryGrid ? grid.underlying_grid : grid)
    px = TX === Periodic || TX === Oceananigans.Grids.FullyConnected && arch.connectivity.west !== nothing
    py = TY === Periodic
    check(ctx, ccall((:csi_tile_set, libcsi), Int32, (Ptr{Cvoid}, Int32, Int32, Int32, Int32, Int32, Int32),
                     ctx.handle, rx, ry, Rx, Ry, px ? 1 : 0, py ? 1 : 0))
    id = zeros(UInt8, 128)
    arch.local_rank == 0 && check(ctx, ccall((:csi_comm_unique_id, libcsi), Int32, (Ptr{UInt8},), id))
    MPI.Bcast!(id, 0, arch.communicator)
    check(ctx, ccall((:csi_comm_init, libcsi), Int32, (Ptr{Cvoid}, Int32, Int32, Ptr{UInt8}), ctx.handle, Rx * Ry, arch.local_rank, id))
end

# Inside the sub-cycle the library moves the halos by peer-direct stores over xGMI by default (it maps its neighbours' arrays over HIP
# IPC itself, at the first sub-cycle, through the communicator given above: halo 4 suffices); `set_halo_transport!(ctx, :rccl)`
# selects the k-batched RCCL exchange instead (halo >= 2 k).  include/csi.h: csi_set_halo_transport.
function set_halo_transport!(ctx, kind::Symbol)
    kind in (:peer, :rccl) || throw(ArgumentError("halo transport :peer or :rccl"))
    check(ctx, ccall((:csi_set_halo_transport, libcsi), Int32, (Ptr{Cvoid}, Int32), ctx.handle, kind === :peer ? 1 : 0))
end

# The memory-ordering tier of the peer transport's flag protocol, the SAME on every rank: -1 automatic (the default: tier 1 whenever a
# neighbour lives in another process or on another device), 0 no fence (explicit opt-in there), 1: + an acquire fence behind the
# flags, 2: + a release fence before them.  include/csi.h: csi_set_peer_tier.
set_peer_tier!(ctx, tier::Integer) = check(ctx, ccall((:csi_set_peer_tier, libcsi), Int32, (Ptr{Cvoid}, Int32), ctx.handle, tier))

# Round 6: the two exact structure cuts (include/csi.h).  Both are on by default; nothing has to be called.
#   set_tile_skipping!(ctx, on): tiles with no ice mass in or around them are left out of the inner launches of a sub-cycle
#   tile_activity(ctx) -> (tiles, live, used): the newest counts that have arrived from the device
#   set_row_constant!(ctx, on, rtol = 0.0): rows of a TripolarGrid's latitude-longitude part whose metric planes hold one value per row are
#       read from per-row vectors; rtol > 0 also marks rows whose columns agree to that relative distance (Oceananigans computes a
#       tripolar grid's metrics per point: its lat-lon rows may carry rounding noise) -- that changes results at the rtol level
set_tile_skipping!(ctx, on::Bool) = check(ctx, ccall((:csi_set_tile_skipping, libcsi), Int32, (Ptr{Cvoid}, Int32), ctx.handle, on ? 1 : 0))
function tile_activity(ctx)
    t, l, u = Ref{Int32}(0), Ref{Int32}(0), Ref{Int32}(0)
    check(ctx, ccall((:csi_tile_activity, libcsi), Int32, (Ptr{Cvoid}, Ref{Int32}, Ref{Int32}, Ref{Int32}), ctx.handle, t, l, u))
    return (tiles = t[], live = l[], used = u[] != 0)
end
set_row_constant!(ctx, on::Bool, rtol::Real = 0.0) =
    check(ctx, ccall((:csi_set_row_constant, libcsi), Int32, (Ptr{Cvoid}, Int32, Cdouble), ctx.handle, on ? 1 : 0, rtol))
function row_constant_rows(ctx)
    n = Ref{Int32}(0)
    check(ctx, ccall((:csi_row_constant_rows, libcsi), Int32, (Ptr{Cvoid}, Ref{Int32}), ctx.handle, n))
    return Int(n[])
end

# Waits for the library's stream and reports a peer-transport wait that gave up (a rank that fell behind or died: the library
# never hangs, it fails).  Every entry point that advances the model reports such an error too, at its start and at its end, so an
# explicit call is needed only where the host must KNOW that a step is complete and valid (before output, before a checkpoint).
synchronize!(ctx) = check(ctx, ccall((:csi_sync, libcsi), Int32, (Ptr{Cvoid},), ctx.handle))

# One PROCESS per tile without RCCL (several ranks on one GPU: RCCL refuses that): a host-channel group over a POSIX shared-memory
# segment every rank names alike; the peer transport maps the neighbours' arrays over HIP IPC as under RCCL.  include/csi.h.
function attach_tiles_host!(ctx, shm_name::AbstractString, rank::Integer, rx, ry, Rx, Ry, px::Bool, py::Bool)
    check(ctx, ccall((:csi_tile_set, libcsi), Int32, (Ptr{Cvoid}, Int32, Int32, Int32, Int32, Int32, Int32),
                     ctx.handle, rx, ry, Rx, Ry, px ? 1 : 0, py ? 1 : 0))
    check(ctx, ccall((:csi_comm_init_host, libcsi), Int32, (Ptr{Cvoid}, Cstring, Int32, Int32), ctx.handle, shm_name, Rx * Ry, rank))
end

# Several tiles driven by ONE process (one task per tile; RCCL refuses two ranks on one device): an in-process tile group instead of
# the RCCL communicator.  `group = local_tile_group(Rx * Ry)` once, then `attach_tiles_local!(ctx, group, rank, ...)` per tile.
function local_tile_group(world::Integer)
    g = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:csi_local_group_create, libcsi), Int32, (Int32, Ptr{Ptr{Cvoid}}), world, g)
    rc == 0 || error("csi_local_group_create failed ($rc)")
    return g[]
end
function attach_tiles_local!(ctx, group::Ptr{Cvoid}, rank::Integer, rx, ry, Rx, Ry, px::Bool, py::Bool)
    check(ctx, ccall((:csi_tile_set, libcsi), Int32, (Ptr{Cvoid}, Int32, Int32, Int32, Int32, Int32, Int32),
                     ctx.handle, rx, ry, Rx, Ry, px ? 1 : 0, py ? 1 : 0))
    check(ctx, ccall((:csi_comm_init_local, libcsi), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int32), ctx.handle, group, rank))
end

# row whose value a halo row images: the wrapped row of a Periodic y direction, the row itself otherwise
wrap_row(grid, j) = topology(grid, 2) == Periodic ? mod1(j, grid.Ny) : j

# One context per model, created on first use.
const CONTEXTS = IdDict{Any, Context}()
context(model) = get!(() -> attach!(model), CONTEXTS, model)

# ---- a solver tag that selects the HIP path by dispatch -------------------------------------------
"""
    HIPSplitExplicitSolver(; substeps = 120)

Drop-in for `SplitExplicitSolver`: `SeaIceMomentumEquation(grid; solver = HIPSplitExplicitSolver())`.
"""
struct HIPSplitExplicitSolver
    substeps::Int
end
HIPSplitExplicitSolver(; substeps = 120) = HIPSplitExplicitSolver(substeps)
const HIPMomentumEquation = SeaIceMomentumEquation{<:HIPSplitExplicitSolver}
# SeaIceModel{GR, TD, SNT, D, TS, ...} (sea_ice_model.jl:22): the dynamics are the FOURTH type parameter, the time stepper the
# fifth.  The reference's own methods dispatch on the fifth (FESeaIceModel / RKSeaIceModel, sea_ice_fe_step.jl:9,
# sea_ice_rk_substep.jl:6), so the HIP methods are defined on the intersections -- more specific than either, no ambiguity.
const HIPSeaIceModel = SeaIceModel{<:Any, <:Any, <:Any, <:HIPMomentumEquation}
const HIPFESeaIceModel = SeaIceModel{<:Any, <:Any, <:Any, <:HIPMomentumEquation, <:ClimaSeaIce.ForwardEulerTimeStepper}
const HIPRKSeaIceModel = SeaIceModel{<:Any, <:Any, <:Any, <:HIPMomentumEquation, <:SplitRungeKuttaTimeStepper}

# time_step_momentum!, SeaIceDynamics/split_explicit_momentum_equations.jl:103-195
function ClimaSeaIce.SeaIceDynamics.time_step_momentum!(model, dynamics::HIPMomentumEquation, Δt)
    ctx = context(model)
    rk = model.timestepper isa SplitRungeKuttaTimeStepper
    # GC.@preserve exactly as the reference does around its own loop (:150)
    GC.@preserve model begin
        check(ctx, ccall((:csi_time_step_momentum, libcsi), Int32, (Ptr{Cvoid}, Cdouble, Int32, Int32),
                         ctx.handle, Δt, dynamics.solver.substeps, rk ? 1 : 0))
    end
    return nothing
end

# compute_tracer_tendencies!, tracer_tendency_kernel_functions.jl:9-25
function ClimaSeaIce.compute_tracer_tendencies!(model::HIPSeaIceModel)
    ctx = context(model)
    order = Oceananigans.Advection.required_halo_size_x(model.advection) == 4 ? 7 : 5     # WENO(order = 7 | 5)
    # precision of the scheme's smoothness / weight arithmetic: f64 unless the user opts in (WENO_WEIGHT_DTYPE below; the library has
    # both modes, include/csi.h: csi_set_weno_weight_dtype, and STRICT equals its oracle bit for bit in either)
    check(ctx, ccall((:csi_set_weno_weight_dtype, libcsi), Int32, (Ptr{Cvoid}, Int32), ctx.handle, weight_dtype(model.advection)))
    GC.@preserve model check(ctx, ccall((:csi_compute_tracer_tendencies, libcsi), Int32, (Ptr{Cvoid}, Int32), ctx.handle, order))
    return nothing
end

# CSI_WEIGHTS_F64 (0) / CSI_WEIGHTS_F32 (1) from the scheme's type parameters: the first parameter that is a float type is FT, a
# second float type -- where the upstream version has one -- is FT2.  Versions with a single float type compute the weights in FT.
# Round 6 (ADVICE round 5): NOT selected automatically.  The f32 arithmetic of the library is a reading of upstream's FT2 that no
# reference run has confirmed (its weights are normalised by a float sum: a constant is reproduced to 6e-8 only), the reference
# pins Oceananigans 0.110 / 0.111, and the Python front end defaults to f64 -- so this stub does too, whatever the scheme's type
# parameters say.  Opt in with `ClimaSeaIceHIP.WENO_WEIGHT_DTYPE[] = :f32` (or :auto for the type-parameter rule below) once
# bench/reference_driver.jl's DONE file and golden outputs have shown which arithmetic the installed upstream runs.
const WENO_WEIGHT_DTYPE = Ref(:f64)
function weight_dtype(scheme)
    WENO_WEIGHT_DTYPE[] === :f64 && return Int32(0)
    WENO_WEIGHT_DTYPE[] === :f32 && return Int32(1)
    floats = [p for p in typeof(scheme).parameters if p isa Type && p <: AbstractFloat]      # :auto
    return (length(floats) >= 2 && floats[2] === Float32) ? Int32(1) : Int32(0)
end
weight_dtype(::Nothing) = Int32(0)

# dynamic_time_step!, sea_ice_fe_step.jl:36-50 (Forward Euler: from the current fields) and sea_ice_rk_substep.jl:134-152 (RK: from Ψ⁻)
function hip_dynamic_time_step!(model, Δt, from_cache)
    ctx = context(model)
    GC.@preserve model check(ctx, ccall((:csi_dynamic_step_tracers, libcsi), Int32, (Ptr{Cvoid}, Cdouble, Int32), ctx.handle, Δt, from_cache))
    return nothing
end
ClimaSeaIce.dynamic_time_step!(model::HIPFESeaIceModel, Δt) = hip_dynamic_time_step!(model, Δt, 0)
ClimaSeaIce.dynamic_time_step!(model::HIPRKSeaIceModel, Δt) = hip_dynamic_time_step!(model, Δt, 1)

# update_state!, sea_ice_model.jl:379-394 (a method of Oceananigans.TimeSteppers.update_state!, as in the reference):
# mask_immersed_field_xy! + fill_halo_regions! of the prognostic fields, the masks of the three mass_fluxes diagnostics
# (:386-389; bound as CSI_F_MASS_FLUX / _SNOW / SNOWFALL_INTERCEPTED in attach!, so csi_update_state masks them too), then
# update_model_field_time_series! (:391), which stays Julia.  On one rank csi_update_state does masks, local boundary
# conditions and the Zipper fold; on a Distributed grid it also sends the halos of h, aice, u, v (, hs) over RCCL with the full
# halo width -- the hand-off Oceananigans' MPI halo pass would do.
function Oceananigans.TimeSteppers.update_state!(model::HIPSeaIceModel, callbacks = [])
    ctx = context(model)
    GC.@preserve model check(ctx, ccall((:csi_update_state, libcsi), Int32, (Ptr{Cvoid},), ctx.handle))
    # No host synchronisation here (round 5; the round-4 stub drained the stream after every RK stage on Distributed grids, which
    # cost the host / device overlap of a whole stage and still proved nothing about ranks that are not direct neighbours: a
    # peer-transport abort reaches the neighbours' abort words only).  csi_update_state -- like every entry point that advances
    # the model -- reports a transport error it already knows of; `validate_state!` below is the collective check for the places
    # that need a state known good on EVERY rank (output, checkpoints).
    Oceananigans.Models.update_model_field_time_series!(model, model.clock)
    # ONCE PER STEP on a Distributed grid (round 6, ADVICE round 5): the first update_state! after the clock has ticked -- the one that
    # ends time_step! (sea_ice_fe_step.jl:30-31; upstream's SplitRungeKutta time_step! likewise) -- runs the collective check, so that
    # whatever reads the fields between two steps (output writers, checkpointers, callbacks) reads a state every rank knows to be
    # good: an aborted sub-cycle throws HERE, on every rank, not at the next step and only on the failed rank's neighbours.  One drain
    # of the stream and one all-reduce per step, not per stage.
    if architecture(model.grid) isa Distributed && model.clock.iteration != ctx.validated_iteration
        ctx.validated_iteration = model.clock.iteration
        validate_state!(model)
    end
    return nothing
end

# Before output writers / checkpointers read the fields of a Distributed model: drain this rank's stream and reduce the transport's
# error word over ALL ranks (csi_validate_all: an all-reduce on the context's communicator), so that every rank takes the same
# decision.  A non-zero status anywhere throws on every rank; the peer transport then stays refused until every rank has called
# csi_set_halo_transport again (include/csi.h), which is a collective decision of the caller.
function validate_state!(model::HIPSeaIceModel)
    ctx = context(model)
    GC.@preserve model check(ctx, ccall((:csi_validate_all, libcsi), Int32, (Ptr{Cvoid},), ctx.handle))
    return nothing
end

# Checkpointing (sea_ice_model.jl:414-445: prognostic_state / restore_prognostic_state!) needs nothing from the library: the
# state lives in the Oceananigans Fields, restore writes into the same parents, and the library holds pointers only.  If a
# restore REPLACES parents (new arrays), drop the context so that the next step re-attaches:
detach!(model) = (delete!(CONTEXTS, model); nothing)

end # module
