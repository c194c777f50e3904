/*
 * csi.h -- C ABI of the MI355X-native sea-ice hot path (libcsi_hip.so).
 *
 * Drop-in boundary for CliMA/ClimaSeaIce.jl's split-explicit EVP momentum sub-cycle,
 * the h / aice advection and their launch loop.  The reference has no FFI of its own:
 * its "operator API" is Julia multiple dispatch, so every entry point below names the
 * Julia method it replaces (paths relative to /root/reference/src); INTEGRATION.md shows
 * the `ccall` methods a maintainer adds on the Julia side.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, fp64 only, no C++ / torch types.
 *  - Every function returns int32_t: CSI_OK (0) or a negative csi_status; the text of the
 *    last failure is available from csi_last_error().  No exception crosses the boundary.
 *  - Ownership: the caller (Julia / AMDGPU.jl ROCArray parents; torch tensors in this repo's
 *    harness) owns all field memory; the library never frees or reallocates it.  Scratch
 *    lives in the context.
 *  - Field layout = Oceananigans parent array: column-major, i fastest, element (i, j)
 *    (1-based) at ptr[(i + Hx - 1) + (j + Hy - 1) * ld], ld = Nx + 2Hx (+1 when the field
 *    is Face-located in a Bounded x direction).
 *  - One context per GPU; calls on one context are serialised by the caller.  All work is
 *    ordered on the context's HIP stream (an external hipStream_t may be supplied); the only
 *    host synchronisation is csi_sync().
 *  - There is NO CPU fallback: every compute entry point needs a HIP device.
 */
#ifndef CSI_H
#define CSI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CSI_VERSION 100 /* 0.1.0 */

typedef enum {
    CSI_OK = 0,
    CSI_ERR_INVALID_ARGUMENT = -1,
    CSI_ERR_NOT_BOUND = -2,     /* a field / grid / parameter needed by the call was never set */
    CSI_ERR_HIP = -3,           /* a HIP runtime call failed; see csi_last_error */
    CSI_ERR_UNSUPPORTED = -4,   /* valid in the reference, not (yet) in this library */
    CSI_ERR_NO_DEVICE = -5,
    CSI_ERR_COMM = -6           /* RCCL failure */
} csi_status;

/* Topology of one horizontal direction (Oceananigans.Grids topologies).  The *_CONNECTED
 * values mark tile edges whose halos are filled by csi_halo_exchange (multi-GPU) rather than
 * by a local boundary condition (split_explicit_momentum_equations.jl:13-16). */
typedef enum {
    CSI_PERIODIC = 0,
    CSI_BOUNDED = 1,
    CSI_FULLY_CONNECTED = 2,
    CSI_LEFT_CONNECTED = 3,   /* low side exchanged, high side Bounded */
    CSI_RIGHT_CONNECTED = 4,  /* low side Bounded, high side exchanged */
    /* y direction of a TripolarGrid: low side Bounded (the southernmost latitude), high side the north FOLD filled by the
     * Zipper boundary condition.  Sign: fields at the velocity points -- (Face, Center) and (Center, Face): u, v, the stress /
     * ocean-velocity / forcing arrays there, u^n, v^n -- change sign across the fold (sea_ice_model.jl:57-64 for u, v;
     * test/distributed_tests_utils.jl:196-197 builds the others with the same conditions), every (Center, Center) and
     * (Face, Face) field does not.  Pivot: the ONE variant implemented has the fold running through the cell CENTRES of row
     * Ny (Oceananigans' RightCenterFolded / LeftConnectedRightCenterFolded, the names the reference imports at
     * split_explicit_momentum_equations.jl:7-16): c[i, Ny + j] = s c[i', Ny - j] for Center-in-y fields (row Ny is stored
     * twice and its two copies are NOT symmetrised: they evolve independently, as in a fill that only writes halos),
     * s c[i', Ny - j + 1] for Face-in-y fields, i' = Nx - i + 1 (Center in x) / Nx - i + 2 (Face in x; column 1 maps onto itself
     * without the sign change).  The F-point pivot (RightFaceFolded) is not implemented.  STATUS: these fill semantics are
     * RECALLED from the un-vendored Oceananigans, restated three times here (oracle/csi_oracle.c fold_north,
     * grids.fold_north, the HIP store images) and checked against each other only -- no reference-run fixture exists; run
     * bench/reference_driver.jl on a tripolar case before relying on them.  x must be Periodic (or, on
     * tiles, unpartitioned: the reference's own distributed tripolar test partitions y only,
     * test/distributed_tests_utils.jl:239).  Fusion level 2 on an untiled RIGHT_FOLDED grid: rows 1 .. Ny - Hy - 4 run
     * through the two-sub-steps kernel, the rows next to the fold through the three kernels on their own stream (they read and
     * store fold images like the reference's kernels; csi_abi.hip FoldBand) -- bit-identical to the three-kernel run of the whole
     * grid; the fold tile of a y partition (LEFT_CONNECTED_RIGHT_FOLDED) does the same when the exchange interval is even.  Fusion
     * level 1 stays on the three kernels. */
    CSI_RIGHT_FOLDED = 5,
    CSI_LEFT_CONNECTED_RIGHT_FOLDED = 6   /* the northernmost tile of a y partition of such a grid */
} csi_topology;

typedef enum {
    CSI_METRIC_UNIFORM = 0,   /* RectilinearGrid, regular spacing */
    CSI_METRIC_PER_J = 1,     /* LatitudeLongitudeGrid, regular: metrics vary with j only */
    CSI_METRIC_FULL = 2       /* orthogonal curvilinear grid (OrthogonalSphericalShellGrid, TripolarGrid and the like): 2-D metric
                               * arrays.  Every operator takes them (same calls as the reference's Oceananigans.Operators).  FAST
                               * mode folds them into twelve per-POINT coefficient planes; the two-sub-steps kernel has
                               * instantiations for them (fusion level 2), level 1 runs the three kernels.  Rows whose planes hold
                               * one value per row are read from per-row vectors (csi_set_row_constant). */
} csi_metric_kind;

/* Host-side description of the grid metrics (copied by csi_grid_set).  PER_J vectors have
 * length Ny + 2Hy + 1; the entry for row j (1-based) sits at [j + Hy - 1]:
 *   dxc = dx at Center rows (dx^cc = dx^fc)   dxf = dx at Face rows (dx^cf = dx^ff)
 *   azc = Az at Center rows (Az^cc = Az^fc)   azf = Az at Face rows (Az^cf = Az^ff)
 * dy is constant for these two kinds. */
typedef struct {
    double dx, dy;
    const double *dxc, *dxf, *azc, *azf;
    /* CSI_METRIC_FULL: the twelve HOST arrays dx, dy, Az at (c,c), (f,c), (c,f), (f,f) -- index 4 * {dx 0, dy 1, Az 2}
     * + (x at Face) + 2 * (y at Face), i.e. dx^ccc, dx^fcc, dx^cfc, dx^ffc, dy^ccc, ... -- each with Ny + 2Hy + 1 rows
     * of leading dimension full_ld >= Nx + 2Hx + 1; element (i, j) at [(i + Hx - 1) + (j + Hy - 1) * full_ld]. */
    const double* full[12];
    int64_t full_ld;
} csi_metrics;

/* Field slots (csi_field_bind).  Locations: (x, y) with c = Center, f = Face. */
typedef enum {
    CSI_F_U = 0,      /* (f,c) model.velocities.u */
    CSI_F_V,          /* (c,f) model.velocities.v */
    CSI_F_H,          /* (c,c) model.ice_thickness */
    CSI_F_A,          /* (c,c) model.ice_concentration */
    CSI_F_S11,        /* (c,c) auxiliaries.fields.sigma11  (elasto_visco_plastic_rheology.jl:147) */
    CSI_F_S22,        /* (c,c) :148 */
    CSI_F_S12,        /* (f,f) :149 */
    CSI_F_UN,         /* (f,c) :150 */
    CSI_F_VN,         /* (c,f) :151 */
    CSI_F_P,          /* (c,c) :152 */
    CSI_F_ALPHA,      /* (c,c) :153 */
    CSI_F_DELTA,      /* (c,c) :154 */
    CSI_F_ZETA_F,     /* (f,f) :157 */
    CSI_F_ZETA_C,     /* (c,c) :158 */
    CSI_F_GH,         /* (c,c) timestepper.G^n.h */
    CSI_F_GA,         /* (c,c) timestepper.G^n.aice */
    CSI_F_HM,         /* (c,c) timestepper.Psi^-.h   (sea_ice_rk_substep.jl:29-42) */
    CSI_F_AM,         /* (c,c) timestepper.Psi^-.aice */
    CSI_F_UM,         /* (f,c) timestepper.Psi^-.u */
    CSI_F_VM,         /* (c,f) timestepper.Psi^-.v */
    CSI_F_TOP_U,      /* (f,c) top stress array or top external velocity u_e */
    CSI_F_TOP_V,      /* (c,f) */
    CSI_F_BOT_U,      /* (f,c) bottom stress array or bottom external velocity u_e */
    CSI_F_BOT_V,      /* (c,f) */
    CSI_F_MASS_FLUX,  /* (c,c) mass_fluxes.thermodynamics.ice */
    CSI_F_HS,         /* (c,c) model.snow_thickness (snow layer; optional) */
    CSI_F_GHS,        /* (c,c) timestepper.G^n.hs */
    CSI_F_HSM,        /* (c,c) timestepper.Psi^-.hs */
    CSI_F_MASS_FLUX_SNOW,   /* (c,c) mass_fluxes.thermodynamics.snow (optional) */
    CSI_F_SNOWFALL_INTERCEPTED, /* (c,c) mass_fluxes.intercepted_snowfall (optional) */
    CSI_F_TU,         /* (c,c) ice_thermodynamics.top_surface_temperature (optional output) */
    CSI_F_TUS,        /* (c,c) snow_thermodynamics.top_surface_temperature (optional output) */
    CSI_F_FORCING_U,  /* (f,c) model.forcing.u given as an array: the `user_forcing` of sum_of_forcing_u
                       * (elasto_visco_plastic_rheology.jl:391-395), an acceleration in m s^-2; optional, both or neither */
    CSI_F_FORCING_V,  /* (c,f) model.forcing.v (:397-401) */
    CSI_F_COUNT
} csi_field_id;

typedef enum { CSI_PRESSURE_REPLACEMENT = 0, CSI_PRESSURE_ICE_STRENGTH = 1 } csi_pressure_kind;

/* ElastoViscoPlasticRheology (elasto_visco_plastic_rheology.jl:14-25, defaults :119-127) +
 * SeaIceMomentumEquation scalars (sea_ice_momentum_equations.jl:67-94) + FPlane coriolis +
 * sea_ice_density (sea_ice_model.jl:142-145). */
typedef struct {
    double ice_compressive_strength;   /* P*      27500 */
    double ice_compaction_hardening;   /* C       20 */
    double yield_curve_eccentricity;   /* e       2 */
    double minimum_plastic_stress;     /* Dmin    2e-9 */
    double min_relaxation_parameter;   /* alpha-  50 */
    double max_relaxation_parameter;   /* alpha+  300 */
    double relaxation_strength;        /* c_alpha pi^2 */
    int32_t pressure_formulation;      /* csi_pressure_kind */
    int32_t has_coriolis;              /* 0: coriolis = nothing, 1: FPlane(f) */
    double coriolis_f;
    double minimum_concentration;      /* 1e-3 */
    double minimum_mass;               /* 1.0 */
    double sea_ice_density;            /* 900 */
} csi_evp_params;

typedef enum {
    CSI_STRESS_NONE = 0,           /* nothing */
    CSI_STRESS_CONST = 1,          /* Number / NamedTuple of Numbers (sea_ice_external_stress.jl:16-17,29-37) */
    CSI_STRESS_FIELD = 2,          /* arrays bound to CSI_F_{TOP,BOT}_{U,V} (:19-20) */
    CSI_STRESS_SEMI_IMPLICIT = 3   /* SemiImplicitStress (:84-130,176-202) */
} csi_stress_kind;
typedef enum { CSI_VEL_ZERO = 0, CSI_VEL_CONST = 1, CSI_VEL_FIELD = 2 } csi_velocity_kind;
typedef enum { CSI_STRESS_TOP = 0, CSI_STRESS_BOTTOM = 1 } csi_stress_side;

typedef struct {
    int32_t kind;                  /* csi_stress_kind */
    int32_t ue_kind, ve_kind;      /* csi_velocity_kind: ZeroField / ConstantField / Field (bound slots) */
    int32_t reserved;
    double tau_u, tau_v;           /* CONST */
    double ue, ve;                 /* SEMI_IMPLICIT with ConstantField external velocity */
    double rho_e, Cd;              /* SEMI_IMPLICIT: 1026, 5.5e-3 */
} csi_stress;

/* Arithmetic mode of the kernels.
 * STRICT: the reference's operation order, no FMA contraction, IEEE division -- bit-for-bit
 *         equal to the CPU oracle (used to anchor parity).
 * FAST:   hoisted reciprocals / FMA contraction, shared strain rates; differs from STRICT
 *         by rounding only (tolerance stated in DESIGN.md and enforced in tests/).  Requires
 *         minimum_mass > 0 (the reference's default is 1 kg m^-2): the mi <= 0 guards of the velocity
 *         tendencies are then implied by the active / marginal ice selection and are not evaluated
 *         (CSI_ERR_UNSUPPORTED otherwise).  Advection, the tracer update and the thermodynamic steps are
 *         computed in the reference's order in both modes. */
typedef enum { CSI_MODE_STRICT = 0, CSI_MODE_FAST = 1 } csi_mode;

typedef enum { CSI_ADVECT_NONE = 0, CSI_ADVECT_UPWIND1 = 1, CSI_ADVECT_WENO3 = 3, CSI_ADVECT_WENO5 = 5, CSI_ADVECT_WENO7 = 7,
               CSI_ADVECT_UPWIND3 = -3, CSI_ADVECT_UPWIND5 = -5 } csi_advection_scheme;

typedef struct csi_context csi_context;

/* ---- lifecycle ------------------------------------------------------------------------- */
int32_t csi_version(void);
/* stream: a hipStream_t owned by the caller, or NULL for a library-owned stream. */
int32_t csi_context_create(int32_t device_id, void* hip_stream, csi_context** out);
int32_t csi_context_destroy(csi_context* ctx);
/* Text of the most recent failure on ctx (or of context creation when ctx == NULL). */
const char* csi_last_error(const csi_context* ctx);
int32_t csi_sync(csi_context* ctx);
/* csi_sync + the halo transport's status reduced over ALL ranks of the context's communicator (collective: every rank calls it
 * between the same two steps; one rank: the same as csi_sync).  A peer-transport wait that gave up reaches only the direct
 * neighbours' abort words; this is the call that lets every rank see it and take the same decision -- what the Julia side calls
 * before output writers / checkpointers read a Distributed model's fields (julia/ClimaSeaIceHIP.jl validate_state!; the
 * reference's own check of a distributed run is done after the fact, test/distributed_tests_utils.jl:40-88). */
int32_t csi_validate_all(csi_context* ctx);
/* Testing aid: leaves the host side in the state a timed-out wait of the peer transport's flag protocol leaves it in (the sticky
 * CSI_ERR_COMM above), without the wait.  Not for production use. */
int32_t csi_debug_peer_abort(csi_context* ctx);
int32_t csi_set_mode(csi_context* ctx, int32_t mode);

/* ---- problem description ----------------------------------------------------------------- */
/* Replaces the grid argument every reference kernel receives (Nx, Ny, halo, topology, metrics). */
int32_t csi_grid_set(csi_context* ctx, int32_t Nx, int32_t Ny, int32_t Hx, int32_t Hy,
                     int32_t topo_x, int32_t topo_y, int32_t metric_kind, const csi_metrics* metrics);
/* Cell-centred activity mask of an ImmersedBoundaryGrid (1 = active), device pointer laid out
 * like a (c,c) parent with leading dimension ld; NULL removes it.
 * (peripheral_node, split_explicit_momentum_equations.jl:226,261; conditional_flux_*,
 * ice_stress_divergence.jl:21-24; mask_immersed_field_xy!, sea_ice_model.jl:381-389) */
int32_t csi_mask_set(csi_context* ctx, const uint8_t* dev_mask, int64_t ld);
/* Bind the parent array of one field: device pointer, leading dimension and parent extents
 * (validated against the grid: ni = Nx + 2Hx [+1], nj = Ny + 2Hy [+1]).  * On a tiled model with the peer halo transport the neighbouring ranks map u, v, sigma11, sigma22, sigma12, alpha, zeta_c, zeta_f and
 * Delta: binding another array to one of these slots makes the next sub-cycle set the transport up again, COLLECTIVELY -- every
 * rank of the decomposition has to re-bind between the same two steps (like any collective). */
int32_t csi_field_bind(csi_context* ctx, int32_t field_id, void* dev_ptr, int64_t ld, int32_t ni, int32_t nj);
int32_t csi_evp_params_set(csi_context* ctx, const csi_evp_params* p);
int32_t csi_stress_set(csi_context* ctx, int32_t side, const csi_stress* s);
/* Boundary condition of the TANGENTIAL velocity at a wall: field_id CSI_F_U with side 0 (south) / 1 (north), CSI_F_V with
 * side 0 (west) / 1 (east).  kind 0: the default no-flux condition (halo = mirror image, free slip); kind 1:
 * ValueBoundaryCondition(value) -- no-slip for value 0, as in examples/ice_advected_on_coastline.jl:96-99 -- whose
 * halo fill sets the first halo cell to 2 * value - c[first interior cell] (upstream fill_halo_regions!, SURVEY.md
 * App. B; a13 of the scope table).  Sides that are not walls ignore it.  Every path takes it (the fused kernels
 * reflect about 2 * value where they otherwise mirror). */
int32_t csi_velocity_bc_set(csi_context* ctx, int32_t field_id, int32_t side, int32_t kind, double value);
/* Immersed boundary conditions of u / v: ImmersedBoundaryCondition(west = FluxBoundaryCondition(number), ...) entering
 * immersed_dj_sigma_1j / immersed_dj_sigma_2j (ice_stress_divergence.jl:65-123; the stress is minus the flux on west / south
 * faces and plus the flux on east / north faces, :115-123).  field_id CSI_F_U or CSI_F_V; all zeros (the default) is the
 * reference's default `nothing`.  Non-zero values need a mask (csi_mask_set) to have any effect; the two-sub-steps kernel takes
 * them (their divergence is evaluated once per sub-cycle into two library arrays), the one-sub-step kernel does not (three kernels). */
int32_t csi_immersed_flux_bc_set(csi_context* ctx, int32_t field_id, double west, double east, double south, double north);
/* Row-dependent Coriolis parameter: BetaPlane, f = f0 + beta * ynode (upstream x_f_cross_U / y_f_cross_U called at
 * momentum_tendencies_kernel_functions.jl:31,64; in the reference's test matrix, test/test_time_stepping.jl:35).
 * f_u: f at the (Face, Center) nodes of each row (u points), f_v: at the (Center, Face) nodes (v points); HOST
 * arrays laid out like the PER_J metric vectors (value for row j at [j + Hy - 1], n = Ny + 2Hy + 1; halo rows hold the
 * value of the row they image: the neighbouring tile's row, the wrapped row of a Periodic direction -- the fused
 * kernels recompute ring rows there and must see their owner's f).  They replace csi_evp_params.coriolis_f while set (has_coriolis must be 1);
 * NULL, NULL returns to the FPlane value.  Call after csi_grid_set (a new grid drops them). */
int32_t csi_coriolis_rows_set(csi_context* ctx, const double* f_u, const double* f_v, int32_t n);
/* Point-dependent Coriolis parameter on CSI_METRIC_FULL grids (f = 2 Omega sin(latitude) of a curvilinear grid whose
 * latitude varies along both indices, e.g. a TripolarGrid): f at the (Face, Center) and (Center, Face) nodes, HOST arrays laid
 * out like the metric planes (Ny + 2Hy + 1 rows of leading dimension ld >= Nx + 2Hx + 1, element (i, j) at
 * [(i + Hx - 1) + (j + Hy - 1) * ld]; halo entries hold the value of the point they image).  Applied with the stencil the
 * reference's hot path exercises, -f * Ixy(v) / +f * Ixy(u) (FPlane / BetaPlane: test/test_time_stepping.jl:35,
 * examples and distributed tests); upstream's HydrostaticSphericalCoriolis enstrophy-conserving stencil appears nowhere in
 * the reference and is not implemented.  Replaces csi_evp_params.coriolis_f / the rows while set; NULL, NULL removes it. */
int32_t csi_coriolis_points_set(csi_context* ctx, const double* f_u, const double* f_v, int64_t ld);

/* ---- the reference's verbs ----------------------------------------------------------------- */
/* initialize_rheology!(model, ::ElastoViscoPlasticRheology), elasto_visco_plastic_rheology.jl:192-219 */
int32_t csi_evp_initialize(csi_context* ctx);
/* The sub-step loop of time_step_momentum!, split_explicit_momentum_equations.jl:170-189:
 * local halo fill of u, v, then `substeps` x [compute_stresses! ; alternating u/v steps with halo
 * fills].  dt is the stage step.  first_substep is the 1-based index of the first sub-step
 * (parity selects the u/v order, :178). */
int32_t csi_evp_subcycle(csi_context* ctx, double dt, int32_t substeps, int32_t first_substep);
/* finalize_rheology!, elasto_visco_plastic_rheology.jl:275-280 (local halo fill of sigma) */
int32_t csi_evp_finalize(csi_context* ctx);
/* time_step_momentum!(model, ::SplitExplicitMomentumEquation, dt), split_explicit_momentum_equations.jl:103-195.
 * rk_reset != 0: reset_velocities! from Psi^- (:89-93). */
int32_t csi_time_step_momentum(csi_context* ctx, double dt, int32_t substeps, int32_t rk_reset);
/* compute_tracer_tendencies!(model), tracer_tendency_kernel_functions.jl:9-45 */
int32_t csi_compute_tracer_tendencies(csi_context* ctx, int32_t scheme);
/* Precision of a WENO scheme's smoothness indicators and nonlinear weights inside csi_compute_tracer_tendencies / the time steppers.
 * CSI_WEIGHTS_F64 (default): everything in double.  CSI_WEIGHTS_F32: the indicators, tau, the ratios, the unnormalised weights and
 * their sum in float on float-converted stencil values; candidates and the final combination in double -- the library's reading
 * of the second float type parameter FT2 (= Float32 by default) that newer Oceananigans versions give WENO{N, FT, FT2, ...}, whose
 * instances the reference only calls (src/sea_ice_advection.jl:51-58).  RECALLED, not verified: the Julia side
 * selects the mode from typeof(model.advection) (julia/ClimaSeaIceHIP.jl weight_dtype), bench/reference_driver.jl dumps that type
 * with its fields so that a reference run shows which mode it was.  Both modes: STRICT equals the oracle bit for bit. */
enum { CSI_WEIGHTS_F64 = 0, CSI_WEIGHTS_F32 = 1 };
int32_t csi_set_weno_weight_dtype(csi_context* ctx, int32_t dtype);
int32_t csi_weno_weight_dtype(csi_context* ctx, int32_t* dtype);
/* dynamic_time_step!(model, dt): sea_ice_fe_step.jl:36-82 (from_cache = 0), sea_ice_rk_substep.jl:134-152 (1) */
int32_t csi_dynamic_step_tracers(csi_context* ctx, double dt, int32_t from_cache);
/* cache_current_fields!(model), sea_ice_rk_substep.jl:29-42 */
int32_t csi_cache_current_fields(csi_context* ctx);
/* update_state!(model), sea_ice_model.jl:379-394: immersed masking + local halo fill of h, aice, u, v */
int32_t csi_update_state(csi_context* ctx);
/* fill_halo_regions!(field; only_local_halos = true) for one bound field */
int32_t csi_fill_halo_local(csi_context* ctx, int32_t field_id);
/* Whole model steps: FE (sea_ice_fe_step.jl:13-34) and the SplitRungeKutta3 stage loop around rk_substep!
 * (sea_ice_rk_substep.jl:81-94).  Without csi_evp_params_set (dynamics = nothing) the velocities are prescribed and
 * the momentum step is skipped: advection-only models.  Thermodynamics: csi_slab_params_set / csi_snow_params_set. */
int32_t csi_time_step_fe(csi_context* ctx, double dt, int32_t substeps, int32_t scheme, int32_t first_iteration);
int32_t csi_time_step_rk3(csi_context* ctx, double dt, int32_t substeps, int32_t scheme);

/* Bare-ice slab thermodynamics with PrescribedTemperature top boundary condition
 * (thermodynamic_time_step.jl:75-118,304-370; slab_thermodynamics_tendencies.jl:28-135). */
typedef struct {
    double conductivity;             /* ConductiveFlux, 2 */
    double sea_ice_density;          /* bulk, 900 */
    double density;                  /* PhaseTransitions.density (pure ice), 917 */
    double liquid_density;           /* 999.8 */
    double liquid_heat_capacity;     /* 4186 */
    double heat_capacity;            /* 2000 */
    double reference_latent_heat;    /* 334e3 */
    double reference_temperature;    /* 0 */
    double liquidus_slope;           /* 0.054 */
    double freshwater_melting_temperature; /* 0 */
    double bottom_salinity;          /* IceWaterThermalEquilibrium salinity, 0 */
    double ice_consolidation_thickness;    /* 0.05 */
    double top_temperature;          /* PrescribedTemperature */
    int32_t top_flux_kind;           /* 0: constant Qu ; 1: internal-flux equilibrium (sea_ice_model.jl:248-256) */
    int32_t bottom_flux_kind;        /* 0: constant Qb ; 1: -(1 - aice) * Qb (examples/freezing_bucket.jl:79-81) */
    double top_heat_flux, bottom_heat_flux;
    /* top boundary condition (slab_thermodynamics_tendencies.jl:107-119).  0: PrescribedTemperature(top_temperature).
     * 1: MeltingConstrainedFluxBalance with a NUMERIC top_heat_flux: the reference's secant solve of
     * Qx - Qi(T) = 0 (top_heat_boundary_conditions.jl:80-97) has, for the linear conductive flux, the closed-form
     * root T = Tb - Qx R (R = h / k, with snow hs / ks + h / k), capped at the melting temperature. */
    int32_t top_bc_kind;
    int32_t pad_;
    double ice_salinity;             /* model.ice_salinity: Tm = melting_temperature(liquidus, S), 0 */
} csi_slab_params;
/* Snow layer on the slab: snow_slab_thermodynamics (slab_sea_ice_thermodynamics.jl:42-49) + snow_density / snowfall
 * of SeaIceModel.  The layered step is _layered_thermodynamic_time_step! (thermodynamic_time_step.jl:131-298); hs is
 * advected and updated like h (tracer_tendency_kernel_functions.jl:49-52, sea_ice_fe_step.jl:86-94) when
 * CSI_F_HS / CSI_F_GHS (/ CSI_F_HSM for RK3) are bound. */
typedef struct {
    double conductivity;             /* 0.31 */
    double snow_density;             /* 330 */
    double snowfall;                 /* kg m^-2 s^-1, constant */
    double top_temperature;          /* PrescribedTemperature of the snow surface (top_bc_kind 0) */
    int32_t top_bc_kind;             /* as csi_slab_params.top_bc_kind, for the snow surface */
    int32_t pad_;
} csi_snow_params;
/* thermodynamic_time_step!(model, ::SlabThermodynamics, ::SlabThermodynamics, dt): needs CSI_F_H, CSI_F_A, CSI_F_HS */
int32_t csi_layered_thermo_step(csi_context* ctx, const csi_slab_params* ice, const csi_snow_params* snow, double dt);
/* With slab parameters set: csi_time_step_fe / _rk3 run the layered step instead of the bare-ice one.  NULL removes it. */
int32_t csi_snow_params_set(csi_context* ctx, const csi_snow_params* p);
int32_t csi_slab_thermo_step(csi_context* ctx, const csi_slab_params* p, double dt);
/* Make csi_time_step_fe / csi_time_step_rk3 run the slab step where the reference does (after the tracer update of
 * every stage: sea_ice_fe_step.jl:28, sea_ice_rk_substep.jl:91).  NULL removes it. */
int32_t csi_slab_params_set(csi_context* ctx, const csi_slab_params* p);

/* ---- multi-GPU tiles (one process per GPU; RCCL point-to-point over xGMI) ----------------- */
/* Position of this context's tile in an Rx x Ry decomposition of a global grid; the
 * *_CONNECTED topologies passed to csi_grid_set must agree with it. */
int32_t csi_tile_set(csi_context* ctx, int32_t rank_x, int32_t rank_y, int32_t Rx, int32_t Ry,
                     int32_t periodic_x, int32_t periodic_y);
/* 128-byte RCCL unique id produced on rank 0 (csi_comm_unique_id) and broadcast by the host. */
int32_t csi_comm_unique_id(uint8_t* id128);
int32_t csi_comm_init(csi_context* ctx, int32_t world_size, int32_t rank, const uint8_t* id128);
/* Ranks of the context's RCCL communicator as RCCL itself reports them (ncclCommCount; 0 before csi_comm_init): what
 * bench.py prints as `rccl_ranks`, so that a run that silently fell back to one rank cannot report n_gpus > 1. */
int32_t csi_comm_count(csi_context* ctx, int32_t* ranks);

/* In-process tile group: several contexts of ONE process -- one host thread each, normally all on one GPU -- exchange their halos
 * through device-to-device copies instead of RCCL (which refuses two ranks on one device), with RCCL's matching rule (messages
 * between a pair of ranks match in the order they were posted: the same send / receive plans run); the peer halo transport
 * addresses the neighbours' arrays directly.  Host-synchronous, built for correctness runs of real decompositions on a one-GPU
 * machine (tests/test_gpu_local_tiles.py: 2 x 2, 1 x 4 with the fold tile, a Bounded x partition) and for a single process that
 * drives several tiles.  Every context of a group calls the sub-cycle from its own thread; a rank that never arrives makes the
 * others fail with CSI_ERR_COMM after two minutes instead of hanging.  The group outlives its contexts' use of it (destroy it
 * after them).  csi_comm_init_local replaces csi_comm_init (csi_tile_set as usual); csi_comm_count reports the group size.
 * With the peer transport the tiles' kernels wait for each other's flags, so their streams must not share a hardware queue (two
 * streams on one queue run in submission order): set GPU_MAX_HW_QUEUES (default 4) to more than the number of tiles before the
 * HIP runtime initialises. */
typedef struct csi_local_group csi_local_group;
int32_t csi_local_group_create(int32_t world_size, csi_local_group** out);
void csi_local_group_destroy(csi_local_group* group);
int32_t csi_comm_init_local(csi_context* ctx, csi_local_group* group, int32_t rank);
/* Host-channel tile group: the ranks are PROCESSES (one context each, any devices -- in particular several on ONE GPU, where RCCL
 * refuses to form a communicator) that share the POSIX shared-memory segment `shm_name` (e.g. "/csi-<job>"; every rank passes
 * the same name; the name is unlinked once all have joined).  The halo exchanges and the two small collectives of the peer
 * set-up travel over the host and HIP IPC copies; the peer halo transport maps the neighbours' arrays and flag words with
 * hipIpcOpenMemHandle exactly as one process per GPU does under RCCL.  Host-synchronous: for correctness runs, not for speed.
 * Stands where the reference's distributed tests start `mpiexec -n 4` on whatever devices there are
 * (test/test_distributed_sea_ice.jl:41-54). */
int32_t csi_comm_init_host(csi_context* ctx, const char* shm_name, int32_t world_size, int32_t rank);
/* Exchange `width` halo layers of the fields in `field_ids` with the neighbouring tiles. */
int32_t csi_halo_exchange(csi_context* ctx, const int32_t* field_ids, int32_t nfields, int32_t width);

/* Velocity of marginal ice (0 < mass, concentration below minimum_mass / minimum_concentration;
 * split_explicit_momentum_equations.jl:219-228).  kind 0 (default): `free_drift = nothing`, zero.  kind 1:
 * StressBalanceFreeDrift built on the model's own top / bottom stresses (stress_balance_free_drift.jl:61-121,
 * materialize_free_drift :44-46): exactly one of them must be a SemiImplicitStress, U = U_e - tau / sqrt(C |tau|).
 * Evaluated once per sub-cycle into library-owned arrays (it depends on the forcing only); the three-kernel paths and
 * the two-sub-steps-per-launch kernel read them. */
int32_t csi_free_drift_set(csi_context* ctx, int32_t kind);

/* FAST mode only.  level 0: always the three-kernel path.  level 1: a sub-step is ONE launch of the fused
 * kernel (stress + both velocity updates, ring recomputation per wavefront, double-buffered u, v, sigma in
 * library scratch) whenever the configuration allows it (no immersed mask, forcing given by numbers;
 * csrc/evp_fused.hip).  level 2 (default): in addition
 * TWO consecutive sub-steps share one launch (csrc/evp_fused2.hip: the first sub-step's results stay in
 * registers; immersed masks, array-valued top stress and array-valued ocean velocities in the bottom drag
 * supported) where the halo is >= 4 (and N >= 2 halo) and, on tiles, the exchange
 * interval is even; an odd trailing sub-step uses the level-1 kernel, or -- with masks, array forcing or per-point metrics -- one more launch
 * of the same kernel whose second wave stores the first sub-step's results instead of computing a second one.  All paths execute the same
 * floating-point operations and give bit-identical results. */
/* (Round 2 built a level 3 -- three sub-steps per launch, three waves per tile chained through two LDS rings; it traded a third of
 * the HBM traffic for 10 % more arithmetic and measured slower than level 2 at every size but 3072^2, so round 4 removed it:
 * DESIGN.md section 3.)  Levels other than 0, 1, 2 are refused. */
int32_t csi_set_fusion(csi_context* ctx, int32_t level);

/* Tile activity (round 6; default on).  Where the ice mass h rho aice is exactly zero -- land after mask_immersed_field_xy!
 * (src/sea_ice_model.jl:379-384), ice-free ocean -- the EVP sub-step is an exact no-op: sigma += ifelse(m > 0, ..., 0)
 * (src/Rheologies/elasto_visco_plastic_rheology.jl:343-347) and the velocity select's zero branch
 * (src/SeaIceDynamics/split_explicit_momentum_equations.jl:217-228, 251-263).  On untiled grids advanced by the two-sub-steps
 * kernel the library tests, before the first launch of every sub-cycle, which 56-column tiles of the launch have no ice mass in
 * or one cell around them (and no -0.0 among their stresses); the first two launches and the last one run every tile, the
 * launches in between only the live ones, on a finer tiling chosen so that the live tiles fill the GPU once.  Results are
 * bit-identical with skipping off (tests/test_gpu_activity.py) as long as the fields are finite.
 * csi_tile_activity: the newest counts that have arrived from the device (tiles of the live launches, live ones among them;
 * -1 live: none yet), and whether the last sub-cycle used live launches (1; 2: its first two launches also left out the tiles that
 * were quiescent from the start -- no ice mass, velocities +0.0 already, no halo image to store --, which takes one copy of u, v,
 * sigma per sub-cycle and is done once a sample has shown quiescent tiles) -- call csi_sync first for the last sub-cycle's counts. */
int32_t csi_set_tile_skipping(csi_context* ctx, int32_t on);
int32_t csi_tile_activity(csi_context* ctx, int32_t* tiles, int32_t* live, int32_t* used);

/* CSI_METRIC_FULL grids, row-constant rows (round 6; default on, rtol 0).  A TripolarGrid is a latitude-longitude grid south of
 * its bipolar cap: there every metric plane holds one value per row.  The library marks the rows in which all Nx + 2Hx + 1
 * columns of all twelve coefficient planes (and of a per-point Coriolis parameter) are EQUAL BIT FOR BIT and lets the
 * two-sub-steps kernel read those rows' values from per-row vectors -- the same operands, the same operations, 96 B per cell and
 * launch less HBM traffic; results are unchanged.  rtol > 0 (at most 1e-6) also marks rows whose columns agree with the first
 * interior column to that relative distance and uses that column's value for the row: for grids whose row-constant part
 * carries rounding noise (metrics computed per point); this CHANGES results at the rtol level and is the caller's decision.
 * csi_row_constant_rows: how many of the Ny + 2Hy + 1 plane rows are marked. */
int32_t csi_set_row_constant(csi_context* ctx, int32_t on, double rtol);
int32_t csi_row_constant_rows(csi_context* ctx, int32_t* rows);

/* RCCL halo exchange of u, v every k sub-steps with width 2k (needs halo >= 2k).  k = 0 (default): automatic -- the peer
 * transport below where it applies, else the largest k <= 16 the halo allows; k >= 1 selects the RCCL exchange with that
 * interval (k = 1: every sub-step); the reference is the k = substeps extreme (halo 2*substeps+3,
 * split_explicit_momentum_equations.jl:51-64). */
int32_t csi_set_exchange_interval(csi_context* ctx, int32_t k);

/* Halo transport of the sub-cycle on tiles (FAST mode, two sub-steps per launch; an odd count ends with one single-sub-step launch
 * of the same kernel).
 * CSI_TRANSPORT_PEER (default): peer-direct halo writes over xGMI.  The neighbouring tiles' u, v, sigma (and alpha, zeta, Delta)
 * arrays are mapped into this process (HIP IPC handles, exchanged once over the context's RCCL communicator); a connected side
 * then behaves like a Periodic one whose halo lives on another GPU: the kernel that owns a cell next to the side stores its halo
 * image straight into the neighbour's array, and per-tile flags in device memory order the launches of neighbouring ranks (only
 * the tiles next to a connected side wait, the interior of a launch overlaps the neighbours' edges).  No pack / unpack kernels,
 * no RCCL kernel inside the sub-cycle, no widened halo: the halo 4 of an untiled run suffices, and one RCCL exchange per
 * sub-cycle remains (what BASELINE.json's north star asks for; the reference's own design is one exchange per sub-cycle with a
 * 2 * substeps + 3 halo, split_explicit_momentum_equations.jl:51-64).  Needs tiles of equal shape (their row strides may differ: the easternmost tile of a Bounded x partition); set up
 * collectively at the first sub-cycle (and again when bound arrays change -- bind on all ranks together); if any rank cannot
 * (no IPC, a tile the two-sub-steps kernel does not take), every rank stays on RCCL.  A tile that waits 3 s for a neighbour gives up, the next csi_sync
 * returns CSI_ERR_COMM.
 * CSI_TRANSPORT_RCCL: pack -> grouped ncclSend / ncclRecv -> unpack of width-2k strips every k sub-steps
 * (csi_set_exchange_interval); what every other path (three kernels, STRICT) uses anyway.
 * Both give results bit-identical to the untiled run.  csi_halo_transport: what the last sub-cycle used. */
enum { CSI_TRANSPORT_RCCL = 0, CSI_TRANSPORT_PEER = 1 };
int32_t csi_set_halo_transport(csi_context* ctx, int32_t kind);
int32_t csi_halo_transport(csi_context* ctx, int32_t* kind);
/* Run-time tiers of the peer transport's memory-ordering protocol.  Every rank of a decomposition must set the SAME tier.
 *  -1 (default): automatic -- tier 1 whenever a neighbour lives in another process or on another device (one process per GPU, the
 *      host-channel group), tier 0 for a tile connected to itself and for the tiles of an in-process group on one device;
 *   0: halo images are write-through stores at system scope, flags follow the drained store queue; a waiting tile loads nothing
 *      before it has seen the flags and issues no cache maintenance (measured fastest).  It rests on an
 *      argument about what CANNOT be cached on the receiving GPU; that argument has only ever run inside one L2 domain, and a
 *      passing tiled == untiled check does not prove it (a violation would be a rare, timing-dependent stale line): across
 *      devices tier 0 is an explicit opt-in (bench.py --peer-tier 0), never the default;
 *   1: + a system-scope acquire fence (buffer_inv sc0 sc1) in every waiting tile once the flags have been seen (edge tiles only);
 *   2: + a system-scope release fence (buffer_wbl2 sc0 sc1) before a tile publishes its flags -- the textbook protocol.
 * csi_peer_tier returns the tier the kernels run (automatic resolved).
 * A wait that gives up after 3 s makes its workgroup leave without storing or publishing, sets this rank's error word and the
 * abort word of every neighbour's flag array; every entry point that advances the model and csi_sync report CSI_ERR_COMM -- and
 * keep reporting it (the error is STICKY: the flags cannot recover by themselves) until the caller has re-armed the transport on
 * EVERY rank with csi_set_halo_transport (CSI_TRANSPORT_PEER: the next sub-cycle runs the collective set-up again, which clears
 * flags, abort words and launch numbers once all ranks have arrived; CSI_TRANSPORT_RCCL: the message exchange) or csi_comm_init*.
 * csi_validate_all spreads the status to ranks that are not neighbours of the one that gave up. */
int32_t csi_set_peer_tier(csi_context* ctx, int32_t tier);
int32_t csi_peer_tier(csi_context* ctx, int32_t* tier);

/* Index ranges (1-based, inclusive: i0, i1, j0, j1) the launch loop uses for a grid of this shape and
 * topology when `valid_width` (V >= 2) layers of u, v beyond the owned cells are valid on connected sides:
 * [0..3] stress kernel (Auxiliaries kernel parameters -H+2:N+H-1, elasto_visco_plastic_rheology.jl:145, on
 * local sides; 2-V : N+V-1 on connected sides), [4..7] the u step when u is updated first, [8..11] the v step
 * when v is updated first, [12..15] the velocity updated second (cf. split_explicit_kernel_size,
 * split_explicit_momentum_equations.jl:40-46).  Pure host function. */
int32_t csi_plan_ranges(int32_t Nx, int32_t Ny, int32_t Hx, int32_t Hy, int32_t topo_x, int32_t topo_y,
                        int32_t valid_width, int32_t* out16);

/* The halo-exchange plan of one tile for one field, eight directions in the library's fixed order
 * (dy outer, dx inner, both -1..1, (0,0) skipped).  For k = 0..7, out40[5k..5k+4] = peer rank (-1: none),
 * i0, j0, ni, nj (1-based start and extents of the strip).  halo = 0: the owned strips this tile SENDS, in
 * send order; halo = 1: the halo strips it RECEIVES, in the order the matching sends were issued (message k
 * arrives from the neighbour in direction -k).  Pure host function: the CPU multi-process tests drive a gloo
 * exchange with exactly the plan the RCCL path uses. */
int32_t csi_plan_exchange(int32_t Nx, int32_t Ny, int32_t Hx, int32_t Hy, int32_t topo_x, int32_t topo_y,
                          int32_t rank_x, int32_t rank_y, int32_t Rx, int32_t Ry, int32_t periodic_x, int32_t periodic_y,
                          int32_t width, int32_t halo, int32_t* out40);

/* ---- introspection used by bench.py / tests ------------------------------------------------ */
/* Device time (ms) of the last csi_evp_subcycle / csi_time_step_momentum call measured with HIP
 * events on the context's stream; valid after csi_sync. */
int32_t csi_last_subcycle_ms(csi_context* ctx, double* ms);
/* Device time of ALL sub-cycles between the two calls (HIP events on the context's stream around every sub-step loop; _end
 * synchronises): their sum in ms, their number and the number of kernel launches inside them -- what bench.py divides to get the
 * dominant kernel's average launch time over the TIMED region itself (at most 4096 sub-cycles are kept). */
int32_t csi_subcycle_stats_begin(csi_context* ctx);
int32_t csi_subcycle_stats_end(csi_context* ctx, double* total_ms, int32_t* cycles, int32_t* launches);
/* How the launch loop would run one PAIR of sub-steps (csi_set_fusion level 2) at position m (even) of an exchange
 * batch of k sub-steps on a grid of this shape: out32[0] = 1 if the pair kernel applies (0: the rest is zero),
 * [1..3] wave-tile geometry (56-column strips, row chunks, rows per chunk), then six index ranges (i0, i1, j0, j1):
 * [4..7] rows / columns the first sub-step computes, [8..11] the second sub-step's compute range (what the wave tiles
 * decompose), [12..15] cells whose stresses are stored, [16..19] / [20..23] first velocity stored when the second
 * sub-step is u-first / v-first, [24..27] second velocity stored; [28] = 1 if a side is a wall.  Periodic and wall
 * sides store the interior (the wall corners of sigma12 included) and refresh halos as images of those stores;
 * connected sides store the ring the next pair needs.  Pure host function. */
int32_t csi_plan_pair(int32_t Nx, int32_t Ny, int32_t Hx, int32_t Hy, int32_t topo_x, int32_t topo_y, int32_t k, int32_t m,
                      int32_t* out32);
/* How a pair launch of the PEER transport cuts a tile of this shape into chunks of rows (pure host function; periodic f-plane tile,
 * sides as the launch loop sees them): out8 = {applies, strips, chunks, rows per chunk, rows of the first chunk (0: as the others),
 * rows kept for the last chunk (0: what is left), chunks in the south side's tile set, in the north side's}; rows_out[2q], [2q + 1]
 * = first / last row of chunk q (at most max_chunks of them).  peer_south / peer_north: a neighbour beyond that y side -- its chunk
 * is four rows shorter, never shorter than the halo. */
int32_t csi_plan_peer_chunks(int32_t Nx, int32_t Ny, int32_t Hx, int32_t Hy, int32_t peer_south, int32_t peer_north, int32_t cus,
                             int32_t* out8, int32_t* rows_out, int32_t max_chunks);

/* Per-phase device time: runs `substeps` (2..64) further EVP sub-steps from the current state with HIP
 * events between the launches on the context's stream and returns the average milliseconds of
 * [0] the stress phase (or, when the fused path is active, one LAUNCH of the fused kernel -- one or two
 *     sub-steps, see csi_last_launches -- then [1] = [2] = 0),
 * [1] the u step, [2] the v step, [3] the halo exchange (0 on an untiled grid).
 * Synchronises; for bench.py's roofline only, never on the timed path. */
int32_t csi_profile_substeps(csi_context* ctx, double dt, int32_t substeps, double* out_ms4);
/* Which path the last sub-cycle took: three kernels (0), fused kernel (1), fused pairs of sub-steps (2); the
 * halo-exchange interval k and the number of exchanges issued.  Any pointer may be NULL. */
int32_t csi_last_path(csi_context* ctx, int32_t* fused, int32_t* exchange_interval, int32_t* exchanges);
/* Kernel launches and sub-steps of the last fused sub-cycle (sub-steps / launches = sub-steps per launch). */
int32_t csi_last_launches(csi_context* ctx, int32_t* launches, int32_t* substeps);
/* Number of kernel launches issued for one sub-step in the current configuration (upper bound). */
int32_t csi_launches_per_substep(csi_context* ctx, int32_t* n);

#ifdef __cplusplus
}
#endif
#endif /* CSI_H */
