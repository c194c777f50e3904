/*
 * csi_oracle.h -- CPU ORACLE for the ClimaSeaIce hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a strict-order, plain-C restatement of the reference algorithm
 * (CliMA/ClimaSeaIce.jl v0.5.8, Julia).  It is NOT the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only
 * as the checker.  The product path (climaseaice.jl_amd/csrc, libcsi_hip.so)
 * never links or calls anything in this directory.
 *
 * PARITY STATUS: "parity unpinned" for absolute values.  The reference is Julia
 * and its arithmetic dependency Oceananigans.jl (compat 0.110/0.111, un-vendored,
 * no Manifest) is absent from /root/reference and from this image, so neither can
 * run here.  The reference's tests hold no golden numbers for this path
 * (SURVEY.md section 8c); what they do pin -- the strain/divergence adjoint identity
 * (test/test_rheology_energy_budget.jl:95-125), the drag bound
 * (test/test_time_stepping.jl:56-80), decomposition invariance
 * (test/distributed_tests_utils.jl:40-137), slab closure -- is checked against
 * this oracle in tests/test_oracle_*.py.  Upstream (Oceananigans) operator
 * semantics used here are the ones recorded in SURVEY.md App. B.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/src).  Operation order inside each expression is the
 * reference's order; build with -O2 -ffp-contract=off (see Makefile).
 *
 * Index convention: all (i, j) are the reference's 1-based indices; a field
 * with halo (Hx, Hy) stores element (i, j) at p[(i + Hx - 1) + (j + Hy - 1) * ld]
 * (column-major Oceananigans parent array, i fastest).
 */
#ifndef CSI_ORACLE_H
#define CSI_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORA_PERIODIC = 0, ORA_BOUNDED = 1, ORA_FULLY_CONNECTED = 2, ORA_LEFT_CONNECTED = 3, ORA_RIGHT_CONNECTED = 4,
       /* y direction of a TripolarGrid: low side a wall (southernmost latitude), high side the north FOLD (Zipper boundary
        * condition, sea_ice_model.jl:57-64); _CONNECTED_FOLDED: the northernmost tile of a y partition */
       ORA_RIGHT_FOLDED = 5, ORA_LEFT_CONNECTED_RIGHT_FOLDED = 6 };
enum { ORA_METRIC_UNIFORM = 0, ORA_METRIC_PER_J = 1, ORA_METRIC_FULL = 2 };
enum { ORA_STRESS_NONE = 0, ORA_STRESS_CONST = 1, ORA_STRESS_FIELD = 2, ORA_STRESS_SEMI_IMPLICIT = 3 };
enum { ORA_VEL_ZERO = 0, ORA_VEL_CONST = 1, ORA_VEL_FIELD = 2 };
enum { ORA_PRESSURE_REPLACEMENT = 0, ORA_PRESSURE_ICE_STRENGTH = 1 };
enum { ORA_LOC_CENTER = 0, ORA_LOC_FACE = 1 };
enum { ORA_BC_PERIODIC = 0, ORA_BC_MIRROR = 1, ORA_BC_NONE = 2, ORA_BC_VALUE = 3, ORA_BC_FOLD = 4 };

typedef struct {
    double* p;     /* parent array start (element (1-Hx, 1-Hy)) */
    int64_t ld;    /* leading dimension (elements) */
} ora_field;

/* An external stress (top = atmosphere, bottom = ocean), sea_ice_external_stress.jl:8-37,84-130 */
typedef struct {
    int32_t kind;            /* ORA_STRESS_* */
    int32_t ue_kind, ve_kind;/* ORA_VEL_* for SEMI_IMPLICIT external velocities */
    int32_t pad;
    double tau_u, tau_v;     /* CONST: stress values (N m^-2) */
    ora_field fu, fv;        /* FIELD: stress arrays at (f,c) and (c,f); SEMI_IMPLICIT+VEL_FIELD: ue, ve */
    double ue, ve;           /* SEMI_IMPLICIT + VEL_CONST */
    double rho_e, Cd;        /* SEMI_IMPLICIT */
} ora_stress;

typedef struct {
    /* ---- grid ---- */
    int32_t Nx, Ny, Hx, Hy;
    int32_t topo_x, topo_y;      /* ORA_PERIODIC / ORA_BOUNDED */
    int32_t metric_kind;         /* ORA_METRIC_* */
    int32_t has_mask;
    double dx, dy;               /* UNIFORM: spacings; PER_J: dy only */
    /* PER_J vectors, element for row j stored at [j + Hy - 1], length Ny + 2Hy + 1:
     * dxc = dx at Center-j rows (dx^cc = dx^fc), dxf = dx at Face-j rows (dx^cf = dx^ff),
     * azc = Az at Center-j rows (Az^cc = Az^fc), azf = Az at Face-j rows (Az^cf = Az^ff). */
    const double *dxc, *dxf, *azc, *azf;
    const uint8_t* mask;         /* cell-centred activity mask (1 = active), same layout as a (c,c) field, ld = mask_ld */
    int64_t mask_ld;

    /* ---- rheology, elasto_visco_plastic_rheology.jl:14-25,119-137 ---- */
    double P_star, C_star, ecc, delta_min, alpha_min, alpha_max, c_alpha;
    int32_t pressure_kind;
    int32_t substeps;
    /* ---- momentum equation, sea_ice_momentum_equations.jl:67-94 ---- */
    double min_mass, min_conc;
    double rho_ice;              /* sea_ice_density (ConstantField), sea_ice_model.jl:142-145,193 */
    double f_coriolis;           /* FPlane f; 0 with has_coriolis = 0 means `nothing` */
    int32_t has_coriolis;
    int32_t free_drift_kind;     /* 0 = nothing (zero), 1 = StressBalanceFreeDrift (exactly one stress semi-implicit) */
    /* BetaPlane (upstream Coriolis; reference test matrix test/test_time_stepping.jl:35):
     * f = f0 + beta * ynode, evaluated at the (Face, Center) nodes for x_f_cross_U and at the (Center, Face) nodes
     * for y_f_cross_U.  The caller evaluates it per row: fu_rows / fv_rows, element for row j at [j + Hy - 1],
     * length Ny + 2Hy + 1; NULL = FPlane (f_coriolis). */
    const double *fu_rows, *fv_rows;
    /* ORA_METRIC_FULL (orthogonal curvilinear grids: OrthogonalSphericalShellGrid and the like): the twelve metric
     * arrays dx, dy, Az at (c,c), (f,c), (c,f), (f,f) -- index = 4 * {dx: 0, dy: 1, Az: 2} + (x at Face) + 2 * (y at
     * Face) -- each laid out like a parent array with leading dimension m2d_ld >= Nx + 2Hx + 1 and Ny + 2Hy + 1 rows. */
    const double* m2d[12];
    int64_t m2d_ld;
    ora_stress top, bottom;

    /* ---- fields ---- */
    ora_field u, v;              /* (f,c), (c,f) */
    ora_field h, aice;           /* (c,c) thickness, concentration */
    ora_field s11, s22, s12;     /* (c,c),(c,c),(f,f) */
    ora_field zeta_c, zeta_f, Delta, alpha, P;  /* (c,c),(f,f),(c,c),(c,c),(c,c) */
    ora_field un, vn;            /* (f,c),(c,f) */
    ora_field Gh, Ga;            /* tracer tendencies (c,c) */
    ora_field hm, am, um, vm;    /* Psi^- cache for RK3: h, aice, u, v */
    ora_field hs, Ghs, hsm;      /* snow thickness, its tendency and Psi^- copy (used when has_snow) */
    int32_t has_snow;
    int32_t weno_weights_f32;    /* 1: WENO smoothness indicators / weights in single precision (upstream's FT2 = Float32, recalled: csi_oracle.c) */
    /* ValueBoundaryCondition on the tangential velocity at a wall (no-slip: examples/ice_advected_on_coastline.jl:96-99):
     * u on the south / north walls, v on the west / east walls; [0] low side, [1] high side.  Upstream fills ONE halo
     * cell, c[0] = 2 val - c[1]; deeper halo cells are left alone.  Default (on = 0): no-flux mirror. */
    int32_t u_value_on[2], v_value_on[2];
    double u_value[2], v_value[2];
    /* model.forcing.u / .v given as arrays (the `user_forcing` of sum_of_forcing_u / _v, evp:391-401): values at the
     * (f,c) / (c,f) points, parent-shaped like u / v; has_forcing = 0: the reference's default (zero forcing). */
    int32_t has_forcing, pad_forcing;
    ora_field forcing_u, forcing_v;
    /* Immersed FluxBoundaryConditions of u and v with NUMBER values, [0..3] = west, east, south, north
     * (immersed_dj_sigma_1j / _2j, ice_stress_divergence.jl:65-123): all zero = the reference's default. */
    double ibc_u[4], ibc_v[4];
    /* per-point Coriolis parameter at the u / v points (metric-plane layout: element (i, j) at [(i + Hx - 1) + (j + Hy - 1) * ld]);
     * NULL: rows / f_coriolis.  Same FPlane / BetaPlane stencil (see include/csi.h csi_coriolis_points_set). */
    const double *fu_points, *fv_points;
    int64_t f_points_ld;
} ora_problem;

/* ---- grid metric accessors (Oceananigans operators) ---- */
double ora_dx(const ora_problem* g, int lx, int ly, int i, int j);
double ora_dy(const ora_problem* g, int lx, int ly, int i, int j);
double ora_az(const ora_problem* g, int lx, int ly, int i, int j);

/* ---- pieces exposed for tests ---- */
double ora_strain_xx(const ora_problem* g, int i, int j);   /* elasto_visco_plastic_rheology.jl:373 */
double ora_strain_yy(const ora_problem* g, int i, int j);   /* :374 */
double ora_strain_xy(const ora_problem* g, int i, int j);   /* :375 */
double ora_div_sigma_1(const ora_problem* g, int i, int j); /* ice_stress_divergence.jl:39-44 */
double ora_div_sigma_2(const ora_problem* g, int i, int j); /* ice_stress_divergence.jl:46-51 */
double ora_immersed_div_sigma_1(const ora_problem* g, int i, int j); /* :65-85, FluxBoundaryCondition numbers :115-123 */
double ora_immersed_div_sigma_2(const ora_problem* g, int i, int j); /* :87-107 */
/* pre-v0.5.8 flux-form divergence kept by the reference's own test for contrast
 * (test/test_rheology_energy_budget.jl:22-32) */
double ora_old_div_sigma_1(const ora_problem* g, int i, int j);
double ora_old_div_sigma_2(const ora_problem* g, int i, int j);
int32_t ora_peripheral_u(const ora_problem* g, int i, int j);
int32_t ora_peripheral_v(const ora_problem* g, int i, int j);

/* ---- the reference's kernels, one function each ---- */
void ora_initialize_rheology(ora_problem* g);                                   /* evp:192-219 */
void ora_compute_viscosities(ora_problem* g, int i0, int i1, int j0, int j1);   /* evp:236-273 */
void ora_compute_stresses(ora_problem* g, double dt, int i0, int i1, int j0, int j1); /* evp:294-354 */
void ora_u_velocity_step(ora_problem* g, double dt, int i0, int i1, int j0, int j1);  /* split_explicit:197-229 */
void ora_v_velocity_step(ora_problem* g, double dt, int i0, int i1, int j0, int j1);  /* split_explicit:231-264 */
/* fill_halo_regions!(...; only_local_halos=true) for one field (upstream) */
void ora_fill_halo(const ora_problem* g, ora_field f, int lx, int ly, int bcx, int bcy);
void ora_fill_halo4(const ora_problem* g, ora_field f, int bxlo, int bxhi, int bylo, int byhi);
void ora_fill_halo_u(ora_problem* g);
void ora_fill_halo_v(ora_problem* g);
void ora_fill_halo_center(ora_problem* g, ora_field f);
/* local halo fill of a field at (lx, ly) with the boundary conditions its location implies on this grid; sign: the Zipper
 * sign of a north fold (+1 scalars and tensor components, -1 vector components) */
void ora_fill_halo_loc(const ora_problem* g, ora_field f, int lx, int ly, int fold_sign);
void ora_finalize_rheology(ora_problem* g);                                     /* evp:275-280 */

/* time_step_momentum!(model, ::SplitExplicitMomentumEquation, dt), split_explicit:103-195.
 * rk_reset != 0 performs reset_velocities! from um, vm (:89-93). */
void ora_time_step_momentum(ora_problem* g, double dt, int rk_reset);
/* the sub-step loop only (:173-189), substeps first..last (1-based parity as in the reference) */
void ora_subcycle(ora_problem* g, double dt, int first, int last);

/* ---- advection + tracer update ---- */
/* _compute_dynamic_tracer_tendencies!, tracer_tendency_kernel_functions.jl:27-45 with
 * horizontal_div_Uc, sea_ice_advection.jl:51-58 and upstream WENO(order);
 * scheme: 5 or 7 = WENO order, 1 = first-order upwind, -5 = UpwindBiased(order=5) */
void ora_compute_tracer_tendencies(ora_problem* g, int scheme);
/* _dynamic_step_tracers!, sea_ice_fe_step.jl:56-82; from_cache: (h^n, aice^n) = Psi^- (sea_ice_rk_substep.jl:140-149) */
void ora_dynamic_step_tracers(ora_problem* g, double dt, int from_cache);
double ora_weno_flux_x(const ora_problem* g, int scheme, ora_field c, int i, int j);
int ora_test_weno(int order, const double* p, double* out);   /* test hook: parts of one reconstruction */
double ora_weno_flux_y(const ora_problem* g, int scheme, ora_field c, int i, int j);

/* ---- whole steps: FE (sea_ice_fe_step.jl:13-34) and RK3 (sea_ice_rk_substep.jl:29-94 + upstream stage loop) ---- */
void ora_update_state(ora_problem* g);                                          /* sea_ice_model.jl:379-394 */
void ora_time_step_fe(ora_problem* g, double dt, int scheme, int first_iteration);
void ora_time_step_rk3(ora_problem* g, double dt, int scheme);

/* ---- bare-ice slab thermodynamics (plumbing), thermodynamic_time_step.jl:75-118,304-370 ---- */
typedef struct {
    double k_ice;        /* conductivity, slab ConductiveFlux */
    double rho_bulk;     /* sea_ice_density (bulk, 900) */
    double rho_pure;     /* PhaseTransitions.density (917) */
    double rho_liquid, c_liquid, c_ice, L0, T0; /* PhaseTransitions, SeaIceThermodynamics.jl:71-170 */
    double liq_slope, liq_T0;                   /* LinearLiquidus: Tm = T0 - slope * S */
    double salinity;
    double h_consolidation;
    int32_t top_bc_kind;   /* 0 = PrescribedTemperature(Tu); 1 = MeltingConstrainedFluxBalance with a NUMERIC external flux:
                            * the secant solve of Qx - Qi(T) = 0 (top_heat_boundary_conditions.jl:80-97, RootSolvers, absent)
                            * has the closed-form root T = Tb - Qx R of the linear conductive flux, capped at Tm */
    int32_t top_flux_kind; /* 0 = const Qu, 1 = internal-flux equilibrium (default for PrescribedTemperature) */
    int32_t bot_flux_kind; /* 0 = const Qb, 1 = FluxFunction -(1 - aice) * Qb (examples/freezing_bucket.jl:79-81, Qb = 1) */
    int32_t pad;
    double Tu, Qu, Qb;
    double ice_salinity;   /* model.ice_salinity (fields.S): Tm = melting_temperature(liquidus, S) caps the solved Tu */
} ora_slab;
/* snow layer on top of the slab: snow_slab_thermodynamics (slab_sea_ice_thermodynamics.jl:42-49), snow_density and
 * snowfall of SeaIceModel (sea_ice_model.jl) */
typedef struct {
    double k_snow;       /* conductivity 0.31 */
    double rho_snow;     /* snow_density 330 */
    double snowfall;     /* kg m^-2 s^-1, constant */
    double Tu;           /* PrescribedTemperature of the snow surface (top_bc_kind 0) */
    int32_t top_bc_kind; /* as ora_slab.top_bc_kind, for the snow surface */
    int32_t pad;
} ora_snow;
/* _layered_thermodynamic_time_step!, thermodynamic_time_step.jl:131-298, on n independent cells; the mass_flux
 * arrays (thermodynamics.ice, thermodynamics.snow, intercepted_snowfall) and the two temperature outputs (ice top =
 * snow-ice interface, snow top) may be NULL */
void ora_layered_thermo_step(const ora_slab* s, const ora_snow* w, int64_t n, double* h, double* aice, double* hs,
                             double* mf_ice, double* mf_snow, double* mf_int, double* tu_ice, double* tu_snow, double dt);
void ora_layered_step_fields(ora_problem* g, const ora_slab* s, const ora_snow* w, double dt);
void ora_time_step_fe_snow(ora_problem* g, double dt, int scheme, int first_iteration, const ora_slab* s, const ora_snow* w);
void ora_time_step_rk3_snow(ora_problem* g, double dt, int scheme, const ora_slab* s, const ora_snow* w);
void ora_slab_thermo_step(const ora_slab* s, int64_t n, double* h, double* aice, double* mass_flux, double dt);
void ora_slab_step_fields(ora_problem* g, const ora_slab* s, double dt);
void ora_time_step_fe_thermo(ora_problem* g, double dt, int scheme, int first_iteration, const ora_slab* s);
void ora_time_step_rk3_thermo(ora_problem* g, double dt, int scheme, const ora_slab* s);

#ifdef __cplusplus
}
#endif
#endif
