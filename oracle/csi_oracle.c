/*
 * csi_oracle.c -- CPU ORACLE (test infrastructure only; see csi_oracle.h header).
 *
 * Strict-order plain-C restatement of the ClimaSeaIce.jl hot path:
 * EVP sub-cycle, WENO advection of h / aice, tracer update, FE / RK3 stage loop.
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile).
 * "parity unpinned": see csi_oracle.h.
 *
 * Each function cites the reference file:line (relative to /root/reference/src)
 * whose arithmetic it restates; the floating-point operation order inside every
 * expression is the reference's (Julia: left-assoc n-ary + and *, x^2 == x*x,
 * x^(-2) == inv(x)*inv(x), a / 2b == a / (2*b)).
 */
#include "csi_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef ORA_OMP
#define OMP_ROWS _Pragma("omp parallel for schedule(static)")
#else
#define OMP_ROWS
#endif

static inline double jmax(double a, double b);
#define C_ ORA_LOC_CENTER
#define F_ ORA_LOC_FACE

/* element (i, j), 1-based reference indices */
#define AT(g, f, i, j) ((f).p[((int64_t)(i) + (g)->Hx - 1) + ((int64_t)(j) + (g)->Hy - 1) * (f).ld])

/* ------------------------------------------------------------------------ */
/* Grid metrics (Oceananigans.Operators)                  */
/* Rectilinear regular: constants, Az = dx*dy.  Lat-lon regular: per-j rows */
/* ------------------------------------------------------------------------ */
static inline double metric2d(const ora_problem* g, int which, int lx, int ly, int i, int j) {
    return g->m2d[4 * which + (lx == F_ ? 1 : 0) + (ly == F_ ? 2 : 0)][((int64_t)i + g->Hx - 1) + ((int64_t)j + g->Hy - 1) * g->m2d_ld];
}
double ora_dx(const ora_problem* g, int lx, int ly, int i, int j) {
    if (g->metric_kind == ORA_METRIC_FULL) return metric2d(g, 0, lx, ly, i, j);
    if (g->metric_kind == ORA_METRIC_UNIFORM) return g->dx;
    return (ly == C_) ? g->dxc[j + g->Hy - 1] : g->dxf[j + g->Hy - 1];
}
double ora_dy(const ora_problem* g, int lx, int ly, int i, int j) {
    if (g->metric_kind == ORA_METRIC_FULL) return metric2d(g, 1, lx, ly, i, j);
    return g->dy;
}
double ora_az(const ora_problem* g, int lx, int ly, int i, int j) {
    if (g->metric_kind == ORA_METRIC_FULL) return metric2d(g, 2, lx, ly, i, j);
    if (g->metric_kind == ORA_METRIC_UNIFORM) return g->dx * g->dy;
    return (ly == C_) ? g->azc[j + g->Hy - 1] : g->azf[j + g->Hy - 1];
}

/* walls of a (tile of a) domain: Bounded has both; a tile at the end of a Bounded direction has one
 * (ORA_RIGHT_CONNECTED: low side wall, ORA_LEFT_CONNECTED: high side wall; connected sides are filled by the
 * halo exchange of the multi-process tests, never by a local boundary condition) */
static int wall_lo(int topo) { return topo == ORA_BOUNDED || topo == ORA_RIGHT_CONNECTED || topo == ORA_RIGHT_FOLDED; }
static int fold_hi(int topo) { return topo == ORA_RIGHT_FOLDED || topo == ORA_LEFT_CONNECTED_RIGHT_FOLDED; }
static int wall_hi(int topo) { return topo == ORA_BOUNDED || topo == ORA_LEFT_CONNECTED; }
static int outside_walls(const ora_problem* g, int i, int j) {
    if ((wall_lo(g->topo_x) && i < 1) || (wall_hi(g->topo_x) && i > g->Nx)) return 1;
    if ((wall_lo(g->topo_y) && j < 1) || (wall_hi(g->topo_y) && j > g->Ny)) return 1;
    return 0;
}
/* inactive_cell: outside a Bounded domain or immersed (upstream Grids.inactive_cell) */
static int inactive_cell(const ora_problem* g, int i, int j) {
    if (outside_walls(g, i, j)) return 1;
    if (g->has_mask) {
        /* clamp into the stored array; halos of the mask are filled by the caller */
        if (i < 1 - g->Hx || i > g->Nx + g->Hx || j < 1 - g->Hy || j > g->Ny + g->Hy) return 1;
        return !g->mask[((int64_t)i + g->Hx - 1) + ((int64_t)j + g->Hy - 1) * g->mask_ld];
    }
    return 0;
}
/* the same for the underlying (non-immersed) grid */
static int inactive_cell_underlying(const ora_problem* g, int i, int j) { return outside_walls(g, i, j); }
/* peripheral_node(i,j,k,grid,Face,Center,Center), split_explicit_momentum_equations.jl:226 */
int32_t ora_peripheral_u(const ora_problem* g, int i, int j) {
    return inactive_cell(g, i, j) | inactive_cell(g, i - 1, j);
}
/* peripheral_node(i,j,k,grid,Center,Face,Center), split_explicit_momentum_equations.jl:261 */
int32_t ora_peripheral_v(const ora_problem* g, int i, int j) {
    return inactive_cell(g, i, j) | inactive_cell(g, i, j - 1);
}
/* immersed_peripheral_node at (c,c) and (f,f): peripheral on the immersed grid but not on
 * the underlying grid (upstream ImmersedBoundaries); used by conditional_flux_ccc/ffc,
 * ice_stress_divergence.jl:21-24 */
static int immersed_peripheral_cc(const ora_problem* g, int i, int j) {
    if (!g->has_mask) return 0;
    return inactive_cell(g, i, j) && !inactive_cell_underlying(g, i, j);
}
static int immersed_peripheral_ff(const ora_problem* g, int i, int j) {
    if (!g->has_mask) return 0;
    int p = inactive_cell(g, i, j) | inactive_cell(g, i - 1, j) | inactive_cell(g, i, j - 1) | inactive_cell(g, i - 1, j - 1);
    int pu = inactive_cell_underlying(g, i, j) | inactive_cell_underlying(g, i - 1, j) |
             inactive_cell_underlying(g, i, j - 1) | inactive_cell_underlying(g, i - 1, j - 1);
    return p && !pu;
}

/* ice_mass, ClimaSeaIce.jl:42 : h * rho * aice (left to right) */
static inline double ice_mass(const ora_problem* g, int i, int j) {
    return AT(g, g->h, i, j) * g->rho_ice * AT(g, g->aice, i, j);
}

/* ------------------------------------------------------------------------ */
/* Strain rates, elasto_visco_plastic_rheology.jl:360-375                    */
/* ------------------------------------------------------------------------ */
static double eps_D(const ora_problem* g, int i, int j) {   /* :365 */
    double a = ora_dy(g, F_, C_, i + 1, j) * AT(g, g->u, i + 1, j) - ora_dy(g, F_, C_, i, j) * AT(g, g->u, i, j);
    double b = ora_dx(g, C_, F_, i, j + 1) * AT(g, g->v, i, j + 1) - ora_dx(g, C_, F_, i, j) * AT(g, g->v, i, j);
    return (a + b) / ora_az(g, C_, C_, i, j);
}
static double eps_T(const ora_problem* g, int i, int j) {   /* :367-368 */
    double dycc = ora_dy(g, C_, C_, i, j), dxcc = ora_dx(g, C_, C_, i, j);
    double a = AT(g, g->u, i + 1, j) / ora_dy(g, F_, C_, i + 1, j) - AT(g, g->u, i, j) / ora_dy(g, F_, C_, i, j);
    double b = AT(g, g->v, i, j + 1) / ora_dx(g, C_, F_, i, j + 1) - AT(g, g->v, i, j) / ora_dx(g, C_, F_, i, j);
    return ((dycc * dycc) * a - (dxcc * dxcc) * b) / ora_az(g, C_, C_, i, j);
}
static double eps_S(const ora_problem* g, int i, int j) {   /* :370-371, at corner (i,j) */
    double dxff = ora_dx(g, F_, F_, i, j), dyff = ora_dy(g, F_, F_, i, j);
    double a = AT(g, g->u, i, j) / ora_dx(g, F_, C_, i, j) - AT(g, g->u, i, j - 1) / ora_dx(g, F_, C_, i, j - 1);
    double b = AT(g, g->v, i, j) / ora_dy(g, C_, F_, i, j) - AT(g, g->v, i - 1, j) / ora_dy(g, C_, F_, i - 1, j);
    return ((dxff * dxff) * a + (dyff * dyff) * b) / ora_az(g, F_, F_, i, j);
}
double ora_strain_xx(const ora_problem* g, int i, int j) { return (eps_D(g, i, j) + eps_T(g, i, j)) / 2; }
double ora_strain_yy(const ora_problem* g, int i, int j) { return (eps_D(g, i, j) - eps_T(g, i, j)) / 2; }
double ora_strain_xy(const ora_problem* g, int i, int j) { return eps_S(g, i, j) / 2; }

/* 4-point averages  Ixy = Iy(Ix(f))  (upstream) */
#define AVG4_FF(fn, g, i, j) ((((fn)(g, (i) - 1, (j) - 1) + (fn)(g, (i), (j) - 1)) / 2 + ((fn)(g, (i) - 1, (j)) + (fn)(g, (i), (j))) / 2) / 2)
#define AVG4_CC(fn, g, i, j) ((((fn)(g, (i), (j)) + (fn)(g, (i) + 1, (j))) / 2 + ((fn)(g, (i), (j) + 1) + (fn)(g, (i) + 1, (j) + 1)) / 2) / 2)

static inline double P_at(const ora_problem* g, int i, int j) { return AT(g, g->P, i, j); }

/* ------------------------------------------------------------------------ */
/* initialize_rheology! / _initialize_evp_rhology!, evp:192-219               */
/* launched over the whole parent array of P (evp:166-167)                    */
/* ------------------------------------------------------------------------ */
void ora_initialize_rheology(ora_problem* g) {
    OMP_ROWS
    for (int j = 1 - g->Hy; j <= g->Ny + g->Hy; ++j)
        for (int i = 1 - g->Hx; i <= g->Nx + g->Hx; ++i) {
            /* ice_strength :219 : P* * h * exp(-C * (1 - aice)) */
            AT(g, g->P, i, j) = g->P_star * AT(g, g->h, i, j) * exp(-g->C_star * (1 - AT(g, g->aice, i, j)));
            AT(g, g->un, i, j) = AT(g, g->u, i, j);
            AT(g, g->vn, i, j) = AT(g, g->v, i, j);
        }
}

/* ------------------------------------------------------------------------ */
/* _compute_evp_viscosities!, evp:236-273                                     */
/* ------------------------------------------------------------------------ */
void ora_compute_viscosities(ora_problem* g, int i0, int i1, int j0, int j1) {
    double ie = 1.0 / g->ecc;
    double em2 = ie * ie;                 /* e^(-2) == inv(e)^2 (literal_pow) */
    double Dm = g->delta_min;
    OMP_ROWS
    for (int j = j0; j <= j1; ++j)
        for (int i = i0; i <= i1; ++i) {
            double e11c = ora_strain_xx(g, i, j);
            double e22c = ora_strain_yy(g, i, j);
            double e12f = ora_strain_xy(g, i, j);
            double e11f = AVG4_FF(ora_strain_xx, g, i, j);
            double e22f = AVG4_FF(ora_strain_yy, g, i, j);
            double e12c = AVG4_CC(ora_strain_xy, g, i, j);

            double dc = e11c + e22c;
            double df = e11f + e22f;
            double sc = sqrt((e11c - e22c) * (e11c - e22c) + 4 * (e12c * e12c));
            double sf = sqrt((e11f - e22f) * (e11f - e22f) + 4 * (e12f * e12f));
            double Dc = jmax(sqrt(dc * dc + (sc * sc) * em2), Dm);
            double Df = jmax(sqrt(df * df + (sf * sf) * em2), Dm);
            double Pc = AT(g, g->P, i, j);
            double Pf = AVG4_FF(P_at, g, i, j);

            AT(g, g->zeta_f, i, j) = Pf / (2 * Df);
            AT(g, g->zeta_c, i, j) = Pc / (2 * Dc);
            AT(g, g->Delta, i, j) = Dc;
        }
}

/* Julia's max(a, b) for floats: NaN if either is NaN (C's fmax would drop the NaN) */
static inline double jmax(double a, double b) { return (a != a || b != b) ? a + b : (a < b ? b : a); }
static inline double jmin(double a, double b) { return (a != a || b != b) ? a + b : (b < a ? b : a); }
static inline double clampd(double x, double lo, double hi) { return x > hi ? hi : (x < lo ? lo : x); }

/* ------------------------------------------------------------------------ */
/* _compute_evp_stresses!, evp:294-354 (+ ice_pressure :282-289)              */
/* ------------------------------------------------------------------------ */
void ora_compute_stresses(ora_problem* g, double dt, int i0, int i1, int j0, int j1) {
    double ie = 1.0 / g->ecc;
    double em2 = ie * ie;
    double ap = g->alpha_max, am = g->alpha_min, ca = g->c_alpha;
    OMP_ROWS
    for (int j = j0; j <= j1; ++j)
        for (int i = i0; i <= i1; ++i) {
            double e11 = ora_strain_xx(g, i, j);
            double e22 = ora_strain_yy(g, i, j);
            double e12 = ora_strain_xy(g, i, j);
            double zc = AT(g, g->zeta_c, i, j);
            double zf = AT(g, g->zeta_f, i, j);

            double Pr;
            if (g->pressure_kind == ORA_PRESSURE_REPLACEMENT) {
                double Pc = AT(g, g->P, i, j), Dc = AT(g, g->Delta, i, j);
                Pr = Pc * Dc / (Dc + g->delta_min);
            } else {
                Pr = AT(g, g->P, i, j);
            }
            double etac = zc * em2;
            double etaf = zf * em2;

            double s11n = 2 * etac * e11 + ((zc - etac) * (e11 + e22) - Pr / 2);
            double s22n = 2 * etac * e22 + ((zc - etac) * (e11 + e22) - Pr / 2);
            double s12n = 2 * etaf * e12;

            double mc = ice_mass(g, i, j);
            double mf = AVG4_FF(ice_mass, g, i, j);

            double g2c = zc * ca * dt / mc / ora_az(g, C_, C_, i, j);
            g2c = isnan(g2c) ? ap * ap : g2c;
            double gc = clampd(sqrt(g2c), am, ap);

            double g2f = zf * ca * dt / mf / ora_az(g, F_, F_, i, j);
            g2f = isnan(g2f) ? ap * ap : g2f;
            double gf = clampd(sqrt(g2f), am, ap);

            double s11s = (s11n - AT(g, g->s11, i, j)) / gc;
            double s22s = (s22n - AT(g, g->s22, i, j)) / gc;
            double s12s = (s12n - AT(g, g->s12, i, j)) / gf;

            AT(g, g->s11, i, j) += (mc > 0) ? s11s : 0.0;
            AT(g, g->s22, i, j) += (mc > 0) ? s22s : 0.0;
            AT(g, g->s12, i, j) += (mf > 0) ? s12s : 0.0;
            AT(g, g->alpha, i, j) = gc;
        }
}

/* ------------------------------------------------------------------------ */
/* Stress divergence, ice_stress_divergence.jl:16-51                          */
/* ------------------------------------------------------------------------ */
static inline double sig11(const ora_problem* g, int i, int j) {     /* _ice_stress_ux :16,21 */
    return immersed_peripheral_cc(g, i, j) ? 0.0 : AT(g, g->s11, i, j);
}
static inline double sig22(const ora_problem* g, int i, int j) {     /* _ice_stress_vy :19,24 */
    return immersed_peripheral_cc(g, i, j) ? 0.0 : AT(g, g->s22, i, j);
}
static inline double sig12(const ora_problem* g, int i, int j) {     /* _ice_stress_uy/vx :17-18,22-23 */
    return immersed_peripheral_ff(g, i, j) ? 0.0 : AT(g, g->s12, i, j);
}
static inline double sigD(const ora_problem* g, int i, int j) { return sig11(g, i, j) + sig22(g, i, j); }  /* :36 */
static inline double sigT(const ora_problem* g, int i, int j) { return sig11(g, i, j) - sig22(g, i, j); }  /* :37 */

double ora_div_sigma_1(const ora_problem* g, int i, int j) {          /* :39-44 */
    double dyfc = ora_dy(g, F_, C_, i, j);
    double d = dyfc * (sigD(g, i, j) - sigD(g, i - 1, j)) / 2;
    double dyc = ora_dy(g, C_, C_, i, j), dycm = ora_dy(g, C_, C_, i - 1, j);
    double T = ((dyc * dyc) * sigT(g, i, j) - (dycm * dycm) * sigT(g, i - 1, j)) / dyfc / 2;
    double dxfn = ora_dx(g, F_, F_, i, j + 1), dxf = ora_dx(g, F_, F_, i, j);
    double S = ((dxfn * dxfn) * sig12(g, i, j + 1) - (dxf * dxf) * sig12(g, i, j)) / ora_dx(g, F_, C_, i, j);
    return (d + T + S) / ora_az(g, F_, C_, i, j);
}
double ora_div_sigma_2(const ora_problem* g, int i, int j) {          /* :46-51 */
    double dxcf = ora_dx(g, C_, F_, i, j);
    double d = dxcf * (sigD(g, i, j) - sigD(g, i, j - 1)) / 2;
    double dxc = ora_dx(g, C_, C_, i, j), dxcm = ora_dx(g, C_, C_, i, j - 1);
    double T = -((dxc * dxc) * sigT(g, i, j) - (dxcm * dxcm) * sigT(g, i, j - 1)) / dxcf / 2;
    double dyfn = ora_dy(g, F_, F_, i + 1, j), dyf = ora_dy(g, F_, F_, i, j);
    double S = ((dyfn * dyfn) * sig12(g, i + 1, j) - (dyf * dyf) * sig12(g, i, j)) / ora_dy(g, C_, F_, i, j);
    return (d + T + S) / ora_az(g, C_, F_, i, j);
}
/* flux-form operator the reference's test keeps for contrast, test/test_rheology_energy_budget.jl:22-32 */
double ora_old_div_sigma_1(const ora_problem* g, int i, int j) {
    double a = ora_dy(g, C_, C_, i, j) * AT(g, g->s11, i, j) - ora_dy(g, C_, C_, i - 1, j) * AT(g, g->s11, i - 1, j);
    double b = ora_dx(g, F_, F_, i, j + 1) * AT(g, g->s12, i, j + 1) - ora_dx(g, F_, F_, i, j) * AT(g, g->s12, i, j);
    return (a + b) / ora_az(g, F_, C_, i, j);
}
double ora_old_div_sigma_2(const ora_problem* g, int i, int j) {
    double a = ora_dy(g, F_, F_, i + 1, j) * AT(g, g->s12, i + 1, j) - ora_dy(g, F_, F_, i, j) * AT(g, g->s12, i, j);
    double b = ora_dx(g, C_, C_, i, j) * AT(g, g->s22, i, j) - ora_dx(g, C_, C_, i, j - 1) * AT(g, g->s22, i, j - 1);
    return (a + b) / ora_az(g, C_, F_, i, j);
}

/* ------------------------------------------------------------------------ */
/* External stresses, sea_ice_external_stress.jl:8-27,176-202                 */
/* ------------------------------------------------------------------------ */
static inline double fld(const ora_problem* g, ora_field f, int i, int j) { return AT(g, f, i, j); }
static inline double ext_ue(const ora_problem* g, const ora_stress* s, int i, int j) {
    return s->ue_kind == ORA_VEL_FIELD ? fld(g, s->fu, i, j) : (s->ue_kind == ORA_VEL_CONST ? s->ue : 0.0);
}
static inline double ext_ve(const ora_problem* g, const ora_stress* s, int i, int j) {
    return s->ve_kind == ORA_VEL_FIELD ? fld(g, s->fv, i, j) : (s->ve_kind == ORA_VEL_CONST ? s->ve : 0.0);
}
/* Ixy^{fc}(q)(i,j): y-average over (j, j+1) of x-averages over (i-1, i) */
#define AVG4_FC(expr_ij) ((( expr_ij(i - 1, j) + expr_ij(i, j)) / 2 + (expr_ij(i - 1, j + 1) + expr_ij(i, j + 1)) / 2) / 2)
/* Ixy^{cf}(q)(i,j): y-average over (j-1, j) of x-averages over (i, i+1) */
#define AVG4_CF(expr_ij) ((( expr_ij(i, j - 1) + expr_ij(i + 1, j - 1)) / 2 + (expr_ij(i, j) + expr_ij(i + 1, j)) / 2) / 2)

/* drag speed sqrt(du^2 + dv^2) at the u point (sea_ice_external_stress.jl:176-181,192-196) */
static double drag_norm_u(const ora_problem* g, const ora_stress* s, int i, int j) {
    double du = ext_ue(g, s, i, j) - AT(g, g->u, i, j);
#define VE_(ii, jj) ext_ve(g, s, ii, jj)
#define V_(ii, jj) AT(g, g->v, ii, jj)
    double dv = AVG4_FC(VE_) - AVG4_FC(V_);
#undef VE_
#undef V_
    return sqrt(du * du + dv * dv);
}
static double drag_norm_v(const ora_problem* g, const ora_stress* s, int i, int j) {   /* :183-188,198-202 */
    double dv = ext_ve(g, s, i, j) - AT(g, g->v, i, j);
#define UE_(ii, jj) ext_ue(g, s, ii, jj)
#define U_(ii, jj) AT(g, g->u, ii, jj)
    double du = AVG4_CF(UE_) - AVG4_CF(U_);
#undef UE_
#undef U_
    return sqrt(du * du + dv * dv);
}
static double explicit_tau_x(const ora_problem* g, const ora_stress* s, int i, int j) {
    switch (s->kind) {
    case ORA_STRESS_CONST: return s->tau_u;                       /* :16 */
    case ORA_STRESS_FIELD: return AT(g, s->fu, i, j);             /* :19 */
    case ORA_STRESS_SEMI_IMPLICIT:                                /* :176-181 */
        return s->rho_e * s->Cd * drag_norm_u(g, s, i, j) * ext_ue(g, s, i, j);
    default: return 0.0;                                          /* :13 */
    }
}
static double explicit_tau_y(const ora_problem* g, const ora_stress* s, int i, int j) {
    switch (s->kind) {
    case ORA_STRESS_CONST: return s->tau_v;
    case ORA_STRESS_FIELD: return AT(g, s->fv, i, j);
    case ORA_STRESS_SEMI_IMPLICIT:                                /* :183-188 */
        return s->rho_e * s->Cd * drag_norm_v(g, s, i, j) * ext_ve(g, s, i, j);
    default: return 0.0;
    }
}
static double implicit_tau_x(const ora_problem* g, const ora_stress* s, int i, int j) {
    if (s->kind == ORA_STRESS_SEMI_IMPLICIT) return s->rho_e * s->Cd * drag_norm_u(g, s, i, j);   /* :192-196 */
    return 0.0;                                                                                     /* :8 */
}
static double implicit_tau_y(const ora_problem* g, const ora_stress* s, int i, int j) {
    if (s->kind == ORA_STRESS_SEMI_IMPLICIT) return s->rho_e * s->Cd * drag_norm_v(g, s, i, j);   /* :198-202 */
    return 0.0;
}

/* ------------------------------------------------------------------------ */
/* StressBalanceFreeDrift closed forms, stress_balance_free_drift.jl:61-129   */
/* kind 1: exactly one of the two stresses is a SemiImplicitStress (which one  */
/* is read off the stress kinds: bottom semi-implicit = TISB :73-95, top       */
/* semi-implicit = BISB :99-121); the other one must not depend on the ice     */
/* velocity.  U_i = U_e - tau / sqrt(C ||tau||), zero where ||tau|| == 0.       */
/* ------------------------------------------------------------------------ */
static double free_drift_u(const ora_problem* g, int i, int j) {
    if (g->free_drift_kind == 0) return 0.0;                                 /* `nothing` :129 */
    const ora_stress* semi = (g->bottom.kind == ORA_STRESS_SEMI_IMPLICIT) ? &g->bottom : &g->top;
    const ora_stress* expl = (g->bottom.kind == ORA_STRESS_SEMI_IMPLICIT) ? &g->top : &g->bottom;
    double tx = explicit_tau_x(g, expl, i, j);                               /* x_momentum_stress :74 */
#define TY_(ii, jj) explicit_tau_y(g, expl, ii, jj)
    double ty = AVG4_FC(TY_);                                                /* Ixy^{fc}(y_momentum_stress) :75 */
#undef TY_
    double t = sqrt(tx * tx + ty * ty);                                      /* :76 */
    double ue = ext_ue(g, semi, i, j);                                       /* :79 */
    double C = semi->rho_e * semi->Cd;                                       /* :80 */
    return ue - ((t == 0) ? t : tx / sqrt(C * t));                           /* :82 */
}
static double free_drift_v(const ora_problem* g, int i, int j) {
    if (g->free_drift_kind == 0) return 0.0;
    const ora_stress* semi = (g->bottom.kind == ORA_STRESS_SEMI_IMPLICIT) ? &g->bottom : &g->top;
    const ora_stress* expl = (g->bottom.kind == ORA_STRESS_SEMI_IMPLICIT) ? &g->top : &g->bottom;
#define TX_(ii, jj) explicit_tau_x(g, expl, ii, jj)
    double tx = AVG4_CF(TX_);                                                /* Ixy^{cf}(x_momentum_stress) :86 */
#undef TX_
    double ty = explicit_tau_y(g, expl, i, j);                               /* :87 */
    double t = sqrt(tx * tx + ty * ty);
    double ve = ext_ve(g, semi, i, j);
    double C = semi->rho_e * semi->Cd;
    return ve - ((t == 0) ? t : ty / sqrt(C * t));                           /* :94 */
}
double ora_free_drift_u(const ora_problem* g, int i, int j) { return free_drift_u(g, i, j); }
double ora_free_drift_v(const ora_problem* g, int i, int j) { return free_drift_v(g, i, j); }

/* ------------------------------------------------------------------------ */
/* immersed_dj_sigma_1j / _2j, ice_stress_divergence.jl:65-107, with            */
/* FluxBoundaryCondition NUMBERS on the immersed boundary (:115-123: the stress */
/* is minus the flux on west / south faces, plus the flux on east / north).     */
/* conditional_flux_*(i, j, k, ibg, ib_flux, intrinsic): the immersed flux      */
/* where the node is an immersed peripheral node, else `intrinsic` (zero here). */
/* index_left / index_right (upstream): Face -> (i - 1, i), Center -> (i, i + 1).*/
/* Areas and volumes of a single-layer grid with dz = 1: Ax = dy, Ay = dx,      */
/* V = Az (dz cancels between the areas and the volume).                         */
/* ------------------------------------------------------------------------ */
double ora_immersed_div_sigma_1(const ora_problem* g, int i, int j) {
    if (!g->has_mask) return 0.0;                                            /* isd:57: zero(grid) on a non-immersed grid */
    double qtW = -g->ibc_u[0], qtE = g->ibc_u[1], qtS = -g->ibc_u[2], qtN = g->ibc_u[3];   /* :115-123 */
    int iW = i - 1, iE = i, jS = j, jN = j + 1;                             /* index_left / _right of (f, c) :72-73 */
    double qW = (immersed_peripheral_cc(g, iW, j) ? qtW : 0.0) * ora_dy(g, C_, C_, iW, j);     /* :76 Ax^{ccc} */
    double qE = (immersed_peripheral_cc(g, iE, j) ? qtE : 0.0) * ora_dy(g, C_, C_, iE, j);
    double qS = (immersed_peripheral_ff(g, i, jS) ? qtS : 0.0) * ora_dx(g, F_, F_, i, jS);     /* :78 Ay^{ffc} */
    double qN = (immersed_peripheral_ff(g, i, jN) ? qtN : 0.0) * ora_dx(g, F_, F_, i, jN);
    return (qE - qW + qN - qS) / ora_az(g, F_, C_, i, j);                    /* :81 / V^{fcc} */
}
double ora_immersed_div_sigma_2(const ora_problem* g, int i, int j) {
    if (!g->has_mask) return 0.0;
    double qtW = -g->ibc_v[0], qtE = g->ibc_v[1], qtS = -g->ibc_v[2], qtN = g->ibc_v[3];
    int iW = i, iE = i + 1, jS = j - 1, jN = j;                             /* (c, f) :94-95 */
    double qW = (immersed_peripheral_ff(g, iW, j) ? qtW : 0.0) * ora_dy(g, F_, F_, iW, j);     /* :98 Ax^{ffc} */
    double qE = (immersed_peripheral_ff(g, iE, j) ? qtE : 0.0) * ora_dy(g, F_, F_, iE, j);
    double qS = (immersed_peripheral_cc(g, i, jS) ? qtS : 0.0) * ora_dx(g, C_, C_, i, jS);     /* :100 Ay^{ccc} */
    double qN = (immersed_peripheral_cc(g, i, jN) ? qtN : 0.0) * ora_dx(g, C_, C_, i, jN);
    return (qE - qW + qN - qS) / ora_az(g, C_, F_, i, j);                    /* :106 / V^{cfc} */
}

/* ------------------------------------------------------------------------ */
/* u_velocity_tendency, momentum_tendencies_kernel_functions.jl:11-41         */
/* ------------------------------------------------------------------------ */
static double u_tendency(const ora_problem* g, int i, int j, double dtau) {
    double ai = (AT(g, g->aice, i - 1, j) + AT(g, g->aice, i, j)) / 2;       /* Ix^f(aice) */
    double mi = (ice_mass(g, i - 1, j) + ice_mass(g, i, j)) / 2;            /* Ix^f(ice_mass) */
    double cor = 0.0;                                                        /* x_f_cross_U, `nothing` -> zero */
    if (g->has_coriolis) {
#define V_(ii, jj) AT(g, g->v, ii, jj)
        double f = g->fu_points ? g->fu_points[((int64_t)i + g->Hx - 1) + ((int64_t)j + g->Hy - 1) * g->f_points_ld]
                                : (g->fu_rows ? g->fu_rows[j + g->Hy - 1] : g->f_coriolis);   /* BetaPlane: f0 + beta * y^{fc}(j) */
        cor = -f * AVG4_FC(V_);                                              /* FPlane: -f * Ixy^{fc}(v) */
#undef V_
    }
    double abar = (AT(g, g->alpha, i - 1, j) + AT(g, g->alpha, i, j)) / 2;  /* Ix^f(alpha), evp:393 */
    /* sum_of_forcing_u, evp:391-395: user forcing (zero, or an array value) + (un - u) / dt / Ix(alpha), with dt := dtau */
    double user = g->has_forcing ? AT(g, g->forcing_u, i, j) : 0.0;
    double forcing = user + (AT(g, g->un, i, j) - AT(g, g->u, i, j)) / dtau / abar;
    double imm = ora_immersed_div_sigma_1(g, i, j) / mi;                     /* isd:57 (zero) / :65-85 with FluxBoundaryCondition numbers */
    double G = (-cor
                - explicit_tau_x(g, &g->top, i, j) / mi * ai
                + explicit_tau_x(g, &g->bottom, i, j) / mi * ai
                + ora_div_sigma_1(g, i, j) / mi
                + imm
                + forcing);
    return (mi <= 0) ? 0.0 : G;                                              /* :38 */
}
static double v_tendency(const ora_problem* g, int i, int j, double dtau) {  /* :44-74 */
    double ai = (AT(g, g->aice, i, j - 1) + AT(g, g->aice, i, j)) / 2;
    double mi = (ice_mass(g, i, j - 1) + ice_mass(g, i, j)) / 2;
    double cor = 0.0;
    if (g->has_coriolis) {
#define U_(ii, jj) AT(g, g->u, ii, jj)
        double f = g->fv_points ? g->fv_points[((int64_t)i + g->Hx - 1) + ((int64_t)j + g->Hy - 1) * g->f_points_ld]
                                : (g->fv_rows ? g->fv_rows[j + g->Hy - 1] : g->f_coriolis);   /* BetaPlane: f0 + beta * y^{cf}(j) */
        cor = f * AVG4_CF(U_);                                               /* FPlane: +f * Ixy^{cf}(u) */
#undef U_
    }
    double abar = (AT(g, g->alpha, i, j - 1) + AT(g, g->alpha, i, j)) / 2;  /* evp:399 */
    double user = g->has_forcing ? AT(g, g->forcing_v, i, j) : 0.0;
    double forcing = user + (AT(g, g->vn, i, j) - AT(g, g->v, i, j)) / dtau / abar;
    double imm = ora_immersed_div_sigma_2(g, i, j) / mi;
    double G = (-cor
                - explicit_tau_y(g, &g->top, i, j) / mi * ai
                + explicit_tau_y(g, &g->bottom, i, j) / mi * ai
                + ora_div_sigma_2(g, i, j) / mi
                + imm
                + forcing);
    return (mi <= 0) ? 0.0 : G;
}

#define EPS64 2.220446049250313e-16

/* ------------------------------------------------------------------------ */
/* _u_velocity_step!, split_explicit_momentum_equations.jl:197-229            */
/* ------------------------------------------------------------------------ */
void ora_u_velocity_step(ora_problem* g, double dt, int i0, int i1, int j0, int j1) {
    OMP_ROWS
    for (int j = j0; j <= j1; ++j)
        for (int i = i0; i <= i1; ++i) {
            double mi = (ice_mass(g, i - 1, j) + ice_mass(g, i, j)) / 2;                 /* :205 */
            double ai = (AT(g, g->aice, i - 1, j) + AT(g, g->aice, i, j)) / 2;           /* :206 */
            double abar = (AT(g, g->alpha, i - 1, j) + AT(g, g->alpha, i, j)) / 2;
            double dtau = dt / abar;                                                     /* :208, evp:384 */
            double Gu = u_tendency(g, i, j, dtau);                                       /* :210 */
            double tau_i = (implicit_tau_x(g, &g->bottom, i, j) - implicit_tau_x(g, &g->top, i, j)) / mi * ai;  /* :214-215 */
            tau_i = (mi <= 0) ? 0.0 : tau_i;                                             /* :217 */
            double uD = (AT(g, g->u, i, j) + dtau * Gu) / (1 + dtau * tau_i);            /* :218 */
            double uF = free_drift_u(g, i, j);                                           /* :219 */
            int marginal = (mi > EPS64) & (ai > EPS64);                                  /* :224 */
            int active_ice = (mi >= g->min_mass) & (ai >= g->min_conc);                  /* :225 */
            int active = !ora_peripheral_u(g, i, j);                                     /* :226 */
            double sel = active_ice ? uD : (marginal ? uF : 0.0);
            /* `sel * active` with a Julia Bool: false is a strong zero (sign of sel kept)   :228 */
            AT(g, g->u, i, j) = active ? sel : copysign(0.0, sel);
        }
}
/* _v_velocity_step!, split_explicit_momentum_equations.jl:231-264 */
void ora_v_velocity_step(ora_problem* g, double dt, int i0, int i1, int j0, int j1) {
    OMP_ROWS
    for (int j = j0; j <= j1; ++j)
        for (int i = i0; i <= i1; ++i) {
            double mi = (ice_mass(g, i, j - 1) + ice_mass(g, i, j)) / 2;
            double ai = (AT(g, g->aice, i, j - 1) + AT(g, g->aice, i, j)) / 2;
            double abar = (AT(g, g->alpha, i, j - 1) + AT(g, g->alpha, i, j)) / 2;
            double dtau = dt / abar;                                                     /* evp:385 */
            double Gv = v_tendency(g, i, j, dtau);
            double tau_i = (implicit_tau_y(g, &g->bottom, i, j) - implicit_tau_y(g, &g->top, i, j)) / mi * ai;
            tau_i = (mi <= 0) ? 0.0 : tau_i;
            double vD = (AT(g, g->v, i, j) + dtau * Gv) / (1 + dtau * tau_i);
            double vF = free_drift_v(g, i, j);
            int marginal = (mi > EPS64) & (ai > EPS64);
            int active_ice = (mi >= g->min_mass) & (ai >= g->min_conc);
            int active = !ora_peripheral_v(g, i, j);
            double sel = active_ice ? vD : (marginal ? vF : 0.0);
            AT(g, g->v, i, j) = active ? sel : copysign(0.0, sel);
        }
}

/* ------------------------------------------------------------------------ */
/* fill_halo_regions!(...; only_local_halos = true) (upstream) */
/* x sides over interior rows first, then y sides over the whole x extent.    */
/* Periodic: wrap.  MIRROR (no-flux, centre location in a Bounded direction): */
/*   c[1-m] = c[m], c[N+m] = c[N+1-m].  NONE: nothing (Face location in a     */
/*   Bounded direction: impenetrable / auxiliary default).                    */
/* ------------------------------------------------------------------------ */
static void fill_halo4v(const ora_problem* g, ora_field f, int bxlo, int bxhi, int bylo, int byhi, const double* vx, const double* vy);
void ora_fill_halo4(const ora_problem* g, ora_field f, int bxlo, int bxhi, int bylo, int byhi) {
    static const double zero[2] = {0.0, 0.0};
    fill_halo4v(g, f, bxlo, bxhi, bylo, byhi, zero, zero);
}
/* North fold of a TripolarGrid (upstream Zipper boundary condition; RECALLED semantics, SURVEY.md App. B -- Oceananigans is
 * not under /root/reference): the fold runs through the cell CENTRES of row Ny ("the Ny line is duplicated"), so for
 * j = 1 .. Hy and i = 1 .. Nx
 *     c[i, Ny + j] = s * c[i', Ny - j]      Center in y          c[i, Ny + j] = s * c[i', Ny - j + 1]      Face in y
 *     i' = Nx - i + 1  (Center in x)        i' = Nx - i + 2  (Face in x; i' = Nx + 1 is column 1 again: there s := |s|)
 * with s = -1 for the velocity components (sea_ice_model.jl:57-64 flips the default) and +1 otherwise; x is Periodic, and
 * the x halos of the folded rows are filled from the folded rows (x pass over rows 1..Ny, fold, x pass over the fold rows). */
static void fold_north(const ora_problem* g, ora_field f, int lx, int ly, int sign) {
    int Nx = g->Nx, Ny = g->Ny, Hx = g->Hx, Hy = g->Hy;
    for (int m = 1; m <= Hy; ++m) {
        int js = (ly == C_) ? Ny - m : Ny - m + 1;
        for (int i = 1; i <= Nx; ++i) {
            int ip = (lx == C_) ? Nx - i + 1 : Nx - i + 2;
            double s = (double)sign;
            if (ip > Nx) { ip -= Nx; s = fabs(s); }
            AT(g, f, i, Ny + m) = s * AT(g, f, ip, js);
        }
        for (int k = 1; k <= Hx; ++k) {                    /* periodic x images of the folded row */
            AT(g, f, 1 - k, Ny + m) = AT(g, f, Nx + 1 - k, Ny + m);
            AT(g, f, Nx + k, Ny + m) = AT(g, f, k, Ny + m);
        }
    }
}
/* vx / vy: values of ORA_BC_VALUE sides ([0] low, [1] high) */
static void fill_halo4v(const ora_problem* g, ora_field f, int bxlo, int bxhi, int bylo, int byhi, const double* vx, const double* vy) {
    int Nx = g->Nx, Ny = g->Ny, Hx = g->Hx, Hy = g->Hy;
    /* x sides over the interior rows (upstream order: x first, then y over the whole x extent); on a tile
     * whose y side is connected (no local y pass there) the rows beyond it are included, because the ring
     * rows recomputed for the neighbours need their x images too. */
    int jlo = (g->topo_y == ORA_FULLY_CONNECTED || g->topo_y == ORA_LEFT_CONNECTED || g->topo_y == ORA_LEFT_CONNECTED_RIGHT_FOLDED) ? 1 - Hy : 1;
    int jhi = (g->topo_y == ORA_FULLY_CONNECTED || g->topo_y == ORA_RIGHT_CONNECTED) ? Ny + Hy : Ny;
    /* a Face-in-y field on a high y wall has one more row of points (the wall faces, row Ny + 1): its x images too */
    if (wall_hi(g->topo_y) && byhi == ORA_BC_NONE && bylo == ORA_BC_NONE) jhi = Ny + 1;
    for (int j = jlo; j <= jhi; ++j)
        for (int m = 1; m <= Hx; ++m) {
            if (bxlo == ORA_BC_PERIODIC) AT(g, f, 1 - m, j) = AT(g, f, Nx + 1 - m, j);
            else if (bxlo == ORA_BC_MIRROR) AT(g, f, 1 - m, j) = AT(g, f, m, j);
            if (bxhi == ORA_BC_PERIODIC) AT(g, f, Nx + m, j) = AT(g, f, m, j);
            else if (bxhi == ORA_BC_MIRROR) AT(g, f, Nx + m, j) = AT(g, f, Nx + 1 - m, j);
            if (m == 1 && bxlo == ORA_BC_VALUE) AT(g, f, 0, j) = 2 * vx[0] - AT(g, f, 1, j);          /* one halo cell */
            if (m == 1 && bxhi == ORA_BC_VALUE) AT(g, f, Nx + 1, j) = 2 * vx[1] - AT(g, f, Nx, j);
        }
    /* y sides cover the full stored x extent (including the extra Face column, if any) */
    int64_t nxs = f.ld;
    for (int m = 1; m <= Hy; ++m)
        for (int64_t ii = 0; ii < nxs; ++ii) {
            int i = (int)ii - Hx + 1;
            if (bylo == ORA_BC_PERIODIC) AT(g, f, i, 1 - m) = AT(g, f, i, Ny + 1 - m);
            else if (bylo == ORA_BC_MIRROR) AT(g, f, i, 1 - m) = AT(g, f, i, m);
            if (byhi == ORA_BC_PERIODIC) AT(g, f, i, Ny + m) = AT(g, f, i, m);
            else if (byhi == ORA_BC_MIRROR) AT(g, f, i, Ny + m) = AT(g, f, i, Ny + 1 - m);
            if (m == 1 && bylo == ORA_BC_VALUE) AT(g, f, i, 0) = 2 * vy[0] - AT(g, f, i, 1);
            if (m == 1 && byhi == ORA_BC_VALUE) AT(g, f, i, Ny + 1) = 2 * vy[1] - AT(g, f, i, Ny);
        }
}
void ora_fill_halo(const ora_problem* g, ora_field f, int lx, int ly, int bcx, int bcy) {
    (void)lx; (void)ly;
    ora_fill_halo4(g, f, bcx, bcx, bcy, bcy);
}
static int bc_side(int topo, int loc, int high) {
    if (topo == ORA_PERIODIC) return ORA_BC_PERIODIC;
    if (high && fold_hi(topo)) return ORA_BC_FOLD;
    int wall = high ? wall_hi(topo) : wall_lo(topo);
    if (!wall) return ORA_BC_NONE;              /* connected: left to the exchange */
    return loc == C_ ? ORA_BC_MIRROR : ORA_BC_NONE;
}
static void fill_loc_sign(const ora_problem* g, ora_field f, int lx, int ly, int sign) {
    int byhi = bc_side(g->topo_y, ly, 1);
    ora_fill_halo4(g, f, bc_side(g->topo_x, lx, 0), bc_side(g->topo_x, lx, 1), bc_side(g->topo_y, ly, 0), byhi == ORA_BC_FOLD ? ORA_BC_NONE : byhi);
    if (byhi == ORA_BC_FOLD) fold_north(g, f, lx, ly, sign);
}
static void fill_loc(const ora_problem* g, ora_field f, int lx, int ly) { fill_loc_sign(g, f, lx, ly, 1); }
void ora_fill_halo_loc(const ora_problem* g, ora_field f, int lx, int ly, int fold_sign) { fill_loc_sign(g, f, lx, ly, fold_sign); }
/* u, v: a ValueBoundaryCondition replaces the no-flux mirror of the tangential component on a wall */
void ora_fill_halo_u(ora_problem* g) {
    int bylo = bc_side(g->topo_y, C_, 0), byhi = bc_side(g->topo_y, C_, 1);
    if (bylo == ORA_BC_MIRROR && g->u_value_on[0]) bylo = ORA_BC_VALUE;
    if (byhi == ORA_BC_MIRROR && g->u_value_on[1]) byhi = ORA_BC_VALUE;
    static const double zero[2] = {0.0, 0.0};
    fill_halo4v(g, g->u, bc_side(g->topo_x, F_, 0), bc_side(g->topo_x, F_, 1), bylo, byhi == ORA_BC_FOLD ? ORA_BC_NONE : byhi, zero, g->u_value);
    if (byhi == ORA_BC_FOLD) fold_north(g, g->u, F_, C_, -1);      /* sea_ice_model.jl:57-64: the velocities flip sign across the fold */
}
void ora_fill_halo_v(ora_problem* g) {
    int bxlo = bc_side(g->topo_x, C_, 0), bxhi = bc_side(g->topo_x, C_, 1);
    if (bxlo == ORA_BC_MIRROR && g->v_value_on[0]) bxlo = ORA_BC_VALUE;
    if (bxhi == ORA_BC_MIRROR && g->v_value_on[1]) bxhi = ORA_BC_VALUE;
    static const double zero[2] = {0.0, 0.0};
    int byhi = bc_side(g->topo_y, F_, 1);
    fill_halo4v(g, g->v, bxlo, bxhi, bc_side(g->topo_y, F_, 0), byhi == ORA_BC_FOLD ? ORA_BC_NONE : byhi, g->v_value, zero);
    if (byhi == ORA_BC_FOLD) fold_north(g, g->v, C_, F_, -1);
}
void ora_fill_halo_center(ora_problem* g, ora_field f) { fill_loc(g, f, C_, C_); }

/* finalize_rheology!, evp:275-280: halo fill of sigma11, sigma12, sigma22 */
void ora_finalize_rheology(ora_problem* g) {
    ora_fill_halo_center(g, g->s11);
    fill_loc(g, g->s12, F_, F_);
    ora_fill_halo_center(g, g->s22);
}

/* ------------------------------------------------------------------------ */
/* the sub-step loop, split_explicit_momentum_equations.jl:173-189            */
/* stress range: Auxiliaries kernel parameters, evp:145                       */
/* ------------------------------------------------------------------------ */
void ora_subcycle(ora_problem* g, double dt, int first, int last) {
    int si0 = -g->Hx + 2, si1 = g->Nx + g->Hx - 1, sj0 = -g->Hy + 2, sj1 = g->Ny + g->Hy - 1;
    for (int s = first; s <= last; ++s) {
        ora_compute_viscosities(g, si0, si1, sj0, sj1);            /* compute_stresses!, evp:222-234 */
        ora_compute_stresses(g, dt, si0, si1, sj0, sj1);
        if ((s % 2) == 0) {                                        /* :178-182 */
            ora_u_velocity_step(g, dt, 1, g->Nx, 1, g->Ny);
            ora_fill_halo_u(g);
            ora_v_velocity_step(g, dt, 1, g->Nx, 1, g->Ny);
            ora_fill_halo_v(g);
        } else {                                                   /* :184-187 */
            ora_v_velocity_step(g, dt, 1, g->Nx, 1, g->Ny);
            ora_fill_halo_v(g);
            ora_u_velocity_step(g, dt, 1, g->Nx, 1, g->Ny);
            ora_fill_halo_u(g);
        }
    }
}

static void copy_parent(const ora_problem* g, ora_field dst, ora_field src) {
    int64_t rows = (int64_t)g->Ny + 2 * g->Hy;
    /* parents of same-location fields have identical shape; copy the common ld x rows block */
    int64_t ld = dst.ld < src.ld ? dst.ld : src.ld;
    for (int64_t r = 0; r < rows; ++r) memcpy(dst.p + r * dst.ld, src.p + r * src.ld, (size_t)ld * sizeof(double));
}

/* time_step_momentum!, split_explicit_momentum_equations.jl:103-195 */
void ora_time_step_momentum(ora_problem* g, double dt, int rk_reset) {
    if (rk_reset) {                          /* reset_velocities! :89-93 */
        copy_parent(g, g->u, g->um);
        copy_parent(g, g->v, g->vm);
    }
    ora_initialize_rheology(g);              /* :130 */
    /* update_external_stress! :133-134 : halo refresh of forcing fields is the caller's job */
    ora_fill_halo_u(g);                      /* :170 */
    ora_fill_halo_v(g);                      /* :171 */
    ora_subcycle(g, dt, 1, g->substeps);     /* :173-189 */
    ora_finalize_rheology(g);                /* :192 */
}

/* ------------------------------------------------------------------------ */
/* Advection: upstream Oceananigans WENO(order = 5 | 7), UpwindBiased(5)      */
/* (SURVEY.md App. B -- recalled semantics, unverifiable here: classical      */
/* Jiang-Shu / Balsara-Shu stencils and smoothness indicators, WENO-Z weights */
/* alpha_s = C_s (1 + (tau / (beta_s + eps))^2), eps = 1e-8).                 */
/* Face i is the west face of cell i; left bias (U > 0) is upwinded to i-1.   */
/* ------------------------------------------------------------------------ */
#define WENO_EPS 1e-8

/* (the *_parts functions expose the candidate values q, the smoothness indicators b, the unnormalised weights a and tau
 * to tests/test_weno_published.py through ora_test_weno; weno5 / weno7 / weno3 combine them in the order they always did) */
static void weno5_parts(const double* p, double* q, double* b, double* a, double* tau_out) {
    /* p[0..4] = psi at (upwind-2, upwind-1, upwind, downwind, downwind+1) */
    q[0] = (2 * p[2] + 5 * p[3] - p[4]) / 6;      /* stencil 0: (upwind, dw, dw+1) */
    q[1] = (-p[1] + 5 * p[2] + 2 * p[3]) / 6;     /* stencil 1 */
    q[2] = (2 * p[0] - 7 * p[1] + 11 * p[2]) / 6; /* stencil 2 */
    b[0] = (p[2] * (10 * p[2] - 31 * p[3] + 11 * p[4]) + p[3] * (25 * p[3] - 19 * p[4]) + p[4] * (4 * p[4])) / 3;
    b[1] = (p[1] * (4 * p[1] - 13 * p[2] + 5 * p[3]) + p[2] * (13 * p[2] - 13 * p[3]) + p[3] * (4 * p[3])) / 3;
    b[2] = (p[0] * (4 * p[0] - 19 * p[1] + 11 * p[2]) + p[1] * (25 * p[1] - 31 * p[2]) + p[2] * (10 * p[2])) / 3;
    double tau = fabs(b[0] - b[2]);
    double r0 = tau / (b[0] + WENO_EPS), r1 = tau / (b[1] + WENO_EPS), r2 = tau / (b[2] + WENO_EPS);
    a[0] = (3.0 / 10) * (1 + r0 * r0);
    a[1] = (3.0 / 5) * (1 + r1 * r1);
    a[2] = (1.0 / 10) * (1 + r2 * r2);
    *tau_out = tau;
}
static double weno5(const double* p) {
    double q[3], b[3], a[3], tau;
    weno5_parts(p, q, b, a, &tau);
    double s = a[0] + a[1] + a[2];
    return (a[0] * q[0] + a[1] * q[1] + a[2] * q[2]) / s;
}
static double upwind5(const double* p) {
    /* UpwindBiased(order=5): optimal linear combination of the three stencils */
    return (2 * p[0] - 13 * p[1] + 47 * p[2] + 27 * p[3] - 3 * p[4]) / 60;
}
static void weno7_parts(const double* p, double* q, double* b, double* a, double* tau_out) {
    /* p[0..6] = psi at (up-3, up-2, up-1, up, dw, dw+1, dw+2) */
    q[0] = (3 * p[3] + 13 * p[4] - 5 * p[5] + p[6]) / 12;
    q[1] = (-p[2] + 7 * p[3] + 7 * p[4] - p[5]) / 12;
    q[2] = (p[1] - 5 * p[2] + 13 * p[3] + 3 * p[4]) / 12;
    q[3] = (-3 * p[0] + 13 * p[1] - 23 * p[2] + 25 * p[3]) / 12;
    b[0] = p[3] * (2.107 * p[3] - 9.402 * p[4] + 7.042 * p[5] - 1.854 * p[6]) +
           p[4] * (11.003 * p[4] - 17.246 * p[5] + 4.642 * p[6]) + p[5] * (7.043 * p[5] - 3.882 * p[6]) + p[6] * (0.547 * p[6]);
    b[1] = p[2] * (0.547 * p[2] - 2.522 * p[3] + 1.922 * p[4] - 0.494 * p[5]) +
           p[3] * (3.443 * p[3] - 5.966 * p[4] + 1.602 * p[5]) + p[4] * (2.843 * p[4] - 1.642 * p[5]) + p[5] * (0.267 * p[5]);
    b[2] = p[1] * (0.267 * p[1] - 1.642 * p[2] + 1.602 * p[3] - 0.494 * p[4]) +
           p[2] * (2.843 * p[2] - 5.966 * p[3] + 1.922 * p[4]) + p[3] * (3.443 * p[3] - 2.522 * p[4]) + p[4] * (0.547 * p[4]);
    b[3] = p[0] * (0.547 * p[0] - 3.882 * p[1] + 4.642 * p[2] - 1.854 * p[3]) +
           p[1] * (7.043 * p[1] - 17.246 * p[2] + 7.042 * p[3]) + p[2] * (11.003 * p[2] - 9.402 * p[3]) + p[3] * (2.107 * p[3]);
    double tau = fabs(b[0] + 3 * b[1] - 3 * b[2] - b[3]);
    double r0 = tau / (b[0] + WENO_EPS), r1 = tau / (b[1] + WENO_EPS), r2 = tau / (b[2] + WENO_EPS), r3 = tau / (b[3] + WENO_EPS);
    a[0] = (4.0 / 35) * (1 + r0 * r0);
    a[1] = (18.0 / 35) * (1 + r1 * r1);
    a[2] = (12.0 / 35) * (1 + r2 * r2);
    a[3] = (1.0 / 35) * (1 + r3 * r3);
    *tau_out = tau;
}
static double weno7(const double* p) {
    double q[4], b[4], a[4], tau;
    weno7_parts(p, q, b, a, &tau);
    double s = a[0] + a[1] + a[2] + a[3];
    return (a[0] * q[0] + a[1] * q[1] + a[2] * q[2] + a[3] * q[3]) / s;
}
static void weno3_parts(const double* p, double* q, double* b, double* a, double* tau_out) {
    /* p[0..2] = psi at (upwind-1, upwind, downwind): WENO(order = 3), the buffer scheme of order 5 */
    q[0] = (p[1] + p[2]) / 2;                     /* stencil 0: (upwind, dw) */
    q[1] = (-p[0] + 3 * p[1]) / 2;                /* stencil 1: (upwind-1, upwind) */
    b[0] = p[1] * (p[1] - 2 * p[2]) + p[2] * p[2];
    b[1] = p[0] * (p[0] - 2 * p[1]) + p[1] * p[1];
    double tau = fabs(b[0] - b[1]);
    double r0 = tau / (b[0] + WENO_EPS), r1 = tau / (b[1] + WENO_EPS);
    a[0] = (2.0 / 3) * (1 + r0 * r0);
    a[1] = (1.0 / 3) * (1 + r1 * r1);
    *tau_out = tau;
}
static double weno3(const double* p) {
    double q[2], b[2], a[2], tau;
    weno3_parts(p, q, b, a, &tau);
    double s = a[0] + a[1];
    return (a[0] * q[0] + a[1] * q[1]) / s;
}
static double upwind3(const double* p) {
    /* UpwindBiased(order = 3), the buffer scheme of UpwindBiased(order = 5) */
    return (-p[0] + 5 * p[1] + 2 * p[2]) / 6;
}
/* ---- WENO weights in single precision (weight_dtype f32; SURVEY.md App. B: newer upstream versions carry a second float type
 * parameter FT2, default Float32, for the smoothness / weight arithmetic of a WENO scheme -- recalled, unverified; the reference
 * constructs its schemes with the upstream default, /root/reference/src/sea_ice_advection.jl:51-58 only calls them).  What this
 * mode ASSUMES, stated so that a reference run can confirm or refute it: the stencil values are converted to float, the smoothness
 * indicators, tau, the ratios and the unnormalised weights alpha_s = C_s (1 + (tau / (beta_s + eps))^2) are evaluated in float with
 * float constants (the same expressions, the same order, eps = 1e-8f), their sum too; the candidate values stay double, and the
 * result is (sum alpha_s q_s) / (sum alpha_s) in double with the alphas widened.  Default is f64 (weight_f32 = 0). */
#define WENO_EPS_F 1e-8f
static void weno5_parts_f32(const double* p, double* q, float* b, float* a, float* tau_out) {
    float f[5];
    for (int k = 0; k < 5; ++k) f[k] = (float)p[k];
    q[0] = (2 * p[2] + 5 * p[3] - p[4]) / 6;
    q[1] = (-p[1] + 5 * p[2] + 2 * p[3]) / 6;
    q[2] = (2 * p[0] - 7 * p[1] + 11 * p[2]) / 6;
    b[0] = (f[2] * (10 * f[2] - 31 * f[3] + 11 * f[4]) + f[3] * (25 * f[3] - 19 * f[4]) + f[4] * (4 * f[4])) / 3;
    b[1] = (f[1] * (4 * f[1] - 13 * f[2] + 5 * f[3]) + f[2] * (13 * f[2] - 13 * f[3]) + f[3] * (4 * f[3])) / 3;
    b[2] = (f[0] * (4 * f[0] - 19 * f[1] + 11 * f[2]) + f[1] * (25 * f[1] - 31 * f[2]) + f[2] * (10 * f[2])) / 3;
    float tau = fabsf(b[0] - b[2]);
    float r0 = tau / (b[0] + WENO_EPS_F), r1 = tau / (b[1] + WENO_EPS_F), r2 = tau / (b[2] + WENO_EPS_F);
    a[0] = (float)(3.0 / 10) * (1 + r0 * r0);
    a[1] = (float)(3.0 / 5) * (1 + r1 * r1);
    a[2] = (float)(1.0 / 10) * (1 + r2 * r2);
    *tau_out = tau;
}
static double weno5_f32(const double* p) {
    double q[3]; float b[3], a[3], tau;
    weno5_parts_f32(p, q, b, a, &tau);
    float s = a[0] + a[1] + a[2];
    return ((double)a[0] * q[0] + (double)a[1] * q[1] + (double)a[2] * q[2]) / (double)s;
}
static void weno7_parts_f32(const double* p, double* q, float* b, float* a, float* tau_out) {
    float f[7];
    for (int k = 0; k < 7; ++k) f[k] = (float)p[k];
    q[0] = (3 * p[3] + 13 * p[4] - 5 * p[5] + p[6]) / 12;
    q[1] = (-p[2] + 7 * p[3] + 7 * p[4] - p[5]) / 12;
    q[2] = (p[1] - 5 * p[2] + 13 * p[3] + 3 * p[4]) / 12;
    q[3] = (-3 * p[0] + 13 * p[1] - 23 * p[2] + 25 * p[3]) / 12;
    b[0] = f[3] * (2.107f * f[3] - 9.402f * f[4] + 7.042f * f[5] - 1.854f * f[6]) +
           f[4] * (11.003f * f[4] - 17.246f * f[5] + 4.642f * f[6]) + f[5] * (7.043f * f[5] - 3.882f * f[6]) + f[6] * (0.547f * f[6]);
    b[1] = f[2] * (0.547f * f[2] - 2.522f * f[3] + 1.922f * f[4] - 0.494f * f[5]) +
           f[3] * (3.443f * f[3] - 5.966f * f[4] + 1.602f * f[5]) + f[4] * (2.843f * f[4] - 1.642f * f[5]) + f[5] * (0.267f * f[5]);
    b[2] = f[1] * (0.267f * f[1] - 1.642f * f[2] + 1.602f * f[3] - 0.494f * f[4]) +
           f[2] * (2.843f * f[2] - 5.966f * f[3] + 1.922f * f[4]) + f[3] * (3.443f * f[3] - 2.522f * f[4]) + f[4] * (0.547f * f[4]);
    b[3] = f[0] * (0.547f * f[0] - 3.882f * f[1] + 4.642f * f[2] - 1.854f * f[3]) +
           f[1] * (7.043f * f[1] - 17.246f * f[2] + 7.042f * f[3]) + f[2] * (11.003f * f[2] - 9.402f * f[3]) + f[3] * (2.107f * f[3]);
    float tau = fabsf(b[0] + 3 * b[1] - 3 * b[2] - b[3]);
    float r0 = tau / (b[0] + WENO_EPS_F), r1 = tau / (b[1] + WENO_EPS_F), r2 = tau / (b[2] + WENO_EPS_F), r3 = tau / (b[3] + WENO_EPS_F);
    a[0] = (float)(4.0 / 35) * (1 + r0 * r0);
    a[1] = (float)(18.0 / 35) * (1 + r1 * r1);
    a[2] = (float)(12.0 / 35) * (1 + r2 * r2);
    a[3] = (float)(1.0 / 35) * (1 + r3 * r3);
    *tau_out = tau;
}
static double weno7_f32(const double* p) {
    double q[4]; float b[4], a[4], tau;
    weno7_parts_f32(p, q, b, a, &tau);
    float s = a[0] + a[1] + a[2] + a[3];
    return ((double)a[0] * q[0] + (double)a[1] * q[1] + (double)a[2] * q[2] + (double)a[3] * q[3]) / (double)s;
}
static void weno3_parts_f32(const double* p, double* q, float* b, float* a, float* tau_out) {
    float f[3];
    for (int k = 0; k < 3; ++k) f[k] = (float)p[k];
    q[0] = (p[1] + p[2]) / 2;
    q[1] = (-p[0] + 3 * p[1]) / 2;
    b[0] = f[1] * (f[1] - 2 * f[2]) + f[2] * f[2];
    b[1] = f[0] * (f[0] - 2 * f[1]) + f[1] * f[1];
    float tau = fabsf(b[0] - b[1]);
    float r0 = tau / (b[0] + WENO_EPS_F), r1 = tau / (b[1] + WENO_EPS_F);
    a[0] = (float)(2.0 / 3) * (1 + r0 * r0);
    a[1] = (float)(1.0 / 3) * (1 + r1 * r1);
    *tau_out = tau;
}
static double weno3_f32(const double* p) {
    double q[2]; float b[2], a[2], tau;
    weno3_parts_f32(p, q, b, a, &tau);
    float s = a[0] + a[1];
    return ((double)a[0] * q[0] + (double)a[1] * q[1]) / (double)s;
}
/* Boundary-order reduction next to walls (upstream topologically_conditional_interpolation, recalled -- SURVEY.md
 * App. B): a scheme with buffer B (order 2B-1) is used at face `idx` (1-based along the line, N cells) only if its
 * biased stencil stays inside the domain, otherwise its buffer scheme (order 2B-3) is tried, down to first-order
 * upwind: left bias needs idx >= B+1 (low wall) and idx <= N+2-B (high wall), right bias idx >= B and idx <= N+1-B. */
static int reduced_buffer(int B, int idx, int N, int left, int wall_lo, int wall_hi) {
    while (B > 1) {
        int lo_ok = !wall_lo || idx >= (left ? B + 1 : B);
        int hi_ok = !wall_hi || idx <= (left ? N + 2 - B : N + 1 - B);
        if (lo_ok && hi_ok) break;
        --B;
    }
    return B;
}
/* Immersed boundaries (ImmersedBoundaryGrid; upstream immersed reconstruction, recalled -- SURVEY.md App. B, used by
 * the reference through _advective_tracer_flux_x/y, sea_ice_advection.jl:1-5,51-54): a biased scheme with buffer B is
 * used at a face only if none of the 2B cells i-B .. i+B-1 around it (the union of its left and right stencils) is
 * inactive -- immersed OR beyond a wall --, otherwise its buffer scheme is tried, down to first-order upwind, which
 * is never reduced.  This replaces the topological rule above on such a grid.  dir: 0 along x, 1 along y. */
static int reduced_buffer_immersed(const ora_problem* g, int B, int i, int j, int dir) {
    while (B > 1) {
        int near = 0;
        for (int k = -B; k <= B - 1; ++k) near |= dir == 0 ? inactive_cell(g, i + k, j) : inactive_cell(g, i, j + k);
        if (!near) break;
        --B;
    }
    return B;
}
/* reconstruct c at a face from a 1-D line of values with the scheme of buffer B; `up` is the upwind cell value index
 * stepping `st` (= +1 for left bias reading towards increasing index). */
static double reconstruct(int scheme, const double* line, int64_t up, int64_t st, int B, int w32) {
    double p[7];
    if (scheme == 1) return line[up];
    const int weno = scheme > 0;
    if (B == 1) return line[up];
    if (B == 2) {
        for (int k = 0; k < 3; ++k) p[k] = line[up + (k - 1) * st];
        return weno ? (w32 ? weno3_f32(p) : weno3(p)) : upwind3(p);
    }
    if (B == 3) {
        for (int k = 0; k < 5; ++k) p[k] = line[up + (k - 2) * st];
        return weno ? (w32 ? weno5_f32(p) : weno5(p)) : upwind5(p);
    }
    for (int k = 0; k < 7; ++k) p[k] = line[up + (k - 3) * st];
    return w32 ? weno7_f32(p) : weno7(p);
}
/* Test hook (tests/test_weno_published.py): the parts of one reconstruction.  order 3 / 5 / 7: WENO; -3 / -5: UpwindBiased
 * (value only).  p: the 2B - 1 stencil values, upwind-most first (B = (|order| + 1) / 2); out: value, then for WENO
 * q[0..B-1], beta[0..B-1], alpha[0..B-1] (unnormalised weights), tau.  Returns the number of doubles written. */
int ora_test_weno(int order, const double* p, double* out) {
    double q[4], b[4], a[4], tau = 0.0;
    int B = 0;
    if (order == -3) { out[0] = upwind3(p); return 1; }
    if (order == -5) { out[0] = upwind5(p); return 1; }
    if (order == 3) { B = 2; out[0] = weno3(p); weno3_parts(p, q, b, a, &tau); }
    else if (order == 5) { B = 3; out[0] = weno5(p); weno5_parts(p, q, b, a, &tau); }
    else if (order == 7) { B = 4; out[0] = weno7(p); weno7_parts(p, q, b, a, &tau); }
    else if (order == 103 || order == 105 || order == 107) {      /* 100 + order: the f32-weight variant, its parts widened to double */
        float bf[4], af[4], tf = 0.f;
        if (order == 103) { B = 2; out[0] = weno3_f32(p); weno3_parts_f32(p, q, bf, af, &tf); }
        else if (order == 105) { B = 3; out[0] = weno5_f32(p); weno5_parts_f32(p, q, bf, af, &tf); }
        else { B = 4; out[0] = weno7_f32(p); weno7_parts_f32(p, q, bf, af, &tf); }
        for (int k = 0; k < B; ++k) { b[k] = bf[k]; a[k] = af[k]; }
        tau = tf;
    }
    else return 0;
    for (int k = 0; k < B; ++k) { out[1 + k] = q[k]; out[1 + B + k] = b[k]; out[1 + 2 * B + k] = a[k]; }
    out[1 + 3 * B] = tau;
    return 2 + 3 * B;
}
static int buffer_at(const ora_problem* g, int scheme, int i, int j, int dir, int left) {
    const int B0 = scheme == 7 ? 4 : (scheme == 1 ? 1 : ((scheme == 3 || scheme == -3) ? 2 : 3));   /* order 2B - 1 */
    if (g->has_mask) return reduced_buffer_immersed(g, B0, i, j, dir);
    const int topo = dir == 0 ? g->topo_x : g->topo_y;
    return reduced_buffer(B0, dir == 0 ? i : j, dir == 0 ? g->Nx : g->Ny, left, wall_lo(topo), wall_hi(topo));
}
/* _advective_tracer_flux_x = Ax^{fcc} * U * c~ (upstream; bias = left iff U > 0); on an immersed grid
 * conditional_flux_fcc: zero at peripheral faces */
double ora_weno_flux_x(const ora_problem* g, int scheme, ora_field c, int i, int j) {
    double uu = AT(g, g->u, i, j);
    const double* base = &AT(g, c, i, j);   /* cell i; upwind of a left-biased face i is cell i-1 */
    if (g->has_mask && ora_peripheral_u(g, i, j)) return 0.0;
    double ct = (uu > 0) ? reconstruct(scheme, base, -1, 1, buffer_at(g, scheme, i, j, 0, 1), g->weno_weights_f32)
                         : reconstruct(scheme, base, 0, -1, buffer_at(g, scheme, i, j, 0, 0), g->weno_weights_f32);
    return ora_dy(g, F_, C_, i, j) * uu * ct;    /* Ax = dy * dz, dz = 1 */
}
double ora_weno_flux_y(const ora_problem* g, int scheme, ora_field c, int i, int j) {
    double vv = AT(g, g->v, i, j);
    const double* base = &AT(g, c, i, j);
    int64_t ld = c.ld;
    if (g->has_mask && ora_peripheral_v(g, i, j)) return 0.0;
    double ct = (vv > 0) ? reconstruct(scheme, base, -ld, ld, buffer_at(g, scheme, i, j, 1, 1), g->weno_weights_f32)
                         : reconstruct(scheme, base, 0, -ld, buffer_at(g, scheme, i, j, 1, 0), g->weno_weights_f32);
    return ora_dx(g, C_, F_, i, j) * vv * ct;    /* Ay = dx^{cf} * dz */
}
/* horizontal_div_Uc, sea_ice_advection.jl:51-54 ; G = -div, tracer_tendency_kernel_functions.jl:39-42 */
static double tendency_of(const ora_problem* g, int scheme, ora_field c, int i, int j) {
    double V = ora_az(g, C_, C_, i, j);          /* V^{ccc} = Az * dz */
    double fx = ora_weno_flux_x(g, scheme, c, i + 1, j) - ora_weno_flux_x(g, scheme, c, i, j);
    double fy = ora_weno_flux_y(g, scheme, c, i, j + 1) - ora_weno_flux_y(g, scheme, c, i, j);
    return -(1 / V * (fx + fy));
}
static void zero_field(const ora_problem* g, ora_field f) {
    if (!f.p) return;
    for (int j = 1; j <= g->Ny; ++j) for (int i = 1; i <= g->Nx; ++i) AT(g, f, i, j) = 0.0;
}
/* advection = nothing: horizontal_div_Uc(..., ::Nothing, ...) = zero(grid) (sea_ice_advection.jl:50), so the tendencies
 * are zero -- and the tracer update still runs (dynamic_time_step! launches unconditionally): it resets h, aice [, hs]
 * to Psi^- at every RK stage and applies the clipping / ridging rules */
static void tendencies_or_zero(ora_problem* g, int scheme) {
    if (scheme) { ora_compute_tracer_tendencies(g, scheme); return; }
    zero_field(g, g->Gh); zero_field(g, g->Ga);
    if (g->has_snow) zero_field(g, g->Ghs);
}
void ora_compute_tracer_tendencies(ora_problem* g, int scheme) {
    OMP_ROWS
    for (int j = 1; j <= g->Ny; ++j)
        for (int i = 1; i <= g->Nx; ++i) {
            if (g->has_snow) AT(g, g->Ghs, i, j) = tendency_of(g, scheme, g->hs, i, j);   /* compute_snow_advection_tendency! :49-52 */
            AT(g, g->Gh, i, j) = tendency_of(g, scheme, g->h, i, j);
            AT(g, g->Ga, i, j) = tendency_of(g, scheme, g->aice, i, j);
        }
}

/* _dynamic_step_tracers!, sea_ice_fe_step.jl:56-82 */
void ora_dynamic_step_tracers(ora_problem* g, double dt, int from_cache) {
    OMP_ROWS
    for (int j = 1; j <= g->Ny; ++j)
        for (int i = 1; i <= g->Nx; ++i) {
            double hn = from_cache ? AT(g, g->hm, i, j) : AT(g, g->h, i, j);
            double an = from_cache ? AT(g, g->am, i, j) : AT(g, g->aice, i, j);
            double hp = hn + dt * AT(g, g->Gh, i, j);
            double ap = an + dt * AT(g, g->Ga, i, j);
            ap = jmax(0.0, ap);
            hp = jmax(0.0, hp);
            ap = (hp == 0) ? 0.0 : ap;
            hp = (ap == 0) ? 0.0 : hp;
            double Vp = hp * ap;
            AT(g, g->aice, i, j) = (ap > 1) ? 1.0 : ap;
            AT(g, g->h, i, j) = (ap > 1) ? Vp : hp;
            if (g->has_snow) {                                       /* dynamic_step_snow!, sea_ice_fe_step.jl:86-94 */
                double sn = from_cache ? AT(g, g->hsm, i, j) : AT(g, g->hs, i, j);
                double sp = sn + dt * AT(g, g->Ghs, i, j);
                sp = jmax(0.0, sp);
                sp = (AT(g, g->aice, i, j) <= 0) ? 0.0 : sp;
                AT(g, g->hs, i, j) = sp;
            }
        }
}

/* update_state!, sea_ice_model.jl:379-394 : mask + halo fill of every prognostic field */
void ora_update_state(ora_problem* g) {
    if (g->has_mask) {
        for (int j = 1; j <= g->Ny; ++j)
            for (int i = 1; i <= g->Nx; ++i) {
                if (inactive_cell(g, i, j)) { AT(g, g->h, i, j) = 0; AT(g, g->aice, i, j) = 0; if (g->has_snow) AT(g, g->hs, i, j) = 0; }
                if (ora_peripheral_u(g, i, j)) AT(g, g->u, i, j) = 0;   /* mask_immersed_field_xy! on (f,c): peripheral nodes */
                if (ora_peripheral_v(g, i, j)) AT(g, g->v, i, j) = 0;
            }
    }
    ora_fill_halo_center(g, g->h);
    ora_fill_halo_center(g, g->aice);
    if (g->has_snow) ora_fill_halo_center(g, g->hs);
    ora_fill_halo_u(g);
    ora_fill_halo_v(g);
}

/* time_step!(::FESeaIceModel), sea_ice_fe_step.jl:13-34 (no thermodynamics) */
void ora_time_step_fe(ora_problem* g, double dt, int scheme, int first_iteration) {
    if (first_iteration) ora_update_state(g);          /* :16 */
    tendencies_or_zero(g, scheme);  /* :19 */
    ora_time_step_momentum(g, dt, 0);                  /* :22 */
    ora_dynamic_step_tracers(g, dt, 0);    /* :25 */
    ora_update_state(g);                               /* :31 */
}
/* SplitRungeKutta3: cache_current_fields! (sea_ice_rk_substep.jl:29-42), then for beta in (3,2,1):
 * rk_substep!(dt/beta) (:81-94) ; update_state!  (upstream stage loop) */
void ora_time_step_rk3(ora_problem* g, double dt, int scheme) {
    copy_parent(g, g->hm, g->h);
    copy_parent(g, g->am, g->aice);
    copy_parent(g, g->um, g->u);
    copy_parent(g, g->vm, g->v);
    for (int beta = 3; beta >= 1; --beta) {
        double dtau = dt / beta;
        tendencies_or_zero(g, scheme);   /* :84 */
        ora_time_step_momentum(g, dtau, 1);                     /* :87 */
        ora_dynamic_step_tracers(g, dtau, 1);       /* :89 */
        ora_update_state(g);
    }
}

/* ------------------------------------------------------------------------ */
/* Bare-ice slab thermodynamics (plumbing)                                    */
/* thermodynamic_time_step.jl:75-118,304-324,358-370;                         */
/* slab_thermodynamics_tendencies.jl:28-135; SeaIceThermodynamics.jl:161-170  */
/* ------------------------------------------------------------------------ */
static double latent_heat(const ora_slab* s, double T) {          /* SeaIceThermodynamics.jl:161-170 */
    return s->L0 + (s->rho_liquid * s->c_liquid / s->rho_pure - s->c_ice) * (T - s->T0);
}
/* slab_internal_heat_flux, slab_heat_and_tracer_fluxes.jl:8-19 */
static double slab_internal_flux(const ora_slab* s, double Tu, double Tb, double h) {
    return (h <= 0) ? 0.0 : -s->k_ice * (Tu - Tb) / h;
}
/* concentration_thermodynamic_step(::ProportionalEvolution), thermodynamic_time_step.jl:358-370 */
static double concentration_step(double dtV, double an, double hn, double hc, double dt) {
    /* `x * flag` with a Julia Bool: false is a strong zero (NaN * false == 0, sign kept) */
    int freezing = (dtV >= 0), melting = (dtV < 0);
    double xf = (1 - an) / hc * dtV, xm = an / (2 * hn) * dtV;
    double daf = freezing ? xf : copysign(0.0, xf);
    double dam = melting ? xm : copysign(0.0, xm);
    double ap = an + dt * (daf + dam);
    return jmax(0.0, ap);
}
/* _ice_thermodynamic_time_step! :75-118 with thermodynamic_tendency (slab_thermodynamics_tendencies.jl:74-135,
 * PrescribedTemperature branch), ice_melt_freeze_tendency (:28-68), ice_volume_update (:304-324) */
static void slab_cell(const ora_slab* s, double* hp_, double* ap_, double* mf_, double dt) {
    double hn = *hp_, an = *ap_, hc = s->h_consolidation;
    int consolidated = hn >= hc;
    double Tb = s->liq_T0 - s->liq_slope * s->salinity;       /* IceWaterThermalEquilibrium: Tm(S) */
    double Tu = s->Tu;                                        /* PrescribedTemperature */
    if (s->top_bc_kind == 1) {
        /* MeltingConstrainedFluxBalance, slab_thermodynamics_tendencies.jl:107-119: consolidated ice solves
         * Qx - Qi(T) = 0 with Qi(T) = -k (T - Tb) / h (numeric Qx: closed-form root), capped at Tm(S_ice);
         * an unconsolidated slab takes the bottom temperature */
        double Tm = s->liq_T0 - s->liq_slope * s->ice_salinity;
        Tu = consolidated ? jmin(Tb - s->Qu * hn / s->k_ice, Tm) : Tb;
    }
    double Eb = s->rho_bulk * latent_heat(s, Tb);
    double Eu = s->rho_bulk * latent_heat(s, Tu);
    double Qi_fun = slab_internal_flux(s, Tu, Tb, hn);        /* internal_flux_function at Tu */
    double Qu = (s->top_flux_kind == 1) ? Qi_fun : s->Qu;     /* sea_ice_model.jl:248-256 default */
    double Qb = (s->bot_flux_kind == 1) ? (-(1 - an)) * s->Qb : s->Qb;
    double Qi = consolidated ? Qi_fun : 0.0;                  /* ice_interior_heat_flux :10-18 */
    double wu = (Qu - Qi) / Eu;
    double wb = (Qi - Qb) / Eb;
    double dtV = wu + wb;
    /* ice_volume_update :304-324 */
    double V1 = hn * an + dt * dtV;
    V1 = jmax(0.0, V1);
    dtV = (V1 - hn * an) / dt;
    double ap = concentration_step(dtV, an, hn, hc, dt);
    double hp = V1 / ap;
    hp = (ap <= 0) ? 0.0 : hp;
    ap = (dtV == 0) ? an : ap;
    hp = (dtV == 0) ? hn : hp;
    ap = (hp == 0) ? 0.0 : ap;
    hp = (ap == 0) ? 0.0 : hp;
    double a1 = (ap > 1) ? 1.0 : ap;
    double h1 = (ap > 1) ? hp * ap : hp;
    *ap_ = a1;
    *hp_ = h1;
    if (mf_) *mf_ = s->rho_bulk * (h1 * a1 - hn * an) / dt;   /* :111 */
}
/* ice_volume_update, thermodynamic_time_step.jl:304-324 */
static void ice_volume_update(double dtV, double hn, double an, double hc, double dt, double* h1, double* a1) {
    double V1 = hn * an + dt * dtV;
    V1 = jmax(0.0, V1);
    dtV = (V1 - hn * an) / dt;
    double ap = concentration_step(dtV, an, hn, hc, dt);
    double hp = V1 / ap;
    hp = (ap <= 0) ? 0.0 : hp;
    ap = (dtV == 0) ? an : ap;
    hp = (dtV == 0) ? hn : hp;
    ap = (hp == 0) ? 0.0 : ap;
    hp = (ap == 0) ? 0.0 : hp;
    *a1 = (ap > 1) ? 1.0 : ap;
    *h1 = (ap > 1) ? hp * ap : hp;
}
/* _layered_thermodynamic_time_step!, thermodynamic_time_step.jl:131-298 (snow on ice, resistors in series) */
static void layered_cell(const ora_slab* s, const ora_snow* w, double* hi_, double* a_, double* hs_,
                         double* mfi, double* mfs, double* mfp, double* tui, double* tus, double dt) {
    const double hin = *hi_, an = *a_, hc = s->h_consolidation;
    double hsn = *hs_;
    const double Vin = hin * an, Vsn = hsn * an;                                 /* :158-159 */
    const int consolidated = hin >= hc;
    const double Tb = s->liq_T0 - s->liq_slope * s->salinity;                    /* bottom_temperature :167 */
    double Tm = s->liq_T0 - s->liq_slope * s->ice_salinity;                      /* melting_temperature :168 */
    const double ks = w->k_snow, ki = s->k_ice;
    const double Qu = s->Qu;                                                     /* numeric top_external_heat_flux */
    Tm = (hsn > 0) ? 0.0 : Tm;                                                   /* :188 */
    const double R = hsn / ks + hin / ki;                                        /* ice_snow_conductive_flux :61-62 */
    double Tus = w->Tu;
    if (w->top_bc_kind == 1) {                                                   /* :190-199, closed-form root of Qu - (Tb - T) / R */
        double Tn = consolidated ? jmin(Tb - Qu * R, Tm) : Tb;
        Tus = Tn;
    }
    /* interface_temperature, slab_heat_and_tracer_fluxes.jl:69-84 */
    const double Ri = hin / ki, Rs = hsn / ks, Rt = Rs + Ri;
    const double Tsi = (Rt <= 0) ? Tb : Tb + (Tus - Tb) * Ri / Rt;
    const double Qic = (R <= 0) ? 0.0 : (Tb - Tus) / R;
    const double Qis = consolidated ? Qic : 0.0;                                 /* :211 */
    const double Qui = Qu;
    const double Qui_per_ice = (an > 0) ? Qui / an : 0.0;                        /* :214 */
    const double dQ = Qui_per_ice - Qis;
    const double melt_energy = jmax(0.0, -dQ);
    const double rs = w->rho_snow, Ls = s->L0;                                   /* reference_latent_heat */
    const double cap = rs * Ls * hsn / dt;
    const double Qs = jmin(melt_energy, cap);
    const double Gsm = Qs / (rs * Ls);
    const double ri = s->rho_bulk, riL = ri * Ls;
    const double Qbi = (s->bot_flux_kind == 1) ? (-(1 - an)) * s->Qb : s->Qb;
    const double alpha = (Qui - Qbi) / riL, beta = Qs / riL;                     /* :238-239 */
    const double Cm = (hin > 0) ? an / (2 * hin) : 0.0;
    const double Cf = (hc > 0) ? (1 - an) / hc : 0.0;
    const double Km = dt * Cm, Kf = dt * Cf;
    const double eps = 2.220446049250313e-16;
    const double Dm = 1 - Km * beta, Df = 1 - Kf * beta;
    const double am = (fabs(Dm) > eps) ? (an + Km * alpha) / Dm : an + Km * alpha;
    const double af = (fabs(Df) > eps) ? (an + Kf * alpha) / Df : an + Kf * alpha;
    const double dtVm = alpha + beta * am;
    const int melting = dtVm < 0;
    const double atmp = melting ? am : af;
    const double Qeff = Qui + Qs * atmp;                                         /* :259 */
    /* ice_melt_freeze_tendency at Tui = Tsi, slab_thermodynamics_tendencies.jl:28-68 */
    const double Eb = ri * latent_heat(s, Tb), Eu = ri * latent_heat(s, Tsi);
    const double Qii = consolidated ? slab_internal_flux(s, Tsi, Tb, hin) : 0.0;
    const double wu = (Qeff - Qii) / Eu, wb = (Qii - Qbi) / Eb;
    double hi1, a1;
    ice_volume_update(wu + wb, hin, an, hc, dt, &hi1, &a1);
    hsn = (a1 > 0) ? hsn * an / a1 : 0.0;                                        /* :274 */
    const double Gsp = (a1 > 0) ? w->snowfall / rs : 0.0;                        /* snow_accumulation :331-334 */
    double hs1 = hsn + dt * (Gsp - Gsm);
    hs1 = jmax(0.0, hs1);
    {   /* snow_ice_formation :336-353 */
        const double rw = s->rho_liquid;
        const double hf = hi1 * (1 - ri / rw) - hs1 * rs / rw;
        double dhs = (hf < 0) ? -hf * ri / rs : 0.0;
        const double hsp = jmax(0.0, hs1 - dhs);
        dhs = hs1 - hsp;
        hi1 = hi1 + dhs * rs / ri;
        hs1 = hsp;
    }
    hs1 = (a1 <= 0) ? 0.0 : hs1;
    *a_ = a1; *hi_ = hi1; *hs_ = hs1;
    const double Pabs = rs * Gsp * a1;
    if (mfi) *mfi = ri * (hi1 * a1 - Vin) / dt;
    if (mfs) *mfs = rs * (hs1 * a1 - Vsn) / dt - Pabs;
    if (mfp) *mfp = Pabs;
    if (tui) *tui = Tsi;
    if (tus) *tus = Tus;
}
void ora_layered_thermo_step(const ora_slab* s, const ora_snow* w, int64_t n, double* h, double* aice, double* hs,
                             double* mf_ice, double* mf_snow, double* mf_int, double* tu_ice, double* tu_snow, double dt) {
    for (int64_t c = 0; c < n; ++c)
        layered_cell(s, w, h + c, aice + c, hs + c, mf_ice ? mf_ice + c : 0, mf_snow ? mf_snow + c : 0, mf_int ? mf_int + c : 0,
                     tu_ice ? tu_ice + c : 0, tu_snow ? tu_snow + c : 0, dt);
}
void ora_layered_step_fields(ora_problem* g, const ora_slab* s, const ora_snow* w, double dt) {
    for (int j = 1; j <= g->Ny; ++j)
        for (int i = 1; i <= g->Nx; ++i)
            layered_cell(s, w, &AT(g, g->h, i, j), &AT(g, g->aice, i, j), &AT(g, g->hs, i, j), 0, 0, 0, 0, 0, dt);
}
void ora_slab_thermo_step(const ora_slab* s, int64_t n, double* h, double* aice, double* mass_flux, double dt) {
    for (int64_t c = 0; c < n; ++c) slab_cell(s, h + c, aice + c, mass_flux ? mass_flux + c : 0, dt);
}
/* thermodynamic_time_step!(model, ::SlabThermodynamics, ::Nothing, dt) over the interior, thermodynamic_time_step.jl:10-31 */
void ora_slab_step_fields(ora_problem* g, const ora_slab* s, double dt) {
    for (int j = 1; j <= g->Ny; ++j)
        for (int i = 1; i <= g->Nx; ++i) slab_cell(s, &AT(g, g->h, i, j), &AT(g, g->aice, i, j), 0, dt);
}
/* whole steps with the thermodynamic step in its place (sea_ice_fe_step.jl:28, sea_ice_rk_substep.jl:91) */
void ora_time_step_fe_thermo(ora_problem* g, double dt, int scheme, int first_iteration, const ora_slab* s) {
    if (first_iteration) ora_update_state(g);
    tendencies_or_zero(g, scheme);
    ora_time_step_momentum(g, dt, 0);
    ora_dynamic_step_tracers(g, dt, 0);
    if (s) ora_slab_step_fields(g, s, dt);
    ora_update_state(g);
}
/* the same with a snow layer: hs is a third advected tracer and joins the Psi^- cache (sea_ice_rk_substep.jl:29-42) */
void ora_time_step_fe_snow(ora_problem* g, double dt, int scheme, int first_iteration, const ora_slab* s, const ora_snow* w) {
    if (first_iteration) ora_update_state(g);
    tendencies_or_zero(g, scheme);
    ora_time_step_momentum(g, dt, 0);
    ora_dynamic_step_tracers(g, dt, 0);
    ora_layered_step_fields(g, s, w, dt);
    ora_update_state(g);
}
void ora_time_step_rk3_snow(ora_problem* g, double dt, int scheme, const ora_slab* s, const ora_snow* w) {
    copy_parent(g, g->hm, g->h);
    copy_parent(g, g->am, g->aice);
    copy_parent(g, g->hsm, g->hs);
    copy_parent(g, g->um, g->u);
    copy_parent(g, g->vm, g->v);
    for (int beta = 3; beta >= 1; --beta) {
        double dtau = dt / beta;
        tendencies_or_zero(g, scheme);
        ora_time_step_momentum(g, dtau, 1);
        ora_dynamic_step_tracers(g, dtau, 1);
        ora_layered_step_fields(g, s, w, dtau);
        ora_update_state(g);
    }
}
void ora_time_step_rk3_thermo(ora_problem* g, double dt, int scheme, const ora_slab* s) {
    copy_parent(g, g->hm, g->h);
    copy_parent(g, g->am, g->aice);
    copy_parent(g, g->um, g->u);
    copy_parent(g, g->vm, g->v);
    for (int beta = 3; beta >= 1; --beta) {
        double dtau = dt / beta;
        tendencies_or_zero(g, scheme);
        ora_time_step_momentum(g, dtau, 1);
        ora_dynamic_step_tracers(g, dtau, 1);
        if (s) ora_slab_step_fields(g, s, dtau);
        ora_update_state(g);
    }
}
