/* abi_client.c -- a plain-C client of libcsi_hip.so (test infrastructure).
 *
 * Everything else in this repository talks to the library through Python's ctypes, whose struct definitions are typed by hand.
 * What Julia's `ccall` binds (the GC.@preserve site of /root/reference/src/SeaIceDynamics/split_explicit_momentum_equations.jl:150
 * is what the stub mirrors) is what a C COMPILER makes of include/csi.h -- so this file includes that header and nothing else of
 * the project, is compiled by gcc as C99, links the shared library, and
 *
 *   abi_client layout            prints sizeof / offsetof of every struct and the enum values as JSON (no GPU needed:
 *                                tests/test_abi.py compares them with climaseaice.jl_amd/_lib.py and julia/ClimaSeaIceHIP.jl);
 *   abi_client run IN OUT        reads a problem (records written by tests/test_gpu_abi_client.py), allocates with hipMalloc,
 *                                runs csi_context_create -> csi_grid_set -> csi_field_bind x 14 -> csi_evp_params_set ->
 *                                csi_stress_set x 2 -> csi_set_mode -> csi_time_step_momentum -> csi_sync and writes u, v, sigma:
 *                                the GPU test compares them with the ctypes path bit for bit.
 *
 * Record format (little endian): char name[16]; int32 dtype (0 int32, 1 float64); int32 pad; int64 count; data.
 */
#include "csi.h"

#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define OFF(T, f) printf("      \"%s\": %zu,\n", #f, offsetof(T, f))

static int layout(void) {
    printf("{\n");
    printf("  \"csi_metrics\": {\n    \"size\": %zu,\n    \"fields\": {\n", sizeof(csi_metrics));
    OFF(csi_metrics, dx); OFF(csi_metrics, dy); OFF(csi_metrics, dxc); OFF(csi_metrics, dxf); OFF(csi_metrics, azc); OFF(csi_metrics, azf);
    OFF(csi_metrics, full);
    printf("      \"full_ld\": %zu\n    }\n  },\n", offsetof(csi_metrics, full_ld));
    printf("  \"csi_evp_params\": {\n    \"size\": %zu,\n    \"fields\": {\n", sizeof(csi_evp_params));
    OFF(csi_evp_params, ice_compressive_strength); OFF(csi_evp_params, ice_compaction_hardening); OFF(csi_evp_params, yield_curve_eccentricity);
    OFF(csi_evp_params, minimum_plastic_stress); OFF(csi_evp_params, min_relaxation_parameter); OFF(csi_evp_params, max_relaxation_parameter);
    OFF(csi_evp_params, relaxation_strength); OFF(csi_evp_params, pressure_formulation); OFF(csi_evp_params, has_coriolis);
    OFF(csi_evp_params, coriolis_f); OFF(csi_evp_params, minimum_concentration); OFF(csi_evp_params, minimum_mass);
    printf("      \"sea_ice_density\": %zu\n    }\n  },\n", offsetof(csi_evp_params, sea_ice_density));
    printf("  \"csi_stress\": {\n    \"size\": %zu,\n    \"fields\": {\n", sizeof(csi_stress));
    OFF(csi_stress, kind); OFF(csi_stress, ue_kind); OFF(csi_stress, ve_kind); OFF(csi_stress, reserved); OFF(csi_stress, tau_u);
    OFF(csi_stress, tau_v); OFF(csi_stress, ue); OFF(csi_stress, ve); OFF(csi_stress, rho_e);
    printf("      \"Cd\": %zu\n    }\n  },\n", offsetof(csi_stress, Cd));
    printf("  \"csi_slab_params\": {\n    \"size\": %zu,\n    \"fields\": {\n", sizeof(csi_slab_params));
    OFF(csi_slab_params, conductivity); OFF(csi_slab_params, sea_ice_density); OFF(csi_slab_params, density); OFF(csi_slab_params, liquid_density);
    OFF(csi_slab_params, liquid_heat_capacity); OFF(csi_slab_params, heat_capacity); OFF(csi_slab_params, reference_latent_heat);
    OFF(csi_slab_params, reference_temperature); OFF(csi_slab_params, liquidus_slope); OFF(csi_slab_params, freshwater_melting_temperature);
    OFF(csi_slab_params, bottom_salinity); OFF(csi_slab_params, ice_consolidation_thickness); OFF(csi_slab_params, top_temperature);
    OFF(csi_slab_params, top_flux_kind); OFF(csi_slab_params, bottom_flux_kind); OFF(csi_slab_params, top_heat_flux);
    OFF(csi_slab_params, bottom_heat_flux); OFF(csi_slab_params, top_bc_kind); OFF(csi_slab_params, pad_);
    printf("      \"ice_salinity\": %zu\n    }\n  },\n", offsetof(csi_slab_params, ice_salinity));
    printf("  \"csi_snow_params\": {\n    \"size\": %zu,\n    \"fields\": {\n", sizeof(csi_snow_params));
    OFF(csi_snow_params, conductivity); OFF(csi_snow_params, snow_density); OFF(csi_snow_params, snowfall); OFF(csi_snow_params, top_temperature);
    OFF(csi_snow_params, top_bc_kind);
    printf("      \"pad_\": %zu\n    }\n  },\n", offsetof(csi_snow_params, pad_));
    printf("  \"enums\": {\"CSI_F_COUNT\": %d, \"CSI_F_FORCING_V\": %d, \"CSI_F_ZETA_C\": %d, \"CSI_LEFT_CONNECTED_RIGHT_FOLDED\": %d, \"CSI_METRIC_FULL\": %d,\n"
           "            \"CSI_STRESS_SEMI_IMPLICIT\": %d, \"CSI_VEL_FIELD\": %d, \"CSI_MODE_FAST\": %d, \"CSI_ADVECT_WENO7\": %d, \"CSI_ADVECT_UPWIND5\": %d,\n"
           "            \"CSI_ERR_COMM\": %d, \"CSI_WEIGHTS_F32\": %d, \"sizeof_enum\": %zu, \"CSI_VERSION\": %d},\n",
           (int)CSI_F_COUNT, (int)CSI_F_FORCING_V, (int)CSI_F_ZETA_C, (int)CSI_LEFT_CONNECTED_RIGHT_FOLDED, (int)CSI_METRIC_FULL,
           (int)CSI_STRESS_SEMI_IMPLICIT, (int)CSI_VEL_FIELD, (int)CSI_MODE_FAST, (int)CSI_ADVECT_WENO7, (int)CSI_ADVECT_UPWIND5,
           (int)CSI_ERR_COMM, (int)CSI_WEIGHTS_F32, sizeof(csi_status), CSI_VERSION);
    printf("  \"library_version\": %d\n}\n", (int)csi_version());
    return 0;
}

/* ---- records ------------------------------------------------------------------------------------------------------------------- */
typedef struct { char name[16]; int32_t dtype, pad; int64_t count; void* data; } rec_t;
static rec_t recs[64];
static int nrecs = 0;

static int read_records(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); return 1; }
    while (nrecs < 64) {
        rec_t* r = &recs[nrecs];
        if (fread(r->name, 1, 16, f) != 16) break;
        if (fread(&r->dtype, 4, 1, f) != 1 || fread(&r->pad, 4, 1, f) != 1 || fread(&r->count, 8, 1, f) != 1) { fclose(f); return 1; }
        const size_t nb = (size_t)r->count * (r->dtype ? 8 : 4);
        r->data = malloc(nb ? nb : 1);
        if (fread(r->data, 1, nb, f) != nb) { fclose(f); return 1; }
        ++nrecs;
    }
    fclose(f);
    return 0;
}
static const rec_t* rec(const char* name) {
    for (int k = 0; k < nrecs; ++k) if (!strncmp(recs[k].name, name, 16)) return &recs[k];
    fprintf(stderr, "abi_client: record %s is missing\n", name);
    exit(2);
}
static int32_t geti(const char* name, int k) { return ((const int32_t*)rec(name)->data)[k]; }
static double getd(const char* name, int k) { return ((const double*)rec(name)->data)[k]; }
static void put(FILE* f, const char* name, const double* v, int64_t n) {
    char nm[16] = {0};
    strncpy(nm, name, 15);
    const int32_t dtype = 1, pad = 0;
    fwrite(nm, 1, 16, f); fwrite(&dtype, 4, 1, f); fwrite(&pad, 4, 1, f); fwrite(&n, 8, 1, f); fwrite(v, 8, (size_t)n, f);
}

#define CHECK(ctx, call) do { int32_t rc_ = (call); if (rc_ != CSI_OK) { fprintf(stderr, "abi_client: %s -> %d: %s\n", #call, (int)rc_, csi_last_error(ctx)); return 3; } } while (0)
#define HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "abi_client: %s: %s\n", #call, hipGetErrorString(e_)); return 4; } } while (0)

static int run(const char* in, const char* out) {
    if (read_records(in)) return 1;
    const int32_t Nx = geti("grid", 0), Ny = geti("grid", 1), Hx = geti("grid", 2), Hy = geti("grid", 3), topo_x = geti("grid", 4), topo_y = geti("grid", 5),
                  metric_kind = geti("grid", 6), substeps = geti("grid", 7), mode = geti("grid", 8);
    csi_context* ctx = NULL;
    CHECK(NULL, csi_context_create(0, NULL, &ctx));

    /* csi_grid_set: the grid argument of every reference kernel */
    csi_metrics met;
    memset(&met, 0, sizeof met);
    met.dx = getd("metrics", 0); met.dy = getd("metrics", 1);
    if (metric_kind == CSI_METRIC_PER_J) {
        met.dxc = (const double*)rec("dxc")->data; met.dxf = (const double*)rec("dxf")->data;
        met.azc = (const double*)rec("azc")->data; met.azf = (const double*)rec("azf")->data;
    }
    CHECK(ctx, csi_grid_set(ctx, Nx, Ny, Hx, Hy, topo_x, topo_y, metric_kind, &met));

    /* the fourteen fields of the EVP sub-cycle: parents owned by THIS program (hipMalloc), as Julia's ROCArray parents would be */
    static const int ids[14] = {CSI_F_U, CSI_F_V, CSI_F_H, CSI_F_A, CSI_F_S11, CSI_F_S22, CSI_F_S12, CSI_F_UN, CSI_F_VN, CSI_F_P, CSI_F_ALPHA,
                                CSI_F_DELTA, CSI_F_ZETA_F, CSI_F_ZETA_C};
    static const char* names[14] = {"u", "v", "h", "a", "s11", "s22", "s12", "un", "vn", "P", "alpha", "Delta", "zeta_f", "zeta_c"};
    double* dev[14];
    int64_t elems[14];
    for (int k = 0; k < 14; ++k) {
        const int32_t ni = geti("extents", 2 * k), nj = geti("extents", 2 * k + 1);
        elems[k] = (int64_t)ni * nj;
        HIP(hipMalloc((void**)&dev[k], (size_t)elems[k] * 8));
        HIP(hipMemset(dev[k], 0, (size_t)elems[k] * 8));
        if (k < 4) {
            const rec_t* r = rec(names[k]);
            if (r->count != elems[k]) { fprintf(stderr, "abi_client: %s has %lld elements, expected %lld\n", names[k], (long long)r->count, (long long)elems[k]); return 2; }
            HIP(hipMemcpy(dev[k], r->data, (size_t)elems[k] * 8, hipMemcpyHostToDevice));
        }
        if (ids[k] == CSI_F_ALPHA) {
            /* rheology_auxiliary_fields pre-fills alpha with max_relaxation_parameter (elasto_visco_plastic_rheology.jl:147-161): the
             * outermost halo layer keeps that value (no kernel writes it) */
            double* host = (double*)malloc((size_t)elems[k] * 8);
            for (int64_t t = 0; t < elems[k]; ++t) host[t] = getd("evp", 5);
            HIP(hipMemcpy(dev[k], host, (size_t)elems[k] * 8, hipMemcpyHostToDevice));
            free(host);
        }
        CHECK(ctx, csi_field_bind(ctx, ids[k], dev[k], ni, ni, nj));
    }

    csi_evp_params p;
    memset(&p, 0, sizeof p);
    p.ice_compressive_strength = getd("evp", 0); p.ice_compaction_hardening = getd("evp", 1); p.yield_curve_eccentricity = getd("evp", 2);
    p.minimum_plastic_stress = getd("evp", 3); p.min_relaxation_parameter = getd("evp", 4); p.max_relaxation_parameter = getd("evp", 5);
    p.relaxation_strength = getd("evp", 6); p.coriolis_f = getd("evp", 7); p.minimum_concentration = getd("evp", 8);
    p.minimum_mass = getd("evp", 9); p.sea_ice_density = getd("evp", 10);
    p.pressure_formulation = geti("evp_i", 0); p.has_coriolis = geti("evp_i", 1);
    CHECK(ctx, csi_evp_params_set(ctx, &p));

    for (int side = 0; side < 2; ++side) {
        const char* nm = side ? "stress_bot" : "stress_top";
        const char* ni = side ? "stress_bot_i" : "stress_top_i";
        csi_stress s;
        memset(&s, 0, sizeof s);
        s.kind = geti(ni, 0); s.ue_kind = geti(ni, 1); s.ve_kind = geti(ni, 2);
        s.tau_u = getd(nm, 0); s.tau_v = getd(nm, 1); s.ue = getd(nm, 2); s.ve = getd(nm, 3); s.rho_e = getd(nm, 4); s.Cd = getd(nm, 5);
        CHECK(ctx, csi_stress_set(ctx, side ? CSI_STRESS_BOTTOM : CSI_STRESS_TOP, &s));
    }
    CHECK(ctx, csi_set_mode(ctx, mode));
    /* time_step_momentum!(model, ::SplitExplicitMomentumEquation, dt), split_explicit_momentum_equations.jl:103-195 */
    CHECK(ctx, csi_time_step_momentum(ctx, getd("dt", 0), substeps, 0));
    CHECK(ctx, csi_sync(ctx));
    double ms = -1.0;
    CHECK(ctx, csi_last_subcycle_ms(ctx, &ms));

    FILE* f = fopen(out, "wb");
    if (!f) { perror(out); return 1; }
    for (int k = 0; k < 14; ++k) {
        if (k == 2 || k == 3) continue;
        double* host = (double*)malloc((size_t)elems[k] * 8);
        HIP(hipMemcpy(host, dev[k], (size_t)elems[k] * 8, hipMemcpyDeviceToHost));
        put(f, names[k], host, elems[k]);
        free(host);
    }
    fclose(f);
    for (int k = 0; k < 14; ++k) HIP(hipFree(dev[k]));
    CHECK(ctx, csi_context_destroy(ctx));
    printf("abi_client: %d sub-steps on %d x %d in %.3f ms (library version %d)\n", (int)substeps, (int)Nx, (int)Ny, ms, (int)csi_version());
    return 0;
}

int main(int argc, char** argv) {
    if (argc == 2 && !strcmp(argv[1], "layout")) return layout();
    if (argc == 4 && !strcmp(argv[1], "run")) return run(argv[2], argv[3]);
    fprintf(stderr, "usage: abi_client layout | abi_client run IN OUT\n");
    return 64;
}
