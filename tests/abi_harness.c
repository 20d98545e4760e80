/* abi_harness.c -- the call sequence a Julia `ccall` shim makes (julia/CarParkingMapsAMD.jl; main.jl:82-102), from plain C against
 * include/cpm.h: pointer types as Julia passes them (Ptr{Float64} / Ptr{Int64} into column-major arrays, Ref{Ptr{Cvoid}} and
 * Ref{Int64} out-parameters, C_NULL for the optional matrices), status codes and cpm_last_error().  Linked against libcpm_hip.so
 * directly, so the compiler checks every call against the header's prototypes.  Test infrastructure (tests/test_abi.py):
 * prints what the calls returned as text; the Python test compares it with what the ctypes path gets for the same inputs.
 *
 *   abi_harness <Z> <cars_per_zone> <seed> [symbols]
 */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cpm.h"

#define CHECK(call)                                                                          \
    do {                                                                                     \
        int32_t st_ = (call);                                                                \
        if (st_ != CPM_OK) {                                                                 \
            fprintf(stderr, "%s -> status %d: %s\n", #call, (int)st_, cpm_last_error());     \
            return 2;                                                                        \
        }                                                                                    \
    } while (0)

static uint64_t fnv(const void *p, size_t n)
{
    const unsigned char *b = (const unsigned char *)p;
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
    return h;
}

int main(int argc, char **argv)
{
    if (argc >= 2 && strcmp(argv[1], "symbols") == 0) { /* every entry of the header resolves (no device needed) */
        const void *fns[] = {(void *)cpm_last_error, (void *)cpm_version, (void *)cpm_device_count, (void *)cpm_device_info, (void *)cpm_create,
                             (void *)cpm_destroy, (void *)cpm_set_option, (void *)cpm_get_info, (void *)cpm_set_stream, (void *)cpm_sync,
                             (void *)cpm_set_p_drive, (void *)cpm_set_p_dest, (void *)cpm_set_datamatrix, (void *)cpm_createdatamatrix_rows,
                             (void *)cpm_parse_uber_csv, (void *)cpm_createdatamatrix_csv, (void *)cpm_get_datamatrix,
                             (void *)cpm_set_distance_from_centroids, (void *)cpm_set_distance, (void *)cpm_get_distance, (void *)cpm_build_p_drive,
                             (void *)cpm_build_p_dest, (void *)cpm_get_p_drive, (void *)cpm_get_cdf_row, (void *)cpm_init_states,
                             (void *)cpm_init_states_strided, (void *)cpm_set_state, (void *)cpm_get_state, (void *)cpm_solve_ivp,
                             (void *)cpm_resample, (void *)cpm_resample_dev, (void *)cpm_solve_ivp_async, (void *)cpm_synth_tables,
                             (void *)cpm_synth_tables_skewed, (void *)cpm_last_kernel_ms, (void *)cpm_algorithmic_bytes_per_hour,
                             (void *)cpm_debug_categorical, (void *)cpm_synth_datamatrix, (void *)cpm_refresh_tables};
        size_t n = sizeof fns / sizeof fns[0], ok = 0;
        for (size_t i = 0; i < n; ++i) ok += fns[i] != NULL;
        printf("symbols %zu of %zu, version %d\n", ok, n, (int)cpm_version());
        return ok == n ? 0 : 1;
    }
    if (argc < 4) {
        fprintf(stderr, "usage: abi_harness <Z> <cars_per_zone> <seed> | abi_harness symbols\n");
        return 64;
    }
    const int64_t Z = atoll(argv[1]), cpz = atoll(argv[2]), T = 24, C = Z * cpz;
    const uint64_t seed = strtoull(argv[3], NULL, 0);
    cpm_ctx *h = NULL;                                   /* Ref{Ptr{Cvoid}}(C_NULL) */
    CHECK(cpm_create(&h, Z, T, 0));
    /* tables the way main.jl:82-85 hands them over: host arrays, column-major Z x T and Z x Z x T */
    double *p_drive = (double *)malloc(sizeof(double) * (size_t)(Z * T));
    double *p_dest = (double *)calloc((size_t)(Z * Z * T), sizeof(double));
    for (int64_t t = 0; t < T; ++t)
        for (int64_t o = 0; o < Z; ++o) {
            p_drive[o + Z * t] = 0.1 + 0.8 * (double)((o * 7 + t * 13) % 97) / 96.0;
            double tot = 0;
            for (int64_t d = 0; d < Z; ++d) tot += (o == d) ? 0.0 : (double)(1 + (o * 31 + d * 17 + t * 5) % 23);
            for (int64_t d = 0; d < Z; ++d) p_dest[o + Z * (d + Z * t)] = (o == d) ? 0.0 : (double)(1 + (o * 31 + d * 17 + t * 5) % 23) / tot;
        }
    CHECK(cpm_set_p_drive(h, p_drive));
    CHECK(cpm_set_p_dest(h, p_dest));
    /* main.jl:88-92: initializestates -> solveinitialvalueproblem -> state_matrix[:,1] = initial_state */
    int64_t *zones = (int64_t *)malloc(sizeof(int64_t) * (size_t)C);
    for (int64_t i = 0; i < C; ++i) zones[i] = i / cpz + 1; /* 1-based, as state_matrix[:,1] holds them */
    CHECK(cpm_init_states(h, C, cpz, 0, C));
    CHECK(cpm_set_state(h, zones));
    int64_t *initial_state = (int64_t *)calloc((size_t)C, sizeof(int64_t));
    CHECK(cpm_solve_ivp(h, seed, initial_state));
    /* main.jl:95-102, fast mode: counts + sum of travel times, no C x T matrices (C_NULL) */
    int64_t *parking = (int64_t *)calloc((size_t)(Z * T), sizeof(int64_t)), *driving = (int64_t *)calloc((size_t)(Z * T), sizeof(int64_t));
    int64_t tt = -1;                                      /* Ref{Int64} */
    CHECK(cpm_resample(h, seed, 0u, parking, driving, &tt, NULL, NULL));
    /* compat mode: the reference's state_matrix (C x T Int64) and transition_matrix (C x T x 4 Float64), filled in place */
    int64_t *state = (int64_t *)calloc((size_t)(C * T), sizeof(int64_t));
    double *trans = (double *)calloc((size_t)(C * T * 4), sizeof(double));
    int64_t *parking2 = (int64_t *)calloc((size_t)(Z * T), sizeof(int64_t)), *driving2 = (int64_t *)calloc((size_t)(Z * T), sizeof(int64_t));
    CHECK(cpm_resample(h, seed, 0u, parking2, driving2, NULL, state, trans));
    int64_t info_kernel = -1;
    CHECK(cpm_get_info(h, CPM_INFO_KERNEL, &info_kernel));
    /* error convention: a status, a message, no abort */
    int32_t bad = cpm_set_state(h, NULL);
    printf("Z %" PRId64 " C %" PRId64 " kernel %" PRId64 " null_state_status %d\n", Z, C, info_kernel, (int)bad);
    printf("initial_state %016" PRIx64 "\n", fnv(initial_state, sizeof(int64_t) * (size_t)C));
    printf("parking %016" PRIx64 " driving %016" PRIx64 " tt %" PRId64 "\n", fnv(parking, sizeof(int64_t) * (size_t)(Z * T)),
           fnv(driving, sizeof(int64_t) * (size_t)(Z * T)), tt);
    printf("compat_counts_equal %d\n", memcmp(parking, parking2, sizeof(int64_t) * (size_t)(Z * T)) == 0 &&
                                           memcmp(driving, driving2, sizeof(int64_t) * (size_t)(Z * T)) == 0);
    printf("state %016" PRIx64 " trans %016" PRIx64 "\n", fnv(state, sizeof(int64_t) * (size_t)(C * T)), fnv(trans, sizeof(double) * (size_t)(C * T * 4)));
    int64_t hour_sum = 0;
    for (int64_t z = 0; z < Z; ++z) hour_sum += parking[z + Z * (T - 1)];
    printf("hour24_cars %" PRId64 "\n", hour_sum);
    CHECK(cpm_destroy(h));
    free(p_drive); free(p_dest); free(zones); free(initial_state); free(parking); free(driving); free(state); free(trans); free(parking2); free(driving2);
    return 0;
}
