/*
 * cpm.h -- C ABI of libcpm_hip.so: the MI355X (gfx950) implementation of the
 * CarParkingMaps HMM traffic-flow sampler path.
 *
 * This is the drop-in boundary.  The reference (Julia, /root/reference) has no
 * FFI of its own; these are the entry points a `ccall` shim binds so that
 * main.jl:82-102 keeps its call surface (julia/CarParkingMapsAMD.jl and
 * INTEGRATION.md show the binding).  Each entry cites the reference interface
 * it replaces.
 *
 * Conventions
 *  - every function returns int32 status: 0 = ok, < 0 = error; the message of the
 *    last error on the calling thread is cpm_last_error().  No exception and no
 *    abort crosses the boundary.
 *  - host arrays are the reference's Julia arrays as they lie in memory:
 *    column-major Float64 / Int64, zone ids 1-based.  They are borrowed for the
 *    duration of the call only (Julia side: GC.@preserve).
 *      p_drive    Z x T          (src/createpdrive.jl:4,36)
 *      p_dest     Z x Z x T      (src/createpdestin.jl:4,48)
 *      datamatrix Z x Z x T x 2  (src/createdatamatrix.jl:7,25)
 *      dist       Z x Z          (src/processgeodata.jl:150)
 *      state      C x T  Int64   (src/initializestates.jl:6,17)
 *      trans      C x T x 4      (src/initializestates.jl:7)
 *      counts     Z x T  Int64   (src/saveresults.jl:8-9, held as Float64 there)
 *  - the library owns all device memory inside the opaque cpm_ctx.  A context is
 *    bound to ONE device and is not thread-safe; distinct contexts may be used
 *    from distinct threads / processes (one process per GPU for multi-GPU runs;
 *    the count tensors are summed by the host layer with one RCCL all-reduce).
 *  - functions without the _dev/_async suffix block until the device is done.
 *  - there is NO CPU fallback: if no HIP device is usable every call fails.
 *
 * RNG contract (the reference is unseeded: bare rand(), src/resampling.jl:13,29):
 * Philox4x32-10, key = seed, counter = (car_lo, car_hi, step, stream) with car
 * the GLOBAL 0-based car id, step 0..T-2 for the initial-value problem and
 * T-1..2T-2 for resampling, stream 0 = (Bernoulli u, categorical u), streams
 * >= 1 = travel time / distance attempts.  Results are therefore independent
 * of thread, workgroup, GPU and shard layout.
 */
#ifndef CPM_H
#define CPM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CPM_OK 0
#define CPM_ERR_ARG (-1)    /* bad argument */
#define CPM_ERR_HIP (-2)    /* HIP runtime error / no device */
#define CPM_ERR_STATE (-3)  /* call order: a required table or state is missing */
#define CPM_ERR_TABLE (-4)  /* p_dest holds NaN / negative entries (the reference would
                               crash with BoundsError one hour later, Appendix A-7) */
#define CPM_ERR_NOMEM (-5)

/* cpm_resample flags */
#define CPM_FLAG_TRAVEL 1u      /* pass 3 of src/resampling.jl:53-78 (needs cpm_set_datamatrix) */

/* kernel families (cpm_set_option CPM_OPT_KERNEL) */
#define CPM_KERNEL_AUTO 0
#define CPM_KERNEL_CAR 1          /* one thread per car, CDF searched in HBM/L2; also what fills the reference's full state_matrix / transition_matrix */
#define CPM_KERNEL_ZONE_LDS 2     /* cars bucketed by zone, f64 CDF row staged in LDS; exact (packed) bucket layout, three launches per hour;
                                   * cannot overflow: the fallback of the grouped path */
#define CPM_KERNEL_ZONE_GROUPED 5 /* what AUTO runs: fixed-stride buckets, row packs (guide + 4-byte CDF high words, exact f64 fallback on ties)
                                   * staged by LDS-DMA, stayers kept by the sampler, drivers placed per destination group; two launches per hour */

#define CPM_OPT_KERNEL 1
#define CPM_OPT_PROFILE 2       /* N >= 1: every N-th hourly launch of the profiled kernel carries a hipEvent pair stamped with the
                                   dispatch's own begin and end (hipExtLaunchKernelGGL); 0: off */
#define CPM_OPT_FUSED 4         /* the grouped path's fused hour (sampler workgroups and the placing blocks of their drivers in ONE launch per hour):
                                   5 (default) on where it pays -- from two rounds of sampler workgroups on (Z >= 12 x the CUs) while the row pack
                                   leaves five blocks per CU (Z <= ~5,600): measured, DESIGN.md 4.1 --, 1 on wherever an instantiation exists,
                                   0 off = two launches per hour, 2 = on with placing blocks that give up waiting at once (the
                                   tests' way into the bail-out: the step comes back with status bit 2 set and the context falls back to 0);
                                   3 = the placing-first form (the PREVIOUS hour's placing blocks in front of the hour's sampler workgroups,
                                   which wait for their zone's destination group; measured slower than 1 at Z = 4,096), 4 = 3
                                   with sampler workgroups that give up waiting at once;
                                   6 = ALL hours of a run but the last in ONE launch (k_grouped_day, csrc/cpm_day.h: an hour's placing blocks in
                                   front of / among the next hour's sampler workgroups, which draw for their stayers first when the placing is
                                   not done yet; measured slower than 1 at every size tried), 8 = 6 with every placing block of an
                                   hour in front of every sampler workgroup of the next, 7 = 6 with blocks that give up waiting at once.
                                   The library reads no environment variable. */
#define CPM_OPT_FUSED_LAG 5     /* chunks of 64 sampler workgroups between a chunk and its placing blocks in the fused launch; at or above the number
                                   of chunks (the default): every sampler workgroup first, then every placing block */
#define CPM_OPT_ZONE_ORDER 6    /* 1: the one-launch hour deals its sampler workgroups the zones LARGEST-FIRST (by their size at the same hour of the
                                   last initial-value problem run in this context); 0: zone order; 2 (default): largest-first on sparse row packs
                                   (datasets), zone order on dense ones.  A scheduling hint: the counts do not depend on it.  Measured (same box,
                                   profiles/round4_notes.md): Melbourne-shaped tables, Z = 2,357 x 1,000 cars per zone 0.686 -> 0.587 ms per resample
                                   (1.5 rounds of workgroups: the hour ended on whatever large bucket came last), x 500 0.443 -> 0.412; dense
                                   4,096 zones 2-4 % SLOWER (zone order is also memory order of the packs and of the runs) */
#define CPM_OPT_PROFILE_KERNEL 3 /* which hourly launch CPM_OPT_PROFILE brackets: */
#define CPM_PROFILE_SAMPLER 0   /*   the sampler (default; every kernel family has one) */
#define CPM_PROFILE_PLACE 1     /*   the grouped path's placing kernel */
#define CPM_PROFILE_TRAVEL 2    /*   the grouped path's travel-time kernel (CPM_FLAG_TRAVEL) */

typedef struct cpm_ctx cpm_ctx;

const char *cpm_last_error(void);
int32_t cpm_version(void);
int32_t cpm_device_count(int32_t *n_out);
/* name[len] <- device name; *cu_count, *hbm_bytes optional */
int32_t cpm_device_info(int32_t device_id, char *name, int32_t len, int32_t *cu_count,
                        int64_t *hbm_bytes);

/* Z = number_zones (main.jl:59), T = 24 (main.jl:42) */
int32_t cpm_create(cpm_ctx **ctx_out, int64_t Z, int64_t T, int32_t device_id);
int32_t cpm_destroy(cpm_ctx *ctx);
int32_t cpm_set_option(cpm_ctx *ctx, int32_t option, int64_t value);
/* what the context would run now: CPM_INFO_KERNEL = the kernel family AUTO resolves to for the current tables / cars
 * (CPM_KERNEL_*; changes to CPM_KERNEL_ZONE_LDS after an overflow that could not be absorbed), CPM_INFO_CAP_MULT = size of a
 * zone's bucket region on the fixed-stride layouts in multiples of the mean bucket size (4; doubled, up to 64, each time a bucket
 * or run outgrew it) */
#define CPM_INFO_KERNEL 1
#define CPM_INFO_CAP_MULT 2
#define CPM_INFO_PARTS 3   /* workgroups per zone of the grouped sampler: 1, or more once a bucket above FOUR times a workgroup's slots was seen
                            * (kHeavy in cpm_grouped.h; lighter overflow stays with the overflow rounds of the zone's own workgroup) */
#define CPM_INFO_SPARSE_TABLES 6  /* 0, or -- the installed p_destin tables were built from a sparse datamatrix's compact rows (csrc/cpm_dataset.h:
                                   * cpm_build_p_dest on a datamatrix whose longest (origin, hour) row holds at most 512 cells and whose sparse
                                   * row pack is at most 60 % of the dense one) -- the 32-bit words of one sparse row pack */
#define CPM_INFO_FUSED_BAILOUTS 5 /* steps so far that came back with status bit 2 (a block of a one-launch form gave up waiting: the context then keeps to two launches per hour) */
#define CPM_INFO_FUSED 4   /* 1 (3: in its placing-first form, 6: all hours in one launch) when the next grouped step runs the fused hour (one launch per hour), 0 when it takes two launches per hour: switched
                            * off (CPM_OPT_FUSED), heavy buckets seen (CPM_INFO_PARTS > 1), rows / groups outside the fused instantiations, or
                            * a placing block once gave up waiting */
int32_t cpm_get_info(cpm_ctx *ctx, int32_t what, int64_t *value_out);
/* run on a caller-owned hipStream_t (e.g. torch's current stream); NULL = ctx's own, which is created when a call first needs it.
 * Contexts meant to run side by side (two resamples interleave on the chip) are each given their stream right behind
 * cpm_create: HIP streams take the process's hardware queues (GPU_MAX_HW_QUEUES, 4 by default) when they are created, and streams
 * that share a queue run one after the other. */
int32_t cpm_set_stream(cpm_ctx *ctx, void *hip_stream);
int32_t cpm_sync(cpm_ctx *ctx);

/* ---- probability tables ------------------------------------------------- */
/* takes the array createpdrive returns (src/createpdrive.jl:36; main.jl:82) */
int32_t cpm_set_p_drive(cpm_ctx *ctx, const double *p_drive);
/* takes the array createpdestin returns (src/createpdestin.jl:48; main.jl:85); builds the
 * canonical CDF on device: sequential left-to-right f64 sum per (origin, hour), the
 * accumulation of src/resampling.jl:39 */
int32_t cpm_set_p_dest(cpm_ctx *ctx, const double *p_dest);
/* uploads createdatamatrix's array (main.jl:79) and processgeodata's distance matrix
 * (main.jl:59); enables cpm_build_* and CPM_FLAG_TRAVEL.  dist may be NULL: the distance matrix
 * already resident (cpm_set_distance*) is kept */
int32_t cpm_set_datamatrix(cpm_ctx *ctx, const double *datamatrix, const double *dist);
/* createdatamatrix(path_to_csv_data, number_zones) (src/createdatamatrix.jl:3-27; main.jl:79) without the
 * dense host array: the Z x Z x T x 2 datamatrix is built in HBM and stays there.
 *   _rows: rawdata = the reference's rawdata[:,1:5] (n_rows x 5 Float64, column-major: sourceid, dstid,
 *          hod, mean_travel_time, standard_deviation_travel_time) -- for a host that has parsed the CSV itself;
 *   _csv : reads the Uber Movement CSV natively (header line, then 7 comma-separated numeric fields per
 *          line of which the first five are used; memory-mapped, parsed by several threads).
 * Zone id 0 -> Z on both endpoints, hod 0 -> 24 (:9-17); of several rows with the same (source, dest, hour)
 * the last one wins (:21-22).  A row whose ids / hour are not integers in range -> CPM_ERR_ARG
 * (reference: InexactError / BoundsError). */
int32_t cpm_createdatamatrix_rows(cpm_ctx *ctx, int64_t n_rows, const double *rawdata);
/* the CSV reader of _csv alone (host only, no device needed): *n_rows_out = data rows in the file; when
 * rawdata_out is not NULL it receives the capacity_rows x 5 column-major matrix (first *n_rows_out rows filled) */
int32_t cpm_parse_uber_csv(const char *path_to_csv_data, int64_t *n_rows_out, double *rawdata_out_or_null,
                           int64_t capacity_rows);
int32_t cpm_createdatamatrix_csv(cpm_ctx *ctx, const char *path_to_csv_data, int64_t *n_rows_out_or_null);
int32_t cpm_get_datamatrix(cpm_ctx *ctx, double *datamatrix_out);
/* the distance part of processgeodata (src/processgeodata.jl:148-166; main.jl:59): centroid_lat / centroid_long
 * [Z] (degrees) -> distance_matrix_km [Z x Z] in HBM, 111.3 * sqrt(cos(mean lat * 0.01745)^2 * dlong^2 + dlat^2),
 * diagonal 1 km.  (The GeoJSON parsing and the centroid sums are O(vertices) host work: host layer.) */
int32_t cpm_set_distance_from_centroids(cpm_ctx *ctx, const double *centroid_lat, const double *centroid_long);
/* uploads a distance matrix computed by the host (the reference's own processgeodata) next to a datamatrix built in HBM */
int32_t cpm_set_distance(cpm_ctx *ctx, const double *distance_matrix_km);
int32_t cpm_get_distance(cpm_ctx *ctx, double *distance_matrix_km_out);
/* createpdrive(datamatrix, distance_matrix_km, number_zones) with the script globals
 * p_min, p_max, e_drive passed explicitly (src/createpdrive.jl:3-38); installs the table and
 * optionally returns it */
int32_t cpm_build_p_drive(cpm_ctx *ctx, double p_min, double p_max, double e_drive,
                          double *p_drive_out_or_null);
/* createpdestin(datamatrix, number_zones) with e_dest explicit (src/createpdestin.jl:3-50);
 * e_is_integer != 0 reproduces Julia's Float64^Int (main.jl:38: e_dest = 2) */
int32_t cpm_build_p_dest(cpm_ctx *ctx, double e_dest, int32_t e_is_integer,
                         double *p_dest_out_or_null);
/* read back the installed tables (Z x T / Z x Z x T, column-major) */
int32_t cpm_get_p_drive(cpm_ctx *ctx, double *p_drive_out);
int32_t cpm_get_cdf_row(cpm_ctx *ctx, int64_t origin1, int64_t hour1, double *cdf_row_out);

/* ---- cars --------------------------------------------------------------- */
/* initializestates(C) (src/initializestates.jl:4-22; main.jl:88): global car g (0-based) starts
 * in zone g / cars_per_zone + 1.  This context simulates the shard
 * [car_begin, car_begin + car_count) of the C_total cars. */
int32_t cpm_init_states(cpm_ctx *ctx, int64_t C_total, int64_t cars_per_zone, int64_t car_begin,
                        int64_t car_count);
/* the same for the interleaved deal of a multi-GPU run: this context simulates the global cars
 * car_first + k * car_stride, k = 0 .. car_count - 1 (rank r of N: car_first = r, car_stride = N).  Every rank then starts with
 * its share of EVERY zone, instead of all cars of 1/N of the zones as with contiguous ranges; results are identical either
 * way (Philox is keyed by the global car id). */
int32_t cpm_init_states_strided(cpm_ctx *ctx, int64_t C_total, int64_t cars_per_zone, int64_t car_first,
                                int64_t car_stride, int64_t car_count);
/* state_matrix[:,1] = initial_state (main.jl:92): zones 1-based, car_count entries */
int32_t cpm_set_state(cpm_ctx *ctx, const int64_t *zones);
int32_t cpm_get_state(cpm_ctx *ctx, int64_t *zones_out);
/* solveinitialvalueproblem (src/solveinitialvalueproblem.jl:4-62; main.jl:91): T-1 steps,
 * advances the context's current state in place and optionally returns it */
int32_t cpm_solve_ivp(cpm_ctx *ctx, uint64_t seed, int64_t *initial_state_out_or_null);
/* resampling + the histogram of saveresults + the sum of averagedrivingtime, fused
 * (src/resampling.jl:3-89; src/saveresults.jl:6-17; src/averagedrivingtime.jl:7-8;
 * main.jl:95,98,102).  Starts from the context's current state and leaves it unchanged, so
 * it can be called repeatedly (the model-selection loops, README.md:1180).
 *   parking_counts, driving_counts : Z x T Int64 (this shard's cars only)
 *   sum_travel_time_q16            : sum of transition_matrix[:,:,3] in 2^-16 s units (0 without
 *                                    CPM_FLAG_TRAVEL); order-free, so shards add exactly
 *   state_out_or_null              : C x T Int64, the reference's state_matrix
 *   trans_out_or_null              : C x T x 4 Float64, the reference's transition_matrix */
int32_t cpm_resample(cpm_ctx *ctx, uint64_t seed, uint32_t flags, int64_t *parking_counts,
                     int64_t *driving_counts, int64_t *sum_travel_time_q16,
                     int64_t *state_out_or_null, double *trans_out_or_null);

/* ---- device-resident forms for the host layer (torch tensors, RCCL) ----- */
/* enqueue the fused resample on the context's stream and return without synchronising.
 * d_counts: DEVICE pointer to int64[2*T*Z + 2] = parking[T][Z] | driving[T][Z] | sum_tt_q16 |
 * status, zeroed and filled by the call (ready for one all-reduce).  status != 0 (after the
 * reduce: on any rank) means a fixed-stride zone kernel met a bucket that outgrew its region
 * (more than 4x the mean zone population in one zone) or a run that outgrew its slot; the counts
 * are then invalid and the caller repeats the step: the context has doubled its regions when it
 * next enqueues (while the problem fits, up to 64x the mean), after that a context running
 * CPM_KERNEL_AUTO uses the exact layout.  cpm_resample and cpm_solve_ivp do all of this by themselves. */
int32_t cpm_resample_dev(cpm_ctx *ctx, uint64_t seed, uint32_t flags, void *d_counts);
int32_t cpm_solve_ivp_async(cpm_ctx *ctx, uint64_t seed);
/* procedural synthetic tables of SURVEY.md 8(d), generated on device (bench inputs):
 * p_drive = 0.1 + 0.8 u ; dense p_dest ~ u^2, zero diagonal, row-normalised */
int32_t cpm_synth_tables(cpm_ctx *ctx, uint64_t table_seed);
/* the same with skewed destination popularity (bench.py --skew): weight u^2 / (skew_q + rank(d)), rank(d) = (7919 d + 13) mod Z
 * -- a few destinations many times as likely as the mean, like real Uber Movement rows (README.md output_24_0.svg) */
int32_t cpm_synth_tables_skewed(cpm_ctx *ctx, uint64_t table_seed, int64_t skew_q);
/* Melbourne-shaped synthetic datamatrix + distance matrix of SURVEY.md 8(d), generated on device (bench inputs for the per-dataset
 * flow main.jl:79-95: 8.68 % of the (o, d, t) cells populated, README.md:302-310); enables cpm_build_* and CPM_FLAG_TRAVEL like
 * cpm_set_datamatrix */
int32_t cpm_synth_datamatrix(cpm_ctx *ctx, uint64_t table_seed, double density);
/* re-derives the samplers' row tables (row totals, running-sum checkpoints, row packs; with_f64_cdf != 0: the canonical f64 CDF rows
 * too) from the p_destin resident in the context: the one pass cpm_set_p_dest / cpm_build_p_dest / cpm_synth_tables end in
 * (src/resampling.jl:39's running sum).  For measurement (bench.py's table_build record) and after cpm_set_option changes nothing it
 * reads; results are those of the installing call. */
int32_t cpm_refresh_tables(cpm_ctx *ctx, int32_t with_f64_cdf);
/* with CPM_OPT_PROFILE: durations (ms) of the hourly launches of the profiled kernel (CPM_OPT_PROFILE_KERNEL) made by
 * cpm_resample* since the option was last set, in launch order (hipEvents on the context's
 * stream); returns the number written through *n_out */
int32_t cpm_last_kernel_ms(cpm_ctx *ctx, float *ms_out, int32_t cap, int32_t *n_out);
/* algorithmic HBM bytes of one hourly sampler launch, with the element
 * size of the rows the selected kernel really streams (4-byte high words on the default grouped
 * path, 8-byte f64 otherwise) */
int32_t cpm_algorithmic_bytes_per_hour(cpm_ctx *ctx, int64_t *bytes_out);
/* diagnostic: the categorical draw (src/resampling.jl:38-45) of the grouped zone sampler for the
 * given 53-bit uniforms (u = k * 2^-53) against the installed row p_dest[origin,:,hour], through
 * the sampler's own staging, high-word tree walk and exact-row fallback.  dest_out[i] = 1-based
 * destination, or 0 for an all-zero row (the sampler then keeps the origin, :35-36);
 * *n_exact_out = how many draws took the exact (f64 row) fallback. */
int32_t cpm_debug_categorical(cpm_ctx *ctx, int64_t origin1, int64_t hour1, int64_t n, const uint64_t *k53,
                              int64_t *dest_out, int32_t *n_exact_out_or_null);

#ifdef __cplusplus
}
#endif
#endif
