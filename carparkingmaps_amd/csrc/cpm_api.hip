// cpm_api.hip -- host side of libcpm_hip.so: context, device memory, launches and the
// C ABI declared in include/cpm.h.  gfx950 only; there is no CPU path in this library.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/cpm.h"
#include "cpm_kernels.h"
#include "cpm_tables.h"
#include "cpm_exact.h"
#include "cpm_grouped.h"
#include "cpm_dataset.h"
#include "cpm_ingest.h"

static hipError_t ensure_stream(cpm_ctx *c);

namespace {

thread_local std::string g_last_error;

int32_t fail(int32_t code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(e_ == hipErrorOutOfMemory ? CPM_ERR_NOMEM : CPM_ERR_HIP, "%s: %s (%s:%d)", #expr, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                            \
    } while (0)

// Every entry that enqueues work: the context's device is current and the context has a stream (its own one is created on first
// need -- a HIP stream takes one of the process's few hardware queues when it is CREATED, GPU_MAX_HW_QUEUES = 4 by default, and
// streams beyond that share queues: two contexts that were each given a caller's stream before doing anything hold two queues and
// their resamples interleave on the chip; with two unused own streams in between they were measured to serialise).
#define CTX_TRY(ctx)                                                 \
    if (!(ctx)) return fail(CPM_ERR_ARG, "null context");            \
    HIP_TRY(hipSetDevice((ctx)->device));                            \
    HIP_TRY(ensure_stream(ctx))

template <typename T>
void dfree(T *&p)
{
    if (p) (void)hipFree(p);
    p = nullptr;
}

inline unsigned nblk(int64_t n, int b) { return static_cast<unsigned>((n + b - 1) / b); }

}  // namespace

struct cpm_ctx {
    int64_t Z = 0, T = 0;
    int Zp = 0;
    int device = 0;
    int cu_count = 0;
    hipStream_t own_stream = nullptr;  // created on first need (ensure_stream)
    hipStream_t stream = nullptr;      // what the context enqueues on: the caller's (cpm_set_stream) or own_stream
    bool have_stream = false;
    // tables
    double *d_pdrive = nullptr;  // [T][Z]
    double *d_p = nullptr;       // [T][Z dest][Z origin] p_destin as the reference lays it out (cpm_set_p_dest, cpm_build_p_dest, the synthetic
                                 // tables).  Resident: the row tables below are derived from it in one pass (k_build_rows), ties walk it
                                 // (search_exact_ckpt)
    double *d_cdf = nullptr;     // [T][Z][Zp] canonical f64 CDF rows: built on first need (ensure_full_cdf) -- the car and exact-layout
    bool cdf_full = false;       // kernels and cpm_get_cdf_row read them, the grouped path never does
    uint32_t *d_hi = nullptr;    // [T][Z][RW] row packs: guide + high words of the CDF (cpm_grouped.h)
    double *d_last = nullptr;    // [T][Z] row totals
    double *d_ckpt = nullptr;    // [T][nck][Z] every 32nd value of the running sums (exact fallback of the grouped path)
    long long *d_thr = nullptr;  // [T][Z] Bernoulli thresholds of p_drive (k_build_thr), rebuilt whenever p_drive changes
    int Zq = 0;
    double *d_dm = nullptr;      // [2][T][Z][Z] (reference layout)
    double *d_dist = nullptr;    // [Z][Z]
    double2 *d_tt = nullptr;          // [T][Z][Z] (mean, std) origin-major: what the grouped path's travel kernel gathers from (cpm_grouped.h)
    bool tt_valid = false;
    // the same as sparse rows (cpm_grouped.h, k_tts_*), when the largest row fits LDS: what the travel kernel stages per (origin, hour)
    uint2 *d_tts_words = nullptr;
    uint32_t *d_tts_off = nullptr;
    cpm::TravelCell *d_tts_cells = nullptr;
    size_t tts_cells_cap = 0, tts_lds = 0;
    bool tts_valid = false;
    bool tts_fixed = false;  // the travel rows are those of the compact dataset: fixed stride kDsCap, counts in d_ds_cnt
    // the current datamatrix as compact rows (cpm_dataset.h: ONE sweep of it; valid until it changes), when it is sparse enough
    cpm::DsCell *d_ds_cells = nullptr;  // [T*Z][kDsCap] cells of every (hour, origin), sorted by destination
    uint32_t *d_ds_cnt = nullptr;       // [T*Z] cells per row; then {longest row, a row outgrew its capacity}
    uint32_t *h_ds_stats = nullptr;     // pinned twin of those two words
    bool ds_valid = false;              // the rows describe the current datamatrix (ds_ok: and are usable)
    bool ds_ok = false;
    int ds_max = 0;                     // longest row
    // ... and the p_destin tables built from them (k_ds_pdest): sparse packs in d_hi, what a tie walks in these three (owned by the
    // TABLE: they stay whole when the next dataset's cells arrive)
    double *d_sp = nullptr;             // [T*Z][kDsCap] normalised p of every row's cells
    uint32_t *d_sj = nullptr;           // ... their destinations
    uint32_t *d_scnt = nullptr;         // [T*Z] cells per row
    bool sparse_tables = false;         // the installed p_destin tables are of that kind: d_p holds p_destin only when p_dense_valid
    bool p_dense_valid = false;
    double tab_e_dest = 2.0;            // the exponent they were built with (cpm_refresh_tables)
    int tab_e_int = 1;
    size_t hi_words_alloc = 0;          // words d_hi was allocated for
    int pk_G = 0, pk_Zc = 0;            // geometry of the packs in d_hi (with Zq): guide bits, entries in front of the pad
    double *d_pdrive_mean = nullptr;  // [T][Z] mean_sum of createpdrive (src/createpdrive.jl:10-21): depends on datamatrix and dist only,
    bool pdrive_mean_valid = false;   // so the model-selection sweep (p_min, p_max, e_drive vary) computes it once
    bool have_pdrive = false, have_cdf = false, have_dmat = false, have_dist = false;
    bool have_dm() const { return have_dmat && have_dist; }
    // cars
    int64_t C_total = 0, cpz = 0, n = 0;
    cpm::CarIndex cars{0, 1};     // local car i = global car cars.begin + i * cars.stride
    uint32_t *d_zone0 = nullptr;  // [n] current zones
    uint32_t *d_ztmp = nullptr;   // [n] ping-pong for the IVP
    uint32_t *d_rec = nullptr;    // [T][n]
    int64_t rec_cap = 0;
    bool have_state = false;
    // results
    int64_t *d_counts = nullptr;  // [2*T*Z + 2]
    int64_t *h_counts = nullptr;  // pinned twin: the blocking calls bring the whole tensor (status word included) over in one copy
    int *d_err = nullptr;
    int *h_err = nullptr;         // pinned twin of the validation flag
    // zone-bucket paths
    cpm::ExactWork zx;
    cpm::GroupedWork zg;
    // AUTO moves from the grouped (fixed-stride) layout to the exact one after an overflow that growing the regions could not
    // absorb: the status word of every grouped step is copied to pinned host memory behind the step and looked at, without
    // waiting, when the next step is enqueued (the overflowed step itself is flagged in its own status word).
    bool grouped_overflowed = false;
    long long *h_status = nullptr;      // pinned [2]: status word, largest heavy bucket of the step (GroupedWork::maxn)
    hipEvent_t status_ev = nullptr;
    bool status_pending = false;
    // An IVP enqueued on the grouped layout writes the new state beside the old one; it is committed (or repeated) by
    // finish_ivp() before anything reads or replaces the state or the tables.
    bool ivp_pending = false;
    uint64_t ivp_seed = 0;
    long long *h_ivp_status = nullptr;  // pinned [2], likewise
    // options
    int kernel = CPM_KERNEL_AUTO;
    bool profile = false;
    int prof_what = CPM_PROFILE_SAMPLER;  // CPM_OPT_PROFILE_KERNEL
    int64_t fused_bailouts = 0;           // steps that came back with status bit 2 (a block of a one-launch form gave up waiting)
    int prof_stride = 1;      // bracket every prof_stride-th hourly sampler launch
    int64_t prof_seen = 0;    // sampler launches since profiling was switched on
    bool prof_open = false;
    std::vector<hipEvent_t> ev;  // 2 per hourly launch
    int n_prof = 0;
};

static hipError_t ensure_stream(cpm_ctx *c)
{
    if (c->have_stream) return hipSuccess;
    if (!c->own_stream) {
        hipError_t e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
        if (e != hipSuccess) return e;
    }
    c->stream = c->own_stream;
    c->have_stream = true;
    return hipSuccess;
}


namespace {

int32_t ensure_cars(cpm_ctx *c, int64_t n)
{
    if (n == c->n && c->d_zone0) return CPM_OK;
    dfree(c->d_zone0);
    dfree(c->d_ztmp);
    dfree(c->d_rec);
    c->rec_cap = 0;
    c->n = n;
    if (n > 0) {
        HIP_TRY(hipMalloc(&c->d_zone0, sizeof(uint32_t) * n));
        HIP_TRY(hipMalloc(&c->d_ztmp, sizeof(uint32_t) * n));
    }
    return CPM_OK;
}

int32_t ensure_rec(cpm_ctx *c)
{
    int64_t need = c->n * c->T;
    if (c->rec_cap >= need && c->d_rec) return CPM_OK;
    dfree(c->d_rec);
    HIP_TRY(hipMalloc(&c->d_rec, sizeof(uint32_t) * std::max<int64_t>(need, 1)));
    c->rec_cap = need;
    return CPM_OK;
}

int32_t check_err_flag(cpm_ctx *c, const char *what, int32_t code)
{
    // (into pinned memory: a copy into pageable memory goes through the runtime's staging path -- ~1 ms per call, measured on createpdestin)
    HIP_TRY(hipMemcpyAsync(c->h_err, c->d_err, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const int h = *c->h_err;
    if (h) {
        HIP_TRY(hipMemsetAsync(c->d_err, 0, sizeof(int), c->stream));
        return fail(code, "%s", what);
    }
    return CPM_OK;
}

// the integer Bernoulli thresholds of the p_drive now in d_pdrive (enqueued on the context's stream)
int32_t update_thr(cpm_ctx *c)
{
    const int64_t n = c->Z * c->T;
    if (!c->d_thr) HIP_TRY(hipMalloc(&c->d_thr, sizeof(long long) * static_cast<size_t>(n)));
    hipLaunchKernelGGL(cpm::k_build_thr, dim3(nblk(n, 256)), dim3(256), 0, c->stream, c->d_pdrive, c->d_thr, n);
    HIP_TRY(hipGetLastError());
    return CPM_OK;
}

template <bool CDF, bool PACK>
hipError_t launch_build_rows(cpm_ctx *c)
{
    static bool attr_done[64] = {};  // LDS opt-in (two 64 x 65 f64 tiles), once per device and instantiation
    if (c->device < 0 || c->device >= 64 || !attr_done[c->device]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(cpm::k_build_rows<CDF, PACK>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           static_cast<int>(cpm::kRowLds));
        if (e != hipSuccess) return e;
        if (c->device >= 0 && c->device < 64) attr_done[c->device] = true;
    }
    dim3 grid(nblk(c->Z, cpm::kRowTile), static_cast<unsigned>(c->T));
    hipLaunchKernelGGL((cpm::k_build_rows<CDF, PACK>), grid, dim3(cpm::kRowBlock), cpm::kRowLds, c->stream, c->d_p, CDF ? c->d_cdf : nullptr,
                       PACK ? c->d_hi : nullptr, c->d_last, c->d_ckpt, static_cast<int>(c->Z), c->Zp, cpm::pack_zq(static_cast<int>(c->Z)),
                       cpm::pack_guide_bits(static_cast<int>(c->Z)), c->d_err);
    return hipGetLastError();
}

int32_t ensure_hi(cpm_ctx *c, size_t words)
{
    if (c->d_hi && c->hi_words_alloc >= words) return CPM_OK;
    dfree(c->d_hi);
    c->hi_words_alloc = 0;
    HIP_TRY(hipMalloc(&c->d_hi, sizeof(uint32_t) * std::max<size_t>(words, 1)));
    c->hi_words_alloc = words;
    return CPM_OK;
}

// The table in d_p -> everything the samplers read: row totals, checkpoints and -- when a row pack fits LDS -- the row packs
// of the grouped path, in one pass.  The f64 CDF rows are built with them only when no pack fits (then every kernel searches f64
// rows) or when the caller asks; otherwise on first need (ensure_full_cdf).
int32_t build_rows(cpm_ctx *c, bool with_cdf)
{
    const int64_t rows = c->T * c->Z;
    const bool pack = cpm::pack_row_fits(static_cast<int>(c->Z));
    const bool cdf = with_cdf || !pack;
    c->have_cdf = false;
    c->cdf_full = false;
    if (!c->d_last) HIP_TRY(hipMalloc(&c->d_last, sizeof(double) * static_cast<size_t>(rows)));
    if (!c->d_ckpt) HIP_TRY(hipMalloc(&c->d_ckpt, sizeof(double) * static_cast<size_t>(rows) * cpm::ckpt_count(static_cast<int>(c->Z))));
    c->sparse_tables = false;  // (dense packs of the table in d_p)
    c->p_dense_valid = true;
    c->Zq = cpm::pack_zq(static_cast<int>(c->Z));
    c->pk_G = cpm::pack_guide_bits(static_cast<int>(c->Z));
    c->pk_Zc = static_cast<int>(c->Z);
    if (pack) {
        int32_t rc_hi = ensure_hi(c, static_cast<size_t>(rows) * cpm::pack_row_words(c->Zq, c->pk_G));
        if (rc_hi != CPM_OK) return rc_hi;
    }
    if (cdf && !c->d_cdf) HIP_TRY(hipMalloc(&c->d_cdf, sizeof(double) * static_cast<size_t>(rows) * c->Zp));
    hipError_t e_rows;
    if (cdf && pack) e_rows = launch_build_rows<true, true>(c);
    else if (cdf) e_rows = launch_build_rows<true, false>(c);
    else e_rows = launch_build_rows<false, true>(c);
    HIP_TRY(e_rows);
    int32_t rc = check_err_flag(c, "p_dest holds NaN or negative entries (reference: BoundsError, Appendix A-7)", CPM_ERR_TABLE);
    if (rc != CPM_OK) return rc;
    c->have_cdf = true;
    c->cdf_full = cdf;
    c->zx.tables_dirty = true;
    return CPM_OK;
}

// p_destin in the reference's layout when the installed tables came from a compact dataset (cpm_dataset.h): only for what reads it --
// a caller that wants the array, the kernels that search f64 rows
int32_t ensure_dense_p(cpm_ctx *c)
{
    if (!c->sparse_tables || c->p_dense_valid) return CPM_OK;
    const size_t bytes = sizeof(double) * c->Z * c->Z * c->T;
    if (!c->d_p) HIP_TRY(hipMalloc(&c->d_p, bytes));
    HIP_TRY(hipMemsetAsync(c->d_p, 0, bytes, c->stream));
    hipLaunchKernelGGL(cpm::k_ds_dense_p, dim3(cpm::ds_grid(c->T * c->Z)), dim3(cpm::kDsThreads * cpm::kDsRows), 0, c->stream, c->d_sp, c->d_sj, c->d_scnt,
                       cpm::kDsCap, c->T * c->Z, static_cast<int>(c->Z), c->d_p);
    HIP_TRY(hipGetLastError());
    c->p_dense_valid = true;
    return CPM_OK;
}

// the canonical f64 CDF rows of the installed table, for the kernels that search them (enqueued on the context's stream)
int32_t ensure_full_cdf(cpm_ctx *c)
{
    if (c->cdf_full) return CPM_OK;
    if (!c->have_cdf) return fail(CPM_ERR_STATE, "p_dest not set");
    {
        int32_t rc_p = ensure_dense_p(c);
        if (rc_p != CPM_OK) return rc_p;
    }
    if (!c->d_cdf) HIP_TRY(hipMalloc(&c->d_cdf, sizeof(double) * static_cast<size_t>(c->T * c->Z) * c->Zp));
    double *keep_last = c->d_last, *keep_ckpt = c->d_ckpt;  // (already built: this pass writes the CDF rows only)
    c->d_last = nullptr;
    c->d_ckpt = nullptr;
    hipError_t e = launch_build_rows<true, false>(c);
    c->d_last = keep_last;
    c->d_ckpt = keep_ckpt;
    HIP_TRY(e);
    c->cdf_full = true;
    c->zx.tables_dirty = true;
    return CPM_OK;
}

bool grouped_fits(const cpm_ctx *c, int cap_mult)
{
    return c->d_hi && c->d_last && (c->d_ckpt || c->sparse_tables) && c->d_thr && cpm::grouped_path_fits(c->n, static_cast<int>(c->Z), cap_mult);
}

// AUTO: a zone-bucketed LDS path when a row fits in LDS and there are enough cars per zone to amortise streaming every row once
// per hour (the grouped fixed-stride form while no overflow went unabsorbed in this context, else the exact layout); otherwise
// one thread per car.
int pick_kernel(const cpm_ctx *c)
{
    if (c->kernel != CPM_KERNEL_AUTO) return c->kernel;
    if (c->n >= 32 * c->Z && c->n < (int64_t(1) << 30)) {
        if (!c->grouped_overflowed && grouped_fits(c, c->zg.cap_mult)) return CPM_KERNEL_ZONE_GROUPED;
        if (cpm::exact_path_fits(static_cast<int>(c->Z))) return CPM_KERNEL_ZONE_LDS;
    }
    return CPM_KERNEL_CAR;
}

// After an overflow of the grouped layout: twice the bucket regions (and runs), while the problem still fits; the next
// enqueue re-allocates the workspace.  false: no room to grow -- the caller falls back to the exact layout for good.
bool grow_grouped(cpm_ctx *c)
{
    const int next = c->zg.cap_mult * 2;
    if (next > cpm::kMaxCapMult || !grouped_fits(c, next)) return false;
    c->zg.cap_mult = next;
    c->zg.buckets0_valid = false;
    return true;
}

cpm::GroupedTables grouped_tables(const cpm_ctx *c)
{
    cpm::GroupedTables tb;
    tb.rp = c->d_hi;
    tb.last = c->d_last;
    tb.thr = c->d_thr;
    tb.ckpt = c->d_ckpt;
    tb.p = c->d_p;
    tb.tt = c->d_tt;
    if (c->tts_valid) {
        tb.tts_words = c->d_tts_words;
        tb.tts_off = c->d_tts_off;
        tb.tts_cells = c->d_tts_cells;
        tb.tts_W = static_cast<int>((c->Z + 31) / 32);
        tb.tts_lds = c->tts_lds;
    }
    if (c->tts_valid && c->tts_fixed) {  // (travel rows of a compact dataset: fixed stride, counts beside them)
        tb.tts_cnt = c->d_ds_cnt;
        tb.tts_stride = cpm::kDsCap;
    }
    tb.Z = static_cast<int>(c->Z);
    tb.Zp = c->Zp;
    tb.Zq = c->Zq;
    tb.T = static_cast<int>(c->T);
    tb.G = c->pk_G;
    tb.Zc = c->pk_Zc;
    tb.smap = c->sparse_tables ? 1 : 0;
    if (c->sparse_tables) {
        tb.sp = c->d_sp;
        tb.sj = c->d_sj;
        tb.scnt = c->d_scnt;
        tb.scap = cpm::kDsCap;
    }
    return tb;
}

constexpr int kMaxProf = 8192;

void prof_begin(cpm_ctx *c, int what = CPM_PROFILE_SAMPLER)
{
    if (what != c->prof_what) return;
    c->prof_open = false;
    if (!c->profile || c->n_prof >= kMaxProf) return;
    if ((c->prof_seen++ % c->prof_stride) != 0) return;  // every prof_stride-th launch of the kind
    size_t k = static_cast<size_t>(c->n_prof) * 2;
    while (c->ev.size() < k + 2) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;  // (no pair: this launch is not timed, nothing is counted)
        c->ev.push_back(e);
    }
    cpm::launch_timer() = cpm::LaunchTimer{c->ev[k], c->ev[k + 1]};  // carried by the launch that follows (cpm::launch)
    c->prof_open = true;                                               // only now: the pair exists and the timer is armed
}

void prof_end(cpm_ctx *c, int what = CPM_PROFILE_SAMPLER)
{
    if (what != c->prof_what || !c->prof_open) return;
    c->prof_open = false;
    const bool taken = cpm::launch_timer().start == nullptr;  // cpm::launch clears the timer when it hands the pair to a dispatch
    cpm::launch_timer() = cpm::LaunchTimer{};
    if (taken) c->n_prof++;  // a pair no launch stamped is reused by the next timed launch, never read
}

// one hourly step of the one-thread-per-car kernel
int32_t launch_step_car(cpm_ctx *c, const uint32_t *zin, uint32_t *out, int t, uint32_t step, uint64_t seed,
                        bool travel, unsigned long long *tt_sum)
{
    const double *pd = c->d_pdrive + static_cast<size_t>(t) * c->Z;
    const double *cdf = c->d_cdf + static_cast<size_t>(t) * c->Z * c->Zp;
    dim3 grid(nblk(c->n, 256)), block(256);
    if (travel)
        cpm::launch(cpm::k_step_car<true>, grid, block, 0, c->stream, zin, out, pd, cdf, static_cast<int>(c->Z),
                           c->Zp, c->n, c->cars, step, seed, c->d_dm, static_cast<int>(c->T), t, tt_sum);
    else
        cpm::launch(cpm::k_step_car<false>, grid, block, 0, c->stream, zin, out, pd, cdf, static_cast<int>(c->Z),
                           c->Zp, c->n, c->cars, step, seed, nullptr, static_cast<int>(c->T), t, nullptr);
    HIP_TRY(hipGetLastError());
    return CPM_OK;
}

int32_t launch_histogram(cpm_ctx *c, int64_t *d_counts)
{
    unsigned long long *parking = reinterpret_cast<unsigned long long *>(d_counts);
    unsigned long long *driving = parking + c->T * c->Z;
    size_t lds = sizeof(uint32_t) * 2 * c->Z;
    if (lds <= 160 * 1024) {
        if (lds > 48 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(cpm::k_histogram),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
        // enough chunks to fill the chip, few enough that the flush (2*Z atomics per block) stays small
        int64_t chunks = std::max<int64_t>(1, std::min<int64_t>((c->n + 16383) / 16384, 2 * c->cu_count / std::max<int64_t>(c->T / 4, 1) + 8));
        int64_t chunk = (c->n + chunks - 1) / chunks;
        dim3 grid(static_cast<unsigned>(chunks), static_cast<unsigned>(c->T));
        hipLaunchKernelGGL(cpm::k_histogram, grid, dim3(1024), lds, c->stream, c->d_zone0, c->d_rec, c->n,
                           static_cast<int>(c->Z), parking, driving, chunk);
    } else {
        dim3 grid(nblk(c->n, 256), static_cast<unsigned>(c->T));
        hipLaunchKernelGGL(cpm::k_histogram_global, grid, dim3(256), 0, c->stream, c->d_zone0, c->d_rec, c->n,
                           static_cast<int>(c->Z), parking, driving);
    }
    HIP_TRY(hipGetLastError());
    return CPM_OK;
}

int32_t finish_ivp(cpm_ctx *c);

// The current datamatrix as compact rows (cpm_dataset.h), once per datamatrix: ONE sweep of it (k_ds_cells), then the rows sorted and
// the travel rows written from them (k_ds_sort).  ds_ok: every row fits its capacity and the sparse pack of the longest row is at most
// 60 % of the dense one -- else the dense builders of cpm_tables.h / cpm_grouped.h take the dataset, as they take every T != 24.
int32_t ensure_dataset(cpm_ctx *c)
{
    if (c->ds_valid) return CPM_OK;
    c->ds_ok = false;
    if (!cpm::ds_fits(c->Z, c->T) || !cpm::pack_row_fits(static_cast<int>(c->Z))) {
        c->ds_valid = true;
        return CPM_OK;
    }
    const int64_t rows = c->T * c->Z;
    const int W = static_cast<int>((c->Z + 31) / 32);
    if (!c->d_ds_cells) HIP_TRY(hipMalloc(&c->d_ds_cells, sizeof(cpm::DsCell) * static_cast<size_t>(rows) * cpm::kDsCap));
    if (!c->d_ds_cnt) HIP_TRY(hipMalloc(&c->d_ds_cnt, sizeof(uint32_t) * static_cast<size_t>(rows + 2)));
    if (!c->h_ds_stats) HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c->h_ds_stats), 2 * sizeof(uint32_t)));
    HIP_TRY(hipMemsetAsync(c->d_ds_cnt, 0, sizeof(uint32_t) * static_cast<size_t>(rows + 2), c->stream));
    hipLaunchKernelGGL(cpm::k_ds_cells, dim3(nblk(c->Z, 64), nblk(c->Z, cpm::kDsJB)), dim3(256), 0, c->stream, c->d_dm, c->d_ds_cells, c->d_ds_cnt,
                       static_cast<int>(c->Z), cpm::kDsCap, c->d_ds_cnt + rows);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(c->h_ds_stats, c->d_ds_cnt + rows, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->ds_max = static_cast<int>(std::max<uint32_t>(c->h_ds_stats[0], 1u));
    c->ds_valid = true;
    const int zq_c = cpm::pack_zq(c->ds_max), g_c = cpm::pack_guide_bits(c->ds_max);
    const int dense_words = cpm::pack_row_words(cpm::pack_zq(static_cast<int>(c->Z)), cpm::pack_guide_bits(static_cast<int>(c->Z)));
    if (c->h_ds_stats[1] != 0 || 10 * cpm::pack_row_words(zq_c, g_c, 1) > 6 * dense_words) return CPM_OK;  // dense: the builders of cpm_tables.h
    if (!c->d_tts_words) HIP_TRY(hipMalloc(&c->d_tts_words, sizeof(uint2) * static_cast<size_t>(rows) * W));
    if (c->tts_cells_cap < static_cast<size_t>(rows) * cpm::kDsCap) {
        dfree(c->d_tts_cells);
        c->tts_cells_cap = 0;
        HIP_TRY(hipMalloc(&c->d_tts_cells, sizeof(cpm::TravelCell) * static_cast<size_t>(rows) * cpm::kDsCap));
        c->tts_cells_cap = static_cast<size_t>(rows) * cpm::kDsCap;
    }
    const size_t lds_sort = sizeof(uint32_t) * 2 * W * cpm::kDsRows;
    hipLaunchKernelGGL(cpm::k_ds_sort, dim3(cpm::ds_grid(rows)), dim3(cpm::kDsThreads * cpm::kDsRows), lds_sort, c->stream, c->d_ds_cells, c->d_ds_cnt, cpm::kDsCap,
                       rows, static_cast<int>(c->Z), c->d_tts_words, W, c->d_tts_cells);
    HIP_TRY(hipGetLastError());
    c->ds_ok = true;
    // the travel rows of this datamatrix are those just written: bitmap words + the longest row's cells must fit the travel kernel's LDS
    const size_t lds = ((static_cast<size_t>(W) * sizeof(uint2) + 15) & ~static_cast<size_t>(15)) + static_cast<size_t>(c->ds_max) * sizeof(cpm::TravelCell);
    if (lds <= 32 * 1024) {
        c->tts_lds = lds;
        c->tts_valid = true;
        c->tts_fixed = true;
        c->tt_valid = true;
        dfree(c->d_tt);  // (a dense table left over from a datamatrix whose rows did not fit)
    }
    return CPM_OK;
}

// createpdestin on the compact rows (k_ds_pdest): sparse packs, row totals and what a tie walks; false when this dataset / exponent
// takes the dense builders (x^0 = 1 also where x = 0: every cell of the dense table then holds weight)
int32_t build_p_dest_sparse(cpm_ctx *c, double e_dest, int32_t e_is_integer, bool *done)
{
    *done = false;
    if (!(e_dest > 0.0)) return CPM_OK;
    int32_t rc = ensure_dataset(c);
    if (rc != CPM_OK || !c->ds_ok) return rc;
    const int64_t rows = c->T * c->Z;
    const int nc = c->ds_max, zq_c = cpm::pack_zq(nc), g_c = cpm::pack_guide_bits(nc);
    if (!c->d_sp) HIP_TRY(hipMalloc(&c->d_sp, sizeof(double) * static_cast<size_t>(rows) * cpm::kDsCap));
    if (!c->d_sj) HIP_TRY(hipMalloc(&c->d_sj, sizeof(uint32_t) * static_cast<size_t>(rows) * cpm::kDsCap));
    if (!c->d_scnt) HIP_TRY(hipMalloc(&c->d_scnt, sizeof(uint32_t) * static_cast<size_t>(rows)));
    if (!c->d_last) HIP_TRY(hipMalloc(&c->d_last, sizeof(double) * static_cast<size_t>(rows)));
    rc = ensure_hi(c, static_cast<size_t>(rows) * cpm::pack_row_words(zq_c, g_c, 1));
    if (rc != CPM_OK) return rc;
    c->have_cdf = false;
    c->cdf_full = false;
    hipLaunchKernelGGL(cpm::k_ds_pdest, dim3(cpm::ds_grid(rows)), dim3(cpm::kDsThreads * cpm::kDsRows), 0, c->stream, c->d_ds_cells, c->d_ds_cnt, cpm::kDsCap,
                       rows, static_cast<int>(c->Z), e_dest, e_is_integer, nc, zq_c, g_c, c->d_hi, c->d_last, c->d_sp, c->d_sj, c->d_scnt, c->d_err);
    HIP_TRY(hipGetLastError());
    c->sparse_tables = true;
    c->p_dense_valid = false;
    c->Zq = zq_c;
    c->pk_G = g_c;
    c->pk_Zc = nc;
    c->tab_e_dest = e_dest;
    c->tab_e_int = e_is_integer;
    rc = check_err_flag(c, "p_dest holds NaN or negative entries (reference: BoundsError, Appendix A-7)", CPM_ERR_TABLE);
    if (rc != CPM_OK) return rc;
    c->have_cdf = true;
    c->zx.tables_dirty = true;
    *done = true;
    return CPM_OK;
}

// The travel table as sparse rows (once per datamatrix, straight from it): kept when the largest row -- its bitmap words
// and its non-zero cells -- fits 32 KB of LDS; the travel kernel then stages an origin's row instead of gathering cells from HBM.
int32_t build_sparse_travel_rows(cpm_ctx *c)
{
    c->tts_valid = false;
    c->tts_fixed = false;
    const int64_t rows = c->T * c->Z;
    const int W = static_cast<int>((c->Z + 31) / 32);
    if (!c->d_tts_words) HIP_TRY(hipMalloc(&c->d_tts_words, sizeof(uint2) * static_cast<size_t>(rows) * W));
    if (!c->d_tts_off) HIP_TRY(hipMalloc(&c->d_tts_off, sizeof(uint32_t) * static_cast<size_t>(rows + 1)));
    uint32_t *d_count = nullptr;
    HIP_TRY(hipMalloc(&d_count, sizeof(uint32_t) * static_cast<size_t>(rows) + 16));
    struct Scratch {
        uint32_t *&p;
        ~Scratch() { dfree(p); }
    } scratch{d_count};
    uint32_t *d_max = d_count + rows;                                            // (+ total: 8 B behind it, 8-aligned or not: read as bytes)
    unsigned long long *d_total = nullptr;
    HIP_TRY(hipMalloc(&d_total, sizeof(unsigned long long)));
    struct Scratch2 {
        unsigned long long *&p;
        ~Scratch2() { dfree(p); }
    } scratch2{d_total};
    const dim3 tgrid(nblk(c->Z, 64), static_cast<unsigned>(c->T), cpm::kTtsSplit);
    hipLaunchKernelGGL(cpm::k_tts_bits, tgrid, dim3(64), 0, c->stream, c->d_dm, c->d_tts_words, static_cast<int>(c->Z), static_cast<int>(c->T), W);
    hipLaunchKernelGGL(cpm::k_tts_prefix, dim3(nblk(rows, 256)), dim3(256), 0, c->stream, c->d_tts_words, d_count, rows, W);
    hipLaunchKernelGGL(cpm::k_tts_offsets, dim3(1), dim3(1024), 0, c->stream, d_count, c->d_tts_off, rows, d_max, d_total);
    HIP_TRY(hipGetLastError());
    uint32_t h_max = 0;
    unsigned long long h_total = 0;
    HIP_TRY(hipMemcpyAsync(&h_max, d_max, sizeof h_max, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(&h_total, d_total, sizeof h_total, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const size_t lds = ((static_cast<size_t>(W) * sizeof(uint2) + 15) & ~static_cast<size_t>(15)) + static_cast<size_t>(h_max) * sizeof(cpm::TravelCell);
    if (lds > 32 * 1024 || h_total >= (1ull << 32)) return CPM_OK;  // dense rows: the travel kernel gathers from the dense table
    if (h_total > c->tts_cells_cap) {
        dfree(c->d_tts_cells);
        c->tts_cells_cap = 0;
        HIP_TRY(hipMalloc(&c->d_tts_cells, sizeof(cpm::TravelCell) * std::max<size_t>(h_total, 1)));
        c->tts_cells_cap = h_total;
    }
    hipLaunchKernelGGL(cpm::k_tts_cells, tgrid, dim3(64), 0, c->stream, c->d_dm, c->d_tts_words, c->d_tts_off, c->d_tts_cells,
                       static_cast<int>(c->Z), static_cast<int>(c->T), W);
    HIP_TRY(hipGetLastError());
    c->tts_lds = lds;
    c->tts_valid = true;
    dfree(c->d_tt);  // (a dense table left over from a datamatrix whose rows did not fit)
    return CPM_OK;
}

// What a non-zero status word of a grouped step asks of the context.  Bit 2 (4): a placing block of the fused hour gave up waiting
// for its sampler workgroups (a dispatch order the hand-off did not expect): two launches per hour from now on.  Bit 1 (2): a bucket
// region or a run overflowed: twice the regions while the problem still fits.  true: the step can be repeated on the grouped path.
bool absorb_status(cpm_ctx *c, long long st)
{
    const bool bailed = (st & 4) != 0;
    if (bailed && c->zg.fused_ok) {
        c->zg.fused_ok = false;
        ++c->fused_bailouts;  // (CPM_INFO_FUSED_BAILOUTS: the context has left the one-launch forms for good)
    }
    // (a step that bailed out AND overflowed: the two-launch path is tried on the regions as they are when they cannot grow)
    const bool grown = (st & ~4ll) != 0 && grow_grouped(c);
    return bailed || grown;
}

int32_t resample_enqueue(cpm_ctx *c, uint64_t seed, uint32_t flags, int64_t *d_counts)
{
    if (!c->have_pdrive || !c->have_cdf) return fail(CPM_ERR_STATE, "resample: p_drive / p_dest not set");
    if (!c->have_state) return fail(CPM_ERR_STATE, "resample: no car state (cpm_init_states / cpm_set_state)");
    {
        int32_t rc_ivp = finish_ivp(c);
        if (rc_ivp != CPM_OK) return rc_ivp;
    }
    bool travel = (flags & CPM_FLAG_TRAVEL) != 0;
    if (travel && !c->have_dm()) return fail(CPM_ERR_STATE, "CPM_FLAG_TRAVEL needs cpm_set_datamatrix");
    size_t nwords = static_cast<size_t>(2 * c->T * c->Z + 2);
    if (c->status_pending && hipEventQuery(c->status_ev) == hipSuccess) {
        c->status_pending = false;
        if (c->h_status[0] != 0 && !absorb_status(c, c->h_status[0])) c->grouped_overflowed = true;
        c->zg.set_parts(static_cast<uint32_t>(c->h_status[1]), static_cast<uint32_t>(c->h_status[1] >> 32));
    }
    const int kernel = pick_kernel(c);
    if (c->n == 0 || kernel != CPM_KERNEL_ZONE_GROUPED)  // (the grouped path zeroes the count tensor with its other counters, in one launch)
        HIP_TRY(hipMemsetAsync(d_counts, 0, sizeof(int64_t) * nwords, c->stream));
    if (c->n == 0) return CPM_OK;
    unsigned long long *tt_sum = reinterpret_cast<unsigned long long *>(d_counts) + 2 * c->T * c->Z;
    if (kernel == CPM_KERNEL_ZONE_GROUPED) {
        if (!grouped_fits(c, c->zg.cap_mult))
            return fail(CPM_ERR_ARG, "CPM_KERNEL_ZONE_GROUPED does not fit this problem (use CPM_KERNEL_ZONE_LDS or CPM_KERNEL_CAR)");
        if (travel && !c->tt_valid) {  // the travel rows of the current datamatrix, once: with its compact rows (cpm_dataset.h) ...
            int32_t rc_ds = ensure_dataset(c);
            if (rc_ds != CPM_OK) return rc_ds;
        }
        if (travel && !c->tt_valid) {  // ... or, datasets those do not take: sparse rows for LDS, or -- rows too large -- the dense table
            int32_t rc_tts = build_sparse_travel_rows(c);
            if (rc_tts != CPM_OK) return rc_tts;
            if (!c->tts_valid) {
                const size_t cells = static_cast<size_t>(c->Z) * c->Z * c->T;
                if (!c->d_tt) HIP_TRY(hipMalloc(&c->d_tt, sizeof(double2) * cells));
                hipLaunchKernelGGL(cpm::k_build_travel_table, dim3(nblk(c->Z, cpm::kTtTile), nblk(c->Z, cpm::kTtTile), static_cast<unsigned>(c->T)),
                                   dim3(cpm::kTtTile * 8), 0, c->stream, c->d_dm, c->d_tt, static_cast<int>(c->Z), static_cast<int>(c->T));
                HIP_TRY(hipGetLastError());
            }
            c->tt_valid = true;
        }
        int32_t rc = cpm::grouped_run(c->zg, c->stream, grouped_tables(c), c->n, c->cars, c->d_zone0, seed, travel, d_counts, c->cu_count,
                                      [&](int what) { prof_begin(c, what); }, [&](int what) { prof_end(c, what); }, g_last_error);
        if (rc == CPM_OK && c->h_status && !c->status_pending) {
            c->h_status[1] = 0;
            if (hipMemcpyAsync(c->h_status, d_counts + nwords - 1, sizeof(long long), hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
                hipMemcpyAsync(c->h_status + 1, c->zg.maxn, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
                hipEventRecord(c->status_ev, c->stream) == hipSuccess)
                c->status_pending = true;
        }
        return rc;
    }
    {   // these kernels search the f64 CDF rows
        int32_t rc_cdf = ensure_full_cdf(c);
        if (rc_cdf != CPM_OK) return rc_cdf;
    }
    if (kernel == CPM_KERNEL_ZONE_LDS) {
        return cpm::exact_run(c->zx, c->stream, c->d_pdrive, c->d_cdf, static_cast<int>(c->Z), c->Zp, static_cast<int>(c->T), c->n, c->cars,
                              c->d_zone0, seed, travel, c->d_dm, d_counts, c->cu_count, [&](int what) { prof_begin(c, what); }, [&](int what) { prof_end(c, what); },
                              g_last_error);
    }
    int32_t rc = ensure_rec(c);
    if (rc != CPM_OK) return rc;
    for (int t = 0; t < c->T; ++t) {
        const uint32_t *zin = (t == 0) ? c->d_zone0 : c->d_rec + static_cast<size_t>(t - 1) * c->n;
        uint32_t *out = c->d_rec + static_cast<size_t>(t) * c->n;
        prof_begin(c);
        rc = launch_step_car(c, zin, out, t, static_cast<uint32_t>(c->T - 1 + t), seed, travel, tt_sum);
        prof_end(c);
        if (rc != CPM_OK) return rc;
    }
    return launch_histogram(c, d_counts);
}

// the IVP on a layout that cannot overflow: exact buckets when the row fits LDS, one thread per car otherwise
int32_t ivp_exact(cpm_ctx *c, uint64_t seed)
{
    {
        int32_t rc_cdf = ensure_full_cdf(c);
        if (rc_cdf != CPM_OK) return rc_cdf;
    }
    if (pick_kernel(c) != CPM_KERNEL_CAR && cpm::exact_path_fits(static_cast<int>(c->Z))) {
        return cpm::exact_run(c->zx, c->stream, c->d_pdrive, c->d_cdf, static_cast<int>(c->Z), c->Zp, static_cast<int>(c->T), c->n, c->cars,
                              c->d_zone0, seed, false, nullptr, c->d_counts, c->cu_count, [](int) {}, [](int) {}, g_last_error, true, c->d_zone0);
    }
    // src/solveinitialvalueproblem.jl:8 : t = 1:(T-1), state update unconditional (:53)
    for (int t = 0; t < c->T - 1; ++t) {
        int32_t rc = launch_step_car(c, c->d_zone0, c->d_ztmp, t, static_cast<uint32_t>(t), seed, false, nullptr);
        if (rc != CPM_OK) return rc;
        std::swap(c->d_zone0, c->d_ztmp);  // flag bit is masked off by every reader
    }
    return CPM_OK;
}

// The IVP on the grouped layout: new state beside the old one (d_ztmp), status word copied to pinned memory behind it.
int32_t ivp_grouped(cpm_ctx *c, uint64_t seed)
{
    c->zx.buckets0_valid = false;  // the current state's grouped buckets may be cached (zg); everything else is stale once the IVP is committed
    int32_t rc = cpm::grouped_run(c->zg, c->stream, grouped_tables(c), c->n, c->cars, c->d_zone0, seed, false, c->d_counts, c->cu_count,
                                  [](int) {}, [](int) {}, g_last_error, true, c->d_ztmp);
    if (rc != CPM_OK) return rc;
    c->h_ivp_status[1] = 0;
    HIP_TRY(hipMemcpyAsync(c->h_ivp_status, c->d_counts + 2 * c->T * c->Z + 1, sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(c->h_ivp_status + 1, c->zg.maxn, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    return CPM_OK;
}

// Commit (or repeat) an IVP that was enqueued on the grouped layout.  Called before anything reads or replaces the car
// state or the tables.  Blocks until the IVP has drained.
int32_t finish_ivp(cpm_ctx *c)
{
    if (!c->ivp_pending) return CPM_OK;
    c->ivp_pending = false;
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->zg.set_parts(static_cast<uint32_t>(c->h_ivp_status[1]), static_cast<uint32_t>(c->h_ivp_status[1] >> 32));
    if (c->h_ivp_status[0] == 0) {
        std::swap(c->d_zone0, c->d_ztmp);
        HIP_TRY(cpm::grouped_commit_ivp(c->zg, c->stream));
        return CPM_OK;
    }
    // A bucket or a run outgrew its region: d_zone0 is untouched.  Run the IVP again with twice the regions while the problem
    // still fits, else on the exact layout (and stay there).
    while (pick_kernel(c) == CPM_KERNEL_ZONE_GROUPED && absorb_status(c, c->h_ivp_status[0])) {
        int32_t rc = ivp_grouped(c, c->ivp_seed);
        if (rc != CPM_OK) return rc;
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->zg.set_parts(static_cast<uint32_t>(c->h_ivp_status[1]), static_cast<uint32_t>(c->h_ivp_status[1] >> 32));
        if (c->h_ivp_status[0] == 0) {
            std::swap(c->d_zone0, c->d_ztmp);
            HIP_TRY(cpm::grouped_commit_ivp(c->zg, c->stream));
            return CPM_OK;
        }
    }
    if (c->kernel == CPM_KERNEL_AUTO) c->grouped_overflowed = true;
    c->zx.buckets0_valid = false;
    c->zg.buckets0_valid = false;
    return ivp_exact(c, c->ivp_seed);
}

int32_t ivp_enqueue(cpm_ctx *c, uint64_t seed)
{
    if (!c->have_pdrive || !c->have_cdf) return fail(CPM_ERR_STATE, "solve_ivp: p_drive / p_dest not set");
    if (!c->have_state) return fail(CPM_ERR_STATE, "solve_ivp: no car state");
    int32_t rc = finish_ivp(c);
    if (rc != CPM_OK) return rc;
    if (c->n == 0) return CPM_OK;
    if (pick_kernel(c) == CPM_KERNEL_ZONE_GROUPED && grouped_fits(c, c->zg.cap_mult)) {
        rc = ivp_grouped(c, seed);
        if (rc != CPM_OK) return rc;
        c->ivp_pending = true;
        c->ivp_seed = seed;
        return CPM_OK;
    }
    c->zx.buckets0_valid = false;
    c->zg.buckets0_valid = false;
    return ivp_exact(c, seed);
}

}  // namespace

extern "C" {

const char *cpm_last_error(void) { return g_last_error.c_str(); }

int32_t cpm_version(void) { return 100; }

int32_t cpm_device_count(int32_t *n_out)
{
    if (!n_out) return fail(CPM_ERR_ARG, "null n_out");
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    *n_out = n;
    return CPM_OK;
}

int32_t cpm_device_info(int32_t device_id, char *name, int32_t len, int32_t *cu_count, int64_t *hbm_bytes)
{
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, device_id));
    if (name && len > 0) {
        std::snprintf(name, static_cast<size_t>(len), "%s (%s)", p.name, p.gcnArchName);
    }
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = static_cast<int64_t>(p.totalGlobalMem);
    return CPM_OK;
}

int32_t cpm_create(cpm_ctx **ctx_out, int64_t Z, int64_t T, int32_t device_id)
{
    if (!ctx_out) return fail(CPM_ERR_ARG, "null ctx_out");
    *ctx_out = nullptr;
    if (Z < 1 || Z > (int64_t(1) << 24)) return fail(CPM_ERR_ARG, "number_zones %lld out of range", (long long)Z);
    if (T < 1 || T > 4096) return fail(CPM_ERR_ARG, "T %lld out of range", (long long)T);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev < 1) return fail(CPM_ERR_HIP, "no HIP device: this library has no CPU path");
    if (device_id < 0 || device_id >= ndev) return fail(CPM_ERR_ARG, "device %d of %d", device_id, ndev);
    HIP_TRY(hipSetDevice(device_id));
    cpm_ctx *c = new (std::nothrow) cpm_ctx();
    if (!c) return fail(CPM_ERR_NOMEM, "host allocation failed");
    c->Z = Z;
    c->T = T;
    c->Zp = static_cast<int>((Z + 15) / 16 * 16);
    c->Zq = cpm::pack_zq(static_cast<int>(Z));
    c->pk_G = cpm::pack_guide_bits(static_cast<int>(Z));
    c->pk_Zc = static_cast<int>(Z);
    c->device = device_id;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device_id) == hipSuccess) c->cu_count = p.multiProcessorCount;
    if (c->cu_count <= 0) c->cu_count = 256;
    hipError_t e = hipMalloc(&c->d_err, sizeof(int));
    if (e == hipSuccess) e = hipMemset(c->d_err, 0, sizeof(int));
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&c->h_err), sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&c->d_counts, sizeof(int64_t) * static_cast<size_t>(2 * T * Z + 2));
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&c->h_counts), sizeof(int64_t) * static_cast<size_t>(2 * T * Z + 2));
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&c->h_status), 2 * sizeof(long long));
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->status_ev, hipEventDisableTiming);
    if (e == hipSuccess) c->h_status[0] = c->h_status[1] = 0;
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&c->h_ivp_status), 2 * sizeof(long long));
    if (e == hipSuccess) c->h_ivp_status[0] = c->h_ivp_status[1] = 0;
    if (e != hipSuccess) {
        cpm_destroy(c);
        return fail(CPM_ERR_HIP, "context setup: %s", hipGetErrorString(e));
    }
#ifdef CPM_DEV_ENV  // development builds only (tools/build_variants.sh ... "-DCPM_DEV_ENV"): the product reads nothing from the environment
    if (const char *v = std::getenv("CPM_FUSED")) cpm_set_option(c, CPM_OPT_FUSED, std::atoi(v));
    if (const char *v = std::getenv("CPM_HEAVY_X")) c->zg.heavy_x_seen = static_cast<uint32_t>(std::max(1, std::min(16, std::atoi(v))));
    if (const char *v = std::getenv("CPM_FUSED_LAG")) cpm_set_option(c, CPM_OPT_FUSED_LAG, std::atoi(v));
#endif
    *ctx_out = c;
    return CPM_OK;
}

int32_t cpm_destroy(cpm_ctx *c)
{
    if (!c) return CPM_OK;
    (void)hipSetDevice(c->device);
    if (c->have_stream) (void)hipStreamSynchronize(c->stream);
    dfree(c->d_pdrive);
    dfree(c->d_cdf);
    dfree(c->d_p);
    dfree(c->d_ckpt);
    dfree(c->d_hi);
    dfree(c->d_last);
    dfree(c->d_thr);
    dfree(c->d_dm);
    dfree(c->d_dist);
    dfree(c->d_pdrive_mean);
    dfree(c->d_ds_cells);
    dfree(c->d_ds_cnt);
    dfree(c->d_sp);
    dfree(c->d_sj);
    dfree(c->d_scnt);
    if (c->h_ds_stats) (void)hipHostFree(c->h_ds_stats);
    dfree(c->d_tt);
    dfree(c->d_tts_words);
    dfree(c->d_tts_off);
    dfree(c->d_tts_cells);
    dfree(c->d_zone0);
    dfree(c->d_ztmp);
    dfree(c->d_rec);
    dfree(c->d_counts);
    dfree(c->d_err);
    c->zx.release();
    c->zg.release();
    if (c->h_err) (void)hipHostFree(c->h_err);
    if (c->h_status) (void)hipHostFree(c->h_status);
    if (c->h_counts) (void)hipHostFree(c->h_counts);
    if (c->h_ivp_status) (void)hipHostFree(c->h_ivp_status);
    if (c->status_ev) (void)hipEventDestroy(c->status_ev);
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return CPM_OK;
}

int32_t cpm_set_option(cpm_ctx *c, int32_t option, int64_t value)
{
    if (!c) return fail(CPM_ERR_ARG, "null context");  // (host state only: no stream is needed, none is created)
    switch (option) {
    case CPM_OPT_KERNEL:
        if (value != CPM_KERNEL_AUTO && value != CPM_KERNEL_CAR && value != CPM_KERNEL_ZONE_LDS && value != CPM_KERNEL_ZONE_GROUPED) return fail(CPM_ERR_ARG, "unknown kernel %lld", (long long)value);
        c->kernel = static_cast<int>(value);
        return CPM_OK;
    case CPM_OPT_FUSED:
        if (value < 0 || value > 8) return fail(CPM_ERR_ARG, "fused hour %lld", (long long)value);
        c->zg.fused_ok = value != 0;
        c->zg.fused_auto = value == 5;                                            // 5: where it pays
        c->zg.fused_pf = value == 3 || value == 4;                                // 3, 4: placing first (k_grouped_hour_pf)
        c->zg.fused_day = value >= 6;                                             // 6 .. 9: all hours of a run in one launch (k_grouped_day)
        c->zg.day_mix = value == 8 ? 0 : 1;                                       // 8: its placing blocks in front of the sampler workgroups instead of among them
        c->zg.fused_spin = (value == 2 || value == 4 || value == 7) ? 0u : cpm::kFusedSpinLimit;  // 2, 4, 7: the waiting side gives up at once (tests of the bail-out)
        return CPM_OK;
    case CPM_OPT_FUSED_LAG:
        if (value < 1 || value > (1 << 20)) return fail(CPM_ERR_ARG, "fused lag %lld", (long long)value);
        c->zg.fused_lag = static_cast<int>(value);
        return CPM_OK;
    case CPM_OPT_ZONE_ORDER:
        if (value < 0 || value > 2) return fail(CPM_ERR_ARG, "zone order %lld (0 zone order, 1 largest-first, 2 largest-first on sparse row packs)", (long long)value);
        c->zg.use_perm = static_cast<int>(value);
        return CPM_OK;
    case CPM_OPT_PROFILE_KERNEL:
        if (value < CPM_PROFILE_SAMPLER || value > CPM_PROFILE_TRAVEL) return fail(CPM_ERR_ARG, "profile kernel %lld", (long long)value);
        c->prof_what = static_cast<int>(value);
        return CPM_OK;
    case CPM_OPT_PROFILE:
        c->profile = value != 0;
        c->prof_stride = value > 1 ? static_cast<int>(value) : 1;
        c->prof_seen = 0;
        c->n_prof = 0;  // (re)start the record; hourly launches append until read or reset
        return CPM_OK;
    default:
        return fail(CPM_ERR_ARG, "unknown option %d", option);
    }
}

int32_t cpm_get_info(cpm_ctx *c, int32_t what, int64_t *value_out)
{
    if (!c || !value_out) return fail(CPM_ERR_ARG, "null argument");
    switch (what) {
    case CPM_INFO_KERNEL:
        *value_out = pick_kernel(c);
        return CPM_OK;
    case CPM_INFO_CAP_MULT:
        *value_out = c->zg.cap_mult;
        return CPM_OK;
    case CPM_INFO_PARTS:
        *value_out = c->zg.parts;
        return CPM_OK;
    case CPM_INFO_FUSED:
        *value_out = (pick_kernel(c) == CPM_KERNEL_ZONE_GROUPED && c->zg.fused_ok && c->zg.parts <= 1 &&
                      cpm::fused_shape_ok(static_cast<int>(c->Z), c->Zq, c->pk_G, c->sparse_tables) &&
                      (!c->zg.fused_auto || cpm::fused_pays(static_cast<int>(c->Z), c->Zq, c->pk_G, c->cu_count, c->sparse_tables, (c->n + c->Z - 1) / std::max<int64_t>(c->Z, 1))))
                         ? (c->zg.fused_day ? 6 : (c->zg.fused_pf ? 3 : 1))
                         : 0;
        return CPM_OK;
    case CPM_INFO_SPARSE_TABLES:
        *value_out = c->sparse_tables ? cpm::pack_row_words(c->Zq, c->pk_G, 1) : 0;
        return CPM_OK;
    case CPM_INFO_FUSED_BAILOUTS:
        *value_out = c->fused_bailouts;
        return CPM_OK;
    default:
        return fail(CPM_ERR_ARG, "unknown info %d", what);
    }
}

int32_t cpm_set_stream(cpm_ctx *c, void *hip_stream)
{
    if (!c) return fail(CPM_ERR_ARG, "null context");
    HIP_TRY(hipSetDevice(c->device));
    if (c->have_stream) {  // whatever was enqueued so far is finished on the stream it was enqueued on
        int32_t rc_ivp = finish_ivp(c);
        if (rc_ivp != CPM_OK) return rc_ivp;
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    c->stream = static_cast<hipStream_t>(hip_stream);
    c->have_stream = hip_stream != nullptr;  // NULL: the context's own stream, created when it is first needed
    return CPM_OK;
}

int32_t cpm_sync(cpm_ctx *c)
{
    CTX_TRY(c);
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CPM_OK;
}

// ------------------------------------------------------------------ tables

int32_t cpm_set_p_drive(cpm_ctx *c, const double *p_drive)
{
    CTX_TRY(c);
    {   // a pending asynchronous IVP must be committed (or, after an overflow, repeated) on the tables it was enqueued with
        int32_t rc_ivp = finish_ivp(c);
        if (rc_ivp != CPM_OK) return rc_ivp;
    }
    if (!p_drive) return fail(CPM_ERR_ARG, "null p_drive");
    size_t bytes = sizeof(double) * c->Z * c->T;
    if (!c->d_pdrive) HIP_TRY(hipMalloc(&c->d_pdrive, bytes));
    HIP_TRY(hipMemcpyAsync(c->d_pdrive, p_drive, bytes, hipMemcpyHostToDevice, c->stream));
    {
        int32_t rc_thr = update_thr(c);
        if (rc_thr != CPM_OK) return rc_thr;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_pdrive = true;
    return CPM_OK;
}

int32_t cpm_set_p_dest(cpm_ctx *c, const double *p_dest)
{
    CTX_TRY(c);
    {   // a pending asynchronous IVP must be committed (or, after an overflow, repeated) on the tables it was enqueued with
        int32_t rc_ivp = finish_ivp(c);
        if (rc_ivp != CPM_OK) return rc_ivp;
    }
    if (!p_dest) return fail(CPM_ERR_ARG, "null p_dest");
    size_t bytes = sizeof(double) * c->Z * c->Z * c->T;
    if (!c->d_p) HIP_TRY(hipMalloc(&c->d_p, bytes));
    c->have_cdf = false;
    HIP_TRY(hipMemcpyAsync(c->d_p, p_dest, bytes, hipMemcpyHostToDevice, c->stream));
    return build_rows(c, false);  // (synchronises: the validation flag is read back)
}

int32_t cpm_set_datamatrix(cpm_ctx *c, const double *datamatrix, const double *dist)
{
    CTX_TRY(c);
    {   // a pending asynchronous IVP must be committed (or, after an overflow, repeated) on the tables it was enqueued with
        int32_t rc_ivp = finish_ivp(c);
        if (rc_ivp != CPM_OK) return rc_ivp;
    }
    if (!datamatrix) return fail(CPM_ERR_ARG, "null datamatrix");
    size_t bytes = sizeof(double) * c->Z * c->Z * c->T * 2;
    size_t dbytes = sizeof(double) * c->Z * c->Z;
    if (!c->d_dm) HIP_TRY(hipMalloc(&c->d_dm, bytes));
    if (dist && !c->d_dist) HIP_TRY(hipMalloc(&c->d_dist, dbytes));
    c->have_dmat = false;
    c->tt_valid = false;
    c->tts_valid = false;
    c->ds_valid = false;
    c->pdrive_mean_valid = false;
    HIP_TRY(hipMemcpyAsync(c->d_dm, datamatrix, bytes, hipMemcpyHostToDevice, c->stream));
    if (dist) HIP_TRY(hipMemcpyAsync(c->d_dist, dist, dbytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_dmat = true;
    if (dist) c->have_dist = true;
    return CPM_OK;
}

// read_uber_csv with the C++ exceptions of its containers and threads turned into a message (nothing throws across the ABI)
static std::string read_csv_noexcept(const char *path, cpm::CsvRows &rows)
{
    try {
        const unsigned hw = std::thread::hardware_concurrency();
        return cpm::read_uber_csv(path, static_cast<int>(std::min(32u, hw ? hw : 4u)), rows, nullptr);
    } catch (const std::exception &e) {
        return std::string(path) + ": " + e.what();
    } catch (...) {
        return std::string(path) + ": unknown failure while reading";
    }
}

// createdatamatrix (src/createdatamatrix.jl:3-27) from rows already resident in d_raw (n x 5, column-major)
static int32_t datamatrix_from_device_rows(cpm_ctx *c, const double *d_raw, int64_t n)
{
    const size_t cells = static_cast<size_t>(c->Z) * c->Z * c->T;
    if (!c->d_dm) HIP_TRY(hipMalloc(&c->d_dm, sizeof(double) * cells * 2));
    c->have_dmat = false;
    c->tt_valid = false;
    c->tts_valid = false;
    c->ds_valid = false;
    c->pdrive_mean_valid = false;
    HIP_TRY(hipMemsetAsync(c->d_dm, 0, sizeof(double) * cells * 2, c->stream));  // zeros(number_zones, number_zones, T, 2) (:7)
    if (n == 0) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->have_dmat = true;
        return CPM_OK;
    }
    uint32_t *d_owner = nullptr;
    HIP_TRY(hipMalloc(&d_owner, sizeof(uint32_t) * cells));
    hipError_t e = hipMemsetAsync(d_owner, 0, sizeof(uint32_t) * cells, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(cpm::k_dm_owner, dim3(nblk(n, 256)), dim3(256), 0, c->stream, d_raw, n, static_cast<int>(c->Z),
                           static_cast<int>(c->T), d_owner, c->d_err);
        hipLaunchKernelGGL(cpm::k_dm_write, dim3(nblk(n, 256)), dim3(256), 0, c->stream, d_raw, n, static_cast<int>(c->Z),
                           static_cast<int>(c->T), d_owner, c->d_dm);
        e = hipGetLastError();
    }
    int32_t rc = (e == hipSuccess) ? check_err_flag(c, "createdatamatrix: a row holds a zone id / hour that is not an integer in range "
                                                        "(reference: InexactError / BoundsError)", CPM_ERR_ARG)
                                   : fail(CPM_ERR_HIP, "createdatamatrix: %s", hipGetErrorString(e));
    (void)hipStreamSynchronize(c->stream);
    dfree(d_owner);
    if (rc == CPM_OK) c->have_dmat = true;
    return rc;
}

int32_t cpm_parse_uber_csv(const char *path_to_csv_data, int64_t *n_rows_out, double *rawdata_out, int64_t capacity_rows)
{
    if (!path_to_csv_data || !n_rows_out) return fail(CPM_ERR_ARG, "parse_uber_csv: null argument");
    // the size query and the copy that follows it parse the file once: the rows of the last query are kept per thread
    thread_local cpm::CsvRows rows;
    thread_local std::string rows_key;
    struct stat st;
    if (stat(path_to_csv_data, &st) != 0) return fail(CPM_ERR_ARG, "parse_uber_csv: %s: %s", path_to_csv_data, strerror(errno));
    const std::string key = std::string(path_to_csv_data) + "|" + std::to_string(st.st_size) + "|" + std::to_string(st.st_mtim.tv_sec) +
                            "." + std::to_string(st.st_mtim.tv_nsec);
    if (key != rows_key) {
        rows_key.clear();
        std::string err = read_csv_noexcept(path_to_csv_data, rows);
        if (!err.empty()) return fail(CPM_ERR_ARG, "parse_uber_csv: %s", err.c_str());
        rows_key = key;
    }
    *n_rows_out = rows.n;
    if (!rawdata_out) return CPM_OK;
    rows_key.clear();  // handed over below: do not keep a second copy
    if (capacity_rows < rows.n) return fail(CPM_ERR_ARG, "parse_uber_csv: %lld rows, room for %lld", (long long)rows.n, (long long)capacity_rows);
    for (int k = 0; k < 5; ++k)
        std::memcpy(rawdata_out + static_cast<size_t>(k) * capacity_rows, rows.col[k].data(), sizeof(double) * static_cast<size_t>(rows.n));
    rows = cpm::CsvRows();
    return CPM_OK;
}

int32_t cpm_createdatamatrix_rows(cpm_ctx *c, int64_t n_rows, const double *rawdata)
{
    CTX_TRY(c);
    {   // a pending asynchronous IVP must be committed (or, after an overflow, repeated) on the tables it was enqueued with
        int32_t rc_ivp = finish_ivp(c);
        if (rc_ivp != CPM_OK) return rc_ivp;
    }
    if (n_rows < 0 || n_rows >= (int64_t(1) << 32) - 1 || (n_rows > 0 && !rawdata)) return fail(CPM_ERR_ARG, "createdatamatrix: bad row list");
    double *d_raw = nullptr;
    if (n_rows > 0) {
        HIP_TRY(hipMalloc(&d_raw, sizeof(double) * 5 * n_rows));
        hipError_t e = hipMemcpyAsync(d_raw, rawdata, sizeof(double) * 5 * n_rows, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) {
            dfree(d_raw);
            return fail(CPM_ERR_HIP, "createdatamatrix: upload: %s", hipGetErrorString(e));
        }
    }
    int32_t rc = datamatrix_from_device_rows(c, d_raw, n_rows);
    dfree(d_raw);
    return rc;
}

int32_t cpm_createdatamatrix_csv(cpm_ctx *c, const char *path_to_csv_data, int64_t *n_rows_out)
{
    CTX_TRY(c);
    {   // a pending asynchronous IVP must be committed (or, after an overflow, repeated) on the tables it was enqueued with
        int32_t rc_ivp = finish_ivp(c);
        if (rc_ivp != CPM_OK) return rc_ivp;
    }
    if (!path_to_csv_data) return fail(CPM_ERR_ARG, "createdatamatrix: null path");
    cpm::CsvRows rows;
    std::string err = read_csv_noexcept(path_to_csv_data, rows);
    if (!err.empty()) return fail(CPM_ERR_ARG, "createdatamatrix: %s", err.c_str());
    if (n_rows_out) *n_rows_out = rows.n;
    if (rows.n >= (int64_t(1) << 32) - 1) return fail(CPM_ERR_ARG, "createdatamatrix: too many rows");
    double *d_raw = nullptr;
    if (rows.n > 0) {
        HIP_TRY(hipMalloc(&d_raw, sizeof(double) * 5 * rows.n));
        for (int k = 0; k < 5; ++k) {
            hipError_t e = hipMemcpyAsync(d_raw + static_cast<size_t>(k) * rows.n, rows.col[k].data(), sizeof(double) * rows.n,
                                          hipMemcpyHostToDevice, c->stream);
            if (e != hipSuccess) {
                (void)hipStreamSynchronize(c->stream);
                dfree(d_raw);
                return fail(CPM_ERR_HIP, "createdatamatrix: upload: %s", hipGetErrorString(e));
            }
        }
    }
    int32_t rc = datamatrix_from_device_rows(c, d_raw, rows.n);
    dfree(d_raw);
    return rc;
}

int32_t cpm_get_datamatrix(cpm_ctx *c, double *datamatrix_out)
{
    CTX_TRY(c);
    if (!datamatrix_out) return fail(CPM_ERR_ARG, "null out");
    if (!c->have_dmat) return fail(CPM_ERR_STATE, "datamatrix not set");
    HIP_TRY(hipMemcpyAsync(datamatrix_out, c->d_dm, sizeof(double) * c->Z * c->Z * c->T * 2, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CPM_OK;
}

int32_t cpm_set_distance_from_centroids(cpm_ctx *c, const double *centroid_lat, const double *centroid_long)
{
    CTX_TRY(c);
    if (!centroid_lat || !centroid_long) return fail(CPM_ERR_ARG, "null centroids");
    const size_t dbytes = sizeof(double) * c->Z * c->Z;
    if (!c->d_dist) HIP_TRY(hipMalloc(&c->d_dist, dbytes));
    double *d_ll = nullptr;
    HIP_TRY(hipMalloc(&d_ll, sizeof(double) * 2 * c->Z));
    hipError_t e = hipMemcpyAsync(d_ll, centroid_lat, sizeof(double) * c->Z, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_ll + c->Z, centroid_long, sizeof(double) * c->Z, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(cpm::k_distance, dim3(nblk(c->Z, 256), static_cast<unsigned>(c->Z)), dim3(256), 0, c->stream, d_ll, d_ll + c->Z,
                           static_cast<int>(c->Z), c->d_dist);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dfree(d_ll);
    if (e != hipSuccess) return fail(CPM_ERR_HIP, "distance matrix: %s", hipGetErrorString(e));
    c->have_dist = true;
    c->pdrive_mean_valid = false;
    return CPM_OK;
}

int32_t cpm_set_distance(cpm_ctx *c, const double *dist)
{
    CTX_TRY(c);
    if (!dist) return fail(CPM_ERR_ARG, "null dist");
    const size_t dbytes = sizeof(double) * c->Z * c->Z;
    if (!c->d_dist) HIP_TRY(hipMalloc(&c->d_dist, dbytes));
    HIP_TRY(hipMemcpyAsync(c->d_dist, dist, dbytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_dist = true;
    c->pdrive_mean_valid = false;
    return CPM_OK;
}

int32_t cpm_get_distance(cpm_ctx *c, double *dist_out)
{
    CTX_TRY(c);
    if (!dist_out) return fail(CPM_ERR_ARG, "null out");
    if (!c->have_dist) return fail(CPM_ERR_STATE, "distance matrix not set");
    HIP_TRY(hipMemcpyAsync(dist_out, c->d_dist, sizeof(double) * c->Z * c->Z, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CPM_OK;
}

int32_t cpm_build_p_drive(cpm_ctx *c, double p_min, double p_max, double e_drive, double *out)
{
    CTX_TRY(c);
    {   // a pending asynchronous IVP must be committed (or, after an overflow, repeated) on the tables it was enqueued with
        int32_t rc_ivp = finish_ivp(c);
        if (rc_ivp != CPM_OK) return rc_ivp;
    }
    if (!c->have_dm()) return fail(CPM_ERR_STATE, "build_p_drive: datamatrix and distance matrix first (cpm_set_datamatrix, or cpm_createdatamatrix_* + cpm_set_distance_from_centroids)");
    size_t bytes = sizeof(double) * c->Z * c->T;
    if (!c->d_pdrive) HIP_TRY(hipMalloc(&c->d_pdrive, bytes));
    if (!c->d_pdrive_mean) HIP_TRY(hipMalloc(&c->d_pdrive_mean, bytes));
    if (!c->pdrive_mean_valid) {  // once per datamatrix / distance matrix: from the dataset's compact rows, or the Z x Z x T pass over the datamatrix
        int32_t rc_ds = ensure_dataset(c);
        if (rc_ds != CPM_OK) return rc_ds;
        if (c->ds_ok) {
            hipLaunchKernelGGL(cpm::k_ds_pdrive, dim3(cpm::ds_grid(c->T * c->Z)), dim3(cpm::kDsThreads * cpm::kDsRows), 0, c->stream, c->d_ds_cells, c->d_ds_cnt,
                               cpm::kDsCap, c->T * c->Z, static_cast<int>(c->Z), c->d_dist, c->d_pdrive_mean);
        } else {
            dim3 grid(nblk(c->Z, 64), static_cast<unsigned>(c->T));
            hipLaunchKernelGGL(cpm::k_pdrive_mean, grid, dim3(64), 0, c->stream, c->d_dm, c->d_dist, c->d_pdrive_mean, static_cast<int>(c->Z));
        }
        HIP_TRY(hipGetLastError());
        c->pdrive_mean_valid = true;
    }
    if (!c->d_thr) HIP_TRY(hipMalloc(&c->d_thr, sizeof(long long) * static_cast<size_t>(c->Z * c->T)));
    hipLaunchKernelGGL(cpm::k_pdrive_final, dim3(nblk(c->Z, 64), static_cast<unsigned>(c->T)), dim3(64), 0, c->stream, c->d_pdrive_mean,
                       c->d_pdrive, c->d_thr, static_cast<int>(c->Z), static_cast<int>(c->T), p_min, p_max, e_drive);  // (thresholds with it)
    hipError_t e = hipGetLastError();
    int32_t rc_thr = CPM_OK;
    if (e == hipSuccess && out) {  // (without a host copy to wait for the call returns as soon as the kernels are enqueued: everything that
        e = hipMemcpyAsync(out, c->d_pdrive, bytes, hipMemcpyDeviceToHost, c->stream);  // uses the table follows on the same stream)
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    if (e != hipSuccess) return fail(CPM_ERR_HIP, "build_p_drive: %s", hipGetErrorString(e));
    if (rc_thr != CPM_OK) return rc_thr;
    c->have_pdrive = true;
    return CPM_OK;
}

int32_t cpm_build_p_dest(cpm_ctx *c, double e_dest, int32_t e_is_integer, double *out)
{
    CTX_TRY(c);
    {   // a pending asynchronous IVP must be committed (or, after an overflow, repeated) on the tables it was enqueued with
        int32_t rc_ivp = finish_ivp(c);
        if (rc_ivp != CPM_OK) return rc_ivp;
    }
    if (!c->have_dmat) return fail(CPM_ERR_STATE, "build_p_dest: datamatrix first (cpm_set_datamatrix or cpm_createdatamatrix_*)");
    size_t bytes = sizeof(double) * c->Z * c->Z * c->T;
    {   // a sparse datamatrix: everything from its compact rows (cpm_dataset.h); p_destin itself only when the caller wants the array
        bool done = false;
        int32_t rc_sp = build_p_dest_sparse(c, e_dest, e_is_integer, &done);
        if (rc_sp != CPM_OK) return rc_sp;
        if (done) {
            if (!out) return CPM_OK;
            rc_sp = ensure_dense_p(c);
            if (rc_sp != CPM_OK) return rc_sp;
            HIP_TRY(hipMemcpyAsync(out, c->d_p, bytes, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            return CPM_OK;
        }
    }
    // createpdestin's array in the reference's layout: weights, then normalised in place.  It stays with the context (a sweep calls
    // this entry once per e_dest value: no allocation on that path); the row tables are derived from it in one more pass.
    if (!c->d_p) HIP_TRY(hipMalloc(&c->d_p, bytes));
    c->have_cdf = false;
    dim3 g1(nblk(c->Z, 256), static_cast<unsigned>(c->Z));
    if (c->T == 24)
        hipLaunchKernelGGL(cpm::k_pdest_weights<24>, g1, dim3(256), 0, c->stream, c->d_dm, c->d_p, static_cast<int>(c->Z), static_cast<int>(c->T), e_dest, e_is_integer);
    else
        hipLaunchKernelGGL(cpm::k_pdest_weights<0>, g1, dim3(256), 0, c->stream, c->d_dm, c->d_p, static_cast<int>(c->Z), static_cast<int>(c->T), e_dest, e_is_integer);
    // (row sums into the row-total array, which the row builder overwrites behind them)
    if (!c->d_last) HIP_TRY(hipMalloc(&c->d_last, sizeof(double) * static_cast<size_t>(c->T * c->Z)));
    hipLaunchKernelGGL(cpm::k_pdest_rowsum, dim3(nblk(c->Z, 64), static_cast<unsigned>(c->T)), dim3(64), 0, c->stream, c->d_p, c->d_last, static_cast<int>(c->Z));
    hipLaunchKernelGGL(cpm::k_pdest_divide, dim3(nblk(c->Z, 256), nblk(c->Z, cpm::kDivBatch), static_cast<unsigned>(c->T)), dim3(256), 0, c->stream, c->d_p,
                       c->d_last, static_cast<int>(c->Z));
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && out) e = hipMemcpyAsync(out, c->d_p, bytes, hipMemcpyDeviceToHost, c->stream);
    if (e != hipSuccess) return fail(CPM_ERR_HIP, "build_p_dest: %s", hipGetErrorString(e));
    return build_rows(c, false);  // (synchronises)
}

int32_t cpm_get_p_drive(cpm_ctx *c, double *out)
{
    CTX_TRY(c);
    if (!out) return fail(CPM_ERR_ARG, "null out");
    if (!c->have_pdrive) return fail(CPM_ERR_STATE, "p_drive not set");
    HIP_TRY(hipMemcpyAsync(out, c->d_pdrive, sizeof(double) * c->Z * c->T, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CPM_OK;
}

int32_t cpm_get_cdf_row(cpm_ctx *c, int64_t origin1, int64_t hour1, double *out)
{
    CTX_TRY(c);
    if (!out) return fail(CPM_ERR_ARG, "null out");
    if (!c->have_cdf) return fail(CPM_ERR_STATE, "p_dest not set");
    if (origin1 < 1 || origin1 > c->Z || hour1 < 1 || hour1 > c->T) return fail(CPM_ERR_ARG, "row index out of range");
    {
        int32_t rc_cdf = ensure_full_cdf(c);
        if (rc_cdf != CPM_OK) return rc_cdf;
    }
    const double *src = c->d_cdf + (static_cast<size_t>(hour1 - 1) * c->Z + (origin1 - 1)) * c->Zp;
    HIP_TRY(hipMemcpyAsync(out, src, sizeof(double) * c->Z, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CPM_OK;
}

int32_t cpm_synth_tables(cpm_ctx *c, uint64_t table_seed) { return cpm_synth_tables_skewed(c, table_seed, 0); }

int32_t cpm_synth_tables_skewed(cpm_ctx *c, uint64_t table_seed, int64_t skew_q)
{
    CTX_TRY(c);
    if (skew_q < 0 || (skew_q > 0 && c->Z % 7919 == 0)) return fail(CPM_ERR_ARG, "synth_tables: skew %lld", (long long)skew_q);
    {   // a pending asynchronous IVP must be committed (or, after an overflow, repeated) on the tables it was enqueued with
        int32_t rc_ivp = finish_ivp(c);
        if (rc_ivp != CPM_OK) return rc_ivp;
    }
    size_t pd_bytes = sizeof(double) * c->Z * c->T;
    if (!c->d_pdrive) HIP_TRY(hipMalloc(&c->d_pdrive, pd_bytes));
    hipLaunchKernelGGL(cpm::k_synth_p_drive, dim3(nblk(c->Z * c->T, 256)), dim3(256), 0, c->stream, c->d_pdrive,
                       static_cast<int>(c->Z), static_cast<int>(c->T), table_seed);
    HIP_TRY(hipGetLastError());
    {
        int32_t rc_thr = update_thr(c);
        if (rc_thr != CPM_OK) return rc_thr;
    }
    c->have_pdrive = true;
    if (!c->d_p) HIP_TRY(hipMalloc(&c->d_p, sizeof(double) * c->Z * c->Z * c->T));
    c->have_cdf = false;
    dim3 grid(nblk(c->Z, 64), static_cast<unsigned>(c->T));
    hipLaunchKernelGGL(cpm::k_synth_p_dest, grid, dim3(64), 0, c->stream, c->d_p, static_cast<int>(c->Z), table_seed, skew_q);
    HIP_TRY(hipGetLastError());
    return build_rows(c, false);
}

int32_t cpm_synth_datamatrix(cpm_ctx *c, uint64_t table_seed, double density)
{
    CTX_TRY(c);
    {
        int32_t rc_ivp = finish_ivp(c);
        if (rc_ivp != CPM_OK) return rc_ivp;
    }
    if (!(density >= 0.0 && density <= 1.0)) return fail(CPM_ERR_ARG, "synth_datamatrix: density %g", density);
    const size_t cells = static_cast<size_t>(c->Z) * c->Z * c->T;
    if (!c->d_dm) HIP_TRY(hipMalloc(&c->d_dm, sizeof(double) * cells * 2));
    if (!c->d_dist) HIP_TRY(hipMalloc(&c->d_dist, sizeof(double) * c->Z * c->Z));
    c->have_dmat = c->have_dist = false;
    c->tt_valid = false;
    c->tts_valid = false;
    c->ds_valid = false;
    c->pdrive_mean_valid = false;
    hipLaunchKernelGGL(cpm::k_synth_datamatrix, dim3(nblk(c->Z, 256), static_cast<unsigned>(c->Z), static_cast<unsigned>(c->T)), dim3(256), 0, c->stream,
                       c->d_dm, static_cast<int>(c->Z), static_cast<int>(c->T), table_seed, density);
    hipLaunchKernelGGL(cpm::k_synth_dist, dim3(nblk(c->Z, 256), static_cast<unsigned>(c->Z)), dim3(256), 0, c->stream, c->d_dist, static_cast<int>(c->Z),
                       table_seed);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_dmat = c->have_dist = true;
    return CPM_OK;
}

int32_t cpm_refresh_tables(cpm_ctx *c, int32_t with_f64_cdf)
{
    CTX_TRY(c);
    {
        int32_t rc_ivp = finish_ivp(c);
        if (rc_ivp != CPM_OK) return rc_ivp;
    }
    if (!c->have_cdf) return fail(CPM_ERR_STATE, "refresh_tables: p_dest not set");
    if (c->sparse_tables) {  // (the tables of a compact dataset: its rows are what they are rebuilt from)
        if (!c->ds_valid || !c->ds_ok) return fail(CPM_ERR_STATE, "refresh_tables: the datamatrix these tables were built from has been replaced");
        bool done = false;
        int32_t rc_sp = build_p_dest_sparse(c, c->tab_e_dest, c->tab_e_int, &done);
        if (rc_sp != CPM_OK || !done) return rc_sp != CPM_OK ? rc_sp : fail(CPM_ERR_STATE, "refresh_tables: compact rows no longer usable");
        return with_f64_cdf ? ensure_full_cdf(c) : CPM_OK;
    }
    if (!c->d_p) return fail(CPM_ERR_STATE, "refresh_tables: p_dest not set");
    return build_rows(c, with_f64_cdf != 0);
}

// ------------------------------------------------------------------ cars

int32_t cpm_init_states_strided(cpm_ctx *c, int64_t C_total, int64_t cars_per_zone, int64_t car_first, int64_t car_stride, int64_t car_count)
{
    CTX_TRY(c);
    if (c->ivp_pending) {  // the state is being replaced: the pending IVP's result is moot
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->ivp_pending = false;
    }
    if (C_total < 0 || cars_per_zone < 1 || car_first < 0 || car_count < 0 || car_stride < 1 || car_stride > 0x7fffffffLL ||
        (car_count > 0 && car_first + (car_count - 1) * car_stride >= C_total))
        return fail(CPM_ERR_ARG, "init_states: bad car set {%lld + k * %lld, k < %lld} of %lld", (long long)car_first, (long long)car_stride,
                    (long long)car_count, (long long)C_total);
    if (C_total > 0 && (C_total - 1) / cars_per_zone >= c->Z)
        return fail(CPM_ERR_ARG, "init_states: C = %lld cars at %lld per zone exceed %lld zones", (long long)C_total,
                    (long long)cars_per_zone, (long long)c->Z);
    if (car_count >= (int64_t(1) << 32)) return fail(CPM_ERR_ARG, "init_states: more than 2^32 cars in one context");
    int32_t rc = ensure_cars(c, car_count);
    if (rc != CPM_OK) return rc;
    c->C_total = C_total;
    c->cpz = cars_per_zone;
    c->cars = cpm::CarIndex{car_first, static_cast<uint32_t>(car_stride)};
    c->zx.buckets0_valid = false;
    c->zg.buckets0_valid = false;
    if (car_count > 0) {
        hipLaunchKernelGGL(cpm::k_init_states, dim3(nblk(car_count, 256)), dim3(256), 0, c->stream, c->d_zone0, c->cars, car_count, cars_per_zone);
        HIP_TRY(hipGetLastError());
    }
    c->have_state = true;
    return CPM_OK;
}

int32_t cpm_init_states(cpm_ctx *c, int64_t C_total, int64_t cars_per_zone, int64_t car_begin, int64_t car_count)
{
    return cpm_init_states_strided(c, C_total, cars_per_zone, car_begin, 1, car_count);
}

int32_t cpm_set_state(cpm_ctx *c, const int64_t *zones)
{
    CTX_TRY(c);
    if (!c->have_state) return fail(CPM_ERR_STATE, "set_state: cpm_init_states first (defines the car range)");
    if (c->ivp_pending) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->ivp_pending = false;
    }
    c->zx.buckets0_valid = false;
    c->zg.buckets0_valid = false;
    if (c->n == 0) return CPM_OK;
    if (!zones) return fail(CPM_ERR_ARG, "null zones");
    int64_t *d_z = nullptr;
    HIP_TRY(hipMalloc(&d_z, sizeof(int64_t) * c->n));
    hipError_t e = hipMemcpyAsync(d_z, zones, sizeof(int64_t) * c->n, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(cpm::k_zones_from_i64, dim3(nblk(c->n, 256)), dim3(256), 0, c->stream, c->d_zone0, d_z, c->n,
                           c->Z, c->d_err);
        e = hipGetLastError();
    }
    int32_t rc = (e == hipSuccess) ? check_err_flag(c, "set_state: zone id outside 1..number_zones", CPM_ERR_ARG)
                                   : fail(CPM_ERR_HIP, "set_state: %s", hipGetErrorString(e));
    (void)hipStreamSynchronize(c->stream);
    dfree(d_z);
    return rc;
}

int32_t cpm_get_state(cpm_ctx *c, int64_t *zones_out)
{
    CTX_TRY(c);
    if (!c->have_state) return fail(CPM_ERR_STATE, "get_state: no car state");
    {
        int32_t rc_ivp = finish_ivp(c);
        if (rc_ivp != CPM_OK) return rc_ivp;
    }
    if (c->n == 0) return CPM_OK;
    if (!zones_out) return fail(CPM_ERR_ARG, "null zones_out");
    int64_t *d_z = nullptr;
    HIP_TRY(hipMalloc(&d_z, sizeof(int64_t) * c->n));
    hipLaunchKernelGGL(cpm::k_zones_to_i64, dim3(nblk(c->n, 256)), dim3(256), 0, c->stream, d_z, c->d_zone0, c->n);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(zones_out, d_z, sizeof(int64_t) * c->n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dfree(d_z);
    if (e != hipSuccess) return fail(CPM_ERR_HIP, "get_state: %s", hipGetErrorString(e));
    return CPM_OK;
}

int32_t cpm_solve_ivp_async(cpm_ctx *c, uint64_t seed)
{
    CTX_TRY(c);
    return ivp_enqueue(c, seed);
}

int32_t cpm_solve_ivp(cpm_ctx *c, uint64_t seed, int64_t *initial_state_out)
{
    CTX_TRY(c);
    int32_t rc = ivp_enqueue(c, seed);
    if (rc != CPM_OK) return rc;
    rc = finish_ivp(c);
    if (rc != CPM_OK) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (initial_state_out) return cpm_get_state(c, initial_state_out);
    return CPM_OK;
}

int32_t cpm_resample_dev(cpm_ctx *c, uint64_t seed, uint32_t flags, void *d_counts)
{
    CTX_TRY(c);
    if (!d_counts) return fail(CPM_ERR_ARG, "null d_counts");
    return resample_enqueue(c, seed, flags, static_cast<int64_t *>(d_counts));
}

int32_t cpm_resample(cpm_ctx *c, uint64_t seed, uint32_t flags, int64_t *parking, int64_t *driving,
                     int64_t *sum_tt_q16, int64_t *state_out, double *trans_out)
{
    CTX_TRY(c);
    if (!parking || !driving) return fail(CPM_ERR_ARG, "null count outputs");
    bool compat = state_out || trans_out;
    struct KernelGuard {  // the kernel choice is overridden for this call only, whichever way the call ends
        cpm_ctx *c;
        int saved;
        ~KernelGuard() { c->kernel = saved; }
    } kernel_guard{c, c->kernel};
    {   // a pending IVP is committed with the caller's kernel choice, not with the override below
        int32_t rc_ivp = finish_ivp(c);
        if (rc_ivp != CPM_OK) return rc_ivp;
    }
    if (compat) c->kernel = CPM_KERNEL_CAR;  // the per-hour records of every car are kept by this path
    int32_t rc = resample_enqueue(c, seed, flags, c->d_counts);
    const size_t zt = static_cast<size_t>(c->Z * c->T), nwords = 2 * zt + 2;
    auto fetch = [&]() -> int32_t {  // the count tensor, Σ time and the status word: one copy into pinned memory, one wait
        HIP_TRY(hipMemcpyAsync(c->h_counts, c->d_counts, sizeof(int64_t) * nwords, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->status_pending) c->zg.set_parts(static_cast<uint32_t>(c->h_status[1]), static_cast<uint32_t>(c->h_status[1] >> 32));  // (how the next grouped step is launched)
        c->status_pending = false;  // this step's status word is dealt with here: resample_enqueue must not grow the regions for it again
        return CPM_OK;
    };
    if (rc == CPM_OK) rc = fetch();
    // a bucket or a run outgrew its region: again with twice the regions while the problem still fits ...
    while (rc == CPM_OK && c->h_counts[nwords - 1] != 0 && pick_kernel(c) == CPM_KERNEL_ZONE_GROUPED && absorb_status(c, c->h_counts[nwords - 1])) {
        rc = resample_enqueue(c, seed, flags, c->d_counts);
        if (rc == CPM_OK) rc = fetch();
    }
    if (rc == CPM_OK && c->h_counts[nwords - 1] != 0) {  // ... else on a layout that cannot overflow
        if (pick_kernel(c) == CPM_KERNEL_ZONE_GROUPED) c->grouped_overflowed = true;
        c->kernel = cpm::exact_path_fits(static_cast<int>(c->Z)) ? CPM_KERNEL_ZONE_LDS : CPM_KERNEL_CAR;
        rc = resample_enqueue(c, seed, flags, c->d_counts);
        if (rc == CPM_OK) rc = fetch();
    }
    if (rc != CPM_OK) return rc;
    std::memcpy(parking, c->h_counts, sizeof(int64_t) * zt);
    std::memcpy(driving, c->h_counts + zt, sizeof(int64_t) * zt);
    if (sum_tt_q16) *sum_tt_q16 = c->h_counts[2 * zt];
    if (compat && c->n > 0) {
        // one hour column at a time: state[:,t] and trans[:,t,1..4] are contiguous runs of C values
        int64_t n = c->n;
        char *d_cols = nullptr;
        HIP_TRY(hipMalloc(&d_cols, static_cast<size_t>(n) * 8 * 5));
        int64_t *d_state = reinterpret_cast<int64_t *>(d_cols);
        double *d_f = reinterpret_cast<double *>(d_cols) + n;
        bool travel = (flags & CPM_FLAG_TRAVEL) != 0;
        hipError_t e = hipSuccess;
        for (int t = 0; t < c->T && e == hipSuccess; ++t) {
            const uint32_t *zsrc = (t == 0) ? c->d_zone0 : c->d_rec + static_cast<size_t>(t - 1) * n;
            hipLaunchKernelGGL(cpm::k_export_hour, dim3(nblk(n, 256)), dim3(256), 0, c->stream, zsrc,
                               c->d_rec + static_cast<size_t>(t) * n, n, c->cars, d_state, d_f, d_f + n, d_f + 2 * n,
                               d_f + 3 * n, travel ? c->d_dm : nullptr, c->d_dist, static_cast<int>(c->Z),
                               static_cast<int>(c->T), t, static_cast<uint32_t>(c->T - 1 + t), seed);
            e = hipGetLastError();
            if (e == hipSuccess && state_out)
                e = hipMemcpyAsync(state_out + static_cast<size_t>(t) * n, d_state, sizeof(int64_t) * n, hipMemcpyDeviceToHost, c->stream);
            for (int k = 0; k < 4 && e == hipSuccess && trans_out; ++k)
                e = hipMemcpyAsync(trans_out + (static_cast<size_t>(k) * c->T + t) * n, d_f + static_cast<size_t>(k) * n,
                                   sizeof(double) * n, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        }
        dfree(d_cols);
        if (e != hipSuccess) return fail(CPM_ERR_HIP, "compat export: %s", hipGetErrorString(e));
    }
    return CPM_OK;
}

int32_t cpm_last_kernel_ms(cpm_ctx *c, float *ms_out, int32_t cap, int32_t *n_out)
{
    CTX_TRY(c);
    if (!ms_out || !n_out) return fail(CPM_ERR_ARG, "null output");
    HIP_TRY(hipStreamSynchronize(c->stream));
    int n = std::min<int>(c->n_prof, cap);
    for (int k = 0; k < n; ++k) HIP_TRY(hipEventElapsedTime(&ms_out[k], c->ev[2 * k], c->ev[2 * k + 1]));
    *n_out = n;
    return CPM_OK;
}

int32_t cpm_algorithmic_bytes_per_hour(cpm_ctx *c, int64_t *bytes_out)
{
    if (!c || !bytes_out) return fail(CPM_ERR_ARG, "null argument");
    // B/T = Z*Z*8 (CDF slab) + Z*8 (p_drive) + C_g*8 (4 B zone in + 4 B out) + 2*Z*8 (counts)
    // The second-generation grouped path streams the 4-byte high-word rows (Zq per row) instead of the f64 rows:
    // its true element size is substituted, as 8(d) prescribes for a variant with a different element size.
    const bool hi_rows = pick_kernel(c) == CPM_KERNEL_ZONE_GROUPED && grouped_fits(c, c->zg.cap_mult);
    const int64_t rows = hi_rows ? c->Z * static_cast<int64_t>(cpm::pack_row_words(c->Zq, c->pk_G, c->sparse_tables)) * 4 + c->Z * 8
                                 : c->Z * c->Z * 8;
    *bytes_out = rows + c->Z * 8 + c->n * 8 + 2 * c->Z * 8;
    return CPM_OK;
}

int32_t cpm_debug_categorical(cpm_ctx *c, int64_t origin1, int64_t hour1, int64_t n, const uint64_t *k53, int64_t *dest_out,
                              int32_t *n_exact_out)
{
    CTX_TRY(c);
    if (!c->have_cdf || !c->d_hi) return fail(CPM_ERR_STATE, "debug_categorical: p_dest not set (or rows too long for the high-word table)");
    if (origin1 < 1 || origin1 > c->Z || hour1 < 1 || hour1 > c->T) return fail(CPM_ERR_ARG, "row index out of range");
    if (n < 0 || (n > 0 && (!k53 || !dest_out))) return fail(CPM_ERR_ARG, "bad draw list");
    if (n_exact_out) *n_exact_out = 0;
    if (n == 0) return CPM_OK;
    const size_t row = static_cast<size_t>(hour1 - 1) * c->Z + (origin1 - 1);
    uint64_t *d_k = nullptr;
    int64_t *d_o = nullptr;
    int *d_n = nullptr;
    hipError_t e = hipMalloc(&d_k, sizeof(uint64_t) * n);
    if (e == hipSuccess) e = hipMalloc(&d_o, sizeof(int64_t) * n);
    if (e == hipSuccess) e = hipMalloc(&d_n, sizeof(int));
    if (e == hipSuccess) e = hipMemsetAsync(d_n, 0, sizeof(int), c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_k, k53, sizeof(uint64_t) * n, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        const int G = c->pk_G;
        const size_t words = static_cast<size_t>(cpm::pack_row_words(c->Zq, G, c->sparse_tables));
        const size_t lds = sizeof(uint32_t) * words;
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(cpm::k_pack_search_debug), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        const size_t th = static_cast<size_t>(hour1 - 1);
        uint32_t h_scnt = 0;
        if (c->sparse_tables) {
            e = hipMemcpyAsync(&h_scnt, c->d_scnt + row, sizeof h_scnt, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        }
        if (e == hipSuccess) {
            const bool sp = c->sparse_tables;
            hipLaunchKernelGGL(cpm::k_pack_search_debug, dim3(1), dim3(512), lds, c->stream, c->d_hi + row * words, c->d_last + row,
                               sp ? nullptr : c->d_ckpt + th * cpm::ckpt_count(static_cast<int>(c->Z)) * c->Z, sp ? nullptr : c->d_p + th * c->Z * c->Z,
                               static_cast<int>(origin1 - 1), static_cast<int>(c->Z), c->Zq, G, c->pk_Zc, sp ? c->d_sp + row * cpm::kDsCap : nullptr,
                               sp ? c->d_sj + row * cpm::kDsCap : nullptr, std::min(h_scnt, cpm::kDsCap), n, d_k, d_o, d_n);
        }
        e = hipGetLastError();
    }
    int h_n = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(dest_out, d_o, sizeof(int64_t) * n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&h_n, d_n, sizeof(int), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dfree(d_k);
    dfree(d_o);
    dfree(d_n);
    if (e != hipSuccess) return fail(CPM_ERR_HIP, "debug_categorical: %s", hipGetErrorString(e));
    if (n_exact_out) *n_exact_out = h_n;
    return CPM_OK;
}

#ifdef CPM_DIAGNOSTIC
// diagnostic builds only (tools/place_stamps.py): a side buffer for the s_memtime stamps of the placing kernel; n_blocks x 8 words
int32_t cpm_diag_place_stamps(cpm_ctx *c, unsigned long long *host_out_or_null, int64_t n_blocks)
{
    CTX_TRY(c);
    static unsigned long long *d_buf = nullptr;
    static int64_t cap = 0;
    if (!host_out_or_null) {  // arm
        if (cap < n_blocks) {
            if (d_buf) (void)hipFree(d_buf);
            HIP_TRY(hipMalloc(&d_buf, sizeof(unsigned long long) * 8 * n_blocks));
            cap = n_blocks;
        }
        HIP_TRY(hipMemset(d_buf, 0, sizeof(unsigned long long) * 8 * n_blocks));
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(cpm::g_place_stamps), &d_buf, sizeof(d_buf)));
        return CPM_OK;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(host_out_or_null, d_buf, sizeof(unsigned long long) * 8 * std::min(n_blocks, cap), hipMemcpyDeviceToHost));
    return CPM_OK;
}
#endif

}  // extern "C"
