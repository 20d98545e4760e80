// cpm_zone6_kernels.h -- second generation of the grouped zone path (CPM_KERNEL_ZONE_GROUPED):
//   * the row a workgroup stages is the HIGH WORD of the CDF row, 4 B per destination instead of 8;
//   * the drivers of an origin zone go into FIXED-SIZE runs, one per destination group, so their
//     position is known the moment their rank is (no scan, no second pass over the cars, no
//     offsets for the placing kernel to fetch before it can fetch the runs).
//
// High-word rows.  The categorical draw of src/resampling.jl:38-45 is "first j with u <= cdf[j]"
// with u = k * 2^-53, k the 53 high bits of two Philox words.  Let hi[j] = floor(cdf[j] * 2^32)
// (0xFFFFFFFF when cdf[j] >= 1) and khi = floor(u * 2^32) = the high Philox word.  Then
//      hi[j]   > khi  =>  cdf[j]   * 2^32 >= khi + 1 > u * 2^32      =>  cdf[j]   >= u  (not cdf[j] < u)
//      hi[j-1] < khi  =>  cdf[j-1] * 2^32 <  hi[j-1] + 1 <= u * 2^32  =>  cdf[j-1] <  u
// so the first j with hi[j] >= khi IS the reference's answer whenever hi[j] > khi strictly.  When
// hi[j] == khi (probability ~ Z * 2^-32 per draw), or no hi[j] >= khi exists (u above the row total),
// the car repeats the search on the f64 row in HBM with exactly the code of the other kernels
// (clamp_u + lower_bound_row).  u == 0 needs no special case: khi = 0, the walk stops at the first j,
// hi[0] > 0 is accepted (cdf[0] > 0: the clamped answer), hi[0] == 0 is a tie.  Results are bit-identical
// to the f64 search by construction; tests/test_gpu_parity.py drives the tie and out-of-range branches
// through cpm_debug_categorical.  The table hi[T][Z][Zq] (Zq = Z rounded up to 32, padded with
// 0xFFFFFFFF) and the row totals last[T][Z] are built next to the CDF.  HBM traffic of an hourly
// launch: Z*Zq*4 B of rows instead of Z*Zp*8.
//
// Fixed-size runs.  Zone z's drivers of destination group g are written at D[(z*kGroups + g)*scap + rank],
// rank from an LDS atomic, as id | (dest - g*zpg) << idbits (4 B; needs n <= 2^idbits).  The 32 run
// lengths go to cntg[z][32].  A run that would outgrow scap raises bit 1 of the status word, as a bucket
// outgrowing cap does; the caller repeats the step on the exact layout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "../../include/cpm.h"
#include "cpm_kernels.h"
#include "cpm_zone_kernels.h"
#include "cpm_zone3_kernels.h"

namespace cpm {

constexpr int kGroups6 = 32;            // destination groups (== kGroups of cpm_zone5_kernels.h)
constexpr int kMaxZonesPerGroup6 = 512;
constexpr uint32_t kHiMax = 0xFFFFFFFFu;

// hi[t][o][:] and last[t][o] from the canonical CDF (one thread per element, coalesced both ways)
__global__ __launch_bounds__(256) void k_build_hi32(const double *__restrict__ cdf, uint32_t *__restrict__ hi,
                                                    double *__restrict__ last, int Z, int Zp, int Zq, int64_t rows)
{
    const int64_t row = blockIdx.y + static_cast<int64_t>(blockIdx.z) * gridDim.y;
    if (row >= rows) return;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= Zq) return;
    uint32_t h = kHiMax;
    if (j < Z) {
        const double c = cdf[row * Zp + j];
        if (c < 1.0) h = static_cast<uint32_t>(floor(c * 0x1.0p32));  // exact scaling; c >= 0 (validated by k_build_cdf)
        if (j == Z - 1) last[row] = c;
    }
    hi[row * Zq + j] = h;
}

struct Zone6Args {
    const uint32_t *ids;      // [Z*cap] this hour's buckets
    const uint32_t *cnt;      // [Z] their sizes
    const uint32_t *hi_t;     // [Z][Zq] high words of this hour's CDF rows
    const double *last_t;     // [Z] row totals (f64)
    const double *pdrive_t, *cdf_t, *dm;
    uint32_t *ids_next;       // [Z*cap]       (grouped)
    uint32_t *cnt_next;       // [Z] stayers   (grouped; k_zone6_place adds the arrivals)
    uint32_t *D;              // [Z][kGroups6][scap] packed drivers (grouped)
    uint32_t *cntg;           // [Z][kGroups6] run lengths (grouped)
    uint32_t *rec_out;        // [Z*cap] dest | drive << 31 per slot (plain form)
    unsigned long long *parking_t, *driving_t, *tt_sum, *status;
    int Z, Zp, Zq, H, T, t, zpg;
    uint32_t cap, scap, idbits, step, gmagic;
    int64_t car_begin;
    uint64_t seed;
    int abl;  // CPM_DIAGNOSTIC builds only (results WRONG): 1 no Philox, 2 no walk, 4 no emit stores, 8 no row load, 16 no ids load,
              // 32 no per-car work at all (everybody stays), 64 no tree build, 128 no rank atomics
};

#ifdef CPM_DIAGNOSTIC
#define CPM_ABL(a, bit) ((a).abl & (bit))
#else
#define CPM_ABL(a, bit) 0
#endif

// request the row: NQ 16-B pieces per thread (clamped: every lane issues every load)
template <int BLOCK, int NQ>
__device__ __forceinline__ void hi_row_load(uint4 (&pc)[NQ], const uint32_t *hi_row, int Zq, int tid)
{
    const uint4 *src = reinterpret_cast<const uint4 *>(hi_row);
#pragma unroll
    for (int m = 0; m < NQ; ++m) pc[m] = src[min(tid + m * BLOCK, Zq / 4 - 1)];
}

// pieces -> breadth-first tree over elements 0..Z-2; tree[0] = element Z-1; ranks >= Z-1 of the tree = 0xFFFFFFFF
template <int BLOCK, int NQ>
__device__ __forceinline__ void hi_tree_store(uint32_t *tree, const uint4 (&pc)[NQ], int Z, int Zq, int H, int tid)
{
    const int P = 1 << H;
    for (int r = Z + tid; r < P; r += BLOCK) {
        int tz = __builtin_ctz(static_cast<unsigned>(r));
        tree[(1u << (H - 1 - tz)) + (static_cast<unsigned>(r) >> (tz + 1))] = kHiMax;
    }
#pragma unroll
    for (int m = 0; m < NQ; ++m) {
        const int j = tid + m * BLOCK;
        if (4 * j < Zq) {
            const uint32_t el = 4 * j;
            const uint32_t q[4] = {pc[m].x, pc[m].y, pc[m].z, pc[m].w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (el + k < static_cast<uint32_t>(Z)) tree[eytz_pos(el + k, Z, H)] = q[k];
        }
    }
}

// The f64 search of the other kernels, on the row where it lies in HBM (rare: ties and u above the row total).
__device__ __noinline__ uint32_t search_exact_row(const double *__restrict__ cdf_row, int Z, double uc, double last)
{
    return static_cast<uint32_t>(lower_bound_row(cdf_row, Z, clamp_u(uc, last)));
}

// CPT simultaneous tree walks; ok[c] = the answer is certain (strictly greater high word)
template <int CPT>
__device__ __forceinline__ void hi_walk(const uint32_t *tree, const uint32_t (&khi)[CPT], int Z, int H, uint32_t hi_last,
                                        uint32_t (&dest)[CPT], bool (&ok)[CPT])
{
    uint32_t i[CPT], cand[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        i[c] = 1;
        cand[c] = kHiMax;
    }
    for (int l = 0; l < H; ++l) {
        uint32_t k[CPT];
#pragma unroll
        for (int c = 0; c < CPT; ++c) k[c] = tree[i[c]];
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            const bool right = k[c] < khi[c];
            cand[c] = right ? cand[c] : k[c];  // value of the last node where the walk went left = value at the answer
            i[c] = 2 * i[c] + (right ? 1u : 0u);
        }
    }
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        dest[c] = eytz_decode(i[c], Z, H);
        if (dest[c] == static_cast<uint32_t>(Z - 1)) cand[c] = hi_last;  // element Z-1 lives in tree[0], its tree slot holds the pad
        ok[c] = cand[c] > khi[c];
    }
}

// Philox words of (car, step, stream 0): Bernoulli integer kb (53 bits of words 0,1) and the categorical words (2,3)
__device__ __forceinline__ void car_draw_words(uint64_t seed, uint64_t car, uint32_t step, long long &kb, uint32_t &clo, uint32_t &chi)
{
    U4 r = philox4x32_10(static_cast<uint32_t>(car), static_cast<uint32_t>(car >> 32), step, 0u,
                         static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32));
    kb = static_cast<long long>(((static_cast<uint64_t>(r.y) << 32) | r.x) >> 11);
    clo = r.z;
    chi = r.w;
}

// One workgroup per origin zone.  (Resident workgroups each walking several zones were measured no faster -- a zone's
// time is its chain of memory round trips, not the dispatch of its waves -- and cost registers.)
// GROUPED: stayers compacted into next hour's bucket of the zone, drivers into the zone's fixed-size runs.  The drivers
// of a zone are first ranked and staged in LDS (kStage6 entries per group) and written out by 16 lanes per run,
// 64 B at a time; ranks beyond kStage6 go to HBM directly.
// !GROUPED: dest | drive << 31 per slot (the last hour of a resample: counts only).
// The bucket size, the ids and the row are requested together: the id loads are clamped to the zone's REGION (cap), not
// to its size, so they do not wait for the size (one HBM round trip less in every workgroup's chain).
constexpr int kStage6 = 32;

template <bool TRAVEL, int BLOCK, int NQ, int CPT, bool GROUPED>
__global__ __launch_bounds__(BLOCK, TRAVEL ? 2 : 8) void k_zone6_sample(Zone6Args a)
{
    extern __shared__ uint32_t tree[];  // 2^H high words
    __shared__ uint32_t s_ndrive, s_nstay;
    __shared__ unsigned long long s_tt;
    __shared__ uint32_t gb[kGroups6];
    __shared__ uint32_t stage[GROUPED ? kGroups6 * kStage6 : 1];
    const int Z = a.Z, H = a.H;
    const int z = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t cap = a.cap;
    const uint32_t b = static_cast<uint32_t>(z) * cap;
    // ids first, then the row: vmcnt retires in order, so the ids have landed when the row has
    uint32_t id[CPT + 1];
#pragma unroll
    for (int c = 0; c <= CPT; ++c) id[c] = CPM_ABL(a, 16) ? b + tid + c * BLOCK : a.ids[b + min(static_cast<uint32_t>(tid + c * BLOCK), cap - 1)];
    uint4 pc[NQ];
    if (!CPM_ABL(a, 8)) hi_row_load<BLOCK, NQ>(pc, a.hi_t + static_cast<size_t>(z) * a.Zq, a.Zq, tid);
    else
        for (int m = 0; m < NQ; ++m) pc[m] = make_uint4(tid * 4u << 18, (tid * 4u + 1) << 18, (tid * 4u + 2) << 18, (tid * 4u + 3) << 18);
    const uint32_t n = min(a.cnt[z], cap);
    const double last = a.last_t[z];
    const long long thr = bernoulli_threshold(a.pdrive_t[z]);
    if (tid == 0) {
        a.parking_t[z] = n;  // every car present at hour t, drivers included (src/saveresults.jl:12)
        s_ndrive = 0;
        s_nstay = 0;
        s_tt = 0;
    }
    if (tid < kGroups6) gb[tid] = 0;
    if (n == 0) {  // driving_t[z] stays 0 (zeroed by the caller)
        if (GROUPED) {
            if (tid == 0) a.cnt_next[z] = 0;
            if (tid < kGroups6) a.cntg[static_cast<size_t>(z) * kGroups6 + tid] = 0;
        }
        return;
    }
    if (!CPM_ABL(a, 64)) hi_tree_store<BLOCK, NQ>(tree, pc, Z, a.Zq, H, tid);
    __syncthreads();
    const uint32_t hi_last = tree[0];
    const double *cdf_row = a.cdf_t + static_cast<size_t>(z) * a.Zp;
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t nd = 0;
    long long tt = 0;
    uint32_t *stay_out = GROUPED ? a.ids_next + static_cast<size_t>(z) * cap : nullptr;
    uint32_t *runs = GROUPED ? a.D + static_cast<size_t>(z) * kGroups6 * a.scap : nullptr;

    auto emit = [&](uint32_t q, uint32_t idc, bool valid, bool drive, uint32_t dest) {
        if (GROUPED) {
            const unsigned long long mS = __ballot(valid && !drive);
            uint32_t bS = 0;
            if (lane == 0 && mS) bS = atomicAdd(&s_nstay, static_cast<uint32_t>(__popcll(mS)));
            bS = __shfl(bS, 0, 64);
            if (valid && !drive && !CPM_ABL(a, 4)) stay_out[bS + static_cast<uint32_t>(__popcll(mS & below))] = idc;
            if (drive) {
                const uint32_t g = (dest * a.gmagic) >> 24;
                const uint32_t rank = atomicAdd(&gb[g], 1u);
                const uint32_t packed = idc | ((dest - g * static_cast<uint32_t>(a.zpg)) << a.idbits);
                if (rank < static_cast<uint32_t>(kStage6) && !CPM_ABL(a, 128)) stage[g * kStage6 + rank] = packed;
                else if (rank < a.scap) runs[g * a.scap + rank] = packed;
            }
        } else {
            if (valid) a.rec_out[b + q] = dest | (drive ? kDriveBit : 0u);
        }
    };

    if (CPM_ABL(a, 32)) {
        if (tid == 0) s_nstay = n;
    } else {  // CPT cars per thread, straight line
        bool valid[CPT], drive[CPT], ok[CPT], any_search = false;
        uint32_t dest[CPT], clo[CPT], khi[CPT];
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            valid[c] = static_cast<uint32_t>(tid + c * BLOCK) < n;
            long long kb;
            if (CPM_ABL(a, 1)) {
                clo[c] = id[c] * 2654435761u;
                khi[c] = (id[c] ^ a.step) * 2246822519u;
                kb = static_cast<long long>(clo[c]) << 21;
            } else {
                car_draw_words(a.seed, static_cast<uint64_t>(a.car_begin) + id[c], a.step, kb, clo[c], khi[c]);
            }
            drive[c] = valid[c] && (kb <= thr);
            any_search |= drive[c] && last != 0.0;
        }
        if (CPM_ABL(a, 2)) {
#pragma unroll
            for (int c = 0; c < CPT; ++c) dest[c] = drive[c] ? (khi[c] >> 8) % static_cast<uint32_t>(Z) : z;
        } else if (__any(any_search)) {
            hi_walk<CPT>(tree, khi, Z, H, hi_last, dest, ok);
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                if (!(drive[c] && last != 0.0)) dest[c] = z;  // stays, or zero row: destination = origin (:35-36)
                else if (!ok[c]) dest[c] = search_exact_row(cdf_row, Z, u53(clo[c], khi[c]), last);
            }
        } else {
#pragma unroll
            for (int c = 0; c < CPT; ++c) dest[c] = z;
        }
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            emit(tid + c * BLOCK, id[c], valid[c], drive[c], dest[c]);
            if (drive[c]) {
                ++nd;
                if (TRAVEL) tt += travel_time_q16(a.dm, Z, a.T, a.t, z, dest[c], a.seed, static_cast<uint64_t>(a.car_begin) + id[c], a.step);
            }
        }
    }
    for (uint32_t q0 = CPT * BLOCK; q0 < n && !CPM_ABL(a, 32); q0 += BLOCK) {  // buckets larger than CPT*BLOCK cars (wave-uniform trips)
        const uint32_t q = q0 + tid;
        const bool valid = q < n;
        const uint32_t idx = (q0 == CPT * BLOCK) ? id[CPT] : (valid ? a.ids[b + q] : 0u);
        const uint64_t car = static_cast<uint64_t>(a.car_begin) + idx;
        long long kb;
        uint32_t clo1[1], khi1[1], dest1[1] = {static_cast<uint32_t>(z)};
        bool ok1[1];
        car_draw_words(a.seed, car, a.step, kb, clo1[0], khi1[0]);
        const bool drive = valid && (kb <= thr);
        if (drive && last != 0.0) {
            hi_walk<1>(tree, khi1, Z, H, hi_last, dest1, ok1);
            if (!ok1[0]) dest1[0] = search_exact_row(cdf_row, Z, u53(clo1[0], khi1[0]), last);
        }
        if (drive) {
            if (TRAVEL) tt += travel_time_q16(a.dm, Z, a.T, a.t, z, dest1[0], a.seed, car, a.step);
            ++nd;
        }
        emit(q, idx, valid, drive, dest1[0]);
    }
    for (int o = 32; o > 0; o >>= 1) nd += __shfl_down(nd, o, 64);
    if (lane == 0 && nd) atomicAdd(&s_ndrive, nd);
    if (TRAVEL) {
        for (int o = 32; o > 0; o >>= 1) tt += __shfl_down(tt, o, 64);
        if (lane == 0 && tt) atomicAdd(&s_tt, static_cast<unsigned long long>(tt));
    }
    __syncthreads();  // ranks, staged drivers and counters are final
    if (GROUPED) {
        // staged drivers -> their runs: 16 lanes per group, 64 B per store
        for (int g = tid >> 4; g < kGroups6; g += BLOCK / 16) {
            const uint32_t lim = min(gb[g], static_cast<uint32_t>(kStage6));
            for (uint32_t i = tid & 15; i < lim; i += 16) runs[g * a.scap + i] = stage[g * kStage6 + i];
        }
        if (tid < kGroups6) {
            const uint32_t c = gb[tid];
            a.cntg[static_cast<size_t>(z) * kGroups6 + tid] = min(c, a.scap);
            if (c > a.scap) atomicOr(a.status, 2ull);  // a run outgrew its slot: the caller repeats on the exact layout
        }
    }
    if (tid == 0) {
        a.driving_t[z] = s_ndrive;
        if (GROUPED) a.cnt_next[z] = s_nstay;  // k_zone6_place adds the arrivals
        if (TRAVEL && s_tt) atomicAdd(a.tt_sum, s_tt);
    }
}

// Pipelined form of k_zone6_sample (GROUPED only): as many workgroups as are resident at once, each walking the zones
// z = blockIdx.x, blockIdx.x + gridDim.x, ...  While a zone is being sampled, the ids, row and scalars of the
// workgroup's NEXT zone are already on their way into registers, so no HBM round trip sits between two zones
// (with one workgroup per zone a zone's time was its chain of round trips -- size/ids/row in, stores drained before
// s_endpgm -- times Z / resident workgroups).  Every global store of a zone is issued in its flush phase, AFTER the
// next zone's registers have been consumed: the sampling phase keeps stayers and drivers in LDS (kStay7 / kStage6 entries
// per zone / per group; what does not fit goes to HBM directly), so the wait for the prefetched registers never
// covers a store that has just been issued (vmcnt retires in order).
constexpr int kStay7 = 1536;

template <bool TRAVEL, int BLOCK, int NQ, int CPT, int WPS>
__global__ __launch_bounds__(BLOCK, TRAVEL ? 2 : WPS) void k_zone7_sample(Zone6Args a)
{
    extern __shared__ uint32_t tree[];  // 2^H high words
    __shared__ uint32_t s_ndrive[2], s_nstay[2];
    __shared__ unsigned long long s_tt[2];
    __shared__ uint32_t gb[2][kGroups6];
    __shared__ uint32_t stage[kGroups6 * kStage6];
    __shared__ uint32_t stay[kStay7];
    const int Z = a.Z, H = a.H;
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t cap = a.cap;
    const unsigned long long below = (1ull << lane) - 1ull;
    const int stride = gridDim.x;
    int z = blockIdx.x;
    if (z >= Z) return;
    // registers of the zone about to be sampled (prefetch target)
    uint32_t idn[CPT + 1];
    uint4 pc[NQ];
    uint32_t n_raw;
    double last_raw, pd_raw;
    auto prefetch = [&](int zz) {
        const uint32_t bb = static_cast<uint32_t>(zz) * cap;
#pragma unroll
        for (int c = 0; c <= CPT; ++c) idn[c] = a.ids[bb + min(static_cast<uint32_t>(tid + c * BLOCK), cap - 1)];
        hi_row_load<BLOCK, NQ>(pc, a.hi_t + static_cast<size_t>(zz) * a.Zq, a.Zq, tid);
        n_raw = a.cnt[zz];
        last_raw = a.last_t[zz];
        pd_raw = a.pdrive_t[zz];
    };
    prefetch(z);
    int par = 0;
    // consume the prefetched registers of zone z: tree, counters, scalars
    uint32_t id[CPT + 1], n;
    double last;
    long long thr;
    auto consume = [&]() {
        n = min(n_raw, cap);
        last = last_raw;
        thr = bernoulli_threshold(pd_raw);
#pragma unroll
        for (int c = 0; c <= CPT; ++c) id[c] = idn[c];
        hi_tree_store<BLOCK, NQ>(tree, pc, Z, a.Zq, H, tid);
        if (tid == 0) {
            s_ndrive[par] = 0;
            s_nstay[par] = 0;
            s_tt[par] = 0;
        }
        if (tid < kGroups6) gb[par][tid] = 0;
    };
    consume();
    {
        const int zn = z + stride;
        prefetch(zn < Z ? zn : z);  // (a harmless repeat of the same zone at the end keeps the code path uniform)
    }
    while (true) {
        __syncthreads();  // A: tree and counters of zone z are in place
        const uint32_t b = static_cast<uint32_t>(z) * cap;
        const uint32_t hi_last = tree[0];
        const double *cdf_row = a.cdf_t + static_cast<size_t>(z) * a.Zp;
        uint32_t nd = 0;
        long long tt = 0;
        uint32_t *stay_out = a.ids_next + static_cast<size_t>(z) * cap;
        uint32_t *runs = a.D + static_cast<size_t>(z) * kGroups6 * a.scap;

        auto emit = [&](uint32_t idc, bool valid, bool drive, uint32_t dest) {
            const unsigned long long mS = __ballot(valid && !drive);
            uint32_t bS = 0;
            if (lane == 0 && mS) bS = atomicAdd(&s_nstay[par], static_cast<uint32_t>(__popcll(mS)));
            bS = __shfl(bS, 0, 64);
            if (valid && !drive) {
                const uint32_t pos = bS + static_cast<uint32_t>(__popcll(mS & below));
                if (pos < static_cast<uint32_t>(kStay7)) stay[pos] = idc;
                else stay_out[pos] = idc;
            }
            if (drive) {
                const uint32_t g = (dest * a.gmagic) >> 24;
                const uint32_t rank = atomicAdd(&gb[par][g], 1u);
                const uint32_t packed = idc | ((dest - g * static_cast<uint32_t>(a.zpg)) << a.idbits);
                if (rank < static_cast<uint32_t>(kStage6)) stage[g * kStage6 + rank] = packed;
                else if (rank < a.scap) runs[g * a.scap + rank] = packed;
            }
        };

        if (n > 0) {
            bool valid[CPT], drive[CPT], ok[CPT], any_search = false;
            uint32_t dest[CPT], clo[CPT], khi[CPT];
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                valid[c] = static_cast<uint32_t>(tid + c * BLOCK) < n;
                long long kb;
                car_draw_words(a.seed, static_cast<uint64_t>(a.car_begin) + id[c], a.step, kb, clo[c], khi[c]);
                drive[c] = valid[c] && (kb <= thr);
                any_search |= drive[c] && last != 0.0;
            }
            if (__any(any_search)) {
                hi_walk<CPT>(tree, khi, Z, H, hi_last, dest, ok);
#pragma unroll
                for (int c = 0; c < CPT; ++c) {
                    if (!(drive[c] && last != 0.0)) dest[c] = z;  // stays, or zero row: destination = origin (:35-36)
                    else if (!ok[c]) dest[c] = search_exact_row(cdf_row, Z, u53(clo[c], khi[c]), last);
                }
            } else {
#pragma unroll
                for (int c = 0; c < CPT; ++c) dest[c] = z;
            }
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                emit(id[c], valid[c], drive[c], dest[c]);
                if (drive[c]) {
                    ++nd;
                    if (TRAVEL) tt += travel_time_q16(a.dm, Z, a.T, a.t, z, dest[c], a.seed, static_cast<uint64_t>(a.car_begin) + id[c], a.step);
                }
            }
            for (uint32_t q0 = CPT * BLOCK; q0 < n; q0 += BLOCK) {  // buckets larger than CPT*BLOCK cars (wave-uniform trips)
                const uint32_t q = q0 + tid;
                const bool valid1 = q < n;
                const uint32_t idx = (q0 == CPT * BLOCK) ? id[CPT] : (valid1 ? a.ids[b + q] : 0u);
                const uint64_t car = static_cast<uint64_t>(a.car_begin) + idx;
                long long kb;
                uint32_t clo1[1], khi1[1], dest1[1] = {static_cast<uint32_t>(z)};
                bool ok1[1];
                car_draw_words(a.seed, car, a.step, kb, clo1[0], khi1[0]);
                const bool drive1 = valid1 && (kb <= thr);
                if (drive1 && last != 0.0) {
                    hi_walk<1>(tree, khi1, Z, H, hi_last, dest1, ok1);
                    if (!ok1[0]) dest1[0] = search_exact_row(cdf_row, Z, u53(clo1[0], khi1[0]), last);
                }
                if (drive1) {
                    if (TRAVEL) tt += travel_time_q16(a.dm, Z, a.T, a.t, z, dest1[0], a.seed, car, a.step);
                    ++nd;
                }
                emit(idx, valid1, drive1, dest1[0]);
            }
            for (int o = 32; o > 0; o >>= 1) nd += __shfl_down(nd, o, 64);
            if (lane == 0 && nd) atomicAdd(&s_ndrive[par], nd);
            if (TRAVEL) {
                for (int o = 32; o > 0; o >>= 1) tt += __shfl_down(tt, o, 64);
                if (lane == 0 && tt) atomicAdd(&s_tt[par], static_cast<unsigned long long>(tt));
            }
        }
        __syncthreads();  // B: ranks, staged cars and counters of zone z are final; nobody reads the tree any more
        const int zdone = z, pdone = par;
        const uint32_t ndone = n;
        const int znext = z + stride;
        const bool more = znext < Z;
        if (more) {  // the next zone's registers -> tree / counters (other set), then ITS next zone's loads
            par ^= 1;
            z = znext;
            consume();
            const int zn2 = z + stride;
            prefetch(zn2 < Z ? zn2 : z);
        }
        // flush zone zdone: every global store of the zone happens here
        {
            uint32_t *so = a.ids_next + static_cast<size_t>(zdone) * cap;
            uint32_t *ru = a.D + static_cast<size_t>(zdone) * kGroups6 * a.scap;
            const uint32_t ns = min(s_nstay[pdone], static_cast<uint32_t>(kStay7));
            for (uint32_t i = tid; i < ns; i += BLOCK) so[i] = stay[i];
            for (int g = tid >> 4; g < kGroups6; g += BLOCK / 16) {
                const uint32_t lim = min(gb[pdone][g], static_cast<uint32_t>(kStage6));
                for (uint32_t i = tid & 15; i < lim; i += 16) ru[g * a.scap + i] = stage[g * kStage6 + i];
            }
            if (tid < kGroups6) {
                const uint32_t c = gb[pdone][tid];
                a.cntg[static_cast<size_t>(zdone) * kGroups6 + tid] = min(c, a.scap);
                if (c > a.scap) atomicOr(a.status, 2ull);
            }
            if (tid == 0) {
                a.parking_t[zdone] = ndone;
                a.driving_t[zdone] = s_ndrive[pdone];
                a.cnt_next[zdone] = s_nstay[pdone];  // k_zone6_place adds the arrivals
                if (TRAVEL && s_tt[pdone]) atomicAdd(a.tt_sum, s_tt[pdone]);
            }
        }
        if (!more) break;
    }
}

// Drivers of destination group g -> their buckets.  blockIdx = j * kGroups6 + g: the blocks of a group share
// blockIdx % 8 (one XCD, one L2: all writes to a bucket merge there; speed only, never correctness).  Block
// (g, j) takes the group-g runs of the origin zones [j*zps, (j+1)*zps): 16 lanes per run, KDEEP entries per
// lane.  Run lengths and run contents sit at addresses known up front, so they are requested together.
constexpr int kPlace6Block = 1024;
constexpr int kPlace6Seg = kPlace6Block / 16;

template <int KRUNS, int KDEEP>
__global__ __launch_bounds__(kPlace6Block) void k_zone6_place(const uint32_t *__restrict__ D, const uint32_t *__restrict__ cntg,
                                                              int zpg, int zps, int Z, uint32_t cap, uint32_t scap, uint32_t idbits,
                                                              uint32_t *__restrict__ cnt_next, uint32_t *__restrict__ ids_next,
                                                              unsigned long long *status)
{
    __shared__ uint32_t bins[kMaxZonesPerGroup6];
    const int tid = threadIdx.x;
    const int g = blockIdx.x % kGroups6, j = blockIdx.x / kGroups6;
    const int zg0 = g * zpg;
    const int nzl = max(0, min(zpg, Z - zg0));
    const int zs0 = j * zps, zs1 = min(Z, zs0 + zps);
    const int sub = tid >> 4, l16 = tid & 15;
    const uint32_t idmask = (idbits >= 32) ? 0xFFFFFFFFu : ((1u << idbits) - 1u);
    for (int k = tid; k < kMaxZonesPerGroup6; k += kPlace6Block) bins[k] = 0;
    if (zs0 >= zs1) return;  // (uniform per block)
    uint32_t c[KRUNS], v[KRUNS][KDEEP];
#pragma unroll
    for (int k = 0; k < KRUNS; ++k) {
        const int zs = zs0 + sub + k * kPlace6Seg;
        const int zc = min(zs, zs1 - 1);
        const size_t run = static_cast<size_t>(zc) * kGroups6 + g;
        c[k] = cntg[run];
        if (zs >= zs1) c[k] = 0;
#pragma unroll
        for (int d = 0; d < KDEEP; ++d) v[k][d] = D[run * scap + l16 + 16 * d];  // scap >= 16 * KDEEP; beyond c[k]: stale, masked
    }
    __syncthreads();
    // pass 1: histogram of the destinations over the group's zones
#pragma unroll
    for (int k = 0; k < KRUNS; ++k) {
#pragma unroll
        for (int d = 0; d < KDEEP; ++d)
            if (static_cast<uint32_t>(l16 + 16 * d) < c[k]) atomicAdd(&bins[v[k][d] >> idbits], 1u);
    }
#pragma unroll
    for (int k = 0; k < KRUNS; ++k) {
        const int zc = min(zs0 + sub + k * kPlace6Seg, zs1 - 1);
        const size_t run = static_cast<size_t>(zc) * kGroups6 + g;
        for (uint32_t i = l16 + 16 * KDEEP; i < c[k]; i += 16) atomicAdd(&bins[D[run * scap + i] >> idbits], 1u);
    }
    __syncthreads();
    if (tid < nzl) {  // ticket: this block's range inside each bucket of the group
        const uint32_t cc = bins[tid];
        uint32_t base = 0;
        if (cc) {
            base = atomicAdd(&cnt_next[zg0 + tid], cc);
            if (base + cc > cap) atomicOr(status, 2ull);
        }
        bins[tid] = base;
    }
    __syncthreads();
    // pass 2: the ids move
#pragma unroll
    for (int k = 0; k < KRUNS; ++k) {
#pragma unroll
        for (int d = 0; d < KDEEP; ++d)
            if (static_cast<uint32_t>(l16 + 16 * d) < c[k]) {
                const uint32_t dl = v[k][d] >> idbits;
                const uint32_t p = atomicAdd(&bins[dl], 1u);
                if (p < cap) ids_next[static_cast<size_t>(zg0 + dl) * cap + p] = v[k][d] & idmask;
            }
    }
#pragma unroll
    for (int k = 0; k < KRUNS; ++k) {
        const int zc = min(zs0 + sub + k * kPlace6Seg, zs1 - 1);
        const size_t run = static_cast<size_t>(zc) * kGroups6 + g;
        for (uint32_t i = l16 + 16 * KDEEP; i < c[k]; i += 16) {
            const uint32_t w = D[run * scap + i];
            const uint32_t dl = w >> idbits;
            const uint32_t p = atomicAdd(&bins[dl], 1u);
            if (p < cap) ids_next[static_cast<size_t>(zg0 + dl) * cap + p] = w & idmask;
        }
    }
}

inline void zone6_launch_place(hipStream_t stream, int bpg, const uint32_t *D, const uint32_t *cntg, int zpg, int Z, uint32_t cap,
                               uint32_t scap, uint32_t idbits, uint32_t *cnt_next, uint32_t *ids_next, unsigned long long *status)
{
    const int zps = (Z + bpg - 1) / bpg;
    const dim3 grid(kGroups6 * bpg), block(kPlace6Block);
    if (zps <= 4 * kPlace6Seg)
        hipLaunchKernelGGL((k_zone6_place<4, 2>), grid, block, 0, stream, D, cntg, zpg, zps, Z, cap, scap, idbits, cnt_next, ids_next, status);
    else
        hipLaunchKernelGGL((k_zone6_place<8, 2>), grid, block, 0, stream, D, cntg, zpg, zps, Z, cap, scap, idbits, cnt_next, ids_next, status);
}

template <bool TRAVEL, int BLOCK, int CPT, bool GROUPED, int NQ>
inline void zone6_launch_nq(const Zone6Args &a, size_t lds, hipStream_t stream)
{
    static bool attr_done = false;
    if (!attr_done && lds > 48 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_zone6_sample<TRAVEL, BLOCK, NQ, CPT, GROUPED>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    hipLaunchKernelGGL((k_zone6_sample<TRAVEL, BLOCK, NQ, CPT, GROUPED>), dim3(a.Z), dim3(BLOCK), lds, stream, a);
}

template <bool TRAVEL, int BLOCK, int CPT, bool GROUPED>
inline void zone6_launch_b(const Zone6Args &a, hipStream_t stream)
{
    const size_t lds = sizeof(uint32_t) * (size_t(1) << a.H);
    const int need = (a.Zq / 4 + BLOCK - 1) / BLOCK;
    if (need <= 1) zone6_launch_nq<TRAVEL, BLOCK, CPT, GROUPED, 1>(a, lds, stream);
    else if (need <= 2) zone6_launch_nq<TRAVEL, BLOCK, CPT, GROUPED, 2>(a, lds, stream);
    else if (need <= 4) zone6_launch_nq<TRAVEL, BLOCK, CPT, GROUPED, 4>(a, lds, stream);
    else if (need <= 8) zone6_launch_nq<TRAVEL, BLOCK, CPT, GROUPED, 8>(a, lds, stream);
    else zone6_launch_nq<TRAVEL, BLOCK, CPT, GROUPED, 16>(a, lds, stream);
}

template <bool TRAVEL, int NQ, int WPS>
inline void zone7_launch_nq(const Zone6Args &a, size_t lds, int cu_count, hipStream_t stream)
{
    static bool attr_done = false;
    static int resident = 0;
    const void *fn = reinterpret_cast<const void *>(k_zone7_sample<TRAVEL, 512, NQ, 2, WPS>);
    if (!attr_done && lds > 32 * 1024) {
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
        attr_done = true;
    }
    if (resident == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 512, lds) != hipSuccess || nb < 1) nb = 1;
        resident = nb;
    }
    const int grid = static_cast<int>(std::min<int64_t>(a.Z, static_cast<int64_t>(resident) * cu_count));
    hipLaunchKernelGGL((k_zone7_sample<TRAVEL, 512, NQ, 2, WPS>), dim3(grid), dim3(512), lds, stream, a);
}

template <bool TRAVEL, int WPS>
inline void zone7_launch(const Zone6Args &a, int cu_count, hipStream_t stream)
{
    const size_t lds = sizeof(uint32_t) * (size_t(1) << a.H);
    const int need = (a.Zq / 4 + 511) / 512;
    if (need <= 1) zone7_launch_nq<TRAVEL, 1, WPS>(a, lds, cu_count, stream);
    else if (need <= 2) zone7_launch_nq<TRAVEL, 2, WPS>(a, lds, cu_count, stream);
    else if (need <= 4) zone7_launch_nq<TRAVEL, 4, WPS>(a, lds, cu_count, stream);
    else if (need <= 8) zone7_launch_nq<TRAVEL, 8, WPS>(a, lds, cu_count, stream);
    else zone7_launch_nq<TRAVEL, 16, WPS>(a, lds, cu_count, stream);
}

// shape: 0 = 512 threads x 2 cars, 1 = 256 threads x 4 cars (travel-time form: 512 x 2 only), 2 = pipelined resident
// workgroups (k_zone7_sample; GROUPED only)
template <bool GROUPED>
inline void zone6_launch(const Zone6Args &a, bool travel, int shape, int cu_count, hipStream_t stream)
{
    if (GROUPED && shape >= 2) {
        if (travel) zone7_launch<true, 2>(a, cu_count, stream);
        else if (shape == 2) zone7_launch<false, 8>(a, cu_count, stream);
        else if (shape == 3) zone7_launch<false, 6>(a, cu_count, stream);
        else zone7_launch<false, 4>(a, cu_count, stream);
    } else if (travel) zone6_launch_b<true, 512, 2, GROUPED>(a, stream);
    else if (shape == 1 && a.Zq / 4 <= 16 * 256) zone6_launch_b<false, 256, 4, GROUPED>(a, stream);
    else zone6_launch_b<false, 512, 2, GROUPED>(a, stream);
}

// rows of at most 16 pieces x 512 threads x 4 words, trees of at most 128 KiB
inline bool zone6_row_fits(int Z) { return Z >= 2 && Z <= 32768; }

// Diagnostic (cpm_debug_categorical): the categorical draw of the sampler for given 53-bit draws k against one
// installed row, through the same staging, walk and exact-row code.  out[i] = destination (1-based), or 0 for a zero row.
__global__ __launch_bounds__(512) void k_zone6_search_debug(const uint32_t *__restrict__ hi_row, const double *__restrict__ last_p,
                                                            const double *__restrict__ cdf_row, int Z, int Zq, int H, int64_t n,
                                                            const uint64_t *__restrict__ k53, int64_t *__restrict__ out,
                                                            int *__restrict__ n_exact)
{
    extern __shared__ uint32_t tree[];
    const int tid = threadIdx.x;
    const uint4 *src = reinterpret_cast<const uint4 *>(hi_row);
    for (int j = tid; j < Zq / 4; j += 512) {
        uint4 pc[1] = {src[j]};
        const uint32_t q[4] = {pc[0].x, pc[0].y, pc[0].z, pc[0].w};
        for (int k = 0; k < 4; ++k)
            if (4 * j + k < Z) tree[eytz_pos(4 * j + k, Z, H)] = q[k];
    }
    for (int r = Z + tid; r < (1 << H); r += 512) {
        int tz = __builtin_ctz(static_cast<unsigned>(r));
        tree[(1u << (H - 1 - tz)) + (static_cast<unsigned>(r) >> (tz + 1))] = kHiMax;
    }
    __syncthreads();
    const double last = *last_p;
    const uint32_t hi_last = tree[0];
    for (int64_t i = tid; i < n; i += 512) {
        const uint64_t k = k53[i] & ((1ull << 53) - 1ull);
        const uint32_t khi[1] = {static_cast<uint32_t>(k >> 21)};
        uint32_t dest[1];
        bool ok[1];
        if (last == 0.0) {
            out[i] = 0;
            continue;
        }
        hi_walk<1>(tree, khi, Z, H, hi_last, dest, ok);
        if (!ok[0]) {
            dest[0] = search_exact_row(cdf_row, Z, static_cast<double>(k) * 0x1.0p-53, last);
            atomicAdd(n_exact, 1);
        }
        out[i] = static_cast<int64_t>(dest[0]) + 1;
    }
}

}  // namespace cpm
