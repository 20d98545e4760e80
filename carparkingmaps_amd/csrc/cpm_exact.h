// cpm_exact.h -- CPM_KERNEL_ZONE_LDS: the zone path on the exact (packed) bucket layout.  The fallback of the grouped path
// (cpm_grouped.h) when a bucket or run outgrows its fixed region and the regions cannot grow any further, and an A/B reference.
//
// The reference walks p_dest[origin,:,t] once per driving car (src/resampling.jl:34-45).  Here
// the cars of one origin zone sit together, so the row is streamed from HBM exactly once per
// hour (coalesced, 16 B per lane), kept in LDS, and every car of the zone searches it there.
// Per hour t:
//   k_exact_sample  one workgroup per origin zone: stage cdf[t][zone][:] in LDS (as a breadth-first
//                   search tree); for each car of the bucket: Philox -> Bernoulli (:11-22) -> categorical
//                   by tree walk in LDS (:26-49); writes dest|drive per slot, parking[t][zone] = bucket
//                   size, driving[t][zone] = drivers (src/saveresults.jl:10-15) -- no histogram atomics.
//   k_zone_hist     counting sort, pass 1: LDS-privatised histogram of the destinations of a
//                   contiguous chunk of slots; one contiguous global atomic per (block, zone)
//                   reserves the block's range in the zone's next bucket (ticket).
//   k_zone_scatter  counting sort, pass 2: every block scans the Z bucket sizes itself (16 KB,
//                   L2-resident), then moves its car ids to their next-hour buckets.
// Buckets are packed (exclusive scan of their sizes), so nothing can overflow.  The bucket sizes ARE the
// parking histogram of the next hour.  The order of ids inside a bucket is arbitrary (it depends on atomic
// arrival order) and does not matter: a car's draw depends only on (seed, global car id, step) and integer
// counts are order-free.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <string>

#include "../../include/cpm.h"
#include "cpm_kernels.h"

namespace cpm {

constexpr int kZoneBlock = 256;
constexpr int kSortBlock = 1024;

// The zone sampler of the exact layout:
//   * the car ids of the bucket are requested BEFORE the row (vmcnt retires in order: by the time
//     the row has landed and the barrier has passed they are there, instead of one exposed HBM
//     round trip per round of cars);
//   * the row is written into LDS as a breadth-first search tree (see cpm_kernels.h);
//   * every thread carries CPT cars through straight-line code: independent Philox chains and
//     tree walks interleave (CPT LDS reads in flight per lane); CPT = 2 measured best;
//   * Bernoulli draw in integers (k <= floor(p * 2^53)).
// NP: 16-B row pieces per thread (>= Zp / 2 / BLOCK).
template <bool TRAVEL, int BLOCK, int NP, int CPT>
__global__ __launch_bounds__(BLOCK) void k_exact_sample(
    const uint32_t *__restrict__ ids, const uint32_t *__restrict__ off, uint32_t *__restrict__ dest_out,
    const double *__restrict__ pdrive_t, const double *__restrict__ cdf_t, int Z, int Zp, int H, CarIndex cars,
    uint32_t step, uint64_t seed, unsigned long long *__restrict__ parking_t,
    unsigned long long *__restrict__ driving_t, const double *__restrict__ dm, int T, int t,
    unsigned long long *tt_sum)
{
    extern __shared__ double row[];  // 2^H doubles: the zone's CDF row as a search tree
    __shared__ uint32_t s_ndrive;
    __shared__ unsigned long long s_tt;
    const int z = blockIdx.x;
    const int tid = threadIdx.x;
    const uint32_t b = off[z];  // bucket z = [off[z], off[z+1])
    const uint32_t n = off[z + 1] - b;
    const uint32_t e = b + n;
    if (tid == 0) {
        parking_t[z] = n;  // every car present at hour t, drivers included (Appendix A-14)
        s_ndrive = 0;
        s_tt = 0;
    }
    if (n == 0) return;  // driving_t[z] stays 0 (zeroed by the caller)
    uint32_t id[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) id[c] = ids[b + min(static_cast<uint32_t>(tid + c * BLOCK), n - 1)];
    // the id of the first overflow round rides along (inside the overflow loop it would be an exposed HBM round trip)
    const uint32_t id_x = ids[b + min(static_cast<uint32_t>(tid + CPT * BLOCK), n - 1)];
    double2 pc[NP];
    const double2 *src = reinterpret_cast<const double2 *>(cdf_t + static_cast<size_t>(z) * Zp);
#pragma unroll
    for (int m = 0; m < NP; ++m) pc[m] = src[min(tid + m * BLOCK, Zp / 2 - 1)];
    const long long thr = bernoulli_threshold(pdrive_t[z]);
    const int P = 1 << H;
    for (int r = Z + tid; r < P; r += BLOCK) {  // ranks Z..2^H-1: +inf
        int tz = __builtin_ctz(static_cast<unsigned>(r));
        row[(1u << (H - 1 - tz)) + (static_cast<unsigned>(r) >> (tz + 1))] = __builtin_huge_val();
    }
#pragma unroll
    for (int m = 0; m < NP; ++m) {
        int j = tid + m * BLOCK;
        if (2 * j < Zp) {
            uint32_t el = 2 * j;
            if (el < static_cast<uint32_t>(Z)) row[eytz_pos(el, Z, H)] = pc[m].x;
            if (el + 1 < static_cast<uint32_t>(Z)) row[eytz_pos(el + 1, Z, H)] = pc[m].y;
        }
    }
    __syncthreads();
    const double last = row[0];
    uint32_t nd = 0;
    long long tt = 0;
    {  // CPT cars per thread, straight line
        bool valid[CPT], drive[CPT], any_search = false;
        uint32_t dest[CPT];
        double ue[CPT];
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            valid[c] = static_cast<uint32_t>(tid + c * BLOCK) < n;
            long long kb;
            double uc;
            car_draws(seed, cars.global(id[c]), step, kb, uc);
            drive[c] = valid[c] && (kb <= thr);
            dest[c] = z;
            ue[c] = clamp_u(uc, last);
            any_search |= drive[c] && last != 0.0;
        }
        if (__any(any_search)) {
            uint32_t i[CPT];
#pragma unroll
            for (int c = 0; c < CPT; ++c) i[c] = 1;
            for (int l = 0; l < H; ++l) {
                double k[CPT];
#pragma unroll
                for (int c = 0; c < CPT; ++c) k[c] = row[i[c]];
#pragma unroll
                for (int c = 0; c < CPT; ++c) i[c] = 2 * i[c] + (k[c] < ue[c] ? 1u : 0u);
            }
#pragma unroll
            for (int c = 0; c < CPT; ++c)
                if (drive[c] && last != 0.0) dest[c] = eytz_decode(i[c], Z, H);
        }
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            if (valid[c]) dest_out[b + tid + c * BLOCK] = dest[c] | (drive[c] ? kDriveBit : 0u);
            if (drive[c]) {
                ++nd;
                if (TRAVEL) tt += travel_time_q16(dm, Z, T, t, z, dest[c], seed, cars.global(id[c]), step);
            }
        }
    }
    for (uint32_t s = b + CPT * BLOCK + tid; s < e; s += BLOCK) {  // buckets larger than CPT*BLOCK cars
        const uint32_t idx = (s < b + (CPT + 1) * BLOCK) ? id_x : ids[s];
        const uint64_t car = cars.global(idx);
        long long kb;
        double uc;
        car_draws(seed, car, step, kb, uc);
        const bool drive = kb <= thr;
        uint32_t dest = z;
        if (drive) {
            if (last != 0.0) {
                const double u1 = clamp_u(uc, last);
                uint32_t i = 1;
                for (int l = 0; l < H; ++l) i = 2 * i + (row[i] < u1 ? 1u : 0u);
                dest = eytz_decode(i, Z, H);
            }
            if (TRAVEL) tt += travel_time_q16(dm, Z, T, t, z, dest, seed, car, step);
            ++nd;
        }
        dest_out[s] = dest | (drive ? kDriveBit : 0u);
    }
    for (int o = 32; o > 0; o >>= 1) nd += __shfl_down(nd, o, 64);
    if ((tid & 63) == 0 && nd) atomicAdd(&s_ndrive, nd);
    if (TRAVEL) {
        for (int o = 32; o > 0; o >>= 1) tt += __shfl_down(tt, o, 64);
        if ((tid & 63) == 0 && tt) atomicAdd(&s_tt, static_cast<unsigned long long>(tt));
    }
    __syncthreads();
    if (tid == 0) {
        driving_t[z] = s_ndrive;
        if (TRAVEL && s_tt) atomicAdd(tt_sum, s_tt);
    }
}

// Wave-aggregated LDS atomics.  The slots of a chunk are ordered by origin zone and about half
// the cars stay where they are, so within one wave-instruction dozens of lanes carry the SAME
// key (their own zone, drive flag clear): plain LDS atomics on one address serialise (measured:
// ~140 cycles per wave-instruction).  Lanes with the flag clear are therefore grouped by key with
// ballots (at most two rounds: a wave's slots span one or two zones); one lane adds the group
// size.  Drivers' keys are scattered over all zones and go one by one.
__device__ __forceinline__ void wave_hist_add(uint32_t *bins, uint32_t keyflag, bool valid)
{
    const int lane = threadIdx.x & 63;
    const uint32_t key = keyflag & kZoneMask;
    bool agg = valid && !(keyflag & kDriveBit);
    unsigned long long todo = __ballot(agg);
    for (int it = 0; it < 2 && todo; ++it) {
        int first = __ffsll(static_cast<long long>(todo)) - 1;
        uint32_t k = __shfl(key, first, 64);
        unsigned long long same = __ballot(agg && key == k) & todo;
        if (lane == first) atomicAdd(&bins[k], static_cast<uint32_t>(__popcll(same)));
        todo &= ~same;
    }
    bool single = valid && ((keyflag & kDriveBit) || ((todo >> lane) & 1ull));
    if (single) atomicAdd(&bins[key], 1u);
}

__device__ __forceinline__ uint32_t wave_slot_take(uint32_t *pos, uint32_t keyflag, bool valid)
{
    const int lane = threadIdx.x & 63;
    const uint32_t key = keyflag & kZoneMask;
    bool agg = valid && !(keyflag & kDriveBit);
    unsigned long long todo = __ballot(agg);
    uint32_t p = 0;
    for (int it = 0; it < 2 && todo; ++it) {
        int first = __ffsll(static_cast<long long>(todo)) - 1;
        uint32_t k = __shfl(key, first, 64);
        unsigned long long same = __ballot(agg && key == k) & todo;
        uint32_t b = 0;
        if (lane == first) b = atomicAdd(&pos[k], static_cast<uint32_t>(__popcll(same)));
        b = __shfl(b, first, 64);
        if ((same >> lane) & 1ull) p = b + static_cast<uint32_t>(__popcll(same & ((1ull << lane) - 1ull)));
        todo &= ~same;
    }
    bool single = valid && ((keyflag & kDriveBit) || ((todo >> lane) & 1ull));
    if (single) p = atomicAdd(&pos[key], 1u);
    return p;
}

// counting sort pass 1.  key[i] & kZoneMask = bucket of slot i.  chunk is a multiple of 4 and
// key is 16-B aligned: 16 B per lane per load, all of a thread's loads issued before its atomics.
constexpr int kSortUnroll = 4;

__global__ __launch_bounds__(kSortBlock) void k_zone_hist(const uint32_t *__restrict__ key, int64_t n, int Z,
                                                          int64_t chunk, uint32_t *__restrict__ cursor,
                                                          uint32_t *__restrict__ base)
{
    extern __shared__ uint32_t bins[];  // Z
    for (int z = threadIdx.x; z < Z; z += kSortBlock) bins[z] = 0;
    __syncthreads();
    const int64_t i0 = static_cast<int64_t>(blockIdx.x) * chunk, i1 = min(i0 + chunk, n);
    const int64_t n4 = (i1 > i0) ? (i1 - i0) / 4 : 0;
    const uint4 *k4 = reinterpret_cast<const uint4 *>(key + i0);
    for (int64_t j0 = 0; j0 < n4; j0 += kSortBlock * kSortUnroll) {  // wave-uniform trip count
        const int64_t j = j0 + threadIdx.x;
        uint4 v[kSortUnroll];
#pragma unroll
        for (int u = 0; u < kSortUnroll; ++u)
            if (j + u * kSortBlock < n4) v[u] = k4[j + u * kSortBlock];
#pragma unroll
        for (int u = 0; u < kSortUnroll; ++u) {
            const bool ok = j + u * kSortBlock < n4;
            wave_hist_add(bins, v[u].x, ok);
            wave_hist_add(bins, v[u].y, ok);
            wave_hist_add(bins, v[u].z, ok);
            wave_hist_add(bins, v[u].w, ok);
        }
    }
    for (int64_t i = i0 + 4 * n4 + threadIdx.x; i < i1; i += kSortBlock) atomicAdd(&bins[key[i] & kZoneMask], 1u);
    __syncthreads();
    // ticket: this block's range inside every bucket.  The four returning atomics of a thread are
    // issued before any result is stored, so they share one HBM-side round trip.  (Out-of-range
    // lanes must NOT add 0 to a clamped address: at Z = 2357 that put 445k atomics on one word.)
    uint32_t *mybase = base + static_cast<size_t>(blockIdx.x) * Z;
    for (int zb = 0; zb < Z; zb += kSortBlock * 4) {
        uint32_t r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int z = zb + threadIdx.x + k * kSortBlock;
            r[k] = 0;
            if (z < Z) {
                const uint32_t c = bins[z];
                if (c) r[k] = atomicAdd(&cursor[z], c);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int z = zb + threadIdx.x + k * kSortBlock;
            if (z < Z) mybase[z] = r[k];
        }
    }
}

// counting sort pass 2.  ids == nullptr: slot index is the car id (initial bucketing).
__global__ __launch_bounds__(kSortBlock) void k_zone_scatter(const uint32_t *__restrict__ key,
                                                             const uint32_t *__restrict__ ids, int64_t n, int Z,
                                                             int64_t chunk, const uint32_t *__restrict__ cursor,
                                                             const uint32_t *__restrict__ base,
                                                             uint32_t *__restrict__ ids_next,
                                                             uint32_t *__restrict__ off_next)
{
    extern __shared__ uint32_t pos[];  // Z
    __shared__ uint32_t wsum[kSortBlock / 64];
    const int tid = threadIdx.x;
    // exclusive scan of the Z bucket sizes, redone by every block (Z*4 B from L2): thread t takes
    // zones t, t+1024, ... so that its loads of cursor[] and base[] are coalesced and all issued
    // before the first use; one block scan per 1024-zone chunk, carry in a register.
    const uint32_t *mybase = base + static_cast<size_t>(blockIdx.x) * Z;
    uint32_t carry = 0;
    for (int zb = 0; zb < Z; zb += kSortBlock * 4) {
        uint32_t cnt[4], bs[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int z = zb + tid + k * kSortBlock;
            const int zc = min(z, Z - 1);
            cnt[k] = cursor[zc];
            bs[k] = mybase[zc];
            if (z >= Z) cnt[k] = 0;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int z = zb + tid + k * kSortBlock;
            uint32_t incl = cnt[k];
            for (int o = 1; o < 64; o <<= 1) {
                uint32_t v = __shfl_up(incl, o, 64);
                if ((tid & 63) >= o) incl += v;
            }
            __syncthreads();  // wsum free
            if ((tid & 63) == 63) wsum[tid >> 6] = incl;
            __syncthreads();
            uint32_t wbase = 0, tot = 0;
#pragma unroll
            for (int w = 0; w < kSortBlock / 64; ++w) {
                const uint32_t sw = wsum[w];
                if (w < (tid >> 6)) wbase += sw;
                tot += sw;
            }
            const uint32_t excl = carry + wbase + incl - cnt[k];
            if (z < Z) {
                pos[z] = excl + bs[k];
                if (blockIdx.x == 0) off_next[z] = excl;
            }
            carry += tot;
            if (zb + (k + 1) * kSortBlock >= Z) break;
        }
    }
    if (blockIdx.x == 0 && tid == 0) off_next[Z] = static_cast<uint32_t>(n);
    __syncthreads();
    const int64_t i0 = static_cast<int64_t>(blockIdx.x) * chunk, i1 = min(i0 + chunk, n);
    const int64_t n4 = (i1 > i0) ? (i1 - i0) / 4 : 0;
    const uint4 *k4 = reinterpret_cast<const uint4 *>(key + i0);
    const uint4 *id4 = reinterpret_cast<const uint4 *>(ids ? ids + i0 : nullptr);
    for (int64_t j0 = 0; j0 < n4; j0 += kSortBlock * kSortUnroll) {  // wave-uniform trip count
        const int64_t j = j0 + tid;
        uint4 v[kSortUnroll], c[kSortUnroll];
#pragma unroll
        for (int u = 0; u < kSortUnroll; ++u)
            if (j + u * kSortBlock < n4) {
                int64_t q = j + u * kSortBlock;
                v[u] = k4[q];
                if (ids) c[u] = id4[q];
                else {
                    uint32_t f = static_cast<uint32_t>(i0 + 4 * q);
                    c[u] = make_uint4(f, f + 1, f + 2, f + 3);
                }
            }
#pragma unroll
        for (int u = 0; u < kSortUnroll; ++u) {
            const bool ok = j + u * kSortBlock < n4;
            const uint32_t p0 = wave_slot_take(pos, v[u].x, ok);
            const uint32_t p1 = wave_slot_take(pos, v[u].y, ok);
            const uint32_t p2 = wave_slot_take(pos, v[u].z, ok);
            const uint32_t p3 = wave_slot_take(pos, v[u].w, ok);
            if (ok) {
                ids_next[p0] = c[u].x;
                ids_next[p1] = c[u].y;
                ids_next[p2] = c[u].z;
                ids_next[p3] = c[u].w;
            }
        }
    }
    for (int64_t i = i0 + 4 * n4 + tid; i < i1; i += kSortBlock) {
        uint32_t p = atomicAdd(&pos[key[i] & kZoneMask], 1u);
        ids_next[p] = ids ? ids[i] : static_cast<uint32_t>(i);
    }
}

// car-indexed state from the buckets: zone0[id] = zone of the bucket holding id
__global__ void k_zone_unbucket(const uint32_t *__restrict__ ids, const uint32_t *__restrict__ off, uint32_t *__restrict__ zone0)
{
    const uint32_t z = blockIdx.x;
    for (uint32_t s = off[z] + threadIdx.x; s < off[z + 1]; s += blockDim.x) zone0[ids[s]] = z;
}

struct ExactWork {
    bool tables_dirty = true;
    bool buckets0_valid = false;  // ids0/off0 describe the context's current car state
    int64_t n = 0;
    int Z = 0, T = 0, nb = 0;
    uint32_t *ids0 = nullptr, *idsA = nullptr, *idsB = nullptr;  // [n]
    uint32_t *dest = nullptr;                                    // [n]
    uint32_t *off0 = nullptr, *offA = nullptr, *offB = nullptr;  // [Z+1]
    uint32_t *cursor = nullptr;                                  // [T+1][Z]
    uint32_t *base = nullptr;                                    // [nb][Z]

    void release()
    {
        for (uint32_t **p : {&ids0, &idsA, &idsB, &dest, &off0, &offA, &offB, &cursor, &base}) {
            if (*p) (void)hipFree(*p);
            *p = nullptr;
        }
        n = 0;
        buckets0_valid = false;
    }

    hipError_t ensure(int64_t n_, int Z_, int T_, int cu_count)
    {
        if (n_ == n && Z_ == Z && T_ == T && ids0) return hipSuccess;
        release();
        n = n_;
        Z = Z_;
        T = T_;
        nb = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(cu_count, (n + 4095) / 4096)));
        hipError_t e = hipSuccess;
        auto alloc = [&](uint32_t **p, size_t words) {
            if (e == hipSuccess) e = hipMalloc(p, sizeof(uint32_t) * std::max<size_t>(words, 1));
        };
        alloc(&ids0, n);
        alloc(&idsA, n);
        alloc(&idsB, n);
        alloc(&dest, n);
        alloc(&off0, Z + 1);
        alloc(&offA, Z + 1);
        alloc(&offB, Z + 1);
        alloc(&cursor, static_cast<size_t>(T + 1) * Z);
        alloc(&base, static_cast<size_t>(nb) * Z);
        if (e != hipSuccess) release();
        return e;
    }
};

// true when the path can run this problem (the row as a search tree, and the sort bins, must fit a CU's 160 KiB of LDS)
inline bool exact_path_fits(int Z)
{
    // (64 KiB of LDS per workgroup is what a launch is granted here: a tree of 8,192 f64 + its 64-byte head asked for 65,600 B and the
    //  launch came back "invalid argument" -- Z = 8,192 with more cars than the grouped path's packed ids hold went down with it)
    return sizeof(double) * (size_t(1) << tree_height(Z)) + 64 <= 64 * 1024 && sizeof(uint32_t) * static_cast<size_t>(Z) + 1024 <= 64 * 1024;
}

// two cars per thread (measured at S4k: 1 -> 38.8 us, 2 -> 36.4, 3 -> 38.5, 4 -> 44.4), 512 threads
template <bool TRAVEL, int NP>
inline void exact_launch_np(hipStream_t stream, size_t lds_tree, const uint32_t *ids, const uint32_t *off, uint32_t *dest_out, const double *pd,
                            const double *cdf, int Z, int Zp, int H, CarIndex cars, uint32_t step, uint64_t seed,
                            unsigned long long *parking_t, unsigned long long *driving_t, const double *dm, int T, int t,
                            unsigned long long *tt_sum)
{
    if (lds_tree > 48 * 1024) {  // LDS opt-in, once per device
        static bool attr_done[64] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 64 || !attr_done[dev]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_exact_sample<TRAVEL, 512, NP, 2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024);
            if (dev >= 0 && dev < 64) attr_done[dev] = true;
        }
    }
    launch(k_exact_sample<TRAVEL, 512, NP, 2>, dim3(Z), dim3(512), lds_tree, stream, ids, off, dest_out, pd, cdf, Z, Zp, H, cars, step,
                       seed, parking_t, driving_t, dm, T, t, tt_sum);
}

template <bool TRAVEL>
inline void exact_launch_sample(hipStream_t stream, const uint32_t *ids, const uint32_t *off, uint32_t *dest_out, const double *pd, const double *cdf,
                                int Z, int Zp, CarIndex cars, uint32_t step, uint64_t seed, unsigned long long *parking_t,
                                unsigned long long *driving_t, const double *dm, int T, int t, unsigned long long *tt_sum)
{
    const int H = tree_height(Z);
    const size_t lds_tree = sizeof(double) * (size_t(1) << H);
    const int need = (Zp / 2 + 511) / 512;
#define CPM_EXACT_ARGS stream, lds_tree, ids, off, dest_out, pd, cdf, Z, Zp, H, cars, step, seed, parking_t, driving_t, dm, T, t, tt_sum
    if (need <= 1) exact_launch_np<TRAVEL, 1>(CPM_EXACT_ARGS);
    else if (need <= 2) exact_launch_np<TRAVEL, 2>(CPM_EXACT_ARGS);
    else if (need <= 4) exact_launch_np<TRAVEL, 4>(CPM_EXACT_ARGS);
    else if (need <= 8) exact_launch_np<TRAVEL, 8>(CPM_EXACT_ARGS);
    else exact_launch_np<TRAVEL, 16>(CPM_EXACT_ARGS);
#undef CPM_EXACT_ARGS
}

// ivp == false: the T-hour resample from the state in d_zone0 (left unchanged); counts -> d_counts.
// ivp == true : solveinitialvalueproblem (src/solveinitialvalueproblem.jl:8,53): T-1 hours, steps
//               0..T-2, every transition applied; the final buckets become the cached bucketing of
//               the new state and are written back car-indexed into d_zone0_out.  d_counts is scratch.
template <typename F1, typename F2>
int32_t exact_run(ExactWork &w, hipStream_t stream, const double *d_pdrive, const double *d_cdf, int Z, int Zp, int T, int64_t n, CarIndex cars,
                  const uint32_t *d_zone0, uint64_t seed, bool travel, const double *d_dm, int64_t *d_counts, int cu_count, F1 prof_begin,
                  F2 prof_end, std::string &err, bool ivp = false, uint32_t *d_zone0_out = nullptr)
{
    auto hip_fail = [&](hipError_t e, const char *what) {
        err = std::string(what) + ": " + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? CPM_ERR_NOMEM : CPM_ERR_HIP;
    };
    if (!exact_path_fits(Z)) {
        err = "CPM_KERNEL_ZONE_LDS: a CDF row of this many zones does not fit in LDS (use CPM_KERNEL_CAR)";
        return CPM_ERR_ARG;
    }
    hipError_t e = w.ensure(n, Z, T, cu_count);
    if (e != hipSuccess) return hip_fail(e, "zone workspace");
    const size_t lds_bins = sizeof(uint32_t) * static_cast<size_t>(Z);
    if (w.tables_dirty) {  // LDS opt-in above 48 KiB, once per context
        if (lds_bins > 48 * 1024) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_zone_hist), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bins));
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_zone_scatter), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bins));
        }
        w.tables_dirty = false;
    }
    const int64_t chunk = ((n + w.nb - 1) / w.nb + 3) / 4 * 4;  // multiple of 4: 16-B aligned chunks
    const dim3 sgrid(static_cast<unsigned>(w.nb)), sblock(kSortBlock);
    e = hipMemsetAsync(w.cursor, 0, sizeof(uint32_t) * static_cast<size_t>(T + 1) * Z, stream);
    if (e != hipSuccess) return hip_fail(e, "memset cursor");
    if (!w.buckets0_valid) {  // bucket the car-indexed state once; reused until the state changes
        uint32_t *cur0 = w.cursor + static_cast<size_t>(T) * Z;
        hipLaunchKernelGGL(k_zone_hist, sgrid, sblock, lds_bins, stream, d_zone0, n, Z, chunk, cur0, w.base);
        hipLaunchKernelGGL(k_zone_scatter, sgrid, sblock, lds_bins, stream, d_zone0, static_cast<const uint32_t *>(nullptr), n, Z, chunk, cur0, w.base,
                           w.ids0, w.off0);
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "initial bucketing");
        w.buckets0_valid = true;
    }
    unsigned long long *parking = reinterpret_cast<unsigned long long *>(d_counts);
    unsigned long long *driving = parking + static_cast<size_t>(T) * Z;
    unsigned long long *tt_sum = parking + 2 * static_cast<size_t>(T) * Z;
    const uint32_t *ids = w.ids0, *off = w.off0;
    const int hours = ivp ? T - 1 : T;
    for (int t = 0; t < hours; ++t) {
        const double *pd = d_pdrive + static_cast<size_t>(t) * Z;
        const double *cdf = d_cdf + static_cast<size_t>(t) * Z * Zp;
        const uint32_t step = static_cast<uint32_t>(ivp ? t : T - 1 + t);
        prof_begin(CPM_PROFILE_SAMPLER);
        if (travel)
            exact_launch_sample<true>(stream, ids, off, w.dest, pd, cdf, Z, Zp, cars, step, seed, parking + static_cast<size_t>(t) * Z,
                                      driving + static_cast<size_t>(t) * Z, d_dm, T, t, tt_sum);
        else
            exact_launch_sample<false>(stream, ids, off, w.dest, pd, cdf, Z, Zp, cars, step, seed, parking + static_cast<size_t>(t) * Z,
                                       driving + static_cast<size_t>(t) * Z, d_dm, T, t, tt_sum);
        prof_end(CPM_PROFILE_SAMPLER);
        if (ivp || t + 1 < T) {  // resampling: hour T's transition is sampled but never applied (src/resampling.jl:81-83)
            uint32_t *cur = w.cursor + static_cast<size_t>(t) * Z;
            uint32_t *ids_next = (t & 1) ? w.idsB : w.idsA;
            uint32_t *off_next = (t & 1) ? w.offB : w.offA;
            hipLaunchKernelGGL(k_zone_hist, sgrid, sblock, lds_bins, stream, w.dest, n, Z, chunk, cur, w.base);
            hipLaunchKernelGGL(k_zone_scatter, sgrid, sblock, lds_bins, stream, w.dest, ids, n, Z, chunk, cur, w.base, ids_next, off_next);
            ids = ids_next;
            off = off_next;
        }
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "zone hour launch");
    }
    if (ivp) {
        // the final buckets describe the new state: keep them as the cached initial bucketing and
        // write the state back car-indexed (initial_state of src/solveinitialvalueproblem.jl:57-58)
        if (ids != w.ids0) {
            if ((e = hipMemcpyAsync(w.ids0, ids, sizeof(uint32_t) * n, hipMemcpyDeviceToDevice, stream)) != hipSuccess) return hip_fail(e, "copy buckets");
            if ((e = hipMemcpyAsync(w.off0, off, sizeof(uint32_t) * (Z + 1), hipMemcpyDeviceToDevice, stream)) != hipSuccess)
                return hip_fail(e, "copy offsets");
        }
        hipLaunchKernelGGL(k_zone_unbucket, dim3(Z), dim3(256), 0, stream, w.ids0, w.off0, d_zone0_out);
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "unbucket");
        w.buckets0_valid = true;
    }
    return CPM_OK;
}

}  // namespace cpm
