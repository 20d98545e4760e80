// cpm_zone5_kernels.h -- CPM_KERNEL_ZONE_GROUPED: fixed-stride buckets, stayers kept by the sampler,
// drivers placed in two levels so that the final id writes merge in one XCD's L2.
//
// Why (profiles/round1_notes.md): in both earlier zone paths the hourly regrouping costs more than
// the sampling (47 us against 36 us at S4k), and its floor is the ~2 M random 4-byte id writes of an
// hour: each one leaves its L2 as a masked 32-byte sector write (WRITE_SIZE 68 MB for 16 MB of
// payload), because the arrivals of one bucket come from every workgroup on every XCD.  Here
//   * a car that does not drive never leaves its zone: the sampler compacts the stayers straight into
//     the zone's fixed region of next hour's id array (coalesced), and they never enter the sort;
//   * the sampler workgroup sorts its drivers by destination GROUP (kGroups groups of consecutive
//     zones) inside its own fixed region of a driver array D: rank per group from LDS atomics, group
//     offsets from a 32-wide wave scan, (id, dest) pairs written at D[z*cap + off_g + rank], and the
//     33 offsets of the zone published with plain stores.  No global atomic, no round trip.
//     (A first version appended to per-group global lists with a ticket per workgroup: all Z
//     workgroups hit the same 32 counters and same-address atomics are served one at a time, 64 us
//     per hour; replicated 32x it still exposed one atomic round trip per workgroup, 51 us.)
//   * k_zone5_place then takes one group per 8 blocks -- blockIdx = j * kGroups + g, so the blocks of a
//     group share blockIdx % 8, i.e. one XCD and one L2.  A block gathers the group-g segments of an
//     eighth of the origin zones (~16 pairs = one 128-B line each) and moves the ids into their
//     buckets.  All writes to a bucket now come from one L2, where they merge before they leave.
// Bucket sizes (= next hour's parking histogram) are the ticket counters, as in cpm_zone3_kernels.h.
// Overflow of a bucket region raises bit 1 of the status word; no out-of-range store is issued; the
// caller repeats the step on the exact layout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <string>

#include "../../include/cpm.h"
#include "cpm_kernels.h"
#include "cpm_zone_kernels.h"
#include "cpm_zone3_kernels.h"
#include "cpm_zone6_kernels.h"

namespace cpm {

constexpr int kGroups = 32;          // destination groups; zones per group <= kMaxZonesPerGroup
constexpr int kBlocksPerGroup = 8;   // blocks of k_zone5_place per group (all on one XCD)
constexpr int kMaxZonesPerGroup = 512;
constexpr int kRankShift = 14;       // per-slot record: dest (14 bits) | rank in (workgroup, group) << 14 | drive << 31
constexpr uint32_t kDest14 = (1u << kRankShift) - 1u;

struct Zone5Args {
    const uint32_t *ids;      // [Z*cap] this hour's buckets
    const uint32_t *cnt;      // [Z] their sizes
    uint32_t *rec;            // [Z*cap] scratch: per-slot record for cars beyond the register-resident ones
    const double *pdrive_t, *cdf_t, *dm;
    uint32_t *ids_next;       // [Z*cap]
    uint32_t *cnt_next;       // [Z]: set to the number of stayers here, drivers are added by k_zone5_place
    uint2 *D;                 // [Z*cap] (id, dest) of this hour's drivers: zone z's, sorted by destination group, at z*cap
    uint32_t *offz;           // [Z][kGroups+1] start of every group's run inside the zone's region of D
    unsigned long long *parking_t, *driving_t, *tt_sum, *status;
    int Z, Zp, H, T, t, zpg;
    uint32_t cap, step, gmagic;  // gmagic: dest / zpg == (dest * gmagic) >> 24 for dest < 2^14
    int64_t car_begin;
    uint64_t seed;
};

// Second launch-bounds argument = waves per SIMD: 8 (four 512-thread workgroups per CU).  Without it the
// kernel took 91 SGPRs, which the hardware admits at only 7 waves per SIMD, i.e. three workgroups per CU.
template <bool TRAVEL, int BLOCK, int NP, int CPT>
__global__ __launch_bounds__(BLOCK, (!TRAVEL && NP <= 8) ? 8 : 2) void k_zone5_sample(Zone5Args a)
{
    extern __shared__ double row[];  // 2^H doubles: the zone's CDF row as a search tree
    __shared__ uint32_t s_ndrive, s_nstay;
    __shared__ unsigned long long s_tt;
    __shared__ uint32_t gb[kGroups];  // drivers of this workgroup per destination group, then their run's start
    const int Z = a.Z, Zp = a.Zp, H = a.H;
    const int z = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t cap = a.cap;
    const uint32_t b = static_cast<uint32_t>(z) * cap;
    const uint32_t n = min(a.cnt[z], cap);
    if (tid == 0) {
        a.parking_t[z] = n;
        s_ndrive = 0;
        s_nstay = 0;
        s_tt = 0;
    }
    if (tid < kGroups) gb[tid] = 0;
    if (n == 0) {
        if (tid == 0) a.cnt_next[z] = 0;
        if (tid <= kGroups) a.offz[static_cast<size_t>(z) * (kGroups + 1) + tid] = 0;
        return;
    }
    uint32_t id[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) id[c] = a.ids[b + min(static_cast<uint32_t>(tid + c * BLOCK), n - 1)];
    // the id of the first overflow round (buckets of CPT*BLOCK+1 .. (CPT+1)*BLOCK cars are common) rides along:
    // loaded here it costs nothing, loaded inside the overflow loop it is an exposed HBM round trip
    const uint32_t id_x = a.ids[b + min(static_cast<uint32_t>(tid + CPT * BLOCK), n - 1)];
    double2 pc[NP];
    const double2 *src = reinterpret_cast<const double2 *>(a.cdf_t + static_cast<size_t>(z) * Zp);
#pragma unroll
    for (int m = 0; m < NP; ++m) pc[m] = src[min(tid + m * BLOCK, Zp / 2 - 1)];
    const long long thr = bernoulli_threshold(a.pdrive_t[z]);
    const int P = 1 << H;
    for (int r = Z + tid; r < P; r += BLOCK) {
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
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t nd = 0;
    long long tt = 0;
    uint32_t *stay_out = a.ids_next + static_cast<size_t>(z) * cap;

    // one car: stayer -> compacted into next hour's bucket of this zone; driver -> ranked inside its
    // destination group.  Returns the record dest | rank << 14 | drive << 31.
    auto place = [&](uint32_t idc, bool valid, bool drive, uint32_t dest) -> uint32_t {
        const unsigned long long mS = __ballot(valid && !drive);
        uint32_t bS = 0;
        if (lane == 0 && mS) bS = atomicAdd(&s_nstay, static_cast<uint32_t>(__popcll(mS)));
        bS = __shfl(bS, 0, 64);
        if (valid && !drive) stay_out[bS + static_cast<uint32_t>(__popcll(mS & below))] = idc;
        uint32_t rec = dest;
        if (drive) {
            const uint32_t rank = atomicAdd(&gb[(dest * a.gmagic) >> 24], 1u);
            rec = dest | (rank << kRankShift) | kDriveBit;
        }
        return rec;
    };

    uint32_t rec[CPT];
    {  // CPT cars per thread, straight line
        bool valid[CPT], drive[CPT], any_search = false;
        uint32_t dest[CPT];
        double ue[CPT];
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            valid[c] = static_cast<uint32_t>(tid + c * BLOCK) < n;
            long long kb;
            double uc;
            car_draws(a.seed, static_cast<uint64_t>(a.car_begin) + id[c], a.step, kb, uc);
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
            rec[c] = place(id[c], valid[c], drive[c], dest[c]);
            if (drive[c]) {
                ++nd;
                if (TRAVEL) tt += travel_time_q16(a.dm, Z, a.T, a.t, z, dest[c], a.seed, static_cast<uint64_t>(a.car_begin) + id[c], a.step);
            }
        }
    }
    uint32_t rec_x = 0;  // record of the first overflow round, kept in a register across the barrier
    for (uint32_t q0 = CPT * BLOCK; q0 < n; q0 += BLOCK) {  // buckets larger than CPT*BLOCK cars (wave-uniform trips)
        const uint32_t q = q0 + tid;
        const bool valid = q < n;
        const bool first = q0 == CPT * BLOCK;
        const uint32_t idx = first ? id_x : (valid ? a.ids[b + q] : 0u);
        const uint64_t car = static_cast<uint64_t>(a.car_begin) + idx;
        long long kb;
        double uc;
        car_draws(a.seed, car, a.step, kb, uc);
        const bool drive = valid && (kb <= thr);
        uint32_t dest = z;
        if (drive) {
            if (last != 0.0) {
                const double u1 = clamp_u(uc, last);
                uint32_t i = 1;
                for (int l = 0; l < H; ++l) i = 2 * i + (row[i] < u1 ? 1u : 0u);
                dest = eytz_decode(i, Z, H);
            }
            if (TRAVEL) tt += travel_time_q16(a.dm, Z, a.T, a.t, z, dest, a.seed, car, a.step);
            ++nd;
        }
        const uint32_t r = place(idx, valid, drive, dest);
        if (first) rec_x = r;
        else if (valid) a.rec[b + q] = r;  // needed again after the barrier
    }
    for (int o = 32; o > 0; o >>= 1) nd += __shfl_down(nd, o, 64);
    if (lane == 0 && nd) atomicAdd(&s_ndrive, nd);
    if (TRAVEL) {
        for (int o = 32; o > 0; o >>= 1) tt += __shfl_down(tt, o, 64);
        if (lane == 0 && tt) atomicAdd(&s_tt, static_cast<unsigned long long>(tt));
    }
    __syncthreads();  // group counts, stayer count final
    // Group offsets inside this zone's region of D = exclusive scan of the 32 group counts.  Every wave
    // scans them for itself (32 LDS reads + 5 shuffles) and keeps offset g in lane g, so no second
    // barrier is needed before the pairs are written; wave 0 publishes the offsets for k_zone5_place.
    const uint32_t gcount = gb[lane & (kGroups - 1)];
    uint32_t incl = gcount;
    for (int o = 1; o < kGroups; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o, 64);
        if ((lane & (kGroups - 1)) >= o) incl += v;
    }
    const uint32_t excl = incl - gcount;  // lanes g and g + 32 both hold the offset of group g
    const uint32_t gtotal = __shfl(incl, kGroups - 1, 64);
    if (tid <= kGroups) a.offz[static_cast<size_t>(z) * (kGroups + 1) + tid] = (tid < kGroups) ? excl : gtotal;
    if (tid == 0) {
        a.driving_t[z] = s_ndrive;
        a.cnt_next[z] = s_nstay;  // k_zone5_place adds the arrivals
        if (TRAVEL && s_tt) atomicAdd(a.tt_sum, s_tt);
    }
    // drivers -> the run of their destination group
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        const uint32_t dest = rec[c] & kDest14, g = (dest * a.gmagic) >> 24;
        const uint32_t base = __shfl(excl, static_cast<int>(g), 64);
        if (rec[c] & kDriveBit) a.D[static_cast<size_t>(b) + base + ((rec[c] & kZoneMask) >> kRankShift)] = make_uint2(id[c], dest);
    }
    for (uint32_t q0 = CPT * BLOCK; q0 < n; q0 += BLOCK) {  // wave-uniform trips (the shuffle needs every lane)
        const uint32_t q = q0 + tid;
        const bool first = q0 == CPT * BLOCK;
        const uint32_t r = first ? rec_x : ((q < n) ? a.rec[b + q] : 0u);
        const uint32_t dest = r & kDest14, g = (dest * a.gmagic) >> 24;
        const uint32_t base = __shfl(excl, static_cast<int>(g), 64);
        if (r & kDriveBit)
            a.D[static_cast<size_t>(b) + base + ((r & kZoneMask) >> kRankShift)] = make_uint2(first ? id_x : a.ids[b + q], dest);
    }
}

// Blocks of the place kernels per destination group: at most 256 origin zones per block (one run per 16 lanes,
// four runs per thread) from 1024 zones on -- measured at S4k: 8 blocks 16.9 us, 16 blocks 15.3 us.
inline int place_bpg(int Z)
{
    if (Z < 1024) return 8;
    int b = 16;
    while (b < 64 && (Z + b - 1) / b > 256) b *= 2;
    return b;
}

// Drivers of destination group g -> their buckets.  blockIdx = j * kGroups + g: the blocks of a group share
// blockIdx % 8 (one XCD, one L2; speed only, never correctness).  Block (g, j) gathers the group-g runs of
// the origin zones [j*zps, (j+1)*zps); 16 lanes per run, KDEEP pairs per lane held in registers.
constexpr int kPlaceBlock = 1024;
constexpr int kPlaceSeg = kPlaceBlock / 16;  // runs in flight per pass

template <int KRUNS, int KDEEP>
__global__ __launch_bounds__(kPlaceBlock) void k_zone5_place(const uint2 *__restrict__ D, const uint32_t *__restrict__ offz,
                                                             int zpg, int zps, int Z, uint32_t cap,
                                                             uint32_t *__restrict__ cnt_next, uint32_t *__restrict__ ids_next,
                                                             unsigned long long *status)
{
    __shared__ uint32_t bins[kMaxZonesPerGroup];
    const int tid = threadIdx.x;
    const int g = blockIdx.x % kGroups, j = blockIdx.x / kGroups;
    const int zg0 = g * zpg;
    const int nzl = max(0, min(zpg, Z - zg0));
    const int zs0 = j * zps, zs1 = min(Z, zs0 + zps);
    const int sub = tid >> 4, l16 = tid & 15;
    for (int k = tid; k < kMaxZonesPerGroup; k += kPlaceBlock) bins[k] = 0;
    __syncthreads();
    // Every thread owns lane l16 of KRUNS runs (origin zones zs0 + sub + k * kPlaceSeg) and the pairs
    // l16, l16 + 16, ... (KDEEP of them) of each: descriptors, then all pairs, are requested together and stay
    // in registers across the ticket.  (With one pair per lane the second sixteen of a run -- runs average ~16
    // at S4k -- were fetched inside the histogram loop, one exposed round trip per run: 20 us per launch.)
    // Pairs beyond 16 * KDEEP of a run take the loops below and are re-read in pass 2.
    uint32_t o0[KRUNS], o1[KRUNS];
    uint2 v[KRUNS][KDEEP];
#pragma unroll
    for (int k = 0; k < KRUNS; ++k) {
        const int zs = zs0 + sub + k * kPlaceSeg;
        const int zc = min(zs, zs1 - 1);
        o0[k] = offz[static_cast<size_t>(zc) * (kGroups + 1) + g];
        o1[k] = offz[static_cast<size_t>(zc) * (kGroups + 1) + g + 1];
        if (zs >= zs1) o1[k] = o0[k];
    }
#pragma unroll
    for (int k = 0; k < KRUNS; ++k) {
        const int zc = min(zs0 + sub + k * kPlaceSeg, zs1 - 1);
#pragma unroll
        for (int d = 0; d < KDEEP; ++d) {
            v[k][d] = make_uint2(0u, static_cast<uint32_t>(zg0));
            if (o0[k] + l16 + 16 * d < o1[k]) v[k][d] = D[static_cast<size_t>(zc) * cap + o0[k] + l16 + 16 * d];
        }
    }
    // pass 1: histogram of the destinations over the group's zones
#pragma unroll
    for (int k = 0; k < KRUNS; ++k) {
#pragma unroll
        for (int d = 0; d < KDEEP; ++d)
            if (o0[k] + l16 + 16 * d < o1[k]) atomicAdd(&bins[v[k][d].y - zg0], 1u);
    }
#pragma unroll
    for (int k = 0; k < KRUNS; ++k) {
        const int zc = min(zs0 + sub + k * kPlaceSeg, zs1 - 1);
        for (uint32_t i = o0[k] + l16 + 16 * KDEEP; i < o1[k]; i += 16) atomicAdd(&bins[D[static_cast<size_t>(zc) * cap + i].y - zg0], 1u);
    }
    __syncthreads();
    if (tid < nzl) {  // ticket: this block's range inside each bucket of the group
        const uint32_t c = bins[tid];
        uint32_t base = 0;
        if (c) {
            base = atomicAdd(&cnt_next[zg0 + tid], c);
            if (base + c > cap) atomicOr(status, 2ull);
        }
        bins[tid] = base;
    }
    __syncthreads();
    // pass 2: the ids move
#pragma unroll
    for (int k = 0; k < KRUNS; ++k) {
#pragma unroll
        for (int d = 0; d < KDEEP; ++d)
            if (o0[k] + l16 + 16 * d < o1[k]) {
                const uint32_t p = atomicAdd(&bins[v[k][d].y - zg0], 1u);
                if (p < cap) ids_next[static_cast<size_t>(v[k][d].y) * cap + p] = v[k][d].x;
            }
    }
#pragma unroll
    for (int k = 0; k < KRUNS; ++k) {
        const int zc = min(zs0 + sub + k * kPlaceSeg, zs1 - 1);
        for (uint32_t i = o0[k] + l16 + 16 * KDEEP; i < o1[k]; i += 16) {
            const uint2 w = D[static_cast<size_t>(zc) * cap + i];
            const uint32_t p = atomicAdd(&bins[w.y - zg0], 1u);
            if (p < cap) ids_next[static_cast<size_t>(w.y) * cap + p] = w.x;
        }
    }
}

// Launch with bpg blocks per group (8 or 16): zps origin zones per block must fit KRUNS * kPlaceSeg.
inline void zone5_launch_place(hipStream_t stream, int bpg, int deep, const uint2 *D, const uint32_t *offz, int zpg, int Z, uint32_t cap,
                               uint32_t *cnt_next, uint32_t *ids_next, unsigned long long *status)
{
    const int zps = (Z + bpg - 1) / bpg;
    const dim3 grid(kGroups * bpg), block(kPlaceBlock);
    if (zps <= 4 * kPlaceSeg) {
        if (deep >= 2) hipLaunchKernelGGL((k_zone5_place<4, 2>), grid, block, 0, stream, D, offz, zpg, zps, Z, cap, cnt_next, ids_next, status);
        else hipLaunchKernelGGL((k_zone5_place<4, 1>), grid, block, 0, stream, D, offz, zpg, zps, Z, cap, cnt_next, ids_next, status);
    } else {
        if (deep >= 2) hipLaunchKernelGGL((k_zone5_place<8, 2>), grid, block, 0, stream, D, offz, zpg, zps, Z, cap, cnt_next, ids_next, status);
        else hipLaunchKernelGGL((k_zone5_place<8, 1>), grid, block, 0, stream, D, offz, zpg, zps, Z, cap, cnt_next, ids_next, status);
    }
}

// run capacity of the second-generation layout: a quarter of a bucket region (= the mean bucket size), >= 64, whole 128-B lines
inline uint32_t zone6_scap(uint32_t cap) { return (std::max<uint32_t>(64u, cap / 4) + 31u) / 32u * 32u; }
// second generation: groups of 2^gshift consecutive zones, at most kGroups of them
inline uint32_t zone6_gshift(int Z)
{
    uint32_t s = 0;
    while ((static_cast<int64_t>(kGroups) << s) < Z) ++s;
    return s;
}
// packed driver = id | (dest mod 2^gshift) << idbits
inline uint32_t zone6_idbits(int Z) { return 32u - std::max(1u, zone6_gshift(Z)); }

struct Zone5Work {
    Zone3Work base;  // bucket arrays, cached initial bucketing, sampler launcher state
    uint2 *D = nullptr;          // [Z*cap]
    uint32_t *offz = nullptr;    // [Z][kGroups+1]
    uint32_t gmagic = 0;
    const uint32_t *ivp_ids = nullptr, *ivp_cnt = nullptr;  // final buckets of the last IVP (see zone5_resample)
    int zpg = 0;
    int bpg = 0, deep = 2;  // place kernels' shape: blocks per group (0 = place_bpg(Z)) and pairs per lane; CPM_OPT_PLACE_SHAPE
    // second generation (cpm_zone6_kernels.h): high-word rows, fixed-size runs
    bool v6 = true;              // CPM_OPT_GROUPED_GEN: 6 (default) or 5
    uint32_t *Dq = nullptr;      // [Z][kGroups][scap] packed drivers
    uint32_t *cntg = nullptr;    // [Z][kGroups] run lengths
    uint32_t scap = 0, idbits = 0, gshift6 = 0, cap_alloc = 0;
    unsigned long long *tt_part = nullptr;  // [kTravelParts] partial travel-time sums (k_zone6_travel), kept zero between resamples
    int64_t n = 0;
    int Z = 0, T = 0;

    void release()
    {
        base.release();
        if (D) (void)hipFree(D);
        if (offz) (void)hipFree(offz);
        if (Dq) (void)hipFree(Dq);
        if (cntg) (void)hipFree(cntg);
        if (tt_part) (void)hipFree(tt_part);
        D = nullptr;
        offz = nullptr;
        Dq = nullptr;
        cntg = nullptr;
        tt_part = nullptr;
        n = 0;
    }

    hipError_t ensure(int64_t n_, int Z_, int T_, int cu_count)
    {
        hipError_t e = base.ensure(n_, Z_, T_, cu_count);
        if (e != hipSuccess) return e;
        if (n_ == n && Z_ == Z && T_ == T && D && cap_alloc == base.cap) return hipSuccess;
        if (D) (void)hipFree(D);
        if (offz) (void)hipFree(offz);
        if (Dq) (void)hipFree(Dq);
        if (cntg) (void)hipFree(cntg);
        D = nullptr;
        offz = nullptr;
        Dq = nullptr;
        cntg = nullptr;
        n = n_;
        Z = Z_;
        T = T_;
        zpg = (Z + kGroups - 1) / kGroups;
        gmagic = (1u << 24) / static_cast<uint32_t>(zpg) + 1u;  // exact for dest < 2^14 (2^24 / zpg >= 2^15)
        e = hipMalloc(&D, sizeof(uint2) * static_cast<size_t>(Z) * base.cap);
        if (e == hipSuccess) e = hipMalloc(&offz, sizeof(uint32_t) * static_cast<size_t>(Z) * (kGroups + 1));
        cap_alloc = base.cap;
        scap = zone6_scap(base.cap);
        idbits = zone6_idbits(Z);
        gshift6 = zone6_gshift(Z);
        if (e == hipSuccess) e = hipMalloc(&Dq, sizeof(uint32_t) * static_cast<size_t>(Z) * kGroups * scap);
        if (e == hipSuccess) e = hipMalloc(&cntg, sizeof(uint32_t) * static_cast<size_t>(Z) * kGroups);
        if (e == hipSuccess && !tt_part) {
            e = hipMalloc(&tt_part, sizeof(unsigned long long) * kTravelParts);
            if (e == hipSuccess) e = hipMemset(tt_part, 0, sizeof(unsigned long long) * kTravelParts);
        }
        if (e != hipSuccess) release();
        return e;
    }
};

inline bool zone5_path_fits(int Zp, int64_t n, int Z, int cap_mult = 4)
{
    const int64_t mean = (n + Z - 1) / Z;
    return zone3_path_fits(Zp, n, Z, cap_mult) && Z <= (1 << kRankShift) && (Z + kGroups - 1) / kGroups <= kMaxZonesPerGroup &&
           (Z + place_bpg(Z) - 1) / place_bpg(Z) <= 8 * kPlaceSeg &&
           n < (int64_t(1) << 30) && std::max<int64_t>(cap_mult * mean, 1024) + 64 <= (int64_t(1) << (31 - kRankShift));  // rank field
}

// second generation: also the packed id field and the run array (Z x 32 x scap x 4 B <= 16 GiB)
inline bool zone6_path_fits(int Zp, int64_t n, int Z, int cap_mult = 4)
{
    if (!zone5_path_fits(Zp, n, Z, cap_mult) || !zone6_row_fits(Z)) return false;
    const int64_t mean = (n + Z - 1) / Z;
    const uint32_t cap = static_cast<uint32_t>((std::max<int64_t>(cap_mult * mean, 1024) + 63) / 64 * 64);
    return n <= (int64_t(1) << zone6_idbits(Z)) && (1 << zone6_gshift(Z)) <= kMaxZonesPerGroup6 && static_cast<int64_t>(Z) * kGroups * zone6_scap(cap) * 4 <= (int64_t(cap_mult <= 4 ? 16 : 48) << 30);
}

template <bool TRAVEL, int NP>
inline void zone5_launch_np(const Zone5Args &a, size_t lds, hipStream_t stream)
{
    static bool attr_done = false;
    if (!attr_done && lds > 64 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_zone5_sample<TRAVEL, 512, NP, 2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    hipLaunchKernelGGL((k_zone5_sample<TRAVEL, 512, NP, 2>), dim3(a.Z), dim3(512), lds, stream, a);
}

template <bool TRAVEL>
inline void zone5_launch(const Zone5Args &a, size_t lds, hipStream_t stream)
{
    const int need = (a.Zp / 2 + 511) / 512;
    if (need <= 1) zone5_launch_np<TRAVEL, 1>(a, lds, stream);
    else if (need <= 2) zone5_launch_np<TRAVEL, 2>(a, lds, stream);
    else if (need <= 4) zone5_launch_np<TRAVEL, 4>(a, lds, stream);
    else if (need <= 8) zone5_launch_np<TRAVEL, 8>(a, lds, stream);
    else zone5_launch_np<TRAVEL, 16>(a, lds, stream);
}

// car-indexed state from fixed-stride buckets
__global__ void k_zone5_unbucket(const uint32_t *__restrict__ ids, const uint32_t *__restrict__ cnt, uint32_t cap,
                                 uint32_t *__restrict__ zone0)
{
    const uint32_t z = blockIdx.x;
    const uint32_t n = min(cnt[z], cap);
    for (uint32_t s = threadIdx.x; s < n; s += blockDim.x) zone0[ids[static_cast<size_t>(z) * cap + s]] = z;
}

// ivp == false: the T-hour resample from the state in d_zone0 (left unchanged); status word as in zone3_resample.
// ivp == true : solveinitialvalueproblem (src/solveinitialvalueproblem.jl:8,53): T-1 hours, steps 0..T-2, every
//               transition applied.  d_zone0 is NOT modified: the new car-indexed state goes to d_zone0_out and the
//               final buckets are remembered in w5.ivp_ids / w5.ivp_cnt.  The caller inspects the status word when
//               the stream has drained and then either commits (zone5_commit_ivp + pointer swap) or repeats the IVP
//               on the exact layout from the untouched d_zone0.
template <typename F1, typename F2>
int32_t zone5_resample(Zone5Work &w5, hipStream_t stream, const double *d_pdrive, const double *d_cdf, int Z, int Zp, int T,
                       int64_t n, int64_t car_begin, const uint32_t *d_zone0, uint64_t seed, bool travel,
                       const double *d_dm, int64_t *d_counts, int cu_count, F1 prof_begin, F2 prof_end, std::string &err,
                       bool ivp = false, uint32_t *d_zone0_out = nullptr, const uint32_t *d_hi = nullptr,
                       const double *d_last = nullptr, int Zq = 0, const long long *d_thr = nullptr)
{
    auto hip_fail = [&](hipError_t e, const char *what) {
        err = std::string(what) + ": " + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? CPM_ERR_NOMEM : CPM_ERR_HIP;
    };
    const bool v6 = w5.v6 && d_hi && d_last && d_thr && zone6_path_fits(Zp, n, Z, w5.base.cap_mult);
    hipError_t e = w5.ensure(n, Z, T, cu_count);
    if (e != hipSuccess) return hip_fail(e, "grouped zone workspace");
    Zone3Work &w = w5.base;
    const size_t lds_bins = sizeof(uint32_t) * static_cast<size_t>(Z);
    if (!w.attrs_set) {
        if (lds_bins > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_zone3_sort<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      static_cast<int>(lds_bins));
        w.attrs_set = true;
    }
    unsigned long long *parking = reinterpret_cast<unsigned long long *>(d_counts);
    unsigned long long *driving = parking + static_cast<size_t>(T) * Z;
    unsigned long long *tt_sum = parking + 2 * static_cast<size_t>(T) * Z;
    unsigned long long *status = tt_sum + 1;
    e = hipMemsetAsync(w.cnt, 0, sizeof(uint32_t) * static_cast<size_t>(T + 1) * Z, stream);
    if (e != hipSuccess) return hip_fail(e, "memset counters");
    if (!w.buckets0_valid) {  // bucket the car-indexed state once; reused until the state changes
        const int64_t chunk = (n + w.nb0 - 1) / w.nb0;
        if ((e = hipMemsetAsync(w.cnt0, 0, sizeof(uint32_t) * Z, stream)) != hipSuccess) return hip_fail(e, "memset cnt0");
        hipLaunchKernelGGL(k_zone3_sort<false>, dim3(w.nb0), dim3(kSort3Block), lds_bins, stream, d_zone0,
                           static_cast<const uint32_t *>(nullptr), static_cast<const uint32_t *>(nullptr), 0, n, chunk, Z, w.cap,
                           w.cnt0, w.ids0, status);
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "initial bucketing");
        w.buckets0_valid = true;
    }
    const int H = tree_height(Z);
    const size_t lds_tree = sizeof(double) * (size_t(1) << H);
    Zone5Args a;
    a.rec = w.dest;
    a.dm = d_dm;
    a.D = w5.D;
    a.offz = w5.offz;
    a.tt_sum = tt_sum;
    a.status = status;
    a.Z = Z;
    a.Zp = Zp;
    a.H = H;
    a.T = T;
    a.zpg = w5.zpg;
    a.cap = w.cap;
    a.gmagic = w5.gmagic;
    a.car_begin = car_begin;
    a.seed = seed;
    const uint32_t *ids = w.ids0, *cnt = w.cnt0;
    const int hours = ivp ? T - 1 : T;
    for (int t = 0; t < hours; ++t) {
        const double *pd = d_pdrive + static_cast<size_t>(t) * Z;
        const double *cdf = d_cdf + static_cast<size_t>(t) * Z * Zp;
        const uint32_t step = static_cast<uint32_t>(ivp ? t : T - 1 + t);
        if (v6) {
            const bool last_hour = !ivp && t + 1 == T;  // sampled, never applied (src/resampling.jl:81-83): counts only
            uint32_t *cnt_next = w.cnt + static_cast<size_t>(t + 1) * Z;
            uint32_t *ids_next = (t & 1) ? w.idsB : w.idsA;
            Zone6Args b;
            b.ids = ids;
            b.cnt = cnt;
            b.rp_t = d_hi + static_cast<size_t>(t) * Z * pack_row_words(Zq, pack_guide_bits(Z));
            b.last_t = d_last + static_cast<size_t>(t) * Z;
            b.thr_t = d_thr ? d_thr + static_cast<size_t>(t) * Z : nullptr;
            b.pdrive_t = pd;
            b.cdf_t = cdf;
            b.dm = d_dm;
            b.ids_next = ids_next;
            b.cnt_next = cnt_next;
            b.D = w5.Dq;
            b.cntg = w5.cntg;
            b.rec_out = w.dest;
            b.parking_t = parking + static_cast<size_t>(t) * Z;
            b.driving_t = driving + static_cast<size_t>(t) * Z;
            b.tt_sum = tt_sum;
            b.status = status;
            b.Z = Z;
            b.Zp = Zp;
            b.Zq = Zq;
            b.G = pack_guide_bits(Z);
            b.T = T;
            b.t = t;
            b.cap = w.cap;
            b.scap = w5.scap;
            b.idbits = w5.idbits;
            b.step = step;
            b.gshift = w5.gshift6;
            b.car_begin = car_begin;
            b.seed = seed;
            b.abl = w.sampler.ablate;
            prof_begin(t);
            // travel times: the last hour's sampler computes its own; for the other hours they are computed from the runs from the
            // runs by k_zone6_travel, and the sampler runs in its faster form without them
            if (last_hour) zone6_launch<false>(b, travel, (n + Z - 1) / Z, stream);
            else zone6_launch<true>(b, false, (n + Z - 1) / Z, stream);
            prof_end(t);
            if (!last_hour) {
                TravelArgs tr{};
                tr.dm = d_dm;
                tr.tt_part = w5.tt_part;
                tr.T = T;
                tr.t = t;
                tr.gshift = static_cast<int>(w5.gshift6);
                tr.step = step;
                tr.car_begin = car_begin;
                tr.seed = seed;
                zone6_launch_place(stream, w5.bpg ? w5.bpg : place_bpg(Z), w5.Dq, w5.cntg, 1 << w5.gshift6, Z, w.cap, w5.scap, w5.idbits, cnt_next,
                                   ids_next, status);
                if (travel) hipLaunchKernelGGL(k_zone6_travel, dim3(Z), dim3(256), 0, stream, w5.Dq, w5.cntg, Z, w5.scap, w5.idbits, tr);
                ids = ids_next;
                cnt = cnt_next;
            }
        } else if (!ivp && t + 1 == T) {  // hour T's transition is sampled but never applied (src/resampling.jl:81-83): counts only
            prof_begin(t);
            launch_zone_sample(w.sampler, stream, travel, ids, cnt, pd, cdf, Z, Zp, car_begin, step, seed,
                               parking + static_cast<size_t>(t) * Z, driving + static_cast<size_t>(t) * Z, d_dm, T, t, tt_sum, 0,
                               w.dest, w.cap);
            prof_end(t);
        } else {
            uint32_t *cnt_next = w.cnt + static_cast<size_t>(t + 1) * Z;
            uint32_t *ids_next = (t & 1) ? w.idsB : w.idsA;
            a.ids = ids;
            a.cnt = cnt;
            a.pdrive_t = pd;
            a.cdf_t = cdf;
            a.ids_next = ids_next;
            a.cnt_next = cnt_next;
            a.parking_t = parking + static_cast<size_t>(t) * Z;
            a.driving_t = driving + static_cast<size_t>(t) * Z;
            a.t = t;
            a.step = step;
            prof_begin(t);
            if (travel) zone5_launch<true>(a, lds_tree, stream);
            else zone5_launch<false>(a, lds_tree, stream);
            prof_end(t);
            zone5_launch_place(stream, w5.bpg ? w5.bpg : place_bpg(Z), w5.deep, w5.D, w5.offz, w5.zpg, Z, w.cap, cnt_next, ids_next, status);
            ids = ids_next;
            cnt = cnt_next;
        }
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "grouped zone hour launch");
    }
    if (v6 && travel && !ivp) {  // the partial sums of k_zone6_travel -> the sum word of the count tensor
        hipLaunchKernelGGL(k_zone6_travel_finish, dim3(1), dim3(kTravelParts), 0, stream, w5.tt_part, tt_sum);
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "travel-time sum");
    }
    if (ivp) {
        hipLaunchKernelGGL(k_zone5_unbucket, dim3(Z), dim3(256), 0, stream, ids, cnt, w.cap, d_zone0_out);
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "unbucket");
        w5.ivp_ids = ids;
        w5.ivp_cnt = cnt;
        w.buckets0_valid = false;  // until zone5_commit_ivp
    }
    return CPM_OK;
}

// After a verified IVP: its final buckets become the cached bucketing of the (new) current state.
inline hipError_t zone5_commit_ivp(Zone5Work &w5, hipStream_t stream)
{
    Zone3Work &w = w5.base;
    hipError_t e = hipSuccess;
    if (w5.ivp_ids != w.ids0)
        e = hipMemcpyAsync(w.ids0, w5.ivp_ids, sizeof(uint32_t) * static_cast<size_t>(w.Z) * w.cap, hipMemcpyDeviceToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(w.cnt0, w5.ivp_cnt, sizeof(uint32_t) * w.Z, hipMemcpyDeviceToDevice, stream);
    if (e == hipSuccess) w.buckets0_valid = true;
    return e;
}

}  // namespace cpm
