// cpm_grouped.h -- CPM_KERNEL_ZONE_GROUPED, the path CPM_KERNEL_AUTO runs: the hourly step of src/resampling.jl:11-49,81-83 (and of
// src/solveinitialvalueproblem.jl:8-53) with the cars kept bucketed by origin zone.
//
// The reference walks p_dest[origin,:,t] once per driving car (src/resampling.jl:34-45).  Here the cars of one origin zone sit
// together, so the zone's row is streamed from HBM once per hour into LDS and every car of the zone searches it there.
//
//   * Fixed-stride buckets: zone z owns the region ids[z*cap .. (z+1)*cap) (cap = cap_mult x the mean bucket, cap_mult = 4, doubled
//     by the context after an overflow).  A bucket has two ends: the cars that STAYED in z last hour fill the region from the bottom
//     (cnt_s[z] of them, written by the zone's own sampler workgroup), the cars that ARRIVED fill it from the top (cnt_a[z], written
//     by the placing blocks) -- neither writer needs the other's count, so the placing of hour t's drivers does not wait for the
//     samplers of their destination zones.  cnt_s + cnt_a IS the parking histogram (src/saveresults.jl:10-12): no histogram atomics.
//   * Row packs: what a workgroup stages is the HIGH WORD of the canonical CDF row (4 B per destination instead of 8) behind a GUIDE
//     table (cut-point method).  With hi[j] = floor(cdf[j] * 2^32) (0xFFFFFFFF when cdf[j] >= 1) and khi = the high Philox word
//     = floor(u * 2^32):   hi[j] > khi => cdf[j] >= u   and   hi[j-1] < khi => cdf[j-1] < u,   so the first j with hi[j] >= khi IS
//     the reference's answer (first j with u <= cdf[j], :38-45) whenever hi[j] > khi strictly.  On a tie (probability ~ Z * 2^-32
//     per draw), or when u lies above the row total, the car repeats the reference's f64 walk on the table itself, from the nearest
//     checkpoint of the running sum (search_exact_ckpt).  Bit-identical to the f64 search by construction; cpm_debug_categorical and
//     tests/test_gpu_parity.py::test_high_word_search_* drive the tie, saturation and out-of-range branches.
//     guide[m] = first j with hi[j] >= m << (32 - G), m = 0 .. 2^G (u16, clamped to Z - 1; G = ceil(log2 Z) - 2).
//     A row pack = [2^G + 8 u16 guide][Zq u32 hi], Zq = Z + at least 31 entries of 0xFFFFFFFF, rounded to 32.
//   * Stayers (about half the cars) never leave their zone: the sampler compacts them into the zone's region of next hour's id
//     array.  Drivers go into FIXED-SIZE runs, one per (origin zone, destination group) -- 32 groups of zpg = ceil(Z / 32) consecutive
//     zones -- at D[(z*32 + g)*scap + rank], rank from an LDS atomic, as id | (dest - g * zpg) << idbits.  k_grouped_place
//     then moves every group's drivers into their buckets; its blocks of one group share an XCD (one L2), where the 4-byte id
//     writes to a bucket merge before they leave.
//   * A bucket that would outgrow cap, or a run that would outgrow scap, raises the status word (no out-of-range store is issued);
//     the context doubles its regions and the step is repeated (cpm_api.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstring>
#include <type_traits>
#include <string>

#include "../../include/cpm.h"
#include "cpm_kernels.h"

namespace cpm {

constexpr int kGroups = 32;              // destination groups
constexpr int kFusedThreads = 256;       // threads of every block of the fused hour (k_grouped_hour)
constexpr int kMaxZonesPerGroup = 1024;  // LDS bins of the place kernel
constexpr uint32_t kHiMax = 0xFFFFFFFFu;
constexpr int kMaxCapMult = 64;
// Words that hand a zone over between workgroups of ONE launch (the day in one launch, cpm_day.h) carry flags beside their count:
//   stayer-count word of (hour, zone): bit 31 = the zone's sampler workgroup of that hour is done (stayers, runs, run lengths complete),
//   bit 30 = ... but it gave up (nothing of the zone was read or written: the step is invalid), bits 26..29 = the XCD it ran on.
//   Every reader of a stayer count masks with kCntMask (counts are < 2^26: grouped_day_fits).
//   hand-off counters: arrivals in the low 16 bits; a workgroup that gave up counts itself in with kDoneAbort on top, and whoever
//   waits on a counter that carries such a mark gives up too (reads nothing, marks its own hand-offs): a bail-out drains the grid.
constexpr uint32_t kCntValid = 0x80000000u, kCntAbort = 0x40000000u;
constexpr int kCntXccShift = 26;
constexpr uint32_t kCntMask = (1u << kCntXccShift) - 1u;
constexpr uint32_t kDoneAbort = 0x10000u, kDoneCount = 0xFFFFu;

// Destination groups: kGroups groups of zpg = ceil(Z / kGroups) CONSECUTIVE zones (a placing block writes to neighbouring buckets).
// Rounds 1-3 took zpg as a power of two (a shift and a mask per driver): Z = 2,357 then used 19 of the 32 groups, its runs were 1.7 x
// as long -- most of them beyond what a placing block's lanes hold -- and 13 of every 32 placing blocks of the one-launch hour had
// nothing to do (tools/hour_stamps.py, profiles/round4_notes.md).  GENERAL groups (any zpg):
//   group(dest) = (dest * M) >> s,  M = ceil(2^s / zpg),  2^s >= Z * zpg   (exact for every dest < Z; both factors < 2^24: one
//   full-rate v_mul_u32_u24),  packed with zpg into ONE argument word: M in bits 0..16, s in 17..21, zpg - 1 in 22..31.
// The multiply and the subtraction per driver cost the dense headline 1.1 % (same box, interleaved), so the general form is what the
// SPARSE sampler instantiations run (datasets: Melbourne's Z = 2,357 and its like) and dense tables keep power-of-two groups -- in the
// same word: M = 1, s = log2 zpg, which the general formula also reads correctly (the kernels that are not hot use it for both).
inline uint32_t grouped_zpg_of(int Z, bool general = false)
{
    if (general) return static_cast<uint32_t>(std::max(1, (Z + kGroups - 1) / kGroups));
    uint32_t s = 0;
    while ((static_cast<int64_t>(kGroups) << s) < Z) ++s;
    return 1u << s;
}
inline uint32_t grouped_gdiv_of(int Z, bool general = false)
{
    const uint32_t zpg = grouped_zpg_of(Z, general);
    uint32_t sh = 0;
    if (!general) {
        while ((1u << sh) < zpg) ++sh;
        return 1u | sh << 17 | (zpg - 1u) << 22;
    }
    sh = 1;
    while ((1ull << sh) < static_cast<unsigned long long>(std::max(Z, 1)) * zpg) ++sh;
    const uint32_t M = static_cast<uint32_t>(((1ull << sh) + zpg - 1) / zpg);
    return M | sh << 17 | (zpg - 1u) << 22;
}
__host__ __device__ __forceinline__ uint32_t gdiv_zpg(uint32_t p) { return (p >> 22) + 1u; }
__host__ __device__ __forceinline__ uint32_t gdiv_group(uint32_t p, uint32_t dest)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(dest, p & 0x1FFFFu) >> ((p >> 17) & 31u);
#else
    return static_cast<uint32_t>((static_cast<unsigned long long>(dest) * (p & 0x1FFFFu)) >> ((p >> 17) & 31u));
#endif
}
// a driver's destination inside its group g (what the packed word carries above the car id)
__host__ __device__ __forceinline__ uint32_t gdiv_local(uint32_t p, uint32_t dest, uint32_t g) { return dest - g * gdiv_zpg(p); }
// GEN = false: the word describes power-of-two groups (a shift and a mask)
template <bool GEN>
__device__ __forceinline__ uint32_t gdiv_group_t(uint32_t p, uint32_t dest)
{
    if constexpr (GEN) return gdiv_group(p, dest);
    else return dest >> ((p >> 17) & 31u);
}
template <bool GEN>
__device__ __forceinline__ uint32_t gdiv_local_t(uint32_t p, uint32_t dest, uint32_t g)
{
    if constexpr (GEN) return gdiv_local(p, dest, g);
    else return dest & (p >> 22);
}

// ------------------------------------------------------------------------------------------------ row packs
__host__ __device__ inline int pack_guide_bits(int Z)
{
    int g = 3;  // >= 8 entries: the guide is a whole number of 16-B pieces
    while ((1 << g) < Z) ++g;
    // a quarter of an entry per destination: measured best at S4k (entries per destination 1: 32.0 us, 1/2: 29.9, 1/4: 29.1 --
    // the shorter pack outweighs the longer bracket)
    g = g - 2 < 3 ? 3 : g - 2;
    return g;
}
// high words per row: Z, then at least 31 entries of 0xFFFFFFFF (the unclamped stride walk of pack_search reads up to 30 past
// its bracket), a whole number of 128-B lines
__host__ __device__ inline int pack_zq(int Z) { return (Z + 31 + 31) / 32 * 32; }
__host__ __device__ inline int pack_guide_words(int G) { return (1 << G) / 2 + 4; }  // 2^G + 1 entries used (+7 pad: whole 16-B pieces)
// smap (sparse pack, cpm_dataset.h): Zq and G of the compact row, and a u16 destination per entry behind the high words
__host__ __device__ inline int pack_row_words(int Zq, int G, int smap = 0)  // at least 1 KiB: one whole LDS-DMA wave-instruction
{
    const int w = pack_guide_words(G) + Zq + (smap ? Zq / 2 : 0);  // (Zq is a multiple of 32, the guide of 4: whole 16-byte pieces)
    return w < 256 ? 256 : w;  // (rows rounded up to whole 128-byte lines: -0.3 % at S4k, within noise -- profiles/round4_notes.md)
}
// a row pack must fit the 150 KiB of LDS a workgroup may ask for, and a guide entry is a u16
inline bool pack_row_fits(int Z)
{
    if (Z < 2 || Z > 32768) return false;
    return sizeof(uint32_t) * static_cast<size_t>(pack_row_words(pack_zq(Z), pack_guide_bits(Z))) <= 150 * 1024;
}

// thr[t][z] = bernoulli_threshold(p_drive[t][z]): the integer the sampler compares the 53-bit draw with (src/resampling.jl:15 as
// k <= floor(p * 2^53); one scalar load per workgroup instead of f64 arithmetic in every thread)
__global__ __launch_bounds__(256) void k_build_thr(const double *__restrict__ pdrive, long long *__restrict__ thr, int64_t n)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) thr[i] = bernoulli_threshold(pdrive[i]);
}

// ------------------------------------------------------------------------------------------------ bucketing
// Car-indexed state -> fixed-stride buckets (once per state; cached until the state changes): LDS histogram of the zones of this
// block's cars -> one batched round of global atomics reserves the block's range in every bucket -> ids move to zone*cap + position.
constexpr int kBucketBlock = 1024;
constexpr int kBucketMaxPass = 512;  // 1024-car passes per block

__global__ __launch_bounds__(kBucketBlock) void k_bucket_cars(const uint32_t *__restrict__ zone0, int64_t n, int64_t chunk, int Z, uint32_t cap,
                                                              uint32_t *__restrict__ cnt, uint32_t *__restrict__ ids, unsigned long long *status)
{
    extern __shared__ uint32_t bins[];  // Z: histogram, then running position inside each bucket
    const int tid = threadIdx.x;
    for (int z = tid; z < Z; z += kBucketBlock) bins[z] = 0;
    __syncthreads();
    const int64_t i0 = static_cast<int64_t>(blockIdx.x) * chunk, i1 = min(i0 + chunk, n);
    constexpr int kU = 8;  // passes in flight per thread: all their loads are issued before the first use
    for (int64_t p0 = i0; p0 < i1; p0 += static_cast<int64_t>(kU) * kBucketBlock) {
        uint32_t v[kU];
        bool ok[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int64_t i = p0 + static_cast<int64_t>(u) * kBucketBlock + tid;
            ok[u] = i < i1;
            if (ok[u]) v[u] = zone0[i];
        }
#pragma unroll
        for (int u = 0; u < kU; ++u)
            if (ok[u]) atomicAdd(&bins[v[u] & kZoneMask], 1u);
    }
    __syncthreads();
    for (int zb = 0; zb < Z; zb += kBucketBlock * 4) {  // ticket: bins[z] becomes this block's first position inside bucket z
        uint32_t r[4], c[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int z = zb + tid + k * kBucketBlock;
            r[k] = 0;
            c[k] = 0;
            if (z < Z) {
                c[k] = bins[z];
                if (c[k]) r[k] = atomicAdd(&cnt[z], c[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int z = zb + tid + k * kBucketBlock;
            if (z < Z) {
                bins[z] = r[k];
                if (r[k] + c[k] > cap) atomicOr(status, 2ull);  // bucket outgrew its region: step invalid
            }
        }
    }
    __syncthreads();
    for (int64_t p0 = i0; p0 < i1; p0 += static_cast<int64_t>(kU) * kBucketBlock) {  // (zones re-read: L2-resident from the histogram pass)
        uint32_t v[kU];
        bool ok[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int64_t i = p0 + static_cast<int64_t>(u) * kBucketBlock + tid;
            ok[u] = i < i1;
            if (ok[u]) v[u] = zone0[i] & kZoneMask;
        }
#pragma unroll
        for (int u = 0; u < kU; ++u)
            if (ok[u]) {
                const uint32_t p = atomicAdd(&bins[v[u]], 1u);
                if (p < cap) ids[static_cast<size_t>(v[u]) * cap + p] = static_cast<uint32_t>(p0 + static_cast<int64_t>(u) * kBucketBlock + tid);
            }
    }
}

// car-indexed state from fixed-stride buckets (end of the IVP): both ends of every region; a bucket whose ends met is flagged here
// (the hour that filled it has no sampler launch behind it to notice)
__global__ void k_unbucket(const uint32_t *__restrict__ ids, uint32_t *__restrict__ cnt_s, const uint32_t *__restrict__ cnt_a, uint32_t cap,
                           uint32_t *__restrict__ zone0, uint32_t n, unsigned long long *status)
{
    const uint32_t z = blockIdx.x;
    const uint32_t ns_raw = cnt_s[z] & kCntMask, na_raw = cnt_a[z];  // (the day launch leaves flags beside the stayer counts ...
    if (threadIdx.x == 0) cnt_s[z] = ns_raw;                         //  ... and they end here: these buckets become the cached ones, which every form of the hour reads)
    const uint32_t ns = min(ns_raw, cap), na = min(na_raw, cap - ns);
    if (threadIdx.x == 0 && static_cast<unsigned long long>(ns_raw) + na_raw > cap) atomicOr(status, 2ull);
    const uint32_t gap = cap - ns - na;
    for (uint32_t s = threadIdx.x; s < ns + na; s += blockDim.x) {
        const uint32_t id = ids[static_cast<size_t>(z) * cap + s + (s >= ns ? gap : 0u)];
        if (id < n) zone0[id] = z;  // (ids of a step that is about to be rejected -- status != 0 -- may be anything)
    }
}

// ------------------------------------------------------------------------------------------------ hourly sampler
// What only the rare branches of the hourly kernels read (the exact fallback of a tie, a heavy bucket, an overflow): in device
// memory, behind ONE pointer of the launch arguments -- the sampler lives at its SGPR limit (106, scalars spilled into VGPR lanes
// and read back with v_readlane in its hot paths), and nine more pointers in its arguments cost it 6 us per launch.  Written by
// k_grouped_zero at the start of every run.
struct GroupedRare {
    const double *ckpt, *p;               // [T][nck][Z] running-sum checkpoints, [T][Z dest][Z origin] p_destin (dense tables)
    const double *sp;                     // sparse tables (cpm_dataset.h): [T*Z][scap] normalised p of every row's cells, in destination order; else null
    const uint32_t *sj;                   // ... their destinations
    const uint32_t *scnt;                 // ... [T*Z] cells per row
    uint32_t scap, pad0_;
    uint32_t *maxn, *heavy_list, *nheavy; // [2] largest heavy bucket / most heavy buckets of an hour; [hgrid] zones of this hour's heavy launch; [T+1] how many per hour
    unsigned long long *status;           // the status word of the run's count tensor
    uint32_t hgrid, parts;                // zones the heavy launch covers, workgroups per heavy zone (GroupedWork)
    int Z, pad_;
};

struct GroupedArgs {
    const uint32_t *ids;      // [Z*cap] this hour's buckets (local car indices): stayers from the bottom of a zone's region, arrivals from its top
    const uint32_t *cnt_s;    // [Z] stayers per bucket
    const uint32_t *cnt_a;    // [Z] arrivals per bucket
    const uint32_t *rp_t;     // [Z][RW] row packs of this hour
    const double *last_t;     // [Z] row totals (f64)
    const long long *thr_t;   // [Z] Bernoulli thresholds floor(p_drive * 2^53) of this hour
    const GroupedRare *rare;  // tables of the exact fallback, heavy-bucket bookkeeping, status word
    uint32_t *ids_next;       // [Z*cap]  next hour's buckets: the stayers (grouped) / dest | drive << 31 per slot (plain)
    uint32_t *cnt_next;       // [Z] next hour's stayers (grouped); the arrivals are counted by k_grouped_place, in an array of their own
    uint32_t *D;              // [Z][kGroups][scap] packed drivers (grouped)
    uint32_t *cntg;           // [Z][kGroups] run lengths (grouped)
    unsigned long long *parking_t, *driving_t;
    int Z, Zq, G;             // (sparse packs, cpm_dataset.h: Zq and G of the COMPACT row)
    int hour;                 // table hour of this launch (0-based): rare->nheavy[hour] counts its heavy buckets
    uint32_t *done_t;         // fused hour: [chunks] sampler workgroups of every chunk of origin zones that have handed their runs over
    int lag;                  // fused hour: the placing blocks of chunk j sit behind the sampler workgroups of chunk j + lag
    uint32_t heavy_x;         // a bucket is heavy above heavy_x x the slots of its sampler workgroup (kHeavy; less once the heavy launch runs anyway)
    uint32_t spin_limit;      // fused hour: polls a placing block makes before it gives up (0: at once -- the tests' way into the bail-out)
    // placing first (k_grouped_hour_pf): the runs of the PREVIOUS hour, which this launch's placing blocks move into THIS hour's buckets
    // (ids / cnt_a above) before its sampler workgroups read them; pchunks = chunks of origin zones to place (0: nothing pending)
    const uint32_t *pD, *pcntg;
    int pchunks;
    uint32_t cap, scap, idbits, gdiv, step;   // (gdiv: the packed group divisor, grouped_gdiv_of)
    // (rare->parts == 1: a launch walks whole buckets in overflow rounds of BLOCK cars.  > 1: of a HEAVY bucket -- more than
    //  kHeavy * CPT * BLOCK cars -- that gets a place in rare->heavy_list it takes the first CPT * BLOCK cars only;
    //  k_grouped_sample_heavy, launched behind it with parts - 1 blocks per listed zone, takes the rest)
    CarIndex cars;
    uint64_t seed;
    // (what the round-4 forms read, BEHIND everything the hourly kernels read: the argument segment is loaded in aligned blocks, and
    //  fields put among the hot ones cost the hourly sampler -- 106 scalar registers, at its ceiling -- six more spills)
    int Zc;                   // entries of a pack row in front of its 0xFFFFFFFF pad: Z, or the longest compact row of a sparse table
    int smap;                 // 1: sparse pack -- the search's answer is an entry, its destination stands in the u16 map behind the high words
    // the day in one launch (k_grouped_day, cpm_day.h): an hour's segment of the grid = the placing blocks of the hour before among
    // this hour's sampler workgroups.  sdone: [chunks] lines, sampler workgroups of THIS hour that have handed their runs over (by
    // chunk of origin zones); psdone: the same of the hour before (what the segment's placing blocks wait for); pdone: [kGroups]
    // lines, the segment's placing blocks that are done (by destination group) and the XCDs they ran on; chained: this hour's
    // buckets were written by blocks of the same launch (flags in cnt_s, every id load past the L1)
    uint32_t *sdone;
    const uint32_t *psdone;
    uint32_t *pdone;
    uint32_t chained;
    const uint32_t *perm_t;   // k_grouped_hour<PERM>: [Z] the zones of this hour, largest first; null: zone order
};


// A barrier for LDS data only: this wave's LDS operations have completed, then the workgroup meets.  __syncthreads() is also a
// workgroup-scope fence, for which hipcc waits until the wave's global stores and ATOMICS have completed (s_waitcnt vmcnt(0)) -- a
// memory round trip in front of every barrier that follows a ticket or a store, although only LDS contents change hands there.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Votes on a condition that already is a lane mask: HIP's __ballot / __any take an int, and hipcc materialises the mask as 0 / 1 in
// a VGPR and compares it again (two VALU instructions per vote; the sampler is bound by what it issues).
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// the XCD this wave runs on (HW_REG_XCC_ID, bits 3:0)
__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) & 15u; }
// Inclusive prefix sum over the 64 lanes of a wave in the vector ALU (DPP: shifts inside a row of 16 lanes, then the last lane of
// a row broadcast to the rows behind it): six dependent additions and no LDS instruction.  __shfl_up is a ds_bpermute per step --
// a round trip through the LDS pipeline, which the sampler workgroups sharing the CU keep busy: the placing blocks' scan of their
// histogram took 4.9 k cycles of a 13 k-cycle block with it (tools/hour_stamps.py).
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x)
{
    int v = static_cast<int>(x);
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1 (lanes without a source keep the 0)
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return static_cast<uint32_t>(v);
}
__device__ __forceinline__ bool any64(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
// p[i] = v with the address as uniform base + 32-bit BYTE offset (i x 4 B < 2^32: a bucket region): the store takes the base from
// scalar registers and one VGPR instead of a 64-bit address pair built per store
__device__ __forceinline__ void put32(uint32_t *p, uint32_t i, uint32_t v)
{
    *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(p) + (i << 2)) = v;
}
// lane 0's value to the whole wave (every lane active): one v_readfirstlane instead of the LDS crossbar of __shfl
__device__ __forceinline__ uint32_t from_lane0(uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v))); }

// ------------------------------------------------------------------------------------------------ row tables in ONE pass
// Everything the samplers read of p_destin is derived here from the table as the reference lays it out, p[o + Z*(d + Z*t)]
// (origin fastest), in one pass over it:
//   * the running sum of a row, left to right in f64 -- range_up = range_up + distribution[j] of src/resampling.jl:39; a tree scan
//     would move boundaries by ulps -- by ONE lane per origin (the 64 origins of a tile are 64 consecutive words of p: coalesced);
//   * CDF && cdf  : the canonical CDF rows cdf[t][o][d] (f64, padded with +inf to Zp) -- only for the kernels that search f64 rows
//                   (one thread per car, exact layout); the grouped path never asks for them;
//   * PACK        : the row packs of the grouped sampler: hi[j] = floor(cdf[j] * 2^32) (0xFFFFFFFF from 1.0 on and in the pad) and the
//                   guide, guide[m] = min(first j with hi[j] >= m << (32 - G), Z - 1), which falls out of the same walk: destination j
//                   owns the entries m with hi[j-1] < m << sh <= hi[j], i.e. (hi[j-1] >> sh) + 1 .. hi[j] >> sh, and Z - 1 owns the rest;
//   * last[t][o]  : the row total;
//   * ckpt[t][k][o] = cdf[min(32 k + 31, Z - 1)]: every 32nd value of the running sum.  A tie of the high-word search repeats the
//                   reference's walk from the checkpoint in front of it (search_exact_ckpt): at most 32 sequential additions of the
//                   same p entries, bit-identical to the full running sum, without 8 bytes per destination in HBM.
// Block = 4 waves: wave 0 sums (32-64 loads in flight per lane), waves 1-3 take the 64 x 64 tile it leaves in LDS (two buffers, one
// barrier per tile) and write it out row-major: consecutive lanes, consecutive destinations of one row.  Two blocks per CU (LDS), two
// waves per SIMD: the summing wave may use 256 VGPRs (its two half tiles in flight are 128 of them; with the 168 of three waves per
// SIMD hipcc spilled the destinations of the hand-written loads -- before their data had landed).
// (Round 2 built the same in three launches -- CDF, high words from the CDF just written, guide by a binary search per entry over the
// high words just written -- 13.4 GB of traffic for the 8.3 GB one pass needs, 3.77 ms at S4k.)
constexpr int kRowTile = 64;
constexpr int kRowBlock = 256;
constexpr int kCkptStride = 32;
__host__ __device__ inline int ckpt_count(int Z) { return (Z + kCkptStride - 1) / kCkptStride; }
constexpr size_t kRowLds = sizeof(double) * 2 * kRowTile * (kRowTile + 1) + sizeof(uint32_t) * kRowTile;

// The summing wave of k_build_rows: lane = origin.  Tiles that lie wholly inside the row take the pipelined path: one half tile of
// 32 loads is always in flight behind the one being added up.  The loads are buffer loads -- descriptor on the tile's first row,
// the row's byte offset in a scalar register, the lane's 8-byte offset in ONE VGPR: with plain pointers hipcc built the 64 row
// addresses of a tile as 64-bit VGPR chains and spilled, and a predicated load among them made it wait vmcnt(0), so that a half
// tile's loads no longer overlapped the other half's additions (3.3 ms at S4k).  Unconditional and of one kind, hipcc counts them
// itself (s_waitcnt vmcnt(32), 31, ... in front of the additions).  The last one or two tiles (the row's end and the pad) take a
// plain path: rows beyond the table are clamped to its last row and their values unused.
typedef uint32_t cpm_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t cpm_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double row_load(__amdgpu_buffer_rsrc_t rows, uint32_t lane_off, uint32_t row_off)
{
    const cpm_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rows, lane_off, row_off, 0);
    return __hiloint2double(static_cast<int>(v.y), static_cast<int>(v.x));
}

__device__ __forceinline__ void row_sums(double (*tile)[kRowTile][kRowTile + 1], const double *__restrict__ src, int lane, bool live,
                                         double *__restrict__ ck, double *__restrict__ last_o, int Z, int nt, int *err)
{
    constexpr int H = kRowTile / 2;
    const uint32_t lane_off = static_cast<uint32_t>(live ? lane : 0) << 3;
    const size_t Zs = static_cast<size_t>(Z);
    const uint32_t rowb = static_cast<uint32_t>(Z) * 8u;  // bytes between consecutive destinations of one origin
    double run = 0.0;
    bool bad = false;
    double xa[H], xb[H];
    auto window = [&](const double *row0) {  // (2 tiles of rows: 128 x Z x 8 B < 2^32)
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(row0), 0, static_cast<int>(2 * kRowTile * rowb), 0x00020000);
    };
    auto sums = [&](double(&x)[H], int buf, int jbase, double *ck_slot) {
#pragma unroll
        for (int u = 0; u < H; ++u) {
            bad |= !(x[u] >= 0.0);
            run = run + x[u];
            tile[buf][lane][jbase + u] = run;
        }
        if (ck_slot && live) *ck_slot = run;  // every 32nd value of the running sum
    };
    const int nfull = Z / kRowTile;  // tiles wholly inside [0, Z)
    const double *row0 = src;
    __amdgpu_buffer_rsrc_t rows = window(row0);
    if (nfull > 0) {
#pragma unroll
        for (int u = 0; u < H; ++u) xa[u] = row_load(rows, lane_off, static_cast<uint32_t>(u) * rowb);
    }
    // (the last whole tile is peeled off: inside the loop the prefetch of the next tile is unconditional, so the waits hipcc
    //  counts in front of the second half's additions leave it in flight)
#pragma unroll 1
    for (int k = 0; k + 1 < nfull; ++k) {
#pragma unroll
        for (int u = 0; u < H; ++u) xb[u] = row_load(rows, lane_off, static_cast<uint32_t>(H + u) * rowb);
        sums(xa, k & 1, 0, ck ? ck + static_cast<size_t>(2 * k) * Zs : nullptr);
#pragma unroll
        for (int u = 0; u < H; ++u) xa[u] = row_load(rows, lane_off, static_cast<uint32_t>(kRowTile + u) * rowb);
        sums(xb, k & 1, H, ck ? ck + static_cast<size_t>(2 * k + 1) * Zs : nullptr);
        lds_barrier();  // (LDS only: the loads of the next tile stay in flight)
        row0 += kRowTile * Zs;
        rows = window(row0);
    }
    if (nfull > 0) {
        const int k = nfull - 1;
#pragma unroll
        for (int u = 0; u < H; ++u) xb[u] = row_load(rows, lane_off, static_cast<uint32_t>(H + u) * rowb);
        sums(xa, k & 1, 0, ck ? ck + static_cast<size_t>(2 * k) * Zs : nullptr);
        sums(xb, k & 1, H, ck ? ck + static_cast<size_t>(2 * k + 1) * Zs : nullptr);
        lds_barrier();
    }
    if (nfull * kRowTile == Z && live && last_o) *last_o = run;
    const int tail_row = min(nfull * kRowTile, Z - 1);
    rows = window(src + static_cast<size_t>(tail_row) * Zs);
#pragma unroll 1
    for (int k = nfull; k < nt; ++k) {  // the row's end and the pad (+inf): half tiles one after the other
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
            const int d0 = k * kRowTile + h * H;
#pragma unroll
            for (int u = 0; u < H; ++u) xa[u] = row_load(rows, lane_off, static_cast<uint32_t>(min(d0 + u, Z - 1) - tail_row) * rowb);
#pragma unroll
            for (int u = 0; u < H; ++u) {
                const int d = d0 + u;
                double v = __builtin_huge_val();
                if (d < Z) {
                    bad |= !(xa[u] >= 0.0);
                    run = run + xa[u];
                    v = run;
                    if (live && (u == H - 1 || d == Z - 1)) {
                        if (ck) ck[static_cast<size_t>(d / kCkptStride) * Zs] = run;
                        if (d == Z - 1 && last_o) *last_o = run;
                    }
                }
                tile[k & 1][lane][h * H + u] = v;
            }
        }
        lds_barrier();
    }
    if (bad && live) atomicOr(err, 1);
}

template <bool CDF, bool PACK>
__global__ __launch_bounds__(kRowBlock, 2) void k_build_rows(const double *__restrict__ p, double *__restrict__ cdf,
                                                          uint32_t *__restrict__ rp, double *__restrict__ last, double *__restrict__ ckpt,
                                                          int Z, int Zp, int Zq, int G, int *err)
{
    extern __shared__ __attribute__((aligned(16))) double row_lds[];
    double(*tile)[kRowTile][kRowTile + 1] = reinterpret_cast<double(*)[kRowTile][kRowTile + 1]>(row_lds);  // [2][64 origins][64 destinations]
    uint32_t *prevh = reinterpret_cast<uint32_t *>(row_lds + 2 * kRowTile * (kRowTile + 1));              // last high word of each row's previous tile
    const int t = blockIdx.y;
    const int o0 = blockIdx.x * kRowTile;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int dmax = PACK ? Zq : Zp;  // (Zq >= Zp)
    const int nt = (dmax + kRowTile - 1) / kRowTile;
    if (wave == 0) {
        const int o = o0 + lane;
        const bool live = o < Z;
        const double *src = p + static_cast<size_t>(t) * Z * Z + o0;  // (wave-uniform: a load is scalar row base + the lane's 8-byte offset)
        double *ck = ckpt ? ckpt + static_cast<size_t>(t) * ckpt_count(Z) * Z + o : nullptr;
        double *last_o = last ? last + static_cast<size_t>(t) * Z + o : nullptr;
        row_sums(tile, src, lane, live, ck, last_o, Z, nt, err);
        return;
    }
    // ---- the tiles out: a lane takes FOUR consecutive destinations of a row, sixteen lanes a row's 64, a wave four rows at a time: one
    // 16-byte store per lane for the high words (and two for the f64 rows) -- with a lane per destination the launch was bound by the
    // store instructions it issued (256 B each: 0.62 ms of 1.56 at S4k) and by ~45 instructions per destination for the guide (0.35 ms).
    const int nrow = min(kRowTile, Z - o0);
    const int sh = 32 - G;
    const int gw = pack_guide_words(G);
    const size_t rw = static_cast<size_t>(pack_row_words(Zq, G));
    const int q16 = lane & 15, rr = lane >> 4;
    const int mtop = (1 << G) + 7;  // the last guide entry (pad included)
    for (int k = 0; k < nt; ++k) {
        lds_barrier();
        const int d = k * kRowTile + 4 * q16;  // the lane's first destination
        for (int grp = wave - 1; grp * 4 < nrow; grp += kRowBlock / 64 - 1) {
            const int r = grp * 4 + rr;
            const bool row_live = r < nrow;
            const int rc = row_live ? r : 0;  // (lanes of rows beyond the tile's last one go through the motions on row 0 and store nothing)
            double c[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = tile[k & 1][rc][4 * q16 + i];
            const size_t row = static_cast<size_t>(t) * Z + o0 + rc;
            if (CDF && row_live && d < Zp) {  // (Zp is a multiple of 16: a quad lies wholly inside or outside)
                double2 *dst = reinterpret_cast<double2 *>(cdf + row * Zp + d);
                dst[0] = make_double2(c[0], c[1]);
                dst[1] = make_double2(c[2], c[3]);
            }
            if (PACK) {
                uint32_t *pack = rp + row * rw;
                uint32_t h[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    h[i] = kHiMax;
                    if (c[i] < 1.0) h[i] = static_cast<uint32_t>(floor(c[i] * 0x1.0p32));  // exact scaling; c >= 0 (validated), +inf in the pad
                }
#ifndef CPM_ROWS_NO_HI  // (ablation builds, tools/build_variants.sh: what the write-out costs)
                if (row_live && d < Zq) *reinterpret_cast<uint4 *>(pack + gw + d) = make_uint4(h[0], h[1], h[2], h[3]);  // (Zq: a multiple of 32)
#endif
#ifndef CPM_ROWS_NO_GUIDE
                // guide: destination j owns the entries m with hi[j-1] < m << sh <= hi[j]; the lane's four destinations own a
                // contiguous range [m_first, m_last] between them, entry m going to the first of them with m <= hi >> sh
                uint32_t hp = __shfl_up(h[3], 1, 64);
                if (q16 == 0) hp = prevh[rc];
                if (q16 == 15 && row_live) prevh[rc] = h[3];  // (row r is this wave's in every tile)
                int m1[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)  // (Z - 1 owns the entries no destination reaches and the pad; what lies behind it owns nothing)
                    m1[i] = (d + i < Z - 1) ? static_cast<int>(h[i] >> sh) : mtop;
                const int m_first = (d == 0) ? 0 : static_cast<int>(hp >> sh) + 1;
                const int m_last = (row_live && d < Z) ? m1[3] : -1;  // (m1 is non-decreasing)
                uint16_t *guide = reinterpret_cast<uint16_t *>(pack);
                auto owner = [&](int m, int a0, int a1, int a2, int dbase) {  // first of the four destinations with m <= its last entry
                    return static_cast<uint16_t>(dbase + (m > a0 ? 1 : 0) + (m > a1 ? 1 : 0) + (m > a2 ? 1 : 0));
                };
                const int n = m_last - m_first + 1;
#pragma unroll
                for (int i = 0; i < 4; ++i)  // (dense rows: one entry per four destinations on average)
                    if (i < n) guide[m_first + i] = owner(m_first + i, m1[0], m1[1], m1[2], d);
                // Longer ranges (sparse rows: a destination with weight among empty ones owns a handful of entries; peaky and empty
                // rows): the entries left over in the wave are dealt over its lanes, entry e of the concatenated ranges to lane
                // e mod 64, which finds the lane it comes from in the running totals.  (One range at a time with the whole wave
                // took eight rounds per pass on Melbourne-shaped tables: 1.5 ms for the kernel against 0.6 on dense ones.)
                if (ballot64(n > 4)) {
                    const int extra = max(n - 4, 0);
                    const int incl = static_cast<int>(wave_incl_scan(static_cast<uint32_t>(extra)));
                    const int total = __shfl(incl, 63, 64);
                    for (int e = lane; e - lane < total; e += 64) {
                        int src = 0;  // first lane whose running total lies above e
#pragma unroll
                        for (int sft = 32; sft; sft >>= 1)
                            if (__shfl(incl, src + sft - 1, 64) <= e) src += sft;
                        const int off = e - (__shfl(incl, src, 64) - __shfl(extra, src, 64));
                        const int m = __shfl(m_first, src, 64) + 4 + off;
                        const int a0 = __shfl(m1[0], src, 64), a1 = __shfl(m1[1], src, 64), a2 = __shfl(m1[2], src, 64);
                        const int dbase = __shfl(d, src, 64);
                        uint16_t *gsrc = reinterpret_cast<uint16_t *>(rp + (static_cast<size_t>(t) * Z + o0 + grp * 4 + (src >> 4)) * rw);
                        if (e < total) gsrc[m] = owner(m, a0, a1, a2, dbase);
                    }
                }
#endif
            }
        }
    }
}

// The categorical draw of src/resampling.jl:38-45 for ONE car on the table itself (ties of the high-word search, draws above the row
// total): first j with cdf[j] >= ue -- found from the checkpoints (binary search over every 32nd value of the running sum), then the
// reference's own walk, run = run + p[j], from the checkpoint in front.  Same additions in the same order as the full running sum.
// hint >= 0: a destination the answer is known not to lie in front of (a tie of the high-word search: every destination below
// the first one with hi >= khi has cdf < u) -- the walk then starts in the hint's block of 32 without searching the checkpoints.
// The loads of a batch do not depend on the running sum: eight are requested together (a tie stalls its workgroup for the round
// trips it takes; with one dependent load per destination the ~2 ties of every hourly launch added microseconds to its tail).
__device__ __forceinline__ uint32_t search_exact_tables(const double *__restrict__ ckpt_t, const double *__restrict__ p_t, int Z, int o, double uc,
                                                        double last, int hint)
{
    const double ue = clamp_u(uc, last);
    int k;
    if (hint >= 0) {
        k = hint / kCkptStride;
    } else {
        int lo = 0, n = ckpt_count(Z);
        while (n > 0) {  // first k with ckpt[k] >= ue (the last one is the row total >= ue)
            const int half = n >> 1;
            if (ckpt_t[static_cast<size_t>(lo + half) * Z + o] < ue) {
                lo += half + 1;
                n -= half + 1;
            } else {
                n = half;
            }
        }
        k = min(lo, ckpt_count(Z) - 1);
    }
    constexpr int kB = 8;
    double run = k ? ckpt_t[static_cast<size_t>(k - 1) * Z + o] : 0.0;
    for (int j0 = k * kCkptStride; j0 < Z; j0 += kB) {  // (the running sum carries on across blocks: same additions, same order)
        double x[kB];
#pragma unroll
        for (int u = 0; u < kB; ++u) x[u] = p_t[static_cast<size_t>(min(j0 + u, Z - 1)) * Z + o];
#pragma unroll
        for (int u = 0; u < kB; ++u) {
            if (j0 + u < Z) {
                run = run + x[u];
                if (run >= ue) return static_cast<uint32_t>(j0 + u);
            }
        }
    }
    return static_cast<uint32_t>(Z - 1);
}

// ... on a sparse table: the same walk over the row's cells in destination order (the zeros in between add nothing: every partial
// sum equals the dense row's there), from the row's first cell (at most scap cells; a tie is a 2^-32 event per cell)
__device__ __forceinline__ uint32_t search_exact_sparse(const double *__restrict__ sp, const uint32_t *__restrict__ sj, uint32_t n, double uc, double last)
{
    const double ue = clamp_u(uc, last);
    double run = 0.0;
    for (uint32_t e = 0; e < n; ++e) {
        run = run + sp[e];
        if (run >= ue) return sj[e];
    }
    return n ? sj[n - 1] : 0u;
}

// the exact fallback of one car of the hourly kernels (rare: out of line, reads its tables through the rare block)
__device__ __noinline__ uint32_t search_exact_ckpt(const GroupedRare *__restrict__ rare, int hour, int o, double uc, double last, int hint)
{
    const int Z = rare->Z;
    const size_t th = static_cast<size_t>(hour);

    return search_exact_tables(rare->ckpt + th * ckpt_count(Z) * Z, rare->p + th * Z * Z, Z, o, uc, last, hint);
}

// ... of the kernels that run on sparse packs (a function of its own: what a noinline callee needs of scalar registers is added to its
// callers', and the dense sampler has none to spare)
__device__ __noinline__ uint32_t search_exact_sparse_rare(const GroupedRare *__restrict__ rare, int hour, int o, double uc, double last)
{
    const size_t row = static_cast<size_t>(hour) * rare->Z + o;
    return search_exact_sparse(rare->sp + row * rare->scap, rare->sj + row * rare->scap, min(rare->scnt[row], rare->scap), uc, last);
}
template <bool SPARSE>
__device__ __forceinline__ uint32_t search_exact_any(const GroupedRare *__restrict__ rare, int hour, int o, double uc, double last, int hint)
{
    if constexpr (SPARSE) return search_exact_sparse_rare(rare, hour, o, uc, last);
    else return search_exact_ckpt(rare, hour, o, uc, last, hint);
}

typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
typedef __attribute__((address_space(3))) const uint16_t lds_cu16;

// Row pack -> LDS by LDS-DMA (global_load_lds_dwordx4): no VGPR destination, no ds_write; one wave-instruction moves
// 64 x 16 B to 1 KiB of consecutive LDS.  The destination is wave-uniform base + lane x 16, the source is per lane.
// Counts in vmcnt like any load.  Every wave issues exactly NQ instructions, unpredicated (so that the count is known at
// compile time and the wave can wait for its OLDER id loads alone with s_waitcnt vmcnt(NQ)): chunk k = 64 pieces from
// min(64 k, pieces - 64); chunks past the end repeat the last one (same bytes to the same LDS words).  pieces >= 64.
template <int BLOCK, int NQ>
__device__ __forceinline__ void pack_dma(uint32_t *lds, const uint32_t *pack, int pieces, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int m = 0; m < NQ; ++m) {
        const int p0 = min((m * (BLOCK / 64) + wave) * 64, pieces - 64);
        // (uniform base + 32-bit byte offset: the load takes the base from scalar registers and one VGPR, no 64-bit add per piece)
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const char *>(pack) + (static_cast<uint32_t>(p0 + lane) << 4),
                                         (__attribute__((address_space(3))) void *)(lds + 4 * p0), 16, 0, 0);
    }
}

// s_waitcnt vmcnt(NQ) carrying the id registers as in/out operands: no use of an id can move above it
template <int N, int NQ>
__device__ __forceinline__ void wait_ids(uint32_t (&id)[N])
{
    static_assert((N == 2 || N == 3 || (N >= 5 && N <= 9)) && NQ <= 63, "written for CPT = 1, 2, 4 .. 8");
    if constexpr (N == 9)
        asm volatile("s_waitcnt vmcnt(%9)" : "+v"(id[0]), "+v"(id[1]), "+v"(id[2]), "+v"(id[3]), "+v"(id[4]), "+v"(id[5]), "+v"(id[6]), "+v"(id[7]), "+v"(id[8]) : "n"(NQ) : "memory");
    else if constexpr (N == 8)
        asm volatile("s_waitcnt vmcnt(%8)" : "+v"(id[0]), "+v"(id[1]), "+v"(id[2]), "+v"(id[3]), "+v"(id[4]), "+v"(id[5]), "+v"(id[6]), "+v"(id[7]) : "n"(NQ) : "memory");
    else if constexpr (N == 7)
        asm volatile("s_waitcnt vmcnt(%7)" : "+v"(id[0]), "+v"(id[1]), "+v"(id[2]), "+v"(id[3]), "+v"(id[4]), "+v"(id[5]), "+v"(id[6]) : "n"(NQ) : "memory");
    else if constexpr (N == 6)
        asm volatile("s_waitcnt vmcnt(%6)" : "+v"(id[0]), "+v"(id[1]), "+v"(id[2]), "+v"(id[3]), "+v"(id[4]), "+v"(id[5]) : "n"(NQ) : "memory");
    else if constexpr (N == 5)
        asm volatile("s_waitcnt vmcnt(%5)" : "+v"(id[0]), "+v"(id[1]), "+v"(id[2]), "+v"(id[3]), "+v"(id[4]) : "n"(NQ) : "memory");
    else if constexpr (N == 3)
        asm volatile("s_waitcnt vmcnt(%3)" : "+v"(id[0]), "+v"(id[1]), "+v"(id[2]) : "n"(NQ) : "memory");
    else
        asm volatile("s_waitcnt vmcnt(%2)" : "+v"(id[0]), "+v"(id[1]) : "n"(NQ) : "memory");
}

// CPT draws against the staged pack, in lockstep (CPT independent LDS reads in flight per step):
// dest[c] = first j with hi[j] >= khi[c], ok[c] = the answer is certain (hi[dest] > khi).  want[c] == false: no search.
// The guide brackets the answer: j in [L, L + n], L = guide[m], n = guide[m+1] - L, m = khi >> sh (entries are clamped to Z-1
// and a draw above the row's last high word never searches, so the bracket is in range).  hi[] is non-decreasing over the WHOLE
// row and padded with 0xFFFFFFFF for >= 31 entries past Z-1 (Zq), so the lower bound inside the bracket is a descending-stride
// walk that needs no upper clamp: with 2^K > n,  for s = 2^(K-1) .. 1:  if (hi[L + o + s - 1] < khi) o += s  ends at o = the
// number of entries from L on that lie below khi = the answer's offset (<= n <= 2^K - 1; the largest index read is L + 2^K - 2).
// K is wave-uniform (the widest bracket among the wave's draws decides): three to five steps of
// {LDS read, compare, select, add} on the dense synthetic rows.  (Two strides per round trip -- three independent probes, the
// highest one below the draw wins -- shortened the dependent chain and lengthened the launch, 27.5 us against 26.8: the kernel is
// bound by what it issues, not by this latency; profiles/round2_notes.md.)  Brackets of 32 entries and more (rows with long runs of
// zero-probability zones) take the same walk from a larger K with the probe index clamped to the row.
// smap != null (sparse pack, cpm_dataset.h): the row is the COMPACT row of the destinations that hold weight; the entry found is mapped
// to its destination at the end (one more LDS read per draw; wave-uniform branch).
template <int CPT>
__device__ __forceinline__ void pack_search(const uint16_t *guide_g, const uint32_t *hi_g, const uint32_t (&khi)[CPT], const bool (&want)[CPT],
                                            int sh, uint32_t hi_last, int Zq, uint32_t (&dest)[CPT], bool (&ok)[CPT], const uint16_t *smap_g = nullptr)
{
    lds_cu16 *smap = (lds_cu16 *)smap_g;
    lds_cu16 *guide = (lds_cu16 *)guide_g;
    lds_cu32 *hi = (lds_cu32 *)hi_g;
    uint32_t kk[CPT], lo[CPT], n[CPT], nor = 0;
    bool in[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        in[c] = want[c] & (khi[c] <= hi_last);
        kk[c] = in[c] ? khi[c] : 0u;  // (a draw that does not search never moves: nothing is below 0)
        const uint32_t m = kk[c] >> sh;
        lo[c] = guide[m];
        n[c] = guide[m + 1];
    }
#pragma unroll
    for (int c = 0; c < CPT; ++c) {  // (masked: a draw that does not search sits in bracket 0, whose width would join the wave's vote on K
                                     //  -- leaving the select out cost 0.9 us per launch, half the lanes hold a stayer)
        n[c] = in[c] ? n[c] - lo[c] : 0u;
        nor |= n[c];
    }
    if (__builtin_expect(any64(nor >= 32u), 0)) {  // wave-uniform
        int K = 6;
        while (any64((nor >> K) != 0u)) ++K;
        for (uint32_t s = 1u << (K - 1); s != 0u; s >>= 1) {
            uint32_t v[CPT];
#pragma unroll
            for (int c = 0; c < CPT; ++c) v[c] = hi[min(lo[c] + s - 1u, static_cast<uint32_t>(Zq - 1))];
#pragma unroll
            for (int c = 0; c < CPT; ++c) lo[c] += (v[c] < kk[c]) ? s : 0u;
        }
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            dest[c] = lo[c];
            ok[c] = in[c] & (hi[lo[c]] > kk[c]);
        }
        if (smap_g) {
#pragma unroll
            for (int c = 0; c < CPT; ++c) dest[c] = smap[dest[c]];
        }
        return;
    }
    lds_cu32 *p[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) p[c] = hi + lo[c];
#define CPM_PACK_STEP(S)                                                                    \
    do {                                                                                    \
        uint32_t v_[CPT];                                                                   \
        _Pragma("unroll") for (int c = 0; c < CPT; ++c) v_[c] = p[c][(S) - 1];              \
        _Pragma("unroll") for (int c = 0; c < CPT; ++c) p[c] += (v_[c] < kk[c]) ? (S) : 0;  \
    } while (0)
    const bool a16 = any64(nor >= 16u);
    const bool a8 = a16 || any64(nor >= 8u);
    if (a16) CPM_PACK_STEP(16);
    if (a8) CPM_PACK_STEP(8);
    CPM_PACK_STEP(4);
    CPM_PACK_STEP(2);
    CPM_PACK_STEP(1);
#undef CPM_PACK_STEP
#pragma unroll
    for (int c = 0; c < CPT; ++c) {  // (LDS addresses are 32 bits wide: a plain pointer difference would be done in 64)
        const uint32_t fin = p[c][0];
        dest[c] = (static_cast<uint32_t>(reinterpret_cast<uintptr_t>(p[c])) - static_cast<uint32_t>(reinterpret_cast<uintptr_t>(hi))) >> 2;
        ok[c] = in[c] & (fin > kk[c]);
    }
    if (smap_g) {
#pragma unroll
        for (int c = 0; c < CPT; ++c) dest[c] = smap[dest[c]];
    }
}

// Philox words of (car, step, stream 0): Bernoulli integer kb (53 bits of words 0,1) and the categorical words (2,3)
__device__ __forceinline__ void car_draw_words(uint64_t seed, uint64_t car, uint32_t step, long long &kb, uint32_t &clo, uint32_t &chi)
{
    U4 r = philox4x32_10(static_cast<uint32_t>(car), static_cast<uint32_t>(car >> 32), step, 0u, static_cast<uint32_t>(seed),
                         static_cast<uint32_t>(seed >> 32));
    kb = static_cast<long long>(((static_cast<uint64_t>(r.y) << 32) | r.x) >> 11);
    clo = r.z;
    chi = r.w;
}

// One workgroup per origin zone.  (Resident workgroups each walking several zones, with or without the next zone's registers or
// pack prefetched, were measured no faster; profiles/round1_notes.md.)
// GROUPED: stayers compacted into next hour's bucket of the zone, drivers into the zone's fixed-size runs.  The drivers are
//          first ranked and staged in LDS (kStage entries per group) and written out by 8 lanes per run, 16 B per lane;
//          ranks beyond kStage go to HBM directly.
// !GROUPED: dest | drive << 31 per slot into ids_next (hour T of a resample: sampled, never applied, src/resampling.jl:81-83).
// Order of a workgroup's life: the two ends' sizes and the threshold (scalar loads), then the ids (their positions depend on the
// sizes since a bucket has two ends; clamped to the zone's region) and the row pack; Philox runs while the pack is landing;
// after the barrier the CPT cars of a thread search in lockstep and take their slots with one stayer ticket per wave and CPT
// rank atomics in flight together.
#ifndef CPM_STAGE
#define CPM_STAGE 32
#endif
constexpr int kStage = CPM_STAGE;
// A bucket is heavy above kHeavy x the slots of its sampler workgroup (4 x ~ the mean bucket: with the default bucket regions of
// 4 x the mean no bucket gets there, so the heavy kernel is launched only in contexts whose regions have had to grow).  Lighter
// overflow -- a quarter of the zones of the flat S4k tables hold a few cars more than CPT * BLOCK, the largest 2.5 x -- stays with
// the overflow rounds of the first workgroup: handing it to a second launch cost more than it saved (profiles/round2_notes.md).
constexpr uint32_t kHeavy = 4;
constexpr int kHeavyCap = 65536;  // work items (chunks of heavy buckets) the heavy launch can be handed in one hour (zones that do not fit stay with their first workgroup)
#ifndef CPM_WPS
#define CPM_WPS 6  // waves per SIMD the register allocator must leave room for (see profiles/round1_notes.md, round2_notes.md)
#endif

#ifdef CPM_NSGPR  // tuning builds: cap the scalar registers (256-thread workgroups per CU <= floor(800 / (ceil(sgpr / 16) * 16 + 16)))
#define CPM_SGPR_ATTR __attribute__((amdgpu_num_sgpr(CPM_NSGPR)))
#else
#define CPM_SGPR_ATTR
#endif
// Diagnostic builds (-DCPM_DIAGNOSTIC, never the product): s_memtime stamps of thread 0 of every block, kept in scalar registers
// and written to a side buffer when the block ends (a stamp that loaded the buffer's address on the spot made hipcc wait for every
// outstanding load and atomic in front of it: the timeline showed the stamps, not the kernel).
#ifdef CPM_DIAGNOSTIC
__device__ unsigned long long *g_place_stamps = nullptr;  // [blocks][8], set by cpm_diag_place_stamps
#define CPM_STAMP_DECL_IMPL unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define CPM_STAMP_IMPL(k) st_[k] = __builtin_amdgcn_s_memtime()
#define CPM_STAMP_FLUSH_IMPL                                                                          \
    do {                                                                                              \
        if (threadIdx.x == 0 && g_place_stamps) {                                                     \
            _Pragma("unroll") for (int k_ = 0; k_ < 8; ++k_) g_place_stamps[static_cast<size_t>(blockIdx.x) * 8 + k_] = st_[k_]; \
        }                                                                                             \
    } while (0)
#else
#define CPM_STAMP_DECL_IMPL \
    do {                    \
    } while (0)
#define CPM_STAMP_IMPL(k) \
    do {                  \
    } while (0)
#define CPM_STAMP_FLUSH_IMPL \
    do {                     \
    } while (0)
#endif
#define CPM_STAMP_NONE \
    do {               \
    } while (0)
#if defined(CPM_STAMP_BOTH)  // (-DCPM_DIAGNOSTIC -DCPM_STAMP_BOTH: both roles of the fused hour, one row per block of the launch; tools/hour_stamps.py)
#define CPM_SSTAMP_DECL CPM_STAMP_DECL_IMPL
#define CPM_SSTAMP(k) CPM_STAMP_IMPL(k)
#define CPM_SSTAMP_FLUSH CPM_STAMP_FLUSH_IMPL
#define CPM_PSTAMP_DECL CPM_STAMP_DECL_IMPL
#define CPM_PSTAMP(k) CPM_STAMP_IMPL(k)
#define CPM_PSTAMP_FLUSH CPM_STAMP_FLUSH_IMPL
#elif defined(CPM_STAMP_SAMPLER)  // (the same side buffer, filled by the sampler instead of the placing kernel: -DCPM_DIAGNOSTIC -DCPM_STAMP_SAMPLER)
#define CPM_SSTAMP_DECL CPM_STAMP_DECL_IMPL
#define CPM_SSTAMP(k) CPM_STAMP_IMPL(k)
#define CPM_SSTAMP_FLUSH CPM_STAMP_FLUSH_IMPL
#define CPM_PSTAMP_DECL CPM_STAMP_NONE
#define CPM_PSTAMP(k) CPM_STAMP_NONE
#define CPM_PSTAMP_FLUSH CPM_STAMP_NONE
#else
#define CPM_SSTAMP_DECL CPM_STAMP_NONE
#define CPM_SSTAMP(k) CPM_STAMP_NONE
#define CPM_SSTAMP_FLUSH CPM_STAMP_NONE
#define CPM_PSTAMP_DECL CPM_STAMP_DECL_IMPL
#define CPM_PSTAMP(k) CPM_STAMP_IMPL(k)
#define CPM_PSTAMP_FLUSH CPM_STAMP_FLUSH_IMPL
#endif

// LDS of a sampler workgroup beside the row pack (one object: the fused kernel overlays it with the placing blocks')
struct alignas(16) SampleLds {
    uint32_t stage[kGroups * kStage];  // (first: read back 16 bytes at a time)
    uint32_t gb[kGroups];
    uint32_t ndrive, nstay, split, pad_;
};

// A store / load of words that another workgroup of the SAME launch reads / has written (FUSED: the runs and run lengths, handed
// from the sampler workgroups to the placing blocks of the fused hour): write-through past this XCD's L2 and past the reader's L1
// (global_store / global_load ... sc1), as MI355X_MICROARCH.md's hand-off table prescribes.  !FUSED: plain.
// (Timing-only builds that left parts of the hand-off out -- placing blocks that do not wait, plain stores, no drain -- were used once
//  to price it (profiles/round3_notes.md) and are gone: a block that reads runs which are not complete indexes memory with what it
//  finds there, and one such build ended in a memory fault on the GPU.)
template <bool FUSED>
__device__ __forceinline__ void hand_store(uint32_t *p, uint32_t v)
{
    if constexpr (FUSED) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
template <bool FUSED>
__device__ __forceinline__ uint32_t hand_load(const uint32_t *p)
{
    if constexpr (FUSED) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
// FUSED: this workgroup's runs are complete -- every storing wave drains its stores, the workgroup meets, ONE lane counts the
// workgroup in (agent-scope add on the counter of its chunk of origin zones; the placing blocks of that chunk poll it)
template <bool FUSED>
__device__ __forceinline__ void hand_off_done(uint32_t *done_chunk, int tid)
{
    if constexpr (FUSED) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (tid == 0) __hip_atomic_fetch_add(done_chunk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// WAIT (placing first, k_grouped_hour_pf): the arrivals of this hour's buckets are written by placing blocks of the SAME launch
// (lower block indices).  A wave asks for the counter of its zone's destination group FIRST, then for the pack (LDS-DMA): with
// vmcnt <= NQ the counter is in its register while the pack is still landing.  Only when every placing block of the group has counted
// itself in (its stores drained) are the arrival count and the ids read -- past the L2 (sc1), as the placing blocks wrote them.
// Slot s of the workgroup: stayer s below the stayers' count, else arrival s - ns from the TOP of the region -- a position that
// does not depend on the arrivals' count, so count and ids are requested together.
// SPARSE: the row packs are sparse packs (cpm_dataset.h); a template parameter, not a.smap read at run time: the dense sampler lives
// at its SGPR ceiling and the pointer, the flag and the branch cost it seven more scalar spills (v_readlane in its hot paths).
template <int BLOCK, int CPT, int NQ, bool GROUPED, bool FUSED, bool WAIT = false, bool SPARSE = false>
__device__ __forceinline__ void grouped_sample_body(const GroupedArgs &a, const int z, uint32_t *pack, SampleLds &sl, uint32_t *done_chunk,
                                                    const uint32_t *wait_on = nullptr, uint32_t wait_need = 0)
{
    uint32_t &s_ndrive = sl.ndrive, &s_nstay = sl.nstay, &s_split = sl.split;
    uint32_t(&gb)[kGroups] = sl.gb;
    uint32_t(&stage)[kGroups * kStage] = sl.stage;
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t cap = a.cap;
    const uint32_t b = static_cast<uint32_t>(z) * cap;
    const int gw = pack_guide_words(a.G), rw = pack_row_words(a.Zq, a.G, SPARSE ? 1 : 0), pieces = rw / 4, sh = 32 - a.G;
    // Scalar loads first, then the id loads -- written in assembly and waited for by hand: with LDS-DMA in flight hipcc
    // (ROCm 7.2) drains vmcnt to 0 at the first use of any ordinary vector load result, which would put Philox behind the whole
    // pack.  The wave issues CPT + 1 id loads, then exactly NQ LDS-DMA instructions; vmcnt retires in order, so vmcnt <= NQ
    // means the ids are in their registers.
    CPM_SSTAMP_DECL;
    CPM_SSTAMP(0);
    uint32_t ns_raw = a.cnt_s[z];
    if constexpr (WAIT) ns_raw &= kCntMask;  // (the hour behind a day launch: flags beside the stayer counts; grouped_run strips them from what it keeps)
    const double last = a.last_t[z];
    const long long thr = a.thr_t[z];
    uint32_t na_raw;
    uint32_t id[CPT + 1];
    if constexpr (WAIT) {
        // ONE wave learns whether the group's placing blocks are done and tells the others through LDS (the verdict must be the
        // workgroup's: waves that disagreed would part ways at the barriers below); the others have their pieces of the pack in
        // flight meanwhile.  A workgroup that gives up READS NOTHING the placing blocks may not have written: it takes its bucket
        // for empty, so no stale or uninitialised word travels on as a car id or a destination (the step is invalid either way).
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        u32x2 seen = {wait_need, 0u};  // {placing blocks of the group that have counted themselves in, the XCDs they ran on}
        const bool poller = wait_need && tid < 64;
        if (poller) asm volatile("global_load_dwordx2 %0, %1, %2 sc1" : "=v"(seen) : "v"(0), "s"(wait_on) : "memory");
        pack_dma<BLOCK, NQ>(pack, a.rp_t + static_cast<size_t>(z) * rw, pieces, tid);
        if (poller) {
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(seen) : "n"(NQ) : "memory");
            uint32_t got = from_lane0(seen.x), where = from_lane0(seen.y);
            for (uint32_t spins = 0; (got & kDoneCount) < wait_need; ++spins) {  // (rare: the placing blocks come first in the launch)
                if (spins >= a.spin_limit) break;  // bounded: the step is then invalid, the context repeats it with two launches per hour
                __builtin_amdgcn_s_sleep(32);
                got = from_lane0(lane == 0 ? __hip_atomic_load(wait_on, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u);
                where = from_lane0(lane == 0 ? __hip_atomic_load(wait_on + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u);
            }
            // (the mask is complete once the count is: a block ORs its XCD in before it counts itself in)
            const bool bad = (got & kDoneCount) < wait_need || (got >> 16) != 0u || where != (1u << xcc_id());
            if (bad && lane == 0) atomicOr(a.rare->status, 4ull);
            if (tid == 0) sl.pad_ = bad ? 1u : 0u;
        }
        bool dead = false;
        if (wait_need) {
            lds_barrier();
            dead = sl.pad_ != 0u;
        }
        const uint32_t nsc = dead ? 0u : min(ns_raw, cap);
        uint32_t nav = 0;
        if (!dead) {
            asm volatile("global_load_dword %0, %1, %2 sc1" : "=v"(nav) : "v"(0), "s"(a.cnt_a + z) : "memory");
#pragma unroll
            for (int c = 0; c <= CPT; ++c) {
                const uint32_t s1 = static_cast<uint32_t>(tid + c * BLOCK);
                const uint32_t pos = s1 < nsc ? s1 : (cap - 1u + nsc - min(s1, cap - 1u + nsc));
                asm volatile("global_load_dword %0, %1, %2 sc1" : "=v"(id[c]) : "v"(pos << 2), "s"(a.ids + b) : "memory");
            }
            if constexpr (CPT == 4)
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(nav), "+v"(id[0]), "+v"(id[1]), "+v"(id[2]), "+v"(id[3]), "+v"(id[4])::"memory");
            else if constexpr (CPT == 2)
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(nav), "+v"(id[0]), "+v"(id[1]), "+v"(id[2])::"memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(nav), "+v"(id[0]), "+v"(id[1])::"memory");
        } else {
#pragma unroll
            for (int c = 0; c <= CPT; ++c) id[c] = 0u;
            ns_raw = 0u;
        }
        na_raw = from_lane0(nav);
    } else {
        na_raw = a.cnt_a[z];
    }
    // the bucket: stayers in slots [0, ns) of the region, arrivals in its last na slots; slot s of the workgroup reads position
    // s + (s >= ns ? gap : 0), gap = cap - ns - na (clamped to the region: slots beyond ns + na hold no car)
    const uint32_t ns = min(ns_raw, cap), na = min(na_raw, cap - ns);
    const uint32_t n_all = ns + na;
    if constexpr (!WAIT) {
        const uint32_t ns4 = ns << 2, gap4 = (cap - n_all) << 2, top4 = (cap - 1) << 2;
#pragma unroll
        for (int c = 0; c <= CPT; ++c) {
            // (the bucket's base in scalar registers + a 32-bit byte offset: cap x 4 B < 2^32)
            const uint32_t s4 = static_cast<uint32_t>(tid + c * BLOCK) << 2;
            const uint32_t off = min(s4 + (s4 >= ns4 ? gap4 : 0u), top4);
            asm volatile("global_load_dword %0, %1, %2" : "=v"(id[c]) : "v"(off), "s"(a.ids + b) : "memory");
        }
        pack_dma<BLOCK, NQ>(pack, a.rp_t + static_cast<size_t>(z) * rw, pieces, tid);
        wait_ids<CPT + 1, NQ>(id);
    }
    CPM_SSTAMP(1);
    // A bucket beyond CPT * BLOCK cars is walked here BLOCK cars at a time -- unless it is heavy, the heavy kernel follows and has
    // room for it: then this workgroup takes the first CPT * BLOCK cars (all its slots are full either way) and lists the zone.
    const bool heavy = n_all > a.heavy_x * (CPT > 4 ? 4 : CPT) * BLOCK;  // (the wide forms run only where no heavy launch follows: same absolute threshold as CPT = 4)
    if (tid == 0) {
        a.parking_t[z] = n_all;  // every car present at hour t, drivers included (src/saveresults.jl:12)
        if (static_cast<unsigned long long>(ns_raw) + na_raw > cap) atomicOr(a.rare->status, 2ull);  // the two ends of the bucket met: step invalid
        s_ndrive = 0;
        s_nstay = 0;
        uint32_t split = 0;
        if (heavy) {
            // the bucket's chunks of CPT * BLOCK cars behind the first one become work items of the heavy launch: (zone, chunk)
            const GroupedRare *r = a.rare;
            const uint32_t items = (n_all + CPT * BLOCK - 1) / (CPT * BLOCK) - 1u;
            atomicMax(&r->maxn[0], n_all);
            const uint32_t idx = atomicAdd(r->nheavy + a.hour, items);
            atomicMax(&r->maxn[1], idx + items);
            if (GROUPED && r->parts > 1) {
                if (idx + items <= r->hgrid) {
                    for (uint32_t q = 0; q < items; ++q) r->heavy_list[idx + q] = static_cast<uint32_t>(z) | (q << 16);
                    split = 1;
                } else {  // no room for all of them: the bucket stays with this workgroup, and the slots it drew are void (not last hour's items)
                    for (uint32_t q = 0; q < items && idx + q < r->hgrid; ++q) r->heavy_list[idx + q] = 0xFFFFFFFFu;
                }
            }
        }
        s_split = split;
    }
    if (tid < kGroups) gb[tid] = 0;
    if (n_all == 0) {  // driving_t[z] stays 0 (zeroed by the caller)
        if (GROUPED) {
            if (tid == 0) a.cnt_next[z] = 0;
            if (tid < kGroups) hand_store<FUSED>(&a.cntg[static_cast<size_t>(z) * kGroups + tid], 0u);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the pack must not land in the LDS of the next workgroup
        hand_off_done<FUSED>(done_chunk, tid);
        return;
    }
    // The register-resident cars: a wave's slots are tid + c * BLOCK, so of its CPT cars per lane the first k hold cars and the
    // rest are empty for the WHOLE wave (k = the number of c with c * BLOCK + 64 * wave < n_all: wave-uniform).  Every empty slot ran
    // Philox and the search like a full one -- at S4k 10.5 % of a launch's (car, wave) groups are empty (the buckets spread from half
    // to two and a half times the mean) -- so the work below is written once for K live cars per lane and a wave runs the K it needs
    // (the K cars of a lane still in lockstep).  Every wave meets the others at the one barrier inside, whatever its K.
    uint32_t nd = 0;
    uint32_t n = 0;
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t *stay_out = a.ids_next + static_cast<size_t>(z) * cap;
    uint32_t *runs = GROUPED ? a.D + static_cast<size_t>(z) * kGroups * a.scap : nullptr;
    const uint16_t *guide = reinterpret_cast<const uint16_t *>(pack);
    const uint32_t *hi = pack + gw;
    const uint16_t *smap = SPARSE ? reinterpret_cast<const uint16_t *>(hi + a.Zq) : nullptr;
    uint32_t hi_last = 0;
    auto first_pass = [&](auto kc) {
        constexpr int K = decltype(kc)::value;
        // Philox: needs the ids only
        bool valid[K ? K : 1], drive[K ? K : 1], want[K ? K : 1], ok[K ? K : 1];
        uint32_t dest[K ? K : 1], clo[K ? K : 1], khi[K ? K : 1];
#pragma unroll
        for (int c = 0; c < K; ++c) {
            valid[c] = static_cast<uint32_t>(tid + c * BLOCK) < n_all;
            long long kb;
            car_draw_words(a.seed, a.cars.global(id[c]), a.step, kb, clo[c], khi[c]);
            drive[c] = valid[c] & (kb <= thr);       // u <= p_drive[origin,t] (src/resampling.jl:15) in integers
            want[c] = drive[c] & (last != 0.0);      // stays, or zero row: destination = origin (:35-36)
        }
        // This wave's pieces of the pack have landed (LDS-DMA counts in vmcnt; s_barrier itself waits for no counter), then the
        // barrier makes every wave's pieces visible to every wave.
        CPM_SSTAMP(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        CPM_SSTAMP(3);
        n = s_split ? static_cast<uint32_t>(CPT * BLOCK) : n_all;  // the cars this workgroup samples
        hi_last = hi[(SPARSE ? a.Zc : a.Z) - 1];
        if constexpr (K > 0) {
            pack_search<K>(guide, hi, khi, want, sh, hi_last, a.Zq, dest, ok, smap);
            {
                bool anyx = false;
#pragma unroll
                for (int c = 0; c < K; ++c) {
                    dest[c] = want[c] ? dest[c] : static_cast<uint32_t>(z);
                    anyx |= want[c] & !ok[c];
                }
                if (__builtin_expect(any64(anyx), 0)) {  // ties and draws above the row total: the f64 row in HBM (wave-uniform, rare)
#pragma unroll
                    for (int c = 0; c < K; ++c)
                        if (want[c] & !ok[c]) dest[c] = search_exact_any<SPARSE>(a.rare, a.hour, z, u53(clo[c], khi[c]), last, khi[c] <= hi_last ? static_cast<int>(dest[c]) : -1);
                }
            }
            CPM_SSTAMP(4);
            if (GROUPED) {
                // stayers: one ticket per wave for all its K slots
                unsigned long long mS[K];
                uint32_t total = 0;
#pragma unroll
                for (int c = 0; c < K; ++c) {
                    mS[c] = ballot64(valid[c] & !drive[c]);
                    total += static_cast<uint32_t>(__popcll(mS[c]));
                }
                uint32_t bS = 0;
                if (lane == 0 && total) bS = atomicAdd(&s_nstay, total);
                bS = from_lane0(bS);
                // drivers: K rank atomics in flight together
                uint32_t rank[K];
#pragma unroll
                for (int c = 0; c < K; ++c) rank[c] = drive[c] ? atomicAdd(&gb[gdiv_group_t<SPARSE>(a.gdiv, dest[c])], 1u) : 0u;
#pragma unroll
                for (int c = 0; c < K; ++c) {
                    if (valid[c] & !drive[c]) put32(stay_out, bS + static_cast<uint32_t>(__popcll(mS[c] & below)), id[c]);
                    bS += static_cast<uint32_t>(__popcll(mS[c]));
                }
#pragma unroll
                for (int c = 0; c < K; ++c) {
                    if (drive[c]) {
                        const uint32_t g = gdiv_group_t<SPARSE>(a.gdiv, dest[c]);
                        const uint32_t packed = id[c] | (gdiv_local_t<SPARSE>(a.gdiv, dest[c], g) << a.idbits);
                        if (rank[c] < static_cast<uint32_t>(kStage)) stage[g * kStage + rank[c]] = packed;
                        else if (rank[c] < a.scap) hand_store<FUSED>(&runs[g * a.scap + rank[c]], packed);
                    }
                }
            } else {
#pragma unroll
                for (int c = 0; c < K; ++c) {
                    if (valid[c]) stay_out[tid + c * BLOCK] = dest[c] | (drive[c] ? kDriveBit : 0u);
                    nd += drive[c] ? 1u : 0u;
                }
            }
        }
    };
    {
        const uint32_t w64 = from_lane0(static_cast<uint32_t>(tid) & ~63u);  // (this wave's first slot of car 0, in a scalar register)
        const uint32_t k = n_all > w64 ? min(static_cast<uint32_t>(CPT), (n_all - w64 + BLOCK - 1) / BLOCK) : 0u;
        if constexpr (CPT == 8) {
            switch (k) {
            case 0: first_pass(std::integral_constant<int, 0>{}); break;
            case 1: first_pass(std::integral_constant<int, 1>{}); break;
            case 2: first_pass(std::integral_constant<int, 2>{}); break;
            case 3: first_pass(std::integral_constant<int, 3>{}); break;
            case 4: first_pass(std::integral_constant<int, 4>{}); break;
            case 5: first_pass(std::integral_constant<int, 5>{}); break;
            case 6: first_pass(std::integral_constant<int, 6>{}); break;
            case 7: first_pass(std::integral_constant<int, 7>{}); break;
            default: first_pass(std::integral_constant<int, 8>{}); break;
            }
        } else if constexpr (CPT == 7) {
            switch (k) {
            case 0: first_pass(std::integral_constant<int, 0>{}); break;
            case 1: first_pass(std::integral_constant<int, 1>{}); break;
            case 2: first_pass(std::integral_constant<int, 2>{}); break;
            case 3: first_pass(std::integral_constant<int, 3>{}); break;
            case 4: first_pass(std::integral_constant<int, 4>{}); break;
            case 5: first_pass(std::integral_constant<int, 5>{}); break;
            case 6: first_pass(std::integral_constant<int, 6>{}); break;
            default: first_pass(std::integral_constant<int, 7>{}); break;
            }
        } else if constexpr (CPT == 6) {
            switch (k) {
            case 0: first_pass(std::integral_constant<int, 0>{}); break;
            case 1: first_pass(std::integral_constant<int, 1>{}); break;
            case 2: first_pass(std::integral_constant<int, 2>{}); break;
            case 3: first_pass(std::integral_constant<int, 3>{}); break;
            case 4: first_pass(std::integral_constant<int, 4>{}); break;
            case 5: first_pass(std::integral_constant<int, 5>{}); break;
            default: first_pass(std::integral_constant<int, 6>{}); break;
            }
        } else if constexpr (CPT == 5) {
            switch (k) {
            case 0: first_pass(std::integral_constant<int, 0>{}); break;
            case 1: first_pass(std::integral_constant<int, 1>{}); break;
            case 2: first_pass(std::integral_constant<int, 2>{}); break;
            case 3: first_pass(std::integral_constant<int, 3>{}); break;
            case 4: first_pass(std::integral_constant<int, 4>{}); break;
            default: first_pass(std::integral_constant<int, 5>{}); break;
            }
        } else if constexpr (CPT == 4) {
            switch (k) {
            case 0: first_pass(std::integral_constant<int, 0>{}); break;
            case 1: first_pass(std::integral_constant<int, 1>{}); break;
            case 2: first_pass(std::integral_constant<int, 2>{}); break;
            case 3: first_pass(std::integral_constant<int, 3>{}); break;
            default: first_pass(std::integral_constant<int, 4>{}); break;
            }
        } else if constexpr (CPT == 2) {
            switch (k) {
            case 0: first_pass(std::integral_constant<int, 0>{}); break;
            case 1: first_pass(std::integral_constant<int, 1>{}); break;
            default: first_pass(std::integral_constant<int, 2>{}); break;
            }
        } else {
            if (k == 0) first_pass(std::integral_constant<int, 0>{});
            else first_pass(std::integral_constant<int, 1>{});
        }
    }
    for (uint32_t q0 = CPT * BLOCK; q0 < n; q0 += BLOCK) {  // buckets larger than CPT*BLOCK cars (wave-uniform trips)
        if (q0 + static_cast<uint32_t>(tid & ~63) >= n) continue;  // none of this wave's 64 slots holds a car (no barrier inside the loop)
        const uint32_t q = q0 + tid;
        const bool valid1 = q < n;
        uint32_t idx;
        if constexpr (WAIT) idx = (q0 == CPT * BLOCK) ? id[CPT] : (valid1 ? hand_load<true>(&a.ids[b + (q < ns ? q : cap - 1u + ns - q)]) : 0u);
        else idx = (q0 == CPT * BLOCK) ? id[CPT] : (valid1 ? a.ids[b + q + (q >= ns ? cap - n_all : 0u)] : 0u);
        const uint64_t car = a.cars.global(idx);
        long long kb;
        uint32_t clo1[1], khi1[1], dest1[1];
        bool ok1[1], want1[1];
        car_draw_words(a.seed, car, a.step, kb, clo1[0], khi1[0]);
        const bool drive1 = valid1 & (kb <= thr);
        want1[0] = drive1 & (last != 0.0);
        pack_search<1>(guide, hi, khi1, want1, sh, hi_last, a.Zq, dest1, ok1, smap);
        if (!want1[0]) dest1[0] = z;
        else if (!ok1[0]) dest1[0] = search_exact_any<SPARSE>(a.rare, a.hour, z, u53(clo1[0], khi1[0]), last, khi1[0] <= hi_last ? static_cast<int>(dest1[0]) : -1);
        if (GROUPED) {
            const unsigned long long m1 = ballot64(valid1 & !drive1);
            uint32_t b1 = 0;
            if (lane == 0 && m1) b1 = atomicAdd(&s_nstay, static_cast<uint32_t>(__popcll(m1)));
            b1 = from_lane0(b1);
            if (valid1 & !drive1) stay_out[b1 + static_cast<uint32_t>(__popcll(m1 & below))] = idx;
            if (drive1) {
                const uint32_t g = gdiv_group_t<SPARSE>(a.gdiv, dest1[0]);
                const uint32_t rank = atomicAdd(&gb[g], 1u);
                const uint32_t packed = idx | (gdiv_local_t<SPARSE>(a.gdiv, dest1[0], g) << a.idbits);
                if (rank < static_cast<uint32_t>(kStage)) stage[g * kStage + rank] = packed;
                else if (rank < a.scap) hand_store<FUSED>(&runs[g * a.scap + rank], packed);
            }
        } else {
            if (valid1) stay_out[q] = dest1[0] | (drive1 ? kDriveBit : 0u);
            nd += drive1 ? 1u : 0u;
        }
    }
    if (!GROUPED) {
        for (int o = 32; o > 0; o >>= 1) nd += __shfl_down(nd, o, 64);
        if (lane == 0 && nd) atomicAdd(&s_ndrive, nd);
    }
    CPM_SSTAMP(5);
    lds_barrier();  // ranks, staged drivers and counters (all in LDS) are final; the stayers' stores need not have landed
    CPM_SSTAMP(6);
    if (GROUPED) {
        // staged drivers -> their runs: 8 lanes per group, 16 bytes per lane (whole pieces: what lies beyond a run's length is never
        // read).  One store instruction per thread; in the fused hour, where the runs are written through to memory (sc1), one fabric
        // write per 16 bytes instead of one per dword.
        static_assert(kStage % 4 == 0, "16-byte pieces");
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(runs, 0, static_cast<int>(kGroups * a.scap * 4u), 0x00020000);
        for (int i = tid; i < kGroups * (kStage / 4); i += BLOCK) {
            const int g = i / (kStage / 4), ch = i % (kStage / 4);
            const uint32_t lim = min(gb[g], static_cast<uint32_t>(kStage));
            if (static_cast<uint32_t>(4 * ch) < lim) {
                const uint4 q = *reinterpret_cast<const uint4 *>(&stage[g * kStage + 4 * ch]);
                cpm_u32x4 qv;
                qv.x = q.x;
                qv.y = q.y;
                qv.z = q.z;
                qv.w = q.w;
                __builtin_amdgcn_raw_buffer_store_b128(qv, rs, (static_cast<uint32_t>(g) * a.scap + 4u * ch) << 2, 0, FUSED ? 16 : 0);  // (aux 16: sc1)
            }
        }
        if (tid < kGroups) {
            const uint32_t c = gb[tid];
            hand_store<FUSED>(&a.cntg[static_cast<size_t>(z) * kGroups + tid], min(c, a.scap));
            if (c > a.scap) atomicOr(a.rare->status, 2ull);  // a run outgrew its slot: the caller grows the regions and repeats
        }
    }
    if (tid == 0) {
        a.driving_t[z] = GROUPED ? n - s_nstay : s_ndrive;  // every car of the bucket either stays or drives
        if (GROUPED) {
            a.cnt_next[z] = s_nstay;
            if (s_nstay > cap) atomicOr(a.rare->status, 2ull);  // (cannot happen: stayers <= the bucket <= cap)
        }
    }
    CPM_SSTAMP(7);
#ifdef CPM_STAMP_BOTH
    if constexpr (FUSED || WAIT)  // (only the fused hour's blocks: the buffer then holds the last fused launch of the resample)
#endif
    CPM_SSTAMP_FLUSH;
    hand_off_done<FUSED>(done_chunk, tid);
}

template <int BLOCK, int CPT, int NQ, bool GROUPED, bool SPARSE = false>
__global__ __launch_bounds__(BLOCK, CPM_WPS) CPM_SGPR_ATTR void k_grouped_sample(GroupedArgs a)
{
    extern __shared__ uint32_t pack[];  // the zone's row pack: guide (u16), then Zq high words
    __shared__ SampleLds sl;
    grouped_sample_body<BLOCK, CPT, NQ, GROUPED, false, false, SPARSE>(a, blockIdx.x, pack, sl, nullptr);
}

// The rest of the HEAVY buckets (real Uber Movement tables are peaky: a central zone can hold tens of times the mean, and one
// workgroup walking it BLOCK cars at a time is the tail of the whole launch).  Launched behind
// k_grouped_sample when the context has seen such buckets (GroupedArgs::parts > 1), with parts - 1 blocks for each of the hgrid
// zones the sampler may list in heavy_list: block (i, q) takes the chunks q, q + (parts - 1), ... of CPT * BLOCK cars behind the first
// one of the i-th listed zone.  It shares the zone's outputs with the first workgroup and its sibling blocks, which have already written theirs
// or are writing them now: stayers take their slots in next hour's bucket with one global ticket per wave (cnt_next[z]), drivers are ranked per chunk in LDS and reserve their range of the zone's runs with one
// global atomic per (chunk, destination group) on the run length, driving counts are added.  Order inside buckets and runs is
// arbitrary anyway (a car's draws depend on its id only, counts are order-free).
template <int BLOCK, int CPT, int NQ, bool SPARSE = false>
__global__ __launch_bounds__(BLOCK, 4) void k_grouped_sample_heavy(GroupedArgs a)
{
    extern __shared__ uint32_t pack[];
    __shared__ uint32_t gb[kGroups], gbase[kGroups];
    __shared__ uint32_t s_stay, s_sbase, s_nd;  // the block's stayers (sum of its waves'), their first slot in the bucket, its drivers
    const GroupedRare *rare = a.rare;
    if (blockIdx.x >= min(rare->nheavy[a.hour], rare->hgrid)) return;  // (the list of this hour is shorter than the grid)
    const uint32_t item = rare->heavy_list[blockIdx.x];                 // one work item = one chunk of one listed zone
    if (item == 0xFFFFFFFFu) return;                                    // (a slot drawn by a zone whose chunks did not all fit the list)
    const int z = static_cast<int>(item & 0xFFFFu);
    const uint32_t q = item >> 16;
    unsigned long long *status = rare->status;
    const int tid = threadIdx.x, lane = tid & 63;
    constexpr uint32_t L = CPT * BLOCK;
    const uint32_t cap = a.cap;
    const uint32_t ns = min(a.cnt_s[z] & kCntMask, cap), na = min(a.cnt_a[z], cap - ns);
    const uint32_t n = ns + na, gap = cap - n;
    const uint32_t start0 = L * (1u + q);
    if (start0 >= n) return;
    const uint32_t b = static_cast<uint32_t>(z) * cap;
    const int gw = pack_guide_words(a.G), rw = pack_row_words(a.Zq, a.G, SPARSE ? 1 : 0), pieces = rw / 4, sh = 32 - a.G;
    const double last = a.last_t[z];
    const long long thr = a.thr_t[z];
    pack_dma<BLOCK, NQ>(pack, a.rp_t + static_cast<size_t>(z) * rw, pieces, tid);
    if (tid < kGroups) gb[tid] = 0;
    if (tid == 0) {
        s_stay = 0;
        s_nd = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const uint16_t *guide = reinterpret_cast<const uint16_t *>(pack);
    const uint32_t *hi = pack + gw;
    const uint16_t *smap = SPARSE ? reinterpret_cast<const uint16_t *>(hi + a.Zq) : nullptr;
    const uint32_t hi_last = hi[(SPARSE ? a.Zc : a.Z) - 1];
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t *stay_out = a.ids_next + static_cast<size_t>(z) * cap;
    uint32_t *runs = a.D + static_cast<size_t>(z) * kGroups * a.scap;
    uint32_t nd = 0;
    for (uint32_t start = start0, once = 0; once < 1u; ++once) {  // (one chunk per block)
        uint32_t id[CPT], dest[CPT], clo[CPT], khi[CPT];
        bool valid[CPT], drive[CPT], want[CPT], ok[CPT];
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            const uint32_t s = start + static_cast<uint32_t>(tid + c * BLOCK);
            valid[c] = s < n;
            const uint32_t sc = min(s, n - 1u);
            id[c] = a.ids[b + sc + (sc >= ns ? gap : 0u)];
        }
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            long long kb;
            car_draw_words(a.seed, a.cars.global(id[c]), a.step, kb, clo[c], khi[c]);
            drive[c] = valid[c] & (kb <= thr);
            want[c] = drive[c] & (last != 0.0);
        }
        pack_search<CPT>(guide, hi, khi, want, sh, hi_last, a.Zq, dest, ok, smap);
        bool anyx = false;
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            dest[c] = want[c] ? dest[c] : static_cast<uint32_t>(z);
            anyx |= want[c] & !ok[c];
        }
        if (__builtin_expect(any64(anyx), 0)) {
#pragma unroll
            for (int c = 0; c < CPT; ++c)
                if (want[c] & !ok[c]) dest[c] = search_exact_any<SPARSE>(a.rare, a.hour, z, u53(clo[c], khi[c]), last, khi[c] <= hi_last ? static_cast<int>(dest[c]) : -1);
        }
        // stayers: ONE global ticket per block (the waves' counts are added up in LDS first).  All the blocks of a heavy zone -- 37
        // of them for the largest bucket of `--skew 32` -- draw on the zone's one stayer counter, its 32 run lengths and its driving
        // count, and atomics on one address are served one after the other at the memory side: with a ticket per WAVE (and a driving
        // count per wave) the heavy launch was 16 us per hour where its blocks' own work is 6.
        unsigned long long mS[CPT];
        uint32_t total = 0;
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            mS[c] = ballot64(valid[c] & !drive[c]);
            total += static_cast<uint32_t>(__popcll(mS[c]));
        }
        uint32_t wS = 0;
        if (lane == 0 && total) wS = atomicAdd(&s_stay, total);
        wS = from_lane0(wS);
        uint32_t rank[CPT];
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            rank[c] = drive[c] ? atomicAdd(&gb[gdiv_group_t<SPARSE>(a.gdiv, dest[c])], 1u) : 0u;
            nd += drive[c] ? 1u : 0u;
        }
        for (int o = 32; o > 0; o >>= 1) nd += __shfl_down(nd, o, 64);
        if (lane == 0 && nd) atomicAdd(&s_nd, nd);
        __syncthreads();  // the chunk's ranks and sums are final
        if (tid == kGroups) {  // (a lane of the same wave as the 32 below: its ticket leaves with theirs)
            const uint32_t all = s_stay;
            uint32_t base = 0;
            if (all) {
                base = atomicAdd(&a.cnt_next[z], all);
                if (base + all > cap) atomicOr(status, 2ull);
            }
            s_sbase = base;
            if (s_nd) atomicAdd(&a.driving_t[z], static_cast<unsigned long long>(s_nd));
        }
        if (tid < kGroups) {
            const uint32_t c = gb[tid];
            uint32_t base = 0;
            if (c) {
                base = atomicAdd(&a.cntg[static_cast<size_t>(z) * kGroups + tid], c);
                if (base + c > a.scap) atomicOr(status, 2ull);
            }
            gbase[tid] = base;
            gb[tid] = 0;
        }
        __syncthreads();
        {
            uint32_t bS = s_sbase + wS;
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                const uint32_t p = bS + static_cast<uint32_t>(__popcll(mS[c] & below));
                if ((valid[c] & !drive[c]) && p < cap) stay_out[p] = id[c];
                bS += static_cast<uint32_t>(__popcll(mS[c]));
            }
        }
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            if (drive[c]) {
                const uint32_t g = gdiv_group_t<SPARSE>(a.gdiv, dest[c]);
                const uint32_t p = gbase[g] + rank[c];
                if (p < a.scap) runs[g * a.scap + p] = id[c] | (gdiv_local_t<SPARSE>(a.gdiv, dest[c], g) << a.idbits);
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------ placing the drivers
// Drivers of destination group g -> their buckets.  blockIdx = j * kGroups + g: the blocks of a group share blockIdx % 8 (one XCD,
// one L2: all writes to a bucket merge there; speed only, never correctness).  Block (g, j) takes the group-g runs of the origin
// zones [j*zps, (j+1)*zps): 8 lanes per run, 2 x KDEEP entries per lane (16 x KDEEP of a run in registers).  Run lengths and run contents sit at addresses known up
// front, so they are requested together.
// Threads per block: 512 (128 origin zones per block, four blocks per CU; measured at S4k against 1024 x two per CU: 11.4 us against
// 12.4 -- the kernel is a chain of round trips and barriers, and four shorter chains per CU interleave better than two; 256: 13.2),
// 1024 when a destination group holds more than 512 zones (one thread per zone takes the ticket and the scan).

// What bounds this kernel is the number of cache lines a store instruction touches (in-kernel stamps, profiles/round2_notes.md:
// with every lane storing its own entry -- 64 buckets per wave-instruction -- issuing the stores took a third of a block's life).
// So the block's entries are first sorted by destination zone in LDS (their rank inside the block is known from pass A, the
// zones' offsets from a scan of the block's histogram) and then written out in that order: consecutive lanes write consecutive
// slots of one bucket.
// LDS of a placing block beside its sorted list (one object: the fused kernel overlays it with the sampler workgroups')
// bins : per destination zone: entries held in registers (pass A), then their first index in the sorted list
// tbins: entries beyond 16 * KDEEP of their run (re-read in pass B), then the running position of those inside the bucket
// delta: (this block's first position inside the zone's bucket) - (the zone's first index in the sorted list)
// lstart: runs longer than the 16 * KDEEP entries their lanes hold (skewed tables: a heavy origin zone, a popular destination group):
//         the entries beyond are dealt over ALL threads of the block -- lstart[r] = first index of run r's surplus in that flat list
template <int PB, int KRUNS, int ZPG>
struct PlaceLds {
    uint32_t bins[ZPG], tbins[ZPG], delta[ZPG];
    uint32_t wsum[PB / 64], total;
    uint32_t lstart[KRUNS * (PB / 16) + 1], any_long, go;
    uint32_t lzone[KRUNS * (PB / 16)];  // PERM (zones dealt largest-first, k_grouped_hour): the origin zone of every run of the block
    using zone_t = typename std::conditional<(ZPG <= 256), uint8_t, uint16_t>::type;  // a zone inside its group, in the sorted list
};

// Runs longer than the 32 entries their lanes hold (rare on flat tables: a few blocks in a thousand) are handled OUT OF LINE, in
// functions the common path calls behind a short conditional jump.  Inlined, the two pieces sat in the middle of every block's
// dependency chain as branches over a few hundred instructions, and a placing block lost ~1.5 k cycles at each of the two (3.2 k +
// 1.5 k of its 11.8 k-cycle life, tools/hour_stamps.py with finer stamps; with the pieces compiled out both vanished) -- whether the
// jump was taken over the block or fell through in front of it: the far end of a branch costs an instruction fetch from memory.
// Entries of a run its own 8 lanes take: the first 16 * KDEEP in registers at once, and in the one-launch hour (KDEEP = 2) the next 32 by a second 16-byte
// load per lane that only the lanes of a run LONGER than 32 issue, once the run lengths have arrived (grouped_place_body: the MEDIUM
// runs).  What lies beyond is the surplus the whole block deals out below.
// Only in the one-launch hour (FUSED), whose blocks have the sampler's 80 registers anyway: the standalone placing kernel went from 64 to
// 81 registers with it -- two 512-thread blocks per CU instead of four -- and lost 3 % at Z = 8,192 and 14 % on skewed tables.
__host__ __device__ constexpr bool place_medium(int kdeep, bool fused) { return kdeep == 2 && fused; }
__host__ __device__ constexpr uint32_t place_held(int kdeep, bool fused) { return place_medium(kdeep, fused) ? 64u : 16u * static_cast<uint32_t>(kdeep); }
// surplus entry e of the block -> (run r, index inside the run): r = the last run with lstart[r] <= e
template <int PB, int KRUNS, int KDEEP, int ZPG, bool FUSED, bool PERM = false>
__device__ __forceinline__ uint32_t place_surplus_entry(const PlaceLds<PB, KRUNS, ZPG> &pl, const uint32_t *D, int g, int zs0, int zs1, uint32_t scap, uint32_t e)
{
    constexpr int kRuns = (KRUNS / 2) * (PB / 8);
    int r = 0;
    constexpr int kTop = kRuns <= 64 ? 32 : kRuns <= 128 ? 64 : kRuns <= 256 ? 128 : 256;  // (kRuns need not be a power of two)
#pragma unroll
    for (int step = kTop; step > 0; step >>= 1)
        if (r + step < kRuns && pl.lstart[r + step] <= e) r += step;
    const int zc = PERM ? static_cast<int>(pl.lzone[min(r, zs1 - zs0 - 1)]) : min(zs0 + r, zs1 - 1);  // (run r of the block: origin zone zs0 + r, or the zone dealt to that position)
    return hand_load<FUSED>(&D[(static_cast<size_t>(zc) * kGroups + g) * scap + place_held(KDEEP, FUSED) + (e - pl.lstart[r])]);
}
// exclusive scan of the surplus lengths (one per thread), then the histogram of the surplus entries (tbins); returns their number
template <int PB, int KRUNS, int KDEEP, int ZPG, bool FUSED, bool PERM = false>
__device__ __noinline__ uint32_t place_surplus_count(PlaceLds<PB, KRUNS, ZPG> &pl, const uint32_t *D, int g, int zs0, int zs1, uint32_t scap, uint32_t idbits)
{
    constexpr int kRuns = (KRUNS / 2) * (PB / 8), kSurplusBatch = 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t len = tid < kRuns ? pl.lstart[tid] : 0u;
    const uint32_t incl = wave_incl_scan(len);
    if (lane == 63) pl.wsum[wave] = incl;
    lds_barrier();
    uint32_t before = 0;
    for (int w = 0; w < wave; ++w) before += pl.wsum[w];
    if (tid < kRuns) pl.lstart[tid] = before + incl - len;
    if (tid == kRuns - 1) pl.lstart[kRuns] = before + incl;
    lds_barrier();
    const uint32_t ltotal = pl.lstart[kRuns];
    for (uint32_t e0 = tid; e0 < ltotal; e0 += kSurplusBatch * PB) {  // (a thread's kSurplusBatch loads are in flight together)
        uint32_t w[kSurplusBatch];
#pragma unroll
        for (int u = 0; u < kSurplusBatch; ++u)
            w[u] = (e0 + u * PB < ltotal) ? place_surplus_entry<PB, KRUNS, KDEEP, ZPG, FUSED, PERM>(pl, D, g, zs0, zs1, scap, e0 + u * PB) : 0u;
#pragma unroll
        for (int u = 0; u < kSurplusBatch; ++u)
            if (e0 + u * PB < ltotal) atomicAdd(&pl.tbins[w[u] >> idbits], 1u);
    }
    lds_barrier();
    return ltotal;
}
// the surplus entries straight to their buckets (tbins: the running position inside each bucket)
template <int PB, int KRUNS, int KDEEP, int ZPG, bool FUSED, bool SIGNAL, bool PERM = false>
__device__ __noinline__ void place_surplus_out(PlaceLds<PB, KRUNS, ZPG> &pl, const uint32_t *D, int g, int zs0, int zs1, uint32_t scap, uint32_t idbits,
                                               int zg0, int nzl, uint32_t cap, uint32_t *__restrict__ ids_next, uint32_t ltotal)
{
    constexpr int kSurplusBatch = 4;
    const int tid = threadIdx.x;
    const uint32_t idmask = (idbits >= 32) ? 0xFFFFFFFFu : ((1u << idbits) - 1u);
    for (uint32_t e0 = tid; e0 < ltotal; e0 += kSurplusBatch * PB) {
        uint32_t w[kSurplusBatch];
#pragma unroll
        for (int u = 0; u < kSurplusBatch; ++u)
            w[u] = (e0 + u * PB < ltotal) ? place_surplus_entry<PB, KRUNS, KDEEP, ZPG, FUSED, PERM>(pl, D, g, zs0, zs1, scap, e0 + u * PB) : 0u;
#pragma unroll
        for (int u = 0; u < kSurplusBatch; ++u)
            if (e0 + u * PB < ltotal) {
                const uint32_t dl = w[u] >> idbits;
                const uint32_t p = atomicAdd(&pl.tbins[dl], 1u);
                if (p < cap && (!SIGNAL || dl < static_cast<uint32_t>(nzl))) ids_next[static_cast<size_t>(zg0 + dl) * cap + (cap - 1u - p)] = w[u] & idmask;
            }
    }
}

// FUSED: the block's input is written by sampler workgroups of the SAME launch (those of chunk j of origin zones: lower block
// indices, dispatched before it).  One wave polls the chunk's counter (relaxed agent-scope loads, s_sleep between them) until all
// `need` workgroups have counted themselves in; the wait is bounded: when it runs out the block raises bit 2 of the status word and
// leaves -- the step is then invalid and the context repeats it with two launches per hour (a placement or dispatch order this
// protocol did not expect can cost time, never a hang).  The runs are then read with sc1 loads (hand_load).
#ifndef CPM_PLACE_SLEEP
#define CPM_PLACE_SLEEP 32  // x 64 cycles between two polls of a placing block of the one-launch hour (~1 us)
#endif
#ifndef CPM_PLACE_NEAR
#define CPM_PLACE_NEAR 4    // ... and once all but this many sampler workgroups of the chunk have handed over:
#endif
#ifndef CPM_PLACE_SLEEP_NEAR
#define CPM_PLACE_SLEEP_NEAR 4
#endif
constexpr uint32_t kFusedSpinLimit = 1u << 15;  // default number of polls: x (one L2 round trip + s_sleep 32) = tens of milliseconds
// SIGNAL (placing first, k_grouped_hour_pf): the buckets this block fills are read by sampler workgroups of the SAME launch -- the
// ids are stored write-through (sc1), and when the block is done every wave drains its stores, the block meets and one lane counts
// the block in on its destination group's counter (done_out), which the group's sampler workgroups ask for before they read.
// PERM (k_grouped_hour with its zones dealt largest-first): the runs of the block are those of the zones perm[zs0 ...] -- the
// sampler workgroups of chunk j by POSITION in the launch.
template <int PB, int KRUNS, int KDEEP, int ZPG, bool FUSED, bool SIGNAL = false, bool PERM = false>
__device__ __forceinline__ void grouped_place_body(const int g, const int j, PlaceLds<PB, KRUNS, ZPG> &pl, uint32_t *sorted_ids, const uint32_t *__restrict__ D,
                                                   const uint32_t *__restrict__ cntg, int zpg, int zps, int Z, uint32_t cap, uint32_t scap, uint32_t idbits,
                                                   uint32_t *__restrict__ cnt_a_next, uint32_t *__restrict__ ids_next, unsigned long long *status,
                                                   const uint32_t *done_chunk, uint32_t need, uint32_t spin_limit, uint32_t *done_out = nullptr,
                                                   const uint32_t *__restrict__ perm = nullptr, const bool slow_poll = false)
{
    // A run's first 32 entries are held by EIGHT lanes, four consecutive entries each: one 16-byte load per lane and run (two 4-byte
    // loads per lane with sixteen lanes per run before: 12 load instructions per thread instead of 4, and in the fused hour, where
    // the runs are read past this CU's L1 (sc1), a fabric request per dword instead of one per 16 bytes).
    // KDEEP = 4: a SECOND 16-byte load per lane and run (entries 32 .. 63), requested with the first whatever the run's length turns
    // out to be.  A run of more than 32 entries is no exception at S4k (~500 drivers of a zone over 32 groups: 16 on average, 28 where
    // p_drive is 0.9): most blocks take the long-run path (place_surplus_*: a scan over the block's runs between two barriers, then
    // dependent loads, twice -- 5 k cycles of a placing block's 12.5 k in the fused hour, tools/hour_stamps.py).  With 64 entries in
    // registers the block's chain is then bound by the ticket's round trip instead, and the hour does not move (kFusedKdeep).
    static_assert((KDEEP == 2 || KDEEP == 4) && KRUNS % 2 == 0, "written for 32 or 64 register entries per run");
    constexpr int kPlaceBlock = PB, kPlaceSeg = PB / 8;    // 8-lane segments: one run each per pass
    constexpr int KR = KRUNS / 2;                          // runs per lane segment (= passes)
    constexpr int kQ = KDEEP / 2, kE = 4 * kQ;             // 16-byte loads / entries per lane and run
    constexpr int kSlots = KRUNS * KDEEP * kPlaceBlock;    // entries a block can hold in registers
    constexpr int kRuns = KR * kPlaceSeg;
    uint32_t(&bins)[ZPG] = pl.bins, (&tbins)[ZPG] = pl.tbins, (&delta)[ZPG] = pl.delta;
    uint32_t(&wsum)[PB / 64] = pl.wsum, &s_total = pl.total;
    uint32_t(&lstart)[kRuns + 1] = pl.lstart, &s_any_long = pl.any_long;
    using zone_t = typename PlaceLds<PB, KRUNS, ZPG>::zone_t;
    zone_t *sorted_zone = reinterpret_cast<zone_t *>(sorted_ids + kSlots);  // [kSlots] their zone inside the group
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int zg0 = g * zpg;
    const int nzl = max(0, min(zpg, Z - zg0));
    const int zs0 = j * zps, zs1 = min(Z, zs0 + zps);
    const int sub = tid >> 3, l8 = tid & 7;
    const uint32_t idmask = (idbits >= 32) ? 0xFFFFFFFFu : ((1u << idbits) - 1u);
    CPM_PSTAMP_DECL;
    CPM_PSTAMP(0);
    for (int k = tid; k < zpg; k += kPlaceBlock) {
        bins[k] = 0;
        tbins[k] = 0;
    }
    if (tid == 0) s_any_long = 0;
    if (zs0 >= zs1) return;  // (uniform per block)
    // (k_grouped_hour, whose placing blocks signal nobody: the plain wait, the code the hourly launch was tuned with -- the abort-marked
    //  form below, the guards of the write-out and the count mask together cost the headline 2.5 %, measured)
    if constexpr (FUSED && !SIGNAL) {
        if (wave == 0) {
            bool ok = false;
            for (uint32_t spins = 0; spins < spin_limit; ++spins) {
                const uint32_t seen = from_lane0(lane == 0 ? __hip_atomic_load(done_chunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u);
                if (seen >= need) {
                    ok = true;
                    break;
                }
                // (~1 us between polls while the chunk is far from complete: 32 placing blocks poll each counter, and polling every 0.25 us
                //  all along cost the Melbourne-shaped hour, whose placing blocks wait while samplers still run, 3.5 %; close to the end
                //  a short interval takes ~1 us off the hand-over: -0.7 % at S4k)
                //  (same box, interleaved: S4k 0.8126 -> 0.7967 ms per resample with 4 / 4; 16 / 4: 0.7994; 8 / 2: 0.8027; 4 / 1: as 4 / 4;
                //   ~2 us while far: S4k +0.9 %, the Melbourne-shaped hour -2.5 % -- that is what slow_poll, the sparse instantiations, takes)
                if (seen + CPM_PLACE_NEAR >= need) __builtin_amdgcn_s_sleep(CPM_PLACE_SLEEP_NEAR);
                else if (slow_poll) __builtin_amdgcn_s_sleep(2 * CPM_PLACE_SLEEP);
                else __builtin_amdgcn_s_sleep(CPM_PLACE_SLEEP);
            }
            if (tid == 0) pl.go = ok ? 1u : 0u;
        }
        lds_barrier();  // (the polling wave's loads come after its poll matched, the other waves' after this barrier)
        if (pl.go == 0u) {
            if (tid == 0) atomicOr(status, 4ull);
            return;
        }
    }
    if constexpr (FUSED && SIGNAL) {
        if (wave == 0) {
            uint32_t go = 0;  // 0: gave up waiting, 1: the chunk's runs are complete, 2: a sampler workgroup of the chunk gave up (kDoneAbort)
            for (uint32_t spins = 0; spins < spin_limit; ++spins) {
                const uint32_t seen = from_lane0(lane == 0 ? __hip_atomic_load(done_chunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u);
                if ((seen & kDoneCount) >= need) {
                    go = (seen >> 16) ? 2u : 1u;
                    break;
                }
                __builtin_amdgcn_s_sleep(32);  // (~1 us between polls)
            }
            if (tid == 0) pl.go = go;
        }
        lds_barrier();  // (the polling wave's loads come after its poll matched, the other waves' after this barrier)
        if (pl.go != 1u) {  // nothing is read, nothing is stored
            if (tid == 0) {
                if (pl.go == 0u) atomicOr(status, 4ull);
                if constexpr (SIGNAL) {  // whoever waits for this block gives up too, at once
                    __hip_atomic_fetch_or(done_out + 1, 1u << xcc_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add(done_out, 1u + kDoneAbort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            return;
        }
    }
    uint32_t c[KR], v[KR][kE], r[KR][kE];
    // (the block's runs behind one buffer descriptor: zone zs0's group-0 run is byte 0; < 2^32 bytes for every region size)
    // (PERM: the whole run array behind the descriptor -- Z x 32 x scap x 4 B < 2^32: perm_fits)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t *>(D) + (PERM ? 0 : static_cast<size_t>(zs0) * kGroups * scap), 0,
        PERM ? -1 : static_cast<int>(static_cast<uint32_t>(zs1 - zs0) * kGroups * scap * 4u), 0x00020000);
    int zrun[KR];
#pragma unroll
    for (int k = 0; k < KR; ++k) {
        const int pidx = min(zs0 + sub + k * kPlaceSeg, zs1 - 1);
        zrun[k] = PERM ? static_cast<int>(perm[pidx]) : pidx;
        if (PERM && l8 == 0) pl.lzone[sub + k * kPlaceSeg] = static_cast<uint32_t>(zrun[k]);
    }
#pragma unroll
    for (int k = 0; k < KR; ++k) {  // (nothing here depends on a loaded value: every request leaves before the first wait)
        const int zc = zrun[k];
        const size_t run = static_cast<size_t>(zc) * kGroups + g;
        c[k] = hand_load<FUSED>(&cntg[run]);
#pragma unroll
        for (int h = 0; h < kQ; ++h) {  // (entries 32 h + 4 l8 ... of the run)
            const cpm_u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rs, ((static_cast<uint32_t>(PERM ? zc : zc - zs0) * kGroups + g) * scap + 32u * h + 4u * l8) << 2, 0,
                                                                      FUSED ? 16 : 0);  // (aux 16: sc1)  scap >= 64; beyond c[k]: stale, masked
            v[k][4 * h + 0] = q.x;
            v[k][4 * h + 1] = q.y;
            v[k][4 * h + 2] = q.z;
            v[k][4 * h + 3] = q.w;
        }
    }
    lds_barrier();
    CPM_PSTAMP(1);
#pragma unroll
    for (int k = 0; k < KR; ++k) {
        c[k] = min(c[k], scap);  // (beyond scap only when the heavy kernel flagged an overflow: the step is repeated)
        if (zs0 + sub + k * kPlaceSeg >= zs1) c[k] = 0;
    }
    // MEDIUM runs (33 .. 64 entries: at S4k a run holds ~16 drivers on average and 28 where p_drive is 0.9, so most blocks hold a few):
    // entries 32 .. 63 by ONE more 16-byte load of the run's own lanes, requested now -- its round trip runs under pass A -- counted in
    // tbins and stored straight to their buckets behind the sorted list.  They took the surplus path before (a scan of the block's runs
    // between two barriers, a binary search over its prefix and a dependent load per entry, TWICE: ~5 k ticks of a placing block's
    // 12.5 k in the fused hour, most blocks).
    constexpr bool kMedium = place_medium(KDEEP, FUSED);
    cpm_u32x4 xq[KR];
    uint32_t nx[KR];
#pragma unroll
    for (int k = 0; k < KR; ++k) {
        nx[k] = 0;
        xq[k] = cpm_u32x4{0u, 0u, 0u, 0u};
        if constexpr (kMedium) {
            const uint32_t first = 32u + 4u * static_cast<uint32_t>(l8);
            // this lane's entries among 32 .. 63 of the run.  (NOT `c > first ? min(c - first, 4u) : 0u`: hipcc (ROCm 7.2) hoists the
            //  subtraction out of the conditional as `sub nuw` into min(), whose result is `noundef` -- and from there concludes c >= 32
            //  for every lane, dropping the `entry < c` guards of pass A below: stale entries were ranked and placed.)
            nx[k] = min(max(c[k], first) - first, 4u);
            if (nx[k])
                xq[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((static_cast<uint32_t>(PERM ? zrun[k] : zrun[k] - zs0) * kGroups + g) * scap + first) << 2, 0, FUSED ? 16 : 0);
        }
    }
    // pass A: rank of every entry among the block's entries for the same destination zone (= the histogram, once all are in)
#pragma unroll
    for (int k = 0; k < KR; ++k) {
#pragma unroll
        for (int d = 0; d < kE; ++d) {
            r[k][d] = 0;
            if (static_cast<uint32_t>(32 * (d / 4) + 4 * l8 + (d % 4)) < c[k]) r[k][d] = atomicAdd(&bins[v[k][d] >> idbits], 1u);
        }
    }
    if constexpr (kMedium) {
#pragma unroll
        for (int k = 0; k < KR; ++k) {
            const uint32_t x4[4] = {xq[k].x, xq[k].y, xq[k].z, xq[k].w};
#pragma unroll
            for (int d = 0; d < 4; ++d)
                if (static_cast<uint32_t>(d) < nx[k]) atomicAdd(&tbins[x4[d] >> idbits], 1u);
        }
    }
    if (l8 == 0) {
#pragma unroll
        for (int k = 0; k < KR; ++k) {
            const uint32_t surplus = c[k] > place_held(KDEEP, FUSED) ? c[k] - place_held(KDEEP, FUSED) : 0u;
            lstart[sub + k * kPlaceSeg] = surplus;
            if (surplus) s_any_long = 1u;
        }
    }
    CPM_PSTAMP(2);
    lds_barrier();
    CPM_PSTAMP(3);
    const bool any_long = __builtin_amdgcn_readfirstlane(static_cast<int>(s_any_long)) != 0;  // (block-uniform, in a scalar register)
    uint32_t ltotal = 0;
    if (any_long) ltotal = place_surplus_count<PB, KRUNS, KDEEP, ZPG, FUSED, PERM>(pl, D, g, zs0, zs1, scap, idbits);
    // The ticket (this block's range inside each bucket of the group) is requested now and needed only when the sorted list is
    // written out: its round trip runs under the block scan of the histogram (the zones' offsets in the sorted list) and the sort.
    const bool zone = tid < nzl;
    const uint32_t cr = zone ? bins[tid] : 0u, ct = zone ? tbins[tid] : 0u;
    uint32_t base = 0;
    if (cr + ct) base = atomicAdd(&cnt_a_next[zg0 + tid], cr + ct);
    uint32_t first;
    {
        const uint32_t incl = wave_incl_scan(cr);
        if (lane == 63) wsum[wave] = incl;
        lds_barrier();
        uint32_t before = 0;
        for (int w = 0; w < wave; ++w) before += wsum[w];  // (16 waves)
        first = before + incl - cr;
        if (zone) bins[tid] = first;
        if (tid == kPlaceBlock - 1) s_total = before + incl;
    }
    CPM_PSTAMP(4);
    lds_barrier();
    CPM_PSTAMP(5);
    // pass B: the entries held in registers go to their place in the sorted list
#pragma unroll
    for (int k = 0; k < KR; ++k) {
#pragma unroll
        for (int d = 0; d < kE; ++d)
            if (static_cast<uint32_t>(32 * (d / 4) + 4 * l8 + (d % 4)) < c[k]) {
                const uint32_t dl = v[k][d] >> idbits;
                const uint32_t li = bins[dl] + r[k][d];
                sorted_ids[li] = v[k][d] & idmask;
                sorted_zone[li] = static_cast<zone_t>(dl);
            }
    }
    if (zone) {  // (first use of the ticket)
        if (base + cr + ct > cap) atomicOr(status, 2ull);
        tbins[tid] = base + cr;
        delta[tid] = base - first;  // (wraps; only base - first + index is used)
    }
    CPM_PSTAMP(6);
    lds_barrier();
    // the sorted list out: consecutive lanes, consecutive slots of one bucket ...
    const uint32_t total = s_total;
    constexpr int kOutBatch = 4;  // (a thread's LDS reads of a batch are in flight together: zone, then delta + id, then the store)
    for (uint32_t i0 = tid; i0 < total; i0 += kOutBatch * kPlaceBlock) {
        uint32_t dl[kOutBatch], p[kOutBatch], idv[kOutBatch];
#pragma unroll
        for (int u = 0; u < kOutBatch; ++u) dl[u] = sorted_zone[min(i0 + u * kPlaceBlock, total - 1u)];  // (total > 0 inside the loop)
#pragma unroll
        for (int u = 0; u < kOutBatch; ++u) {
            p[u] = delta[dl[u]] + i0 + u * kPlaceBlock;
            idv[u] = sorted_ids[min(i0 + u * kPlaceBlock, total - 1u)];
        }
#pragma unroll
        for (int u = 0; u < kOutBatch; ++u)
            if (i0 + u * kPlaceBlock < total && p[u] < cap && (!SIGNAL || dl[u] < static_cast<uint32_t>(nzl)))
                ids_next[static_cast<size_t>(zg0 + dl[u]) * cap + (cap - 1u - p[u])] = idv[u];  // arrivals fill a region from its top
    }
    // ... the entries 32 .. 63 of the medium runs straight to their buckets (tbins: the running position behind the sorted ones) ...
    if constexpr (kMedium) {
#pragma unroll
        for (int k = 0; k < KR; ++k) {
            const uint32_t x4[4] = {xq[k].x, xq[k].y, xq[k].z, xq[k].w};
#pragma unroll
            for (int d = 0; d < 4; ++d)
                if (static_cast<uint32_t>(d) < nx[k]) {
                    const uint32_t dl = x4[d] >> idbits;
                    const uint32_t p = atomicAdd(&tbins[dl], 1u);
                    if (p < cap && (!SIGNAL || dl < static_cast<uint32_t>(nzl))) ids_next[static_cast<size_t>(zg0 + dl) * cap + (cap - 1u - p)] = x4[d] & idmask;
                }
        }
    }
    // ... and the surplus of the long runs straight to their buckets
    if (any_long) place_surplus_out<PB, KRUNS, KDEEP, ZPG, FUSED, SIGNAL, PERM>(pl, D, g, zs0, zs1, scap, idbits, zg0, nzl, cap, ids_next, ltotal);
    CPM_PSTAMP(7);
#if defined(CPM_DIAGNOSTIC) && !defined(CPM_STAMP_SAMPLER) && !defined(CPM_STAMP_BOTH)
    st_[7] = (st_[7] & ~1ull) | (any_long ? 1ull : 0ull);  // (the tick's lowest bit: did this block take the long-run path)
#endif
#if defined(CPM_DIAGNOSTIC) && defined(CPM_STAMP_BOTH)
    st_[0] = (st_[0] & ~0xFFull) | ((__builtin_amdgcn_s_getreg((4) | (8 << 6) | (7 << 11))) & 0xFFu);  // HW_REG_HW_ID bits 8..15: cu, sh, se
    st_[7] = (st_[7] & ~0x1Full) | xcc_id();
#endif
    CPM_PSTAMP_FLUSH;
    if constexpr (SIGNAL) {
        // the ids were stored plainly: they lie in THIS XCD's L2 (all writes to a bucket merge there), which is where a sampler
        // workgroup of the same XCD reads them (sc1 loads are served by the L2).  The block leaves its XCD's number beside the counter;
        // a sampler workgroup that finds another XCD's there than its own gives the step up (status bit 2).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (tid == 0) {
            __hip_atomic_fetch_or(done_out + 1, 1u << xcc_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(done_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int PB, int KRUNS, int KDEEP>
__global__ __launch_bounds__(PB) void k_grouped_place(const uint32_t *__restrict__ D, const uint32_t *__restrict__ cntg, int zpg, int zps,
                                                               int Z, uint32_t cap, uint32_t scap, uint32_t idbits,
                                                               uint32_t *__restrict__ cnt_a_next, uint32_t *__restrict__ ids_next,
                                                               unsigned long long *status)
{
    __shared__ PlaceLds<PB, KRUNS, kMaxZonesPerGroup> pl;
    extern __shared__ uint32_t sorted_ids[];  // [kSlots] ids in destination order, then [kSlots] u16: their zone inside the group
    grouped_place_body<PB, KRUNS, KDEEP, kMaxZonesPerGroup, false>(blockIdx.x % kGroups, blockIdx.x / kGroups, pl, sorted_ids, D, cntg, zpg, zps, Z, cap,
                                                                    scap, idbits, cnt_a_next, ids_next, status, nullptr, 0u, 0u);
}

// ------------------------------------------------------------------------------------------------ the fused hour
// ONE launch per hour: the sampler workgroups of hour t AND the placing blocks that move hour t's drivers into hour t + 1's
// buckets.  A placing block (destination group g, chunk j of kFusedChunk origin zones) needs the runs of exactly the kFusedChunk
// sampler workgroups of its chunk -- not the whole launch -- so it can run while later zones are still being sampled: the placing
// kernel is a chain of memory round trips that leaves the vector units idle, the sampler is bound by what it issues, and side by
// side they fill each other's gaps (two INDEPENDENT resamples on two streams gain 24 % that way, profiles/round2_notes.md; this
// is the same overlap inside one resample).  Block order: chunk c = kFusedChunk sampler workgroups, then the kGroups placing
// blocks of chunk c - lag (lag >= the number of chunks, the default: all sampler workgroups, then all placing blocks); a block only
// ever waits for blocks of LOWER index, which the dispatcher has started before it, so the wait cannot deadlock however the blocks
// are placed; it is bounded all the same (grouped_place_body).
// MEASURED (S4k, one box, profiles/round3_notes.md): two launches per hour 28.2 + 13.2 us; fused with lag 2 / 8 / 16 / 32 / 48 / 64:
// 62 / 60 / 48 / 40 / 37.2 / 37.0 us.  A placing block that arrives before its samplers are done holds one of the CU's six block
// slots while it waits -- every block of the launch carries the sampler's 106 scalar registers, so six blocks per CU is all there
// is -- and the sampler, bound by what it issues, needs those slots to hide its latencies.  What pays is not the interleave but the
// missing kernel boundary: placing blocks start in the slots the last sampler workgroups free, with no second launch ramp.  Chunk size + kGroups is a
// multiple of 8: the placing blocks of one group share blockIdx % 8 = one XCD = one L2, as in the two-launch form (speed only).
// Hand-off (MI355X_MICROARCH.md, inter-workgroup visibility): runs and run lengths stored sc1 by the samplers, every storing wave
// drains (s_waitcnt vmcnt(0)), workgroup barrier, one agent-scope add on the chunk's counter; the placing block polls that counter
// with relaxed agent-scope loads and reads the runs with sc1 loads.  Everything else crosses a kernel boundary as before: the
// buckets the placing blocks fill are read by the NEXT launch.
#ifndef CPM_FUSED_KRUNS
#define CPM_FUSED_KRUNS 4
#endif
constexpr int kFusedKruns = CPM_FUSED_KRUNS;
#ifndef CPM_FUSED_KDEEP
#define CPM_FUSED_KDEEP 2
#endif
// 2: 32 entries of a run in registers, 4: 64 (a second 16-byte load per lane).  MEASURED (S4k, one box, interleaved runs): 4 takes the
// long-run path out of most blocks' chains (a placing block lives 11.6 k cycles instead of 12.5 k) and the hour is the same to 0.1 %
// (0.8849 against 0.8861 ms per resample) for 128 B more read per run: 2 it stays.
constexpr int kFusedKdeep = CPM_FUSED_KDEEP;
constexpr int kFusedChunk = 16 * kFusedKruns;  // origin zones per chunk = runs a 256-thread placing block takes (8 lanes per run, KRUNS / 2 passes)
constexpr int kDoneStride = 32;  // words between the hand-off counters of consecutive chunks: a 128-B line each (64 adds and the polls of 32
                                 // placing blocks per counter; with all of an hour's counters in two lines every add and every poll of the
                                 // launch queued at one memory channel: 300 us per launch instead of 30)
constexpr int kFusedZpg = 256;   // zones per destination group the fused form is built for (Z <= 8,192)

// PERM: sampler workgroup b takes zone a.perm_t[b] -- the zones of the hour dealt LARGEST-FIRST (k_zone_order: by their size at the same
// hour of the initial-value problem, the same tables a day earlier) -- so that the workgroups that enter last are the short ones: buckets
// spread from half to 2.5 x the mean, and in zone order the sampler phase ended on whatever came last.  Chunks are chunks of POSITIONS:
// they still complete in launch order, and a placing block finds its chunk's zones through the same list.
template <int CPT, int NQ, bool SPARSE = false, bool PERM = false>
__global__ __launch_bounds__(kFusedThreads, CPM_WPS) CPM_SGPR_ATTR void k_grouped_hour(GroupedArgs a)
{
    extern __shared__ uint32_t dyn[];  // sampler: the zone's row pack; placing block: its sorted list
    __shared__ union {
        SampleLds s;
        PlaceLds<kFusedThreads, kFusedKruns, kFusedZpg> p;
    } u;
    constexpr int per = kFusedChunk + kGroups;
    const int nchunk = (a.Z + kFusedChunk - 1) / kFusedChunk;
    int z = -1, g = 0, j = -1;  // the block's role: sampler workgroup of zone z, or placing block (g, j)
    if (a.lag >= nchunk) {
        // all sampler workgroups, then all placing blocks (chunk by chunk: the first ones find their runs long complete)
        const int zr = (a.Z + 7) & ~7;  // (placing blocks of one group keep blockIdx % 8)
        if (static_cast<int>(blockIdx.x) < zr) {
            z = blockIdx.x;
            if (z >= a.Z) return;
            if constexpr (PERM) z = static_cast<int>(a.perm_t[blockIdx.x]);
        } else {
            const int b = blockIdx.x - zr;
            g = b % kGroups;
            j = b / kGroups;
        }
    } else {
        const int c = blockIdx.x / per, q = blockIdx.x % per;
        if (q < kFusedChunk) {
            z = c * kFusedChunk + q;
            if (c >= nchunk || z >= a.Z) return;
        } else {
            g = q - kFusedChunk;
            j = c - a.lag;
            if (j < 0) return;
        }
    }
    if (z >= 0) {
        grouped_sample_body<kFusedThreads, CPT, NQ, true, true, false, SPARSE>(
            a, z, dyn, u.s, a.done_t + static_cast<size_t>((PERM ? static_cast<int>(blockIdx.x) : z) / kFusedChunk) * kDoneStride);
    } else {
        const uint32_t need = static_cast<uint32_t>(min(kFusedChunk, a.Z - j * kFusedChunk));
#ifdef CPM_PLACE_PRIO
        __builtin_amdgcn_s_setprio(CPM_PLACE_PRIO);
#endif
        grouped_place_body<kFusedThreads, kFusedKruns, kFusedKdeep, kFusedZpg, true, false, PERM>(
            g, j, u.p, dyn, a.D, a.cntg, static_cast<int>(gdiv_zpg(a.gdiv)), kFusedChunk, a.Z, a.cap, a.scap, a.idbits, a.cnt_next + a.Z, a.ids_next, a.rare->status,
            a.done_t + static_cast<size_t>(j) * kDoneStride, need, a.spin_limit, nullptr, PERM ? a.perm_t : nullptr, SPARSE);
    }
}

// ------------------------------------------------------------------------------------------------ the hour, placing first
// ONE launch per hour the other way round: the placing blocks that move the PREVIOUS hour's drivers into this hour's buckets,
// then this hour's sampler workgroups.  At the head of a launch every placing block has its input complete (the runs were written
// by the previous launch) and all of them start at once, whereas behind the samplers of their own hour (k_grouped_hour) they
// trickle in as chunks complete and queue for the block slots the last samplers free: there the hour ends with ~1.6 placing-block
// lifetimes in which the vector units idle (tools/hour_stamps.py), here it begins with one.  A sampler workgroup waits for the
// placing blocks of ITS zone's destination group only (grouped_sample_body<WAIT>); blocks are laid out 8 groups at a time --
// block = (g / 8) * 8 * chunks + j * 8 + g % 8 -- so the groups of the first sets are complete while later sets still run (and the
// blocks of one group share blockIdx % 8 = one XCD).  A block only ever waits for blocks of LOWER index, which never wait themselves.
// GROUPED = false: the last hour of a resample (sampled, never applied) behind the placing of the hour before it.
template <int CPT, int NQ, bool GROUPED, bool SPARSE = false>
__global__ __launch_bounds__(kFusedThreads, CPM_WPS) CPM_SGPR_ATTR void k_grouped_hour_pf(GroupedArgs a)
{
    extern __shared__ uint32_t dyn[];  // sampler: the zone's row pack; placing block: its sorted list
    __shared__ union {
        SampleLds s;
        PlaceLds<kFusedThreads, kFusedKruns, kFusedZpg> p;
    } u;
    const int npl = a.pchunks * kGroups;  // (a multiple of 8)
    if (static_cast<int>(blockIdx.x) < npl) {
        const int b = blockIdx.x;
        const int g = (b / (8 * a.pchunks)) * 8 + (b & 7), j = (b >> 3) % a.pchunks;
        grouped_place_body<kFusedThreads, kFusedKruns, kFusedKdeep, kFusedZpg, false, true>(
            g, j, u.p, dyn, a.pD, a.pcntg, static_cast<int>(gdiv_zpg(a.gdiv)), kFusedChunk, a.Z, a.cap, a.scap, a.idbits, const_cast<uint32_t *>(a.cnt_a),
            const_cast<uint32_t *>(a.ids), a.rare->status, nullptr, 0u, 0u, a.done_t + static_cast<size_t>(g) * kDoneStride);
    } else {
        // sampler block 8 q + x takes a zone of a group g with g % 8 == x: the XCD its arrivals were placed on (blockIdx % 8)
        const int bs = blockIdx.x - npl, zpg = static_cast<int>(gdiv_zpg(a.gdiv));
        const int g = (bs & 7) + 8 * ((bs >> 3) / zpg), z = g * zpg + ((bs >> 3) % zpg);
        if (g >= kGroups || z >= a.Z) return;
        grouped_sample_body<kFusedThreads, CPT, NQ, GROUPED, false, true, SPARSE>(a, z, dyn, u.s, nullptr, a.done_t + static_cast<size_t>(g) * kDoneStride,
                                                                          static_cast<uint32_t>(a.pchunks));
    }
}

// Geometry of a placing launch: threads per block, runs per 16 threads (KRUNS = 4 or 8: KRUNS / 2 passes of the 8-lane segments), blocks per destination group
struct PlaceShape {
    int pb, kruns, bpg;
};
inline PlaceShape place_shape(int Z)
{
    const int zpg = static_cast<int>(grouped_zpg_of(Z));
    PlaceShape p;
    p.pb = zpg <= 512 ? 512 : 1024;
    const int seg = p.pb / 16;
    p.bpg = Z < 1024 ? 8 : 16;
    while (p.bpg < 128 && (Z + p.bpg - 1) / p.bpg > 4 * seg) p.bpg *= 2;
    p.kruns = (Z + p.bpg - 1) / p.bpg <= 4 * seg ? 4 : 8;
    return p;
}
inline bool place_shape_fits(int Z)
{
    const PlaceShape p = place_shape(Z);
    return (Z + p.bpg - 1) / p.bpg <= 8 * (p.pb / 16);
}

template <int PB, int KRUNS>
inline void grouped_launch_place_t(hipStream_t stream, int bpg, const uint32_t *D, const uint32_t *cntg, int zpg, int Z, uint32_t cap, uint32_t scap,
                                   uint32_t idbits, uint32_t *cnt_next, uint32_t *ids_next, unsigned long long *status)
{
    const int zps = (Z + bpg - 1) / bpg;
    const size_t lds = static_cast<size_t>(6) * KRUNS * 2 * PB;  // sorted ids (4 B) + their zones (2 B) per slot
    if (lds > 48 * 1024) {  // LDS opt-in, once per device
        static bool attr_done[64] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 64 || !attr_done[dev]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_grouped_place<PB, KRUNS, 2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      static_cast<int>(lds));
            if (dev >= 0 && dev < 64) attr_done[dev] = true;
        }
    }
    launch(k_grouped_place<PB, KRUNS, 2>, dim3(kGroups * bpg), dim3(PB), lds, stream, D, cntg, zpg, zps, Z, cap, scap, idbits, cnt_next,
                       ids_next, status);
}

inline void grouped_launch_place(hipStream_t stream, const uint32_t *D, const uint32_t *cntg, int zpg, int Z, uint32_t cap, uint32_t scap,
                                 uint32_t idbits, uint32_t *cnt_next, uint32_t *ids_next, unsigned long long *status)
{
    const PlaceShape p = place_shape(Z);
#define CPM_PLACE_ARGS stream, p.bpg, D, cntg, zpg, Z, cap, scap, idbits, cnt_next, ids_next, status
    if (p.pb == 512 && p.kruns == 4) grouped_launch_place_t<512, 4>(CPM_PLACE_ARGS);
    else if (p.pb == 512) grouped_launch_place_t<512, 8>(CPM_PLACE_ARGS);
    else if (p.kruns == 4) grouped_launch_place_t<1024, 4>(CPM_PLACE_ARGS);
    else grouped_launch_place_t<1024, 8>(CPM_PLACE_ARGS);
#undef CPM_PLACE_ARGS
}

// ------------------------------------------------------------------------------------------------ travel times
// Travel times of an hour's drivers (src/resampling.jl:53-69), read back from the runs by a kernel of their own (inside the
// sampler they cost it its occupancy: 138 VGPRs).  The sum is an integer in 2^-16 s units: order-free, bit-exact.
constexpr int kTravelParts = 256;
// Threads of the block of one (origin zone, hour): one wave while a lane has at most ~16 drivers (the ~500 drivers of a 1000-car zone
// are two batches of four per lane; 256 threads made 4 x the waves for the same drivers -- 226 k per resample at Z = 2,357 -- and the
// launch of waves was a third of the kernel: 349 us against 305), more for larger buckets, 256 once the context has seen heavy ones.
inline int travel_block(int64_t mean_bucket, bool heavy_seen)
{
    if (heavy_seen || mean_bucket > 4096) return 256;
    return mean_bucket > 2048 ? 128 : 64;
}
// LDS of a travel block on sparse rows: the largest row
inline size_t travel_lds_bytes(size_t row_bytes, int block)
{
    (void)block;
    return row_bytes;
}

// The travel kernel gathers (mean, std) of its drivers' (origin, destination, hour) cells.  In the reference's datamatrix layout
// [2][T][dest][origin] the two values lie Z*Z*T*8 B apart and consecutive destinations of one origin Z*8 B apart: two cache lines per
// driver, none shared.  travel table = the same numbers as tt[t][origin][dest] = (mean, std): the drivers of an origin zone (one block)
// read one 16-B cell each inside one contiguous Z*16-B row, so lines are shared and stay in L2.  Built once per datamatrix.
constexpr int kTtTile = 32;
__global__ __launch_bounds__(kTtTile * 8) void k_build_travel_table(const double *__restrict__ dm, double2 *__restrict__ tt, int Z, int T)
{
    __shared__ double2 tile[kTtTile][kTtTile + 1];
    const int t = blockIdx.z;
    const int o0 = blockIdx.x * kTtTile, d0 = blockIdx.y * kTtTile;
    const int tx = threadIdx.x & (kTtTile - 1), ty = threadIdx.x / kTtTile;  // 32 x 8
    const size_t sd_off = static_cast<size_t>(Z) * Z * T;
    for (int r = ty; r < kTtTile; r += 8) {  // read: origin on the lane (the reference's fastest index)
        const int o = o0 + tx, d = d0 + r;
        double2 v = make_double2(0.0, 0.0);
        if (o < Z && d < Z) {
            const size_t cell = static_cast<size_t>(o) + static_cast<size_t>(Z) * (d + static_cast<size_t>(Z) * t);
            v = make_double2(dm[cell], dm[cell + sd_off]);
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < kTtTile; r += 8) {  // write: destination on the lane
        const int o = o0 + r, d = d0 + tx;
        if (o < Z && d < Z) tt[(static_cast<size_t>(t) * Z + o) * Z + d] = tile[tx][r];
    }
}

// The travel table as SPARSE rows, for staging in LDS: real Uber Movement tables hold ~9 % of the (origin, destination, hour) cells
// (README.md:302-310), so an origin's row of an hour is a bitmap of Z bits + ~200 cells of 16 B -- a few KB that the block of that
// (origin, hour) brings into LDS once, instead of one scattered 16-B global load per driver.  Per row (t, o): words[w] = (bitmap of
// destinations 32 w .. 32 w + 31 with a non-zero cell, number of non-zero cells of the row in front of word w), cells in destination
// order at cells[off[row] ...].
// Built straight from the datamatrix (reference layout [2][T][dest][origin]: the origin is the fastest index, so a wave takes 64
// origins, lane = origin, and walks destinations -- every load a whole line): bitmaps (k_tts_bits), the cells in front of every
// word and the rows' sizes (k_tts_prefix), the rows' offsets (k_tts_offsets, one block), the cells (k_tts_cells).  The dense
// origin-major table this used to go through (2.1 GB written and read twice at Melbourne's size, allocated and freed per dataset)
// is only built when a row does not fit LDS.
constexpr int kTtsSplit = 4;   // parts of a row's words, one wave each (888 waves of 64 origins would leave most of the chip empty)
constexpr int kTtsBatch = 16;  // destinations whose loads a lane has in flight together
__global__ __launch_bounds__(64) void k_tts_bits(const double *__restrict__ dm, uint2 *__restrict__ words, int Z, int T, int W)
{
    const int o = blockIdx.x * 64 + threadIdx.x, t = blockIdx.y;
    const int wper = (W + kTtsSplit - 1) / kTtsSplit, w0 = blockIdx.z * wper, w1 = min(W, w0 + wper);
    if (o >= Z) return;
    const size_t sd_off = static_cast<size_t>(Z) * Z * T;
    const double *src = dm + static_cast<size_t>(o) + static_cast<size_t>(Z) * Z * t;  // + Z * d
    uint2 *dst = words + (static_cast<size_t>(t) * Z + o) * W;
    for (int w = w0; w < w1; ++w) {
        uint32_t bits = 0;
#pragma unroll
        for (int h = 0; h < 32; h += kTtsBatch) {
            double m[kTtsBatch], sdv[kTtsBatch];
#pragma unroll
            for (int u = 0; u < kTtsBatch; ++u) {
                const int d = min(32 * w + h + u, Z - 1);
                m[u] = src[static_cast<size_t>(Z) * d];
                sdv[u] = src[static_cast<size_t>(Z) * d + sd_off];
            }
#pragma unroll
            for (int u = 0; u < kTtsBatch; ++u)
                if (32 * w + h + u < Z && (m[u] != 0.0 || sdv[u] != 0.0)) bits |= 1u << (h + u);
        }
        dst[w] = make_uint2(bits, 0u);
    }
}
// words[row][w].y = non-zero cells of the row in front of word w; count[row] = the row's cells.  One lane per row.
__global__ __launch_bounds__(256) void k_tts_prefix(uint2 *__restrict__ words, uint32_t *__restrict__ count, int64_t rows, int W)
{
    const int64_t row = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (row >= rows) return;
    uint2 *wr = words + row * W;
    uint32_t run = 0;
    for (int w = 0; w < W; ++w) {
        const uint32_t bits = wr[w].x;
        wr[w].y = run;
        run += static_cast<uint32_t>(__popc(bits));
    }
    count[row] = run;
}
// off[r] = sum of count[0 .. r-1], off[rows] = total; *max_out = largest row.  One block (rows is T x Z: ~10^5).
__global__ __launch_bounds__(1024) void k_tts_offsets(const uint32_t *__restrict__ count, uint32_t *__restrict__ off, int64_t rows, uint32_t *__restrict__ max_out,
                                                      unsigned long long *__restrict__ total_out)
{
    __shared__ unsigned long long part[1024];
    __shared__ uint32_t pmax[1024];
    const int tid = threadIdx.x;
    const int64_t per = (rows + 1023) / 1024, r0 = tid * per, r1 = min(r0 + per, rows);
    unsigned long long s = 0;
    uint32_t mx = 0;
    for (int64_t r = r0; r < r1; ++r) {
        s += count[r];
        mx = max(mx, count[r]);
    }
    part[tid] = s;
    pmax[tid] = mx;
    __syncthreads();
    if (tid == 0) {
        unsigned long long acc = 0;
        uint32_t m = 0;
        for (int k = 0; k < 1024; ++k) {
            const unsigned long long v = part[k];
            part[k] = acc;
            acc += v;
            m = max(m, pmax[k]);
        }
        *max_out = m;
        *total_out = acc;
    }
    __syncthreads();
    unsigned long long acc = part[tid];
    for (int64_t r = r0; r < r1; ++r) {
        off[r] = static_cast<uint32_t>(acc);
        acc += count[r];
    }
    if (tid == 1023) off[rows] = static_cast<uint32_t>(acc);
}
__global__ __launch_bounds__(64) void k_tts_cells(const double *__restrict__ dm, const uint2 *__restrict__ words, const uint32_t *__restrict__ off,
                                                  TravelCell *__restrict__ cells, int Z, int T, int W)
{
    const int o = blockIdx.x * 64 + threadIdx.x, t = blockIdx.y;
    const int wper = (W + kTtsSplit - 1) / kTtsSplit, w0 = blockIdx.z * wper, w1 = min(W, w0 + wper);
    if (o >= Z) return;
    const size_t sd_off = static_cast<size_t>(Z) * Z * T;
    const double *src = dm + static_cast<size_t>(o) + static_cast<size_t>(Z) * Z * t;
    const size_t row = static_cast<size_t>(t) * Z + o;
    const uint2 *wr = words + row * W;
    TravelCell *dst = cells + off[row];
    for (int w = w0; w < w1; ++w) {
        const uint2 word = wr[w];
        uint32_t at = word.y;
#pragma unroll
        for (int h = 0; h < 32; h += kTtsBatch) {
            double m[kTtsBatch], sdv[kTtsBatch];
#pragma unroll
            for (int u = 0; u < kTtsBatch; ++u) {
                const int d = min(32 * w + h + u, Z - 1);
                m[u] = src[static_cast<size_t>(Z) * d];
                sdv[u] = src[static_cast<size_t>(Z) * d + sd_off];
            }
#pragma unroll
            for (int u = 0; u < kTtsBatch; ++u)
                if (word.x & (1u << (h + u))) {
                    // a cell as the travel kernel uses it: (mean, sigma, mass of the window) with sigma = std, or a tenth of the mean where the
                    // data hold no std (src/resampling.jl:65-67) -- the window's mass (truncnormal_mass: an erf) once per cell instead of per driver
                    const double s1 = (sdv[u] == 0) ? 0.1 * m[u] : sdv[u];
                    TravelCell tc;
                    tc.mu = m[u];
                    tc.sigma = s1;
                    tc.mass = truncnormal_mass(m[u], s1);
                    dst[at++] = tc;
                }
        }
    }
}

struct TravelArgs {
    const double2 *tt;            // [T][Z][Z] (mean, std), origin-major
    const uint2 *tts_words;       // sparse rows (k_tts_*): [T*Z][W] (bitmap, cells in front), or null: gather from tt
    const uint32_t *tts_off;      // [T*Z + 1] first cell of every row -- or, fixed-stride rows (tts_stride != 0, cpm_dataset.h): [T*Z] cells per row
    uint32_t tts_stride;          // ... the row's cells start at row * tts_stride
    const TravelCell *tts_cells;  // the non-zero cells, row by row: (mean, sigma, mass of the window)
    int W;                        // bitmap words per row
    uint32_t list_off;            // sparse rows: byte offset of the threads' driver lists in the block's LDS (behind the largest row)
    unsigned long long *tt_part;  // [kTravelParts] partial sums, zero between resamples
    size_t d_stride, c_stride;    // words between the runs / run lengths of consecutive hours (one launch for all hours), or 0
    int t0, zpg;                  // hour of blockIdx.y = 0; zones per destination group
    uint32_t step0;               // its Philox step
    CarIndex cars;
    uint64_t seed;
};

// one block (travel_block threads) per (origin zone, hour).  When the runs of all hours of a resample are kept (GroupedWork::history) ONE launch
// at the end serves them all: 24 x fewer launches and a grid 24 x as deep (at Z = 2,357 an hourly grid is 1.15 rounds of blocks).
template <bool SPARSE>
__global__ __launch_bounds__(256) void k_grouped_travel(const uint32_t *__restrict__ D, const uint32_t *__restrict__ cntg, int Z, uint32_t scap,
                                                        uint32_t idbits, TravelArgs tr)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char travel_lds[];  // SPARSE: the origin's row of this hour: W words, then its cells
    const int hour = tr.t0 + static_cast<int>(blockIdx.y);
    const uint32_t step = tr.step0 + blockIdx.y;
    D += tr.d_stride * blockIdx.y;
    cntg += tr.c_stride * blockIdx.y;
    // The zone's drivers are dealt evenly over the threads whatever the run lengths are (popular destination groups
    // hold most of them): driver i of the zone sits in run g with prefix[g] <= i < prefix[g+1].
    __shared__ uint32_t prefix[kGroups + 1];
    __shared__ unsigned long long s_tt;
    const int z = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    uint2 *row_words = reinterpret_cast<uint2 *>(travel_lds);
    TravelCell *row_cells = reinterpret_cast<TravelCell *>(travel_lds + ((static_cast<size_t>(tr.W) * sizeof(uint2) + 15) & ~static_cast<size_t>(15)));
    if constexpr (SPARSE) {  // (requested first: lands under the prefix scan of the run lengths)
        const size_t row = static_cast<size_t>(hour) * Z + z;
        const size_t c0 = tr.tts_stride ? row * tr.tts_stride : tr.tts_off[row];
        const uint32_t nc = tr.tts_stride ? min(tr.tts_off[row], tr.tts_stride) : tr.tts_off[row + 1] - tr.tts_off[row];
        for (int i = tid; i < tr.W; i += blockDim.x) row_words[i] = tr.tts_words[row * tr.W + i];
        for (uint32_t i = tid; i < nc; i += blockDim.x) row_cells[i] = tr.tts_cells[c0 + i];
    }
    if (tid < 64) {
        const uint32_t c = (lane < kGroups) ? min(cntg[static_cast<size_t>(z) * kGroups + lane], scap) : 0u;
        uint32_t incl = c;
        for (int o = 1; o < kGroups; o <<= 1) {
            const uint32_t v = __shfl_up(incl, o, 64);
            if (lane >= o) incl += v;
        }
        if (lane < kGroups) prefix[lane + 1] = incl;
        if (lane == 0) {
            prefix[0] = 0;
            s_tt = 0;
        }
    }
    __syncthreads();
    const uint32_t total = prefix[kGroups];
    const uint32_t idmask = (idbits >= 32) ? 0xFFFFFFFFu : ((1u << idbits) - 1u);
    long long tt = 0;
    if constexpr (SPARSE) {
        // Batches of kTravelList drivers per thread: the run entries (requested together), the destinations, the cell of each in the
        // staged row, ONE draw each (truncnormal_draw: the window's mass comes with the cell); drivers inside the zone take 300 s.
        // (The rejection sampler of rounds 1-3 kept per-lane lists of (cell, car) in LDS so that a lane made one attempt per trip
        //  whatever driver it was at; with one draw per driver there is nothing to balance.)
        constexpr int kTravelList = 8;
        constexpr uint32_t kNoCell = 0xFFFFFFFFu;
        const uint32_t nthr = blockDim.x;
        for (uint32_t base = 0; base < total; base += kTravelList * nthr) {
            uint32_t w[kTravelList], gq[kTravelList];
            bool live[kTravelList];
#pragma unroll
            for (int u = 0; u < kTravelList; ++u) {
                const uint32_t i = base + tid + u * nthr;
                live[u] = i < total;
                uint32_t g = 0;
#pragma unroll
                for (int step = kGroups / 2; step > 0; step >>= 1)
                    if (prefix[g + step] <= i) g += step;
                g = live[u] ? g : 0u;
                gq[u] = g;
                w[u] = live[u] ? D[(static_cast<size_t>(z) * kGroups + g) * scap + (i - prefix[g])] : 0u;
            }
#pragma unroll 2
            for (int u = 0; u < kTravelList; ++u) {
                if (!live[u]) continue;
                const uint32_t dest = gq[u] * static_cast<uint32_t>(tr.zpg) + (w[u] >> idbits);
                if (dest == static_cast<uint32_t>(z)) {  // same zone: 300 s (src/resampling.jl:58-60)
                    tt += q16(300.0);
                } else {
                    const uint2 rw = row_words[dest >> 5];
                    const uint32_t bit = 1u << (dest & 31u);
                    const uint32_t cell = (rw.x & bit) ? rw.y + static_cast<uint32_t>(__popc(rw.x & (bit - 1u))) : kNoCell;
                    // (no data for the pair: mean 0, sigma 0 -> the draw is the mean, as the sampler has it)
                    double mu = 0.0, sigma = 0.0, mass = 0.0;
                    if (cell != kNoCell) {
                        const TravelCell c = row_cells[cell];
                        mu = c.mu;
                        sigma = c.sigma;
                        mass = c.mass;
                    }
                    double u1, u2;
                    car_uniforms(tr.seed, tr.cars.global(w[u] & idmask), step, 1u, u1, u2);
                    tt += q16(truncnormal_draw(u1, mu, sigma, mass));
                }
            }
        }
    } else {
    // Batches of kTravelBatch drivers per thread: their run entries, then their two datamatrix cells, are requested together
    // (a driver's chain entry -> cell -> mean, std -> draws is otherwise three exposed round trips).
    constexpr int kTravelBatch = 4;
    const double2 *tt_row = tr.tt + (static_cast<size_t>(hour) * Z + z) * Z;
    for (uint32_t i0 = tid; i0 < total; i0 += kTravelBatch * blockDim.x) {
        uint32_t w[kTravelBatch], dest[kTravelBatch];
        bool live[kTravelBatch];
#pragma unroll
        for (int u = 0; u < kTravelBatch; ++u) {
            const uint32_t i = i0 + u * blockDim.x;
            live[u] = i < total;
            uint32_t g = 0;
#pragma unroll
            for (int step = kGroups / 2; step > 0; step >>= 1)
                if (prefix[g + step] <= i) g += step;
            g = live[u] ? g : 0u;
            w[u] = live[u] ? D[(static_cast<size_t>(z) * kGroups + g) * scap + (i - prefix[g])] : 0u;
            dest[u] = g * static_cast<uint32_t>(tr.zpg) + (w[u] >> idbits);
        }
        double mean[kTravelBatch], sd[kTravelBatch];
#pragma unroll
        for (int u = 0; u < kTravelBatch; ++u) {
            const bool moving = live[u] && dest[u] != static_cast<uint32_t>(z);
            double2 cell = make_double2(0.0, 0.0);
            if (moving) cell = tt_row[dest[u]];
            mean[u] = cell.x;
            sd[u] = cell.y;
        }
#pragma unroll
        for (int u = 0; u < kTravelBatch; ++u) {
            if (!live[u]) continue;
            if (dest[u] == static_cast<uint32_t>(z)) {  // same zone: 300 s (src/resampling.jl:58-60)
                tt += q16(300.0);
            } else {
                const double s1 = (sd[u] == 0) ? 0.1 * mean[u] : sd[u];  // :65-67
                tt += q16(truncnormal_pm10(tr.seed, tr.cars.global(w[u] & idmask), step, 1, mean[u], s1));
            }
        }
    }
    }
    // one global atomic per block, spread over kTravelParts words (atomics on ONE word are served one at a time at the memory
    // side: four per block on the sum itself made this kernel 119 us per hour); k_grouped_travel_finish adds the parts up
    for (int o = 32; o > 0; o >>= 1) tt += __shfl_down(tt, o, 64);
    if (lane == 0 && tt) atomicAdd(&s_tt, static_cast<unsigned long long>(tt));
    __syncthreads();
    if (tid == 0 && s_tt) atomicAdd(&tr.tt_part[(z + 31 * blockIdx.y) % kTravelParts], s_tt);
}

__global__ __launch_bounds__(256) void k_grouped_travel_finish(unsigned long long *__restrict__ tt_part, unsigned long long *__restrict__ tt_sum)
{
    unsigned long long v = tt_part[threadIdx.x];
    tt_part[threadIdx.x] = 0;  // ready for the next resample
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(tt_sum, v);
}

// ------------------------------------------------------------------------------------------------ launch helpers
constexpr int kSampleBlock = 256;  // measured at S4k: 512 threads x 2 cars: 31 us, 256 x 4: 28, 128 x 8: 44

template <bool GROUPED, int CPT, int NQ, bool SPARSE = false>
inline void grouped_launch_nq(const GroupedArgs &a, size_t lds, hipStream_t stream)
{
    if (lds > 48 * 1024) {  // LDS opt-in, once per device (contexts of several devices may live in one process)
        static bool attr_done[64] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 64 || !attr_done[dev]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_grouped_sample<kSampleBlock, CPT, NQ, GROUPED, SPARSE>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            if (dev >= 0 && dev < 64) attr_done[dev] = true;
        }
    }
    launch(k_grouped_sample<kSampleBlock, CPT, NQ, GROUPED, SPARSE>, dim3(a.Z), dim3(kSampleBlock), lds, stream, a);
}

template <bool GROUPED, int CPT>
inline void grouped_launch_c(const GroupedArgs &a, hipStream_t stream)
{
    const int words = pack_row_words(a.Zq, a.G, a.smap);
    const size_t lds = sizeof(uint32_t) * static_cast<size_t>(words);
    const int need = (words / 4 + kSampleBlock - 1) / kSampleBlock;
    if (a.smap) {  // (a sparse pack is at most kDsCap entries: one LDS-DMA instruction per wave; sparse_shape_ok)
        grouped_launch_nq<GROUPED, CPT, 1, true>(a, lds, stream);
        return;
    }
    if (need <= 1) grouped_launch_nq<GROUPED, CPT, 1>(a, lds, stream);
    else if (need <= 2) grouped_launch_nq<GROUPED, CPT, 2>(a, lds, stream);
    else if (need <= 3) grouped_launch_nq<GROUPED, CPT, 3>(a, lds, stream);
    else if (need <= 4) grouped_launch_nq<GROUPED, CPT, 4>(a, lds, stream);
    else if (need <= 5) grouped_launch_nq<GROUPED, CPT, 5>(a, lds, stream);
    else if (need <= 6) grouped_launch_nq<GROUPED, CPT, 6>(a, lds, stream);
    else if (need <= 8) grouped_launch_nq<GROUPED, CPT, 8>(a, lds, stream);
    else if (need <= 12) grouped_launch_nq<GROUPED, CPT, 12>(a, lds, stream);
    else if (need <= 20) grouped_launch_nq<GROUPED, CPT, 20>(a, lds, stream);
    else grouped_launch_nq<GROUPED, CPT, 40>(a, lds, stream);
}

inline int grouped_cpt(int64_t mean) { return mean <= 224 ? 1 : (mean <= 560 ? 2 : 4); }
// ... and where no heavy bucket has been seen (the heavy launch shares the sampler's chunking: it stays with grouped_cpt): slots for
// 1.5 x the mean bucket.  Buckets spread from half to 2.5 x the mean on flat tables, and a bucket beyond its workgroup's slots pays a
// whole serial round (ids, Philox, search, ranks) per 256 cars more, while a wave only runs the K <= CPT cars per lane it holds
// (first_pass): at S4k 42 % of the buckets needed such rounds with 4 cars per lane, 0.851 -> 0.815 ms per resample with 6 (same box,
// interleaved; 5: 0.827, 7: 0.824 with 9 vector registers spilled, 8 at five waves per SIMD: 0.856 -- profiles/round4_notes.md).
#ifndef CPM_CPT_RULE
#define CPM_CPT_RULE 1
#endif
inline int grouped_cpt_wide(int64_t mean)
{
    if (CPM_CPT_RULE == 0) return grouped_cpt(mean);
    return mean <= 170 ? 1 : (mean <= 340 ? 2 : (mean <= 700 ? 4 : 6));
}

// the fused hour (k_grouped_hour): (chunks + lag) x (kFusedChunk sampler workgroups + kGroups placing blocks)
inline size_t fused_lds_bytes(int Zq, int G, int smap = 0)
{
    return std::max(sizeof(uint32_t) * static_cast<size_t>(pack_row_words(Zq, G, smap)), static_cast<size_t>(4 + sizeof(PlaceLds<kFusedThreads, kFusedKruns, kFusedZpg>::zone_t)) * kFusedKruns * kFusedKdeep * kFusedThreads);
}
template <int CPT, int NQ, bool SPARSE = false, bool PERM = false>
inline void grouped_launch_hour_nq(const GroupedArgs &a, hipStream_t stream)
{
    if constexpr (!PERM) {
        if (a.perm_t && a.lag >= (a.Z + kFusedChunk - 1) / kFusedChunk) {  // (zones dealt largest-first: only in the samplers-then-placing order)
            grouped_launch_hour_nq<CPT, NQ, SPARSE, true>(a, stream);
            return;
        }
    }
    const size_t lds = fused_lds_bytes(a.Zq, a.G, a.smap);
    if (lds > 48 * 1024) {
        static bool attr_done[64] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 64 || !attr_done[dev]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_grouped_hour<CPT, NQ, SPARSE, PERM>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            if (dev >= 0 && dev < 64) attr_done[dev] = true;
        }
    }
    const int nchunk = (a.Z + kFusedChunk - 1) / kFusedChunk;
    const unsigned blocks = a.lag >= nchunk ? static_cast<unsigned>(((a.Z + 7) & ~7) + nchunk * kGroups)
                                            : static_cast<unsigned>((nchunk + a.lag) * (kFusedChunk + kGroups));
    launch(k_grouped_hour<CPT, NQ, SPARSE, PERM>, dim3(blocks), dim3(kFusedThreads), lds, stream, a);
}
// true when an instantiation exists for this problem (the common pack sizes; others take two launches per hour)
inline bool fused_shape_ok(int Z, int Zq, int G, int smap = 0)
{
    const int need = (pack_row_words(Zq, G, smap) / 4 + kSampleBlock - 1) / kSampleBlock;
    return static_cast<int>(grouped_zpg_of(Z)) <= kFusedZpg && need <= 12;
}
// ... and where it PAYS.  MEASURED (one launch against two per hour, 1,000 cars per zone, interleaved runs on one box, ms per
// resample): Z = 1,536: 0.524 / 0.493, 2,357: 0.664 / 0.650, 3,072: 0.710 / 0.727, 4,096: 0.868 / 0.912 (500 cars per zone: 0.703 /
// 0.760, 2,000: 2.21 / 2.21), 5,120: 1.283 / 1.362, 6,144: 1.680 / 1.615, 8,192 x 500: 2.19 / 2.07.  It pays from two rounds of sampler
// workgroups on (12 per CU) while the blocks' LDS -- every block of the fused launch carries the row pack's -- leaves five per
// CU: below, the one launch has no second round to tuck its placing blocks behind; above, the placing blocks sit four to a CU.
// Sparse packs (cpm_dataset.h: a few KB per row): measured at Z = 2,357 on Melbourne-shaped tables, one launch / two launches per hour,
// ms per resample: 1,000 cars per zone 0.721 / 0.742, 100 cars per zone 0.331 / 0.322 -- one launch from ~500 cars per zone on.
inline bool fused_pays(int Z, int Zq, int G, int cu_count, int smap = 0, int64_t mean = 0)
{
    // (round 4, with the zones dealt largest-first and groups of any size: one launch / two launches per hour at Z = 2,357, Melbourne-shaped,
    //  100 cars per zone 0.331 / 0.309, 200: 0.352 / 0.342, 300: 0.372 / 0.406, 400: 0.395 / 0.442, 500: 0.412 / 0.471 -- one launch from 256 on)
    if (smap) return mean >= 256 && Z >= 6 * std::max(cu_count, 1);
    const size_t lds = fused_lds_bytes(Zq, G, smap) + 4608;  // (+ the static part: SampleLds / PlaceLds)
    return Z >= 12 * std::max(cu_count, 1) && 5 * lds <= 160 * 1024;
}
template <int CPT>
inline void grouped_launch_hour_c(const GroupedArgs &a, hipStream_t stream)
{
    const int need = (pack_row_words(a.Zq, a.G, a.smap) / 4 + kSampleBlock - 1) / kSampleBlock;
    if (a.smap) {
        grouped_launch_hour_nq<CPT, 1, true>(a, stream);
        return;
    }
    if (need <= 1) grouped_launch_hour_nq<CPT, 1>(a, stream);
    else if (need <= 2) grouped_launch_hour_nq<CPT, 2>(a, stream);
    else if (need <= 3) grouped_launch_hour_nq<CPT, 3>(a, stream);
    else if (need <= 4) grouped_launch_hour_nq<CPT, 4>(a, stream);
    else if (need <= 5) grouped_launch_hour_nq<CPT, 5>(a, stream);
    else if (need <= 6) grouped_launch_hour_nq<CPT, 6>(a, stream);
    else if (need <= 8) grouped_launch_hour_nq<CPT, 8>(a, stream);
    else grouped_launch_hour_nq<CPT, 12>(a, stream);
}
inline void grouped_launch_hour(const GroupedArgs &a, int64_t mean, hipStream_t stream)
{
    switch (grouped_cpt_wide(mean)) {  // (one launch per hour only while no heavy bucket has been seen)
    case 1: grouped_launch_hour_c<1>(a, stream); break;
    case 2: grouped_launch_hour_c<2>(a, stream); break;
    case 4: grouped_launch_hour_c<4>(a, stream); break;
    default: grouped_launch_hour_c<6>(a, stream); break;
    }
}

template <int CPT, int NQ, bool GROUPED, bool SPARSE = false>
inline void grouped_launch_hour_pf_nq(const GroupedArgs &a, hipStream_t stream)
{
    const size_t lds = fused_lds_bytes(a.Zq, a.G, a.smap);
    if (lds > 48 * 1024) {
        static bool attr_done[64] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 64 || !attr_done[dev]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_grouped_hour_pf<CPT, NQ, GROUPED, SPARSE>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            if (dev >= 0 && dev < 64) attr_done[dev] = true;
        }
    }
    launch(k_grouped_hour_pf<CPT, NQ, GROUPED, SPARSE>, dim3(static_cast<unsigned>(a.pchunks * kGroups + kGroups * static_cast<int>(gdiv_zpg(a.gdiv)))), dim3(kFusedThreads), lds, stream, a);
}
template <int CPT, bool GROUPED>
inline void grouped_launch_hour_pf_c(const GroupedArgs &a, hipStream_t stream)
{
    const int need = (pack_row_words(a.Zq, a.G, a.smap) / 4 + kSampleBlock - 1) / kSampleBlock;
    if (a.smap) {
        grouped_launch_hour_pf_nq<CPT, 1, GROUPED, true>(a, stream);
        return;
    }
    if (need <= 1) grouped_launch_hour_pf_nq<CPT, 1, GROUPED>(a, stream);
    else if (need <= 2) grouped_launch_hour_pf_nq<CPT, 2, GROUPED>(a, stream);
    else if (need <= 3) grouped_launch_hour_pf_nq<CPT, 3, GROUPED>(a, stream);
    else if (need <= 4) grouped_launch_hour_pf_nq<CPT, 4, GROUPED>(a, stream);
    else if (need <= 5) grouped_launch_hour_pf_nq<CPT, 5, GROUPED>(a, stream);
    else if (need <= 6) grouped_launch_hour_pf_nq<CPT, 6, GROUPED>(a, stream);
    else if (need <= 8) grouped_launch_hour_pf_nq<CPT, 8, GROUPED>(a, stream);
    else grouped_launch_hour_pf_nq<CPT, 12, GROUPED>(a, stream);
}
template <bool GROUPED>
inline void grouped_launch_hour_pf(const GroupedArgs &a, int64_t mean, hipStream_t stream)
{
    switch (grouped_cpt(mean)) {
    case 1: grouped_launch_hour_pf_c<1, GROUPED>(a, stream); break;
    case 2: grouped_launch_hour_pf_c<2, GROUPED>(a, stream); break;
    default: grouped_launch_hour_pf_c<4, GROUPED>(a, stream); break;
    }
}

template <int CPT, int NQ, bool SPARSE = false>
inline void grouped_launch_heavy_nq(const GroupedArgs &a, int parts, int hgrid, size_t lds, hipStream_t stream)
{
    if (lds > 48 * 1024) {
        static bool attr_done[64] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 64 || !attr_done[dev]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_grouped_sample_heavy<kSampleBlock, CPT, NQ, SPARSE>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            if (dev >= 0 && dev < 64) attr_done[dev] = true;
        }
    }
    hipLaunchKernelGGL((k_grouped_sample_heavy<kSampleBlock, CPT, NQ, SPARSE>), dim3(hgrid), dim3(kSampleBlock), lds, stream, a);
}

template <int CPT>
inline void grouped_launch_heavy_c(const GroupedArgs &a, int parts, int hgrid, hipStream_t stream)
{
    const int words = pack_row_words(a.Zq, a.G, a.smap);
    const size_t lds = sizeof(uint32_t) * static_cast<size_t>(words);
    const int need = (words / 4 + kSampleBlock - 1) / kSampleBlock;
    if (a.smap) {
        grouped_launch_heavy_nq<CPT, 2, true>(a, parts, hgrid, lds, stream);
        return;
    }
    if (need <= 2) grouped_launch_heavy_nq<CPT, 2>(a, parts, hgrid, lds, stream);
    else if (need <= 5) grouped_launch_heavy_nq<CPT, 5>(a, parts, hgrid, lds, stream);
    else if (need <= 12) grouped_launch_heavy_nq<CPT, 12>(a, parts, hgrid, lds, stream);
    else grouped_launch_heavy_nq<CPT, 40>(a, parts, hgrid, lds, stream);
}


inline void grouped_launch_heavy(const GroupedArgs &a, int parts, int hgrid, int64_t mean, hipStream_t stream)
{
    if (parts <= 1 || hgrid == 0) return;
    switch (grouped_cpt(mean)) {
    case 1: grouped_launch_heavy_c<1>(a, parts, hgrid, stream); break;
    case 2: grouped_launch_heavy_c<2>(a, parts, hgrid, stream); break;
    default: grouped_launch_heavy_c<4>(a, parts, hgrid, stream); break;
    }
}

// Cars per thread by the mean bucket size (cars of this GPU / zones): 256 x 4 slots for ~1000 cars per zone, 256 x 2 and 256 x 1
// for smaller buckets (every slot runs Philox whether a car sits in it or not); larger buckets take the overflow rounds.
template <bool GROUPED>
inline void grouped_launch_sample(const GroupedArgs &a, int64_t mean, bool heavy_follows, hipStream_t stream)
{
    switch (heavy_follows ? grouped_cpt(mean) : grouped_cpt_wide(mean)) {
    case 1: grouped_launch_c<GROUPED, 1>(a, stream); break;
    case 2: grouped_launch_c<GROUPED, 2>(a, stream); break;
    case 4: grouped_launch_c<GROUPED, 4>(a, stream); break;
    default: grouped_launch_c<GROUPED, 6>(a, stream); break;
    }
}

// Diagnostic (cpm_debug_categorical): the categorical draw of the sampler for given 53-bit draws k against one
// installed row, through the same staging, search and exact-row code.  out[i] = destination (1-based), or 0 for a zero row.
// sparse tables (sp != null): Zq, G, Zc of the compact row, the row's normalised cells for the ties
__global__ __launch_bounds__(512) void k_pack_search_debug(const uint32_t *__restrict__ pack_g, const double *__restrict__ last_p,
                                                           const double *__restrict__ ckpt_t, const double *__restrict__ p_t, int origin, int Z, int Zq,
                                                           int G, int Zc, const double *__restrict__ sp, const uint32_t *__restrict__ sj, uint32_t scnt,
                                                           int64_t n, const uint64_t *__restrict__ k53, int64_t *__restrict__ out, int *__restrict__ n_exact)
{
    extern __shared__ uint32_t pack[];
    const int tid = threadIdx.x;
    const int gw = pack_guide_words(G), pieces = pack_row_words(Zq, G, sp ? 1 : 0) / 4, sh = 32 - G;
    {
        const int lane = tid & 63;
        for (int p0 = tid - lane; p0 < pieces; p0 += 512) {  // wave-uniform trips
            const int p = p0 + lane;
            if (p < pieces)
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint4 *>(pack_g) + p,
                                                 (__attribute__((address_space(3))) void *)(pack + 4 * p0), 16, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const uint16_t *guide = reinterpret_cast<const uint16_t *>(pack);
    const uint32_t *hi = pack + gw;
    const double last = *last_p;
    const uint32_t hi_last = hi[Zc - 1];
    const uint16_t *smap = sp ? reinterpret_cast<const uint16_t *>(hi + Zq) : nullptr;
    for (int64_t i0 = 0; i0 < n; i0 += 512) {  // wave-uniform trips (pack_search votes across the wave)
        const int64_t i = i0 + tid;
        const bool live = i < n;
        const uint64_t k = live ? k53[i] & ((1ull << 53) - 1ull) : 0ull;
        uint32_t dest[1];
        bool ok[1];
        const uint32_t khi[1] = {static_cast<uint32_t>(k >> 21)};
        const bool want[1] = {live && last != 0.0};
        pack_search<1>(guide, hi, khi, want, sh, hi_last, Zq, dest, ok, smap);
        if (!live) continue;
        if (last == 0.0) {
            out[i] = 0;
            continue;
        }
        if (!ok[0]) {
            if (sp) dest[0] = search_exact_sparse(sp, sj, scnt, static_cast<double>(k) * 0x1.0p-53, last);
            else dest[0] = search_exact_tables(ckpt_t, p_t, Z, origin, static_cast<double>(k) * 0x1.0p-53, last, khi[0] <= hi_last ? static_cast<int>(dest[0]) : -1);
            atomicAdd(n_exact, 1);
        }
        out[i] = static_cast<int64_t>(dest[0]) + 1;
    }
}

// perm[t][.] = the zones of table hour t, LARGEST bucket first: a counting sort by size class of parking[t][z] -- the bucket sizes of that
// hour in the run that wrote `parking` (the initial-value problem: the same tables a day earlier).  What k_grouped_hour<PERM> deals its
// sampler workgroups by; any permutation gives the same counts (the order inside a class is whatever the atomics make it).
__global__ __launch_bounds__(1024) void k_zone_order(const unsigned long long *__restrict__ parking, uint32_t *__restrict__ perm, int Z, uint32_t shift)
{
    __shared__ uint32_t bins[256];
    const int t = blockIdx.x, tid = threadIdx.x;
    const unsigned long long *n = parking + static_cast<size_t>(t) * Z;
    if (tid < 256) bins[tid] = 0;
    __syncthreads();
    auto key = [&](int z) {
        const uint32_t c = static_cast<uint32_t>(min(n[z] >> shift, 255ull));
        return 255u - c;
    };
    for (int z = tid; z < Z; z += 1024) atomicAdd(&bins[key(z)], 1u);
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (int k = 0; k < 256; ++k) {
            const uint32_t c = bins[k];
            bins[k] = run;
            run += c;
        }
    }
    __syncthreads();
    for (int z = tid; z < Z; z += 1024) {
        const uint32_t pos = atomicAdd(&bins[key(z)], 1u);
        perm[static_cast<size_t>(t) * Z + pos] = static_cast<uint32_t>(z);
    }
}

}  // namespace cpm
#include "cpm_day.h"  // the hours of a run in ONE launch (k_grouped_day), built from the bodies above
namespace cpm {

// ------------------------------------------------------------------------------------------------ workspace and driver
// run capacity: a quarter of a bucket region (= the mean bucket size at cap_mult 4), >= 64, whole 128-B lines
// ... an ODD number of 128-B lines: with 1,024 entries per run (S4k) the 32 runs of a zone lay 4 KiB apart and the same group's runs of
// consecutive zones 128 KiB apart -- powers of two, a sampler's flush and a placing block's 64 run heads all on the same few memory
// channels: 0.887 -> 0.858 ms per resample with one line of padding per run (same box, four interleaved runs each, +-0.001; on another
// box 0.874 -> 0.851: the unpadded form also flipped between 0.86 and 0.89 from process to process, the padded one does not)
inline uint32_t odd_lines(uint32_t words) { return ((words / 32u) & 1u) == 0u ? words + 32u : words; }
inline uint32_t grouped_scap(uint32_t cap) { return odd_lines((std::max<uint32_t>(64u, cap / 4) + 31u) / 32u * 32u); }
// packed driver = id | (dest - group * zpg) << idbits: the local destination takes ceil(log2 zpg) bits, at least one
inline uint32_t grouped_idbits(int Z, bool general = false)
{
    uint32_t b = 1;
    while ((1u << b) < grouped_zpg_of(Z, general)) ++b;
    return 32u - b;
}
inline uint32_t grouped_cap(int64_t n, int Z, int cap_mult)
{
    const int64_t mean = (n + Z - 1) / Z;
    // (an odd number of lines per bucket region as well -- 2,048-entry regions, Z = 8,192 x 500, lie 8 KiB apart -- was measured: no difference)
    return static_cast<uint32_t>((std::max<int64_t>(cap_mult * mean, 1024) + 63) / 64 * 64);
}

// true when the path can run this problem: the row pack fits LDS, ids fit the packed driver word, the bucket and run arrays fit
// their budget (3 id arrays of Z x cap x 4 B and Z x 32 x scap x 4 B of runs: <= 24 GiB, 80 GiB once grown)
inline bool grouped_path_fits(int64_t n, int Z, int cap_mult = 4)
{
    if (!pack_row_fits(Z) || n < 1 || n >= (int64_t(1) << 30)) return false;
    const uint32_t cap = grouped_cap(n, Z, cap_mult);
    if (n > (int64_t(1) << grouped_idbits(Z)) || static_cast<int>(grouped_zpg_of(Z)) > kMaxZonesPerGroup) return false;
    if (!place_shape_fits(Z)) return false;
    const int64_t bytes = static_cast<int64_t>(Z) * cap * 4 * 3 + 2 * static_cast<int64_t>(Z) * kGroups * grouped_scap(cap) * 4;
    return bytes <= (int64_t(cap_mult <= 4 ? 24 : 80) << 30);
}

struct GroupedWork {
    bool attrs_set = false;
    bool buckets0_valid = false;  // ids0 / cnt0 describe the context's current car state
    int64_t n = 0;
    int Z = 0, T = 0, nb0 = 0;
    int cap_mult = 4;        // bucket region = cap_mult x the mean bucket size; doubled by the context after an overflow (up to kMaxCapMult)
    int cap_mult_alloc = 0;  // what the arrays below were sized for
    uint32_t cap = 0, scap = 0, idbits = 0, gdiv = 0, zpg = 1;
    uint32_t *ids0 = nullptr, *idsA = nullptr, *idsB = nullptr;  // [Z*cap]: cached initial bucketing, ping-pong
    uint32_t *cnt0 = nullptr;                                    // [2][Z] the cached initial buckets: stayers (all cars after bucketing) | arrivals
    uint32_t *cnt = nullptr;                                     // [T+1][2][Z] per hour: stayers | arrivals of every bucket; then [T][chunks] the fused hour's hand-off counters
    uint32_t heavy_x_seen = 2;                                   // the heavy threshold (x a workgroup's slots) once heavy buckets were seen: the heavy launch runs
                                                                 // anyway then, and a bucket of 2-4 x the slots walked by ONE workgroup is the sampler's tail
    bool fused_ok = true;                                        // the fused hour is used (CPM_OPT_FUSED; cleared for good when a placing block gave up waiting)
    bool fused_auto = true;                                      // ... where it pays (fused_pays); false: wherever an instantiation exists (CPM_OPT_FUSED set by the caller)
    bool fused_pf = false;                                       // ... in its placing-first form (k_grouped_hour_pf): the previous hour's placing blocks, then the samplers
    bool fused_day = false;                                      // ... as ONE launch for all hours of a run but the last (k_grouped_day, cpm_day.h)
    int day_mix = 1;                                             // its block order: placing blocks among the sampler workgroups (1) or in front of them (0)
    GroupedArgs *day_hours = nullptr;                            // [T] the hours' arguments of a day launch (device memory, filled by k_grouped_zero)
    uint32_t *perm = nullptr;                                    // [T][Z] the zones of every table hour, largest bucket first (k_zone_order, behind every IVP)
    bool perm_valid = false;
    int use_perm = 2;  // 0 zone order, 1 largest-first, 2 (default) largest-first for sparse row packs only                   // ... written at least once since the arrays were allocated; CPM_OPT_ZONE_ORDER
    int fused_lag = 1 << 20;                                     // chunks of sampler workgroups between a chunk and its placing blocks; >= all chunks (default):
                                                                 // every sampler workgroup first, then every placing block
    uint32_t fused_spin = kFusedSpinLimit;
    uint32_t *Dq = nullptr;                                      // [Z][kGroups][scap] packed drivers
    uint32_t *cntg = nullptr;                                    // [Z][kGroups] run lengths
    unsigned long long *tt_part = nullptr;                       // [kTravelParts] partial travel-time sums, kept zero between resamples
    GroupedRare *rare = nullptr;                                 // what the rare branches of the hourly kernels read (written by k_grouped_zero per run)
    uint32_t *maxn = nullptr;                                    // [2] of the current run: largest heavy bucket (> kHeavy x the sampler workgroup's slots), most heavy buckets in one hour
    uint32_t *heavy_list = nullptr, *nheavy = nullptr;           // [kHeavyCap] zones handed to the heavy kernel this hour; [T+1] how many, per hour
    int run_hours = kDayRunCopies;                               // copies of Dq / cntg: 3 (rotating), or T when the runs of every hour of a resample are kept (ensure_history)
    int parts = 1;                                               // workgroups per heavy zone: 1 + blocks of the heavy kernel (set_parts)
    int hgrid = 0;                                               // zones the heavy launch covers
    const uint32_t *ivp_ids = nullptr, *ivp_cnt = nullptr;       // final buckets of the last IVP (grouped_commit_ivp)

    // destination groups of a run: general (any zones per group) for sparse row packs, power-of-two for dense ones (grouped_gdiv_of)
    void set_groups(bool general)
    {
        idbits = grouped_idbits(Z, general);
        gdiv = grouped_gdiv_of(Z, general);
        zpg = grouped_zpg_of(Z, general);
    }
    // what the last run saw -> how the next one is launched: enough workgroups per zone for the largest bucket, at most 32
    void set_parts(uint32_t largest_heavy_bucket, uint32_t most_heavy_buckets)
    {
        const int64_t slots = static_cast<int64_t>(grouped_cpt((n + Z - 1) / std::max(Z, 1))) * kSampleBlock;
        parts = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(32, (largest_heavy_bucket + slots - 1) / slots)));
        hgrid = parts > 1 ? static_cast<int>(std::min<int64_t>(kHeavyCap, most_heavy_buckets + most_heavy_buckets / 4 + 32)) : 0;  // (work items)
    }

    size_t fused_chunks() const { return static_cast<size_t>(std::max((Z + kFusedChunk - 1) / kFusedChunk, kGroups)); }  // (counter lines per hour: by chunk, or by group)
    size_t done_base() const { return (static_cast<size_t>(T + 1) * 2 * Z + kDoneStride - 1) / kDoneStride * kDoneStride; }  // (whole lines)
    size_t pdone_base() const { return done_base() + static_cast<size_t>(T) * fused_chunks() * kDoneStride; }  // (day launch: [T][kGroups] lines behind the chunk counters)
    size_t cnt_words() const { return pdone_base() + static_cast<size_t>(T + 1) * kGroups * kDoneStride; }
    size_t run_words() const { return static_cast<size_t>(Z) * kGroups * scap; }
    size_t len_words() const { return static_cast<size_t>(Z) * kGroups; }
    // Room for the runs of all T hours (travel times in one launch at the end of a resample), when they fit 24 GiB; else one copy.
    bool ensure_history()
    {
        if (run_hours == T) return true;
        if (run_words() * 4 * static_cast<size_t>(T) > (size_t(24) << 30)) return false;
        uint32_t *d = nullptr, *c = nullptr;
        if (hipMalloc(&d, sizeof(uint32_t) * run_words() * T) != hipSuccess) return false;
        if (hipMalloc(&c, sizeof(uint32_t) * len_words() * T) != hipSuccess) {
            (void)hipFree(d);
            return false;
        }
        (void)hipFree(Dq);
        (void)hipFree(cntg);
        Dq = d;
        cntg = c;
        run_hours = T;
        return true;
    }

    void release()
    {
        for (uint32_t **p : {&ids0, &idsA, &idsB, &cnt0, &cnt, &Dq, &cntg, &heavy_list, &nheavy}) {
            if (*p) (void)hipFree(*p);
            *p = nullptr;
        }
        if (tt_part) (void)hipFree(tt_part);
        tt_part = nullptr;
        if (maxn) (void)hipFree(maxn);
        maxn = nullptr;
        if (rare) (void)hipFree(rare);
        rare = nullptr;
        if (day_hours) (void)hipFree(day_hours);
        day_hours = nullptr;
        if (perm) (void)hipFree(perm);
        perm = nullptr;
        perm_valid = false;
        n = 0;
        run_hours = kDayRunCopies;
        buckets0_valid = false;
    }

    hipError_t ensure(int64_t n_, int Z_, int T_, int cu_count)
    {
        if (n_ == n && Z_ == Z && T_ == T && ids0 && cap_mult_alloc == cap_mult) return hipSuccess;
        release();
        n = n_;
        Z = Z_;
        T = T_;
        cap_mult_alloc = cap_mult;
        cap = grouped_cap(n, Z, cap_mult);
        scap = grouped_scap(cap);
        set_groups(false);

        nb0 = static_cast<int>(std::max<int64_t>({int64_t(1), std::min<int64_t>(2 * cu_count, (n + 4095) / 4096),
                                                  (n + int64_t(kBucketMaxPass) * 1024 - 1) / (int64_t(kBucketMaxPass) * 1024)}));
        hipError_t e = hipSuccess;
        auto alloc = [&](uint32_t **p, size_t words) {
            if (e == hipSuccess) e = hipMalloc(p, sizeof(uint32_t) * std::max<size_t>(words, 1));
        };
        const size_t slots = static_cast<size_t>(Z) * cap;
        alloc(&ids0, slots);
        alloc(&idsA, slots);
        alloc(&idsB, slots);
        alloc(&cnt0, 2 * static_cast<size_t>(Z));
        alloc(&cnt, cnt_words());
        alloc(&Dq, kDayRunCopies * static_cast<size_t>(Z) * kGroups * scap);  // (three hours: what the day launch rotates over, cpm_day.h; the hourly launches alternate between two)
        alloc(&cntg, kDayRunCopies * static_cast<size_t>(Z) * kGroups);
        run_hours = kDayRunCopies;
        alloc(&heavy_list, kHeavyCap);
        alloc(&nheavy, static_cast<size_t>(T) + 1);
        if (e == hipSuccess) e = hipMalloc(&tt_part, sizeof(unsigned long long) * kTravelParts);
        if (e == hipSuccess) e = hipMemset(tt_part, 0, sizeof(unsigned long long) * kTravelParts);
        if (e == hipSuccess) e = hipMalloc(&maxn, 2 * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc(&rare, sizeof(GroupedRare));
        if (e == hipSuccess) e = hipMalloc(&day_hours, sizeof(GroupedArgs) * static_cast<size_t>(std::max(T, 1)));
        if (e == hipSuccess) e = hipMalloc(&perm, sizeof(uint32_t) * static_cast<size_t>(std::max(T, 1)) * Z);
        if (e != hipSuccess) release();
        return e;
    }
};

struct GroupedTables {
    const uint32_t *rp;      // [T][Z][RW]
    const double *last;      // [T][Z]
    const long long *thr;    // [T][Z]
    const double *ckpt;      // [T][nck][Z] checkpoints of the running sums
    const double *p;         // [T][Z dest][Z origin] p_destin (reference layout)
    const double2 *tt;       // [T][Z][Z] travel table (k_build_travel_table) or nullptr
    const uint2 *tts_words = nullptr;   // its sparse rows (k_tts_*), when a row fits LDS: what the travel kernel then stages per (origin, hour)
    const uint32_t *tts_off = nullptr;
    const TravelCell *tts_cells = nullptr;
    int tts_W = 0;
    size_t tts_lds = 0;      // bytes of the largest row (words + cells)
    const uint32_t *tts_cnt = nullptr;  // fixed-stride travel rows (cpm_dataset.h): cells per row; the row's cells start at row * tts_stride
    uint32_t tts_stride = 0;
    int Z, Zp, Zq, T;
    int G = 0, Zc = 0, smap = 0;        // pack geometry: guide bits, entries in front of the pad, sparse packs (cpm_dataset.h: Zq, G, Zc of the compact row)
    const double *sp = nullptr;         // sparse tables: the rows' normalised cells / cells / counts (what a tie walks)
    const uint32_t *sj = nullptr;
    const uint32_t *scnt = nullptr;
    uint32_t scap = 0;
};

// Everything a run starts from zero, in ONE launch: the count tensor (driving counts, time sum and status word are added to),
// the largest-bucket words, the per-hour heavy counts and the per-hour bucket counts (the arrivals are added to; the stayers are
// stored by the sampler).  (Five hipMemsetAsync calls were five ~5 us fill kernels in the stream of every resample, 2.5 % of it.)
__global__ __launch_bounds__(256) void k_grouped_zero(unsigned long long *__restrict__ counts, size_t nwords, uint32_t *__restrict__ maxn,
                                                      uint32_t *__restrict__ nheavy, int nh, uint32_t *__restrict__ cnt, size_t ncnt,
                                                      GroupedRare *__restrict__ rare, GroupedRare rare_now, GroupedArgs *__restrict__ hours, GroupedDay day)
{
    const size_t stride = static_cast<size_t>(gridDim.x) * 256;
    for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < nwords; i += stride) counts[i] = 0ull;
    for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < ncnt; i += stride) cnt[i] = 0u;  // (the arrival counts are added to)
    if (blockIdx.x == 0) {
        if (threadIdx.x < 2) maxn[threadIdx.x] = 0u;
        if (threadIdx.x == 0) *rare = rare_now;  // (what the rare branches of this run's kernels read)
        for (int k = threadIdx.x; k < nh; k += 256) nheavy[k] = 0u;
        for (int k = threadIdx.x; k < day.nhours; k += 256) day_fill_hour(day, k, hours + k);  // (the arguments of a day launch's hours)
    }
}

// d_counts is zeroed here (k_grouped_zero), not by the caller.
// ivp == false: the T-hour resample from the state in d_zone0 (left unchanged).  On overflow bit 1 of the status word
//               (d_counts[2*T*Z+1]) is set and the counts are invalid.
// ivp == true : solveinitialvalueproblem (src/solveinitialvalueproblem.jl:8,53): T-1 hours, steps 0..T-2, every transition
//               applied.  d_zone0 is NOT modified: the new car-indexed state goes to d_zone0_out and the final buckets are
//               remembered in w.ivp_ids / w.ivp_cnt.  The caller inspects the status word when the stream has drained and then
//               either commits (grouped_commit_ivp + pointer swap) or repeats the IVP from the untouched d_zone0.
template <typename F1, typename F2>
int32_t grouped_run(GroupedWork &w, hipStream_t stream, const GroupedTables &tb, int64_t n, CarIndex cars, const uint32_t *d_zone0,
                    uint64_t seed, bool travel, int64_t *d_counts, int cu_count, F1 prof_begin, F2 prof_end, std::string &err, bool ivp = false,
                    uint32_t *d_zone0_out = nullptr)
{
    auto hip_fail = [&](hipError_t e, const char *what) {
        err = std::string(what) + ": " + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? CPM_ERR_NOMEM : CPM_ERR_HIP;
    };
    const int Z = tb.Z, T = tb.T;
    hipError_t e = w.ensure(n, Z, T, cu_count);
    if (e != hipSuccess) return hip_fail(e, "grouped zone workspace");
    w.set_groups(tb.smap != 0);
    const size_t lds_bins = sizeof(uint32_t) * static_cast<size_t>(Z);
    if (!w.attrs_set) {
        if (lds_bins > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_bucket_cars), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bins));
        w.attrs_set = true;
    }
    unsigned long long *parking = reinterpret_cast<unsigned long long *>(d_counts);
    unsigned long long *driving = parking + static_cast<size_t>(T) * Z;
    unsigned long long *tt_sum = parking + 2 * static_cast<size_t>(T) * Z;
    unsigned long long *status = tt_sum + 1;
    // travel times: from the runs, by one launch per hour -- or, when the runs of all hours fit, by one launch at the end
    const bool history = travel && !ivp && w.ensure_history();
    const int G = tb.G;
    const size_t rw = static_cast<size_t>(pack_row_words(tb.Zq, G, tb.smap));
    const int64_t mean = (n + Z - 1) / Z;
    const int hours = ivp ? T - 1 : T;
    // one launch for the hour (sampler workgroups + the placing blocks of their drivers) while no heavy bucket has been seen
    const bool shape = w.fused_ok && w.parts <= 1 && fused_shape_ok(Z, tb.Zq, G, tb.smap) && (!w.fused_auto || fused_pays(Z, tb.Zq, G, cu_count, tb.smap, mean));
    // ... and one launch for ALL hours that are applied (k_grouped_day): the IVP's T - 1, a resample's first T - 1 (hour T is sampled,
    // never applied: the plain form behind the placing of hour T - 1, k_grouped_hour_pf); hourly travel launches need hourly boundaries
    const int day_n = (shape && w.fused_day && w.cap < (1u << kCntXccShift) && (!travel || history)) ? (ivp ? hours : hours - 1) : 0;
    const int nchunk = (Z + kFusedChunk - 1) / kFusedChunk;
    GroupedDay day{};
    auto hour_base = [&](GroupedArgs &a) {  // what all hours of a run share
        a.rare = w.rare;
        a.Z = Z;
        a.Zq = tb.Zq;
        a.G = G;
        a.Zc = tb.Zc;
        a.smap = tb.smap;
        a.cap = w.cap;
        a.scap = w.scap;
        a.idbits = w.idbits;
        a.gdiv = w.gdiv;
        a.cars = cars;
        a.seed = seed;
        a.lag = w.fused_lag;
        a.heavy_x = w.parts > 1 ? w.heavy_x_seen : kHeavy;
        a.spin_limit = w.fused_spin;
        a.sdone = a.pdone = nullptr;
        a.psdone = nullptr;
        a.chained = 0;
        a.perm_t = nullptr;
    };
    // zones dealt largest-first (k_grouped_hour<PERM>): a list exists (an IVP has run on these arrays) and the run array fits 32-bit offsets
    const bool permute = (w.use_perm == 1 || (w.use_perm == 2 && tb.smap != 0)) && w.perm_valid && static_cast<uint64_t>(Z) * kGroups * w.scap * 4u < (1ull << 32);
    if (day_n >= 2) {
        std::memset(&day.base, 0, sizeof(day.base));
        hour_base(day.base);
        day.ids0 = w.ids0;
        day.cnt0 = w.cnt0;
        day.idsA = w.idsA;
        day.idsB = w.idsB;
        day.cnt = w.cnt;
        day.rp = tb.rp;
        day.last = tb.last;
        day.thr = tb.thr;
        day.Dq = w.Dq;
        day.cntg = w.cntg;
        day.parking = parking;
        day.driving = driving;
        day.sdone0 = w.cnt + w.done_base();
        day.pdone0 = w.cnt + w.pdone_base();
        day.rw = rw;
        day.run_words = w.run_words();
        day.len_words = w.len_words();
        day.sdone_stride = w.fused_chunks() * kDoneStride;
        day.pdone_stride = static_cast<size_t>(kGroups) * kDoneStride;
        day.copies = w.run_hours;
        day.nchunk = nchunk;
        day.nhours = day_n;
        day.step0 = static_cast<uint32_t>(ivp ? 0 : T - 1);
    }
    {
        const size_t nwords = 2 * static_cast<size_t>(T) * Z + 2;
        const unsigned zgrid = static_cast<unsigned>(std::min<size_t>((nwords + 255) / 256, 1024));
        GroupedRare rn{};
        rn.ckpt = tb.ckpt;
        rn.p = tb.p;
        rn.sp = tb.sp;
        rn.sj = tb.sj;
        rn.scnt = tb.scnt;
        rn.scap = tb.scap;
        rn.maxn = w.maxn;
        rn.heavy_list = w.heavy_list;
        rn.nheavy = w.nheavy;
        rn.status = status;
        rn.hgrid = static_cast<uint32_t>(w.hgrid);
        rn.parts = static_cast<uint32_t>(w.parts);
        rn.Z = Z;
        hipLaunchKernelGGL(k_grouped_zero, dim3(zgrid), dim3(256), 0, stream, parking, nwords, w.maxn, w.nheavy, T + 1, w.cnt, w.cnt_words(), w.rare, rn, w.day_hours, day);
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "zeroing the counters");
    }
    if (!w.buckets0_valid) {  // bucket the car-indexed state once; reused until the state changes
        const int64_t chunk = (n + w.nb0 - 1) / w.nb0;
        if ((e = hipMemsetAsync(w.cnt0, 0, sizeof(uint32_t) * 2 * Z, stream)) != hipSuccess) return hip_fail(e, "memset cnt0");
        hipLaunchKernelGGL(k_bucket_cars, dim3(w.nb0), dim3(kBucketBlock), lds_bins, stream, d_zone0, n, chunk, Z, w.cap, w.cnt0, w.ids0, status);
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "initial bucketing");
        w.buckets0_valid = true;
    }
    const uint32_t *ids = w.ids0, *cnt = w.cnt0;
    // placing first: the drivers of the hour before, still in their runs, wait for the next launch to move them into `ids` / `cnt`
    const uint32_t *pend_D = nullptr, *pend_cntg = nullptr;
    auto flush_pending = [&]() {  // ... or for a placing launch of their own, when that launch is not of the placing-first kind
        if (!pend_D) return;
        prof_begin(CPM_PROFILE_PLACE);
        grouped_launch_place(stream, pend_D, pend_cntg, static_cast<int>(w.zpg), Z, w.cap, w.scap, w.idbits, const_cast<uint32_t *>(cnt) + Z, const_cast<uint32_t *>(ids), status);
        prof_end(CPM_PROFILE_PLACE);
        pend_D = pend_cntg = nullptr;
    };
    int t_first = 0;
    if (day_n >= 2) {
        prof_begin(CPM_PROFILE_SAMPLER);
        grouped_launch_day(w.day_hours, day_n, Z, tb.Zq, G, tb.smap, static_cast<int>(w.zpg), nchunk, w.day_mix, mean, stream);
        prof_end(CPM_PROFILE_SAMPLER);
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "grouped zone day launch");
        // the drivers of its last hour are still in their runs: placed in front of hour T's sampler workgroups, or by a launch of their own
        t_first = day_n;
        ids = ((day_n - 1) & 1) ? w.idsB : w.idsA;
        cnt = w.cnt + static_cast<size_t>(day_n) * 2 * Z;
        pend_D = w.Dq + w.run_words() * static_cast<size_t>((day_n - 1) % w.run_hours);
        pend_cntg = w.cntg + w.len_words() * static_cast<size_t>((day_n - 1) % w.run_hours);
    }
    for (int t = t_first; t < hours; ++t) {
        const uint32_t step = static_cast<uint32_t>(ivp ? t : T - 1 + t);
        // hour T of a resample is sampled, never applied (src/resampling.jl:81-83): counts only -- unless its travel times are wanted,
        // which are computed from the runs
        const bool last_hour = !ivp && t + 1 == T;
        const bool grouped = !last_hour || travel || w.parts > 1;  // (heavy buckets: the grouped form splits them over workgroups, the plain one walks them)
        uint32_t *cnt_next = w.cnt + static_cast<size_t>(t + 1) * 2 * Z;  // stayers; the arrivals Z words behind
        uint32_t *ids_next = (t & 1) ? w.idsB : w.idsA;
        GroupedArgs a;
        hour_base(a);
        a.ids = ids;
        a.cnt_s = cnt;
        a.cnt_a = cnt + Z;
        a.rp_t = tb.rp + static_cast<size_t>(t) * Z * rw;
        a.last_t = tb.last + static_cast<size_t>(t) * Z;
        a.thr_t = tb.thr + static_cast<size_t>(t) * Z;
        a.hour = t;
        a.ids_next = ids_next;
        a.cnt_next = cnt_next;
        // (the hourly launches alternate between TWO copies: rotating over the three the day launch needs touched half as much memory
        //  again per resample and cost the headline 3 %)
        const int copy = history ? t : (day_n >= 2 ? t % w.run_hours : (t & 1));  // (behind a day launch its rotation carries on: the pending runs lie in copy (t - 1) % 3)
        a.D = w.Dq + w.run_words() * copy;
        a.cntg = w.cntg + w.len_words() * copy;
        a.parking_t = parking + static_cast<size_t>(t) * Z;
        a.driving_t = driving + static_cast<size_t>(t) * Z;
        a.step = step;
        if (permute && t < T - 1) a.perm_t = w.perm + static_cast<size_t>(t) * Z;
        const bool after_day = day_n >= 2 && t == t_first;  // (the hour behind a day launch: its placing rides in front, whatever the context's own form)
        const bool pf = shape && (w.fused_pf || after_day);  // (also the last hour, in its plain form: the placing of the hour before it rides in front)
        const bool fuse = grouped && !last_hour && shape && !pf;
        if (!pf) flush_pending();
        a.pD = pend_D;
        a.pcntg = pend_cntg;
        a.pchunks = pend_D ? nchunk : 0;
        pend_D = pend_cntg = nullptr;
        // hand-off counters of the hour: by chunk (k_grouped_hour) or by group (k_grouped_hour_pf; behind a day launch: a segment of their own)
        a.done_t = after_day ? w.cnt + w.pdone_base() + static_cast<size_t>(t) * kGroups * kDoneStride
                             : w.cnt + w.done_base() + static_cast<size_t>(t) * w.fused_chunks() * kDoneStride;
        prof_begin(CPM_PROFILE_SAMPLER);
        if (pf && grouped) grouped_launch_hour_pf<true>(a, mean, stream);
        else if (pf) grouped_launch_hour_pf<false>(a, mean, stream);
        else if (fuse) grouped_launch_hour(a, mean, stream);
        else if (grouped) grouped_launch_sample<true>(a, mean, w.parts > 1, stream);
        else grouped_launch_sample<false>(a, mean, w.parts > 1, stream);
        prof_end(CPM_PROFILE_SAMPLER);
        if (grouped && !fuse && !pf) grouped_launch_heavy(a, w.parts, w.hgrid, mean, stream);
        if (!last_hour) {
            if (pf) {
                pend_D = a.D;
                pend_cntg = a.cntg;
            } else if (!fuse) {
                prof_begin(CPM_PROFILE_PLACE);
                grouped_launch_place(stream, a.D, a.cntg, static_cast<int>(w.zpg), Z, w.cap, w.scap, w.idbits, cnt_next + Z, ids_next, status);
                prof_end(CPM_PROFILE_PLACE);
            }
            ids = ids_next;
            cnt = cnt_next;
        }
        if (travel && grouped && !history) {
            TravelArgs tr{};
            tr.tt = tb.tt;
            tr.tts_words = tb.tts_words;
            tr.tts_off = tb.tts_stride ? tb.tts_cnt : tb.tts_off;
            tr.tts_stride = tb.tts_stride;
            tr.tts_cells = tb.tts_cells;
            tr.W = tb.tts_W;
            tr.tt_part = w.tt_part;
            tr.d_stride = tr.c_stride = 0;
            tr.t0 = t;
            tr.zpg = static_cast<int>(w.zpg);
            tr.step0 = step;
            tr.cars = cars;
            tr.seed = seed;
            prof_begin(CPM_PROFILE_TRAVEL);
            tr.list_off = static_cast<uint32_t>(tb.tts_lds);
            if (tb.tts_words) launch(k_grouped_travel<true>, dim3(Z, 1), dim3(travel_block(mean, w.parts > 1)), travel_lds_bytes(tb.tts_lds, travel_block(mean, w.parts > 1)), stream, a.D, a.cntg, Z, w.scap, w.idbits, tr);
            else launch(k_grouped_travel<false>, dim3(Z, 1), dim3(travel_block(mean, w.parts > 1)), 0, stream, a.D, a.cntg, Z, w.scap, w.idbits, tr);
            prof_end(CPM_PROFILE_TRAVEL);
        }
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "grouped zone hour launch");
    }
    if (history) {  // every hour's drivers are still in their runs: one launch
        TravelArgs tr{};
        tr.tt = tb.tt;
        tr.tts_words = tb.tts_words;
        tr.tts_off = tb.tts_stride ? tb.tts_cnt : tb.tts_off;
        tr.tts_stride = tb.tts_stride;
        tr.tts_cells = tb.tts_cells;
        tr.W = tb.tts_W;
        tr.tt_part = w.tt_part;
        tr.d_stride = w.run_words();
        tr.c_stride = w.len_words();
        tr.t0 = 0;
        tr.zpg = static_cast<int>(w.zpg);
        tr.step0 = static_cast<uint32_t>(T - 1);
        tr.cars = cars;
        tr.seed = seed;
        prof_begin(CPM_PROFILE_TRAVEL);
        tr.list_off = static_cast<uint32_t>(tb.tts_lds);
        if (tb.tts_words) launch(k_grouped_travel<true>, dim3(Z, T), dim3(travel_block(mean, w.parts > 1)), travel_lds_bytes(tb.tts_lds, travel_block(mean, w.parts > 1)), stream, w.Dq, w.cntg, Z, w.scap, w.idbits, tr);
        else launch(k_grouped_travel<false>, dim3(Z, T), dim3(travel_block(mean, w.parts > 1)), 0, stream, w.Dq, w.cntg, Z, w.scap, w.idbits, tr);
        prof_end(CPM_PROFILE_TRAVEL);
    }
    if (travel && !ivp) {  // the partial sums of k_grouped_travel -> the sum word of the count tensor
        hipLaunchKernelGGL(k_grouped_travel_finish, dim3(1), dim3(kTravelParts), 0, stream, w.tt_part, tt_sum);
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "travel-time sum");
    }
    flush_pending();  // (an IVP ends on a placing: its final buckets are read below)
    if (ivp && T >= 2) {  // the zones of every table hour, largest first, for the runs that follow (a hint: any order gives the same counts)
        uint32_t shift = 0;
        while ((static_cast<uint64_t>(4 * std::max<int64_t>(mean, 1)) >> shift) > 255) ++shift;
        hipLaunchKernelGGL(k_zone_order, dim3(T - 1), dim3(1024), 0, stream, parking, w.perm, Z, shift);
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "zone order");
        w.perm_valid = true;
    }
    if (ivp) {
        hipLaunchKernelGGL(k_unbucket, dim3(Z), dim3(256), 0, stream, ids, const_cast<uint32_t *>(cnt), cnt + Z, w.cap, d_zone0_out, static_cast<uint32_t>(n), status);
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "unbucket");
        w.ivp_ids = ids;
        w.ivp_cnt = cnt;
        w.buckets0_valid = false;  // until grouped_commit_ivp
    }
    return CPM_OK;
}

// After a verified IVP: its final buckets become the cached bucketing of the (new) current state.
inline hipError_t grouped_commit_ivp(GroupedWork &w, hipStream_t stream)
{
    hipError_t e = hipSuccess;
    if (w.ivp_ids != w.ids0) e = hipMemcpyAsync(w.ids0, w.ivp_ids, sizeof(uint32_t) * static_cast<size_t>(w.Z) * w.cap, hipMemcpyDeviceToDevice, stream);
    if (e == hipSuccess && w.ivp_cnt != w.cnt0) e = hipMemcpyAsync(w.cnt0, w.ivp_cnt, sizeof(uint32_t) * 2 * w.Z, hipMemcpyDeviceToDevice, stream);
    if (e == hipSuccess) w.buckets0_valid = true;
    return e;
}

}  // namespace cpm
