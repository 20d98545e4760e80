// cpm_zone6_kernels.h -- second generation of the grouped zone path (CPM_KERNEL_ZONE_GROUPED):
//   * the row a workgroup stages is the HIGH WORD of the CDF row, 4 B per destination instead of 8, preceded by a
//     GUIDE table (cut-point method) that brackets a draw's answer, so that a draw costs a few LDS probes instead of the
//     12-level tree walk of the first generation;
//   * the drivers of an origin zone go into FIXED-SIZE runs, one per destination group, so their position is
//     known the moment their rank is (no scan, no second pass over the cars, no offsets for the placing
//     kernel to fetch before it can fetch the runs).
// Why: measured with ablations (tools/micro/sampler_bench.hip, profiles/round1_notes.md) the sampler is bound by
// VALU issue (~650 vector instructions per thread and zone, ~4 cycles each), not by HBM (a kernel that only streams
// the same bytes takes 14.6 us against 40) and not by latency (prefetching the next zone's registers in resident
// workgroups changed nothing).  So the layout in HBM is chosen to make the instructions few: the row arrives in the
// order it is used in (a straight 16-B copy into LDS, no per-element index arithmetic) and the search is short.
//
// High-word rows.  The categorical draw of src/resampling.jl:38-45 is "first j with u <= cdf[j]"
// with u = k * 2^-53, k the 53 high bits of two Philox words.  Let hi[j] = floor(cdf[j] * 2^32)
// (0xFFFFFFFF when cdf[j] >= 1) and khi = floor(u * 2^32) = the high Philox word.  Then
//      hi[j]   > khi  =>  cdf[j]   * 2^32 >= khi + 1 > u * 2^32      =>  cdf[j]   >= u  (not cdf[j] < u)
//      hi[j-1] < khi  =>  cdf[j-1] * 2^32 <  hi[j-1] + 1 <= u * 2^32  =>  cdf[j-1] <  u
// so the first j with hi[j] >= khi IS the reference's answer whenever hi[j] > khi strictly.  When
// hi[j] == khi (probability ~ Z * 2^-32 per draw), or no hi[j] >= khi exists (u above the row total),
// the car repeats the search on the f64 row in HBM with exactly the code of the other kernels
// (clamp_u + lower_bound_row).  u == 0 needs no special case: khi = 0, the scan stops at the first j,
// hi[0] > 0 is accepted (cdf[0] > 0: the clamped answer), hi[0] == 0 is a tie.  Results are bit-identical
// to the f64 search by construction; tests/test_gpu_parity.py drives the tie and out-of-range branches
// through cpm_debug_categorical.
// Guide.  guide[m] = first j with hi[j] >= m << (32 - G), m = 0 .. 2^G (u16, clamped to Z - 1; G = guide bits = ceil(log2 Z) - 2,
// a quarter of an entry per destination).  The answer of a draw lies in [guide[m], guide[m+1]], m = khi >> (32 - G): every
// j' < guide[m] has hi[j'] < m << (32-G) <= khi, and hi[guide[m+1]] >= (m+1) << (32-G) > khi; a lower bound inside that bracket
// finds it.  Draws with khi above the row's last high word never search (they take the exact fallback).
// A row pack = [2^G + 8 u16 guide][Zq u32 hi] (Zq = Z rounded up to 32, padded with 0xFFFFFFFF), built next
// to the CDF; HBM traffic of an hourly launch: Z * (2^G * 2 + Zq * 4) B of rows instead of Z * Zp * 8.
//
// Fixed-size runs.  Zone z's drivers of destination group g are written at D[(z*kGroups + g)*scap + rank],
// (groups of 2^gshift consecutive zones), rank from an LDS atomic, as id | (dest mod 2^gshift) << idbits (4 B; n <= 2^idbits).  The 32 run
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
constexpr int kMaxZonesPerGroup6 = 1024;
constexpr uint32_t kHiMax = 0xFFFFFFFFu;

// row pack geometry: guide words (two u16 entries per word), then Zq high words
__host__ __device__ inline int pack_guide_bits(int Z)
{
    int g = 3;  // >= 8 entries: the guide is a whole number of 16-B pieces
    while ((1 << g) < Z) ++g;
    // a quarter of an entry per destination: measured best at S4k (entries per destination 1: 32.0 us, 1/2: 29.9, 1/4: 29.1 --
    // the shorter pack (18 instead of 24 KiB of HBM and LDS per zone) outweighs the ~2 extra probes of a draw)
#ifndef CPM_GUIDE_SHIFT
#define CPM_GUIDE_SHIFT 2
#endif
    g = g - CPM_GUIDE_SHIFT < 3 ? 3 : g - CPM_GUIDE_SHIFT;
    return g;
}
// high words per row: Z, then at least 31 entries of 0xFFFFFFFF (the unclamped stride walk of pack_search reads up to 30 past
// its bracket), a whole number of 128-B lines
__host__ __device__ inline int pack_zq(int Z) { return (Z + 31 + 31) / 32 * 32; }
__host__ __device__ inline int pack_guide_words(int G) { return (1 << G) / 2 + 4; }  // 2^G + 1 entries used (+7 pad: whole 16-B pieces)
__host__ __device__ inline int pack_row_words(int Zq, int G)  // at least 1 KiB: one whole LDS-DMA wave-instruction
{
    const int w = pack_guide_words(G) + Zq;
    return w < 256 ? 256 : w;
}

// hi part of every row pack and last[t][o], from the canonical CDF (one thread per element, coalesced both ways)
__global__ __launch_bounds__(256) void k_build_hi32(const double *__restrict__ cdf, uint32_t *__restrict__ rp,
                                                    double *__restrict__ last, int Z, int Zp, int Zq, int G, int64_t rows)
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
    rp[row * pack_row_words(Zq, G) + pack_guide_words(G) + j] = h;
}

// thr[t][z] = bernoulli_threshold(p_drive[t][z]): the integer the sampler compares the 53-bit draw with (one scalar load per
// workgroup instead of f64 arithmetic in every thread)
__global__ __launch_bounds__(256) void k_build_thr(const double *__restrict__ pdrive, long long *__restrict__ thr, int64_t n)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) thr[i] = bernoulli_threshold(pdrive[i]);
}

// guide part: guide[m] = min(first j in [0, Z) with hi[j] >= m << (32 - G), Z - 1), m = 0 .. 2^G (entry 2^G and the pad: Z - 1)
__global__ __launch_bounds__(256) void k_build_guide(uint32_t *__restrict__ rp, int Z, int Zq, int G, int64_t rows)
{
    const int64_t row = blockIdx.y + static_cast<int64_t>(blockIdx.z) * gridDim.y;
    if (row >= rows) return;
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= (1 << G) + 8) return;
    uint32_t *pack = rp + row * pack_row_words(Zq, G);
    const uint32_t *hi = pack + pack_guide_words(G);
    int lo = Z - 1;
    if (m < (1 << G)) {
        const uint32_t edge = static_cast<uint32_t>(m) << (32 - G);
        int n = Z;
        lo = 0;
        while (n > 0) {
            const int half = n >> 1;
            if (hi[lo + half] < edge) {
                lo += half + 1;
                n -= half + 1;
            } else {
                n = half;
            }
        }
        lo = min(lo, Z - 1);
    }
    reinterpret_cast<uint16_t *>(pack)[m] = static_cast<uint16_t>(lo);
}

struct Zone6Args {
    const uint32_t *ids;      // [Z*cap] this hour's buckets
    const uint32_t *cnt;      // [Z] their sizes
    const uint32_t *rp_t;     // [Z][RW] row packs of this hour: guide, then the high words of the CDF row
    const double *last_t;     // [Z] row totals (f64)
    const long long *thr_t;   // [Z] Bernoulli thresholds floor(p_drive * 2^53) of this hour (bernoulli_threshold), or nullptr
    const double *pdrive_t, *cdf_t, *dm;
    uint32_t *ids_next;       // [Z*cap]       (grouped)
    uint32_t *cnt_next;       // [Z] stayers   (grouped; k_zone6_place adds the arrivals)
    uint32_t *D;              // [Z][kGroups6][scap] packed drivers (grouped)
    uint32_t *cntg;           // [Z][kGroups6] run lengths (grouped)
    uint32_t *rec_out;        // [Z*cap] dest | drive << 31 per slot (plain form)
    unsigned long long *parking_t, *driving_t, *tt_sum, *status;
    int Z, Zp, Zq, G, T, t;  // G: guide bits; RW = 2^G / 2 + Zq words per row pack
    uint32_t cap, scap, idbits, step, gshift;  // destination group = dest >> gshift (2^gshift zones per group, at most 32 groups)
    int64_t car_begin;
    uint64_t seed;
    unsigned long long *stamps;  // CPM_DIAGNOSTIC builds only: [Z][8] s_memtime stamps of wave 0 (tools/micro/sampler_bench.hip)
    int abl;  // CPM_DIAGNOSTIC builds only (results WRONG): 1 no Philox, 2 no search, 16 synthetic ids, 32 no search / slot taking at all
              // (everybody stays)
};

#ifdef CPM_DIAGNOSTIC
#define CPM_ABL(a, bit) ((a).abl & (bit))
#define CPM_STAMP(a, z, k)                                                                    \
    do {                                                                                       \
        if ((a).stamps && threadIdx.x == 0) (a).stamps[static_cast<size_t>(z) * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define CPM_ABL(a, bit) 0
#define CPM_STAMP(a, z, k) \
    do {                   \
    } while (0)
#endif

// Row pack -> LDS by LDS-DMA (global_load_lds_dwordx4): no VGPR destination, no ds_write; one wave-instruction moves
// 64 x 16 B to 1 KiB of consecutive LDS.  The destination is wave-uniform base + lane x 16, the source is per lane.
// Counts in vmcnt like any load.  NQ (compile time) >= pieces / BLOCK wave-instructions per wave: with a static count
// the compiler can wait for the OLDER id loads alone (s_waitcnt vmcnt(NQ)) and run Philox while the pack is landing.
template <int BLOCK, int NQ>
__device__ __forceinline__ void pack_dma(uint32_t *lds, const uint32_t *pack, int pieces, int tid)
{
    // Every wave issues exactly NQ instructions, unpredicated (so that the count is known at compile time): chunk k
    // = 64 pieces from min(64 k, pieces - 64); chunks past the end repeat the last one (same bytes to the same LDS words).
    // pieces >= 64: a pack is at least 1 KiB (pack_row_words).
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int m = 0; m < NQ; ++m) {
        const int p0 = min((m * (BLOCK / 64) + wave) * 64, pieces - 64);
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint4 *>(pack) + p0 + lane,
                                         (__attribute__((address_space(3))) void *)(lds + 4 * p0), 16, 0, 0);
    }
}

// s_waitcnt vmcnt(NQ) carrying the id registers as in/out operands (see k_zone6_sample)
template <int N, int NQ>
__device__ __forceinline__ void wait_ids(uint32_t (&id)[N])
{
    static_assert((N == 2 || N == 3 || N == 5) && NQ <= 63, "written for CPT = 1, 2, 4");
    if constexpr (N == 5)
        asm volatile("s_waitcnt vmcnt(%5)" : "+v"(id[0]), "+v"(id[1]), "+v"(id[2]), "+v"(id[3]), "+v"(id[4]) : "n"(NQ) : "memory");
    else if constexpr (N == 3)
        asm volatile("s_waitcnt vmcnt(%3)" : "+v"(id[0]), "+v"(id[1]), "+v"(id[2]) : "n"(NQ) : "memory");
    else
        asm volatile("s_waitcnt vmcnt(%2)" : "+v"(id[0]), "+v"(id[1]) : "n"(NQ) : "memory");
}

// The f64 search of the other kernels, on the row where it lies in HBM (rare: ties and u above the row total).
__device__ __noinline__ uint32_t search_exact_row(const double *__restrict__ cdf_row, int Z, double uc, double last)
{
    return static_cast<uint32_t>(lower_bound_row(cdf_row, Z, clamp_u(uc, last)));
}

// CPT draws against the staged pack, in lockstep (CPT independent LDS reads in flight per step):
// dest[c] = first j with hi[j] >= khi[c], ok[c] = the answer is certain (hi[dest] > khi).  want[c] == false: no search.
// The guide brackets the answer: j in [L, L + n], L = guide[m], n = guide[m+1] - L, m = khi >> sh (entries are clamped to Z-1
// and a draw above the row's last high word never searches, so the bracket is in range).  hi[] is non-decreasing over the WHOLE
// row and padded with 0xFFFFFFFF for >= 31 entries past Z-1 (Zq), so the lower bound inside the bracket is a descending-stride
// walk that needs no upper clamp: with 2^K > n,  for s = 2^(K-1) .. 1:  if (hi[L + o + s - 1] < khi) o += s  ends at o = the
// number of entries from L on that lie below khi = the answer's offset (<= n <= 2^K - 1; the largest index read is L + 2^K - 2).
// K is wave-uniform (the widest bracket among the wave's draws decides): three to five steps of
// {LDS read, compare, select, add} on the dense synthetic rows.  Brackets of 32 entries and more (rows with long runs of
// zero-probability zones) take the same walk from a larger K with the probe index clamped to the row.
typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
typedef __attribute__((address_space(3))) const uint16_t lds_cu16;

template <int CPT>
__device__ __forceinline__ void pack_search(const uint16_t *guide_g, const uint32_t *hi_g, const uint32_t (&khi)[CPT], const bool (&want)[CPT],
                                            int sh, uint32_t hi_last, int Zq, uint32_t (&dest)[CPT], bool (&ok)[CPT])
{
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
    for (int c = 0; c < CPT; ++c) {
        n[c] = in[c] ? n[c] - lo[c] : 0u;
        nor |= n[c];
    }
    if (__builtin_expect(__any(nor >= 32u), 0)) {  // wave-uniform
        int K = 6;
        while (__any((nor >> K) != 0u)) ++K;
        for (uint32_t s = 1u << (K - 1); s != 0u; s >>= 1) {
            uint32_t v[CPT];
#pragma unroll
            for (int c = 0; c < CPT; ++c) v[c] = hi[min(lo[c] + s - 1u, static_cast<uint32_t>(Zq - 1))];
#pragma unroll
            for (int c = 0; c < CPT; ++c) lo[c] += (v[c] < kk[c]) ? s : 0u;
        }
    } else {
        lds_cu32 *p[CPT];
#pragma unroll
        for (int c = 0; c < CPT; ++c) p[c] = hi + lo[c];
#define CPM_PACK_STEP(S)                                                    \
    do {                                                                    \
        uint32_t v_[CPT];                                                   \
        _Pragma("unroll") for (int c = 0; c < CPT; ++c) v_[c] = p[c][(S) - 1]; \
        _Pragma("unroll") for (int c = 0; c < CPT; ++c) p[c] += (v_[c] < kk[c]) ? (S) : 0; \
    } while (0)
        const bool a16 = __any(nor >= 16u);
        const bool a8 = a16 || __any(nor >= 8u);
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
        return;
    }
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        dest[c] = lo[c];
        ok[c] = in[c] & (hi[lo[c]] > kk[c]);
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

// One workgroup per origin zone.  (Resident workgroups each walking several zones, with or without the next zone's
// registers prefetched, were measured no faster; profiles/round1_notes.md.)
// GROUPED: stayers compacted into next hour's bucket of the zone, drivers into the zone's fixed-size runs.  The drivers
// of a zone are first ranked and staged in LDS (kStage6 entries per group) and written out by 16 lanes per run,
// 64 B at a time; ranks beyond kStage6 go to HBM directly.
// !GROUPED: dest | drive << 31 per slot (the last hour of a resample: counts only).
// Order of a workgroup's life (in-kernel stamps, S4k: loads 3.5 us, Philox 1.0, search 2.3, emit 1.2, rest 1.1 before
// this ordering): the bucket size, the ids and the row pack are requested together (the id loads are clamped to the
// zone's REGION, not to its size, so they do not wait for the size); Philox runs while the pack is landing; after the
// barrier the CPT cars of a thread search in lockstep and take their slots with one stayer ticket per wave and CPT
// rank atomics in flight together.
#ifndef CPM_STAGE6
#define CPM_STAGE6 32
#endif
constexpr int kStage6 = CPM_STAGE6;

// Waves per SIMD the compiler must leave room for.  LDS admits 7 of these workgroups per CU at S4k (28 waves = 7 per SIMD);
// asking for 8 squeezed the kernel into 94 SGPRs with 31 of them spilled to VGPR lanes (v_readlane / v_writelane on the
// VALU this kernel is short of): measured 30.4 us at 8, 29.7 at 7, 28.2 at 6 (106 SGPRs, 7 waves per SIMD still fit).
#ifndef CPM_WPS
#define CPM_WPS 6
#endif
template <bool TRAVEL, int BLOCK, int CPT, int NQ, bool GROUPED>
__global__ __launch_bounds__(BLOCK, TRAVEL ? 2 : CPM_WPS) void k_zone6_sample(Zone6Args a)
{
    extern __shared__ uint32_t pack[];  // the zone's row pack: guide (u16), then Zq high words
    __shared__ uint32_t s_ndrive, s_nstay;
    __shared__ unsigned long long s_tt;
    __shared__ uint32_t gb[kGroups6];
    __shared__ uint32_t stage[GROUPED ? kGroups6 * kStage6 : 1];
    const int Z = a.Z;
    const int z = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t cap = a.cap;
    const uint32_t b = static_cast<uint32_t>(z) * cap;
    const int gw = pack_guide_words(a.G), rw = pack_row_words(a.Zq, a.G), pieces = rw / 4, sh = 32 - a.G;
    CPM_STAMP(a, z, 0);
    // The id loads are written in assembly and waited for by hand: with LDS-DMA in flight hipcc (ROCm 7.2) drains
    // vmcnt to 0 at the first use of any ordinary load result, which would put Philox behind the whole pack.  Here the
    // wave issues CPT + 1 id loads, then exactly NQ LDS-DMA instructions; vmcnt retires in order, so vmcnt <= NQ means the
    // ids are in their registers.  (The waitcnt statement names the ids as in/out operands: no use can move above it.)
    const uint32_t n_raw = a.cnt[z];  // (scalar loads: requested before the statements below fence memory operations)
    const double last = a.last_t[z];
    const long long thr = a.thr_t[z];  // scalar too: a vector load here would be waited for with vmcnt(0), i.e. behind the whole pack
    uint32_t id[CPT + 1];
#pragma unroll
    for (int c = 0; c <= CPT; ++c) {
        const uint32_t *src = a.ids + b + min(static_cast<uint32_t>(tid + c * BLOCK), cap - 1);
        asm volatile("global_load_dword %0, %1, off" : "=v"(id[c]) : "v"(src) : "memory");
    }
    pack_dma<BLOCK, NQ>(pack, a.rp_t + static_cast<size_t>(z) * rw, pieces, tid);
    wait_ids<CPT + 1, NQ>(id);
    const uint32_t n = min(n_raw, cap);
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
    // Philox of the register-resident cars: needs the ids only
    bool valid[CPT], drive[CPT], want[CPT], ok[CPT];
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
        drive[c] = valid[c] & (kb <= thr);
        want[c] = drive[c] & (last != 0.0);  // stays, or zero row: destination = origin (:35-36)
    }
    CPM_STAMP(a, z, 1);
    // This wave's pieces of the pack have landed (LDS-DMA counts in vmcnt; s_barrier itself waits for no counter), then the
    // barrier makes every wave's pieces visible to every wave.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    CPM_STAMP(a, z, 2);
    const uint16_t *guide = reinterpret_cast<const uint16_t *>(pack);
    const uint32_t *hi = pack + gw;
    const uint32_t hi_last = hi[Z - 1];
    const double *cdf_row = a.cdf_t + static_cast<size_t>(z) * a.Zp;
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t nd = 0;
    long long tt = 0;
    uint32_t *stay_out = GROUPED ? a.ids_next + static_cast<size_t>(z) * cap : nullptr;
    uint32_t *runs = GROUPED ? a.D + static_cast<size_t>(z) * kGroups6 * a.scap : nullptr;
    CPM_STAMP(a, z, 3);
    if (!CPM_ABL(a, 32)) {
        if (CPM_ABL(a, 2)) {
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                dest[c] = (khi[c] >> 8) % static_cast<uint32_t>(Z);
                ok[c] = true;
            }
        } else {
            pack_search<CPT>(guide, hi, khi, want, sh, hi_last, a.Zq, dest, ok);
        }
        bool anyx = false;
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            dest[c] = want[c] ? dest[c] : static_cast<uint32_t>(z);
            anyx |= want[c] & !ok[c];
        }
        if (__builtin_expect(__any(anyx), 0)) {  // ties and draws above the row total: the f64 row in HBM (wave-uniform, rare)
#pragma unroll
            for (int c = 0; c < CPT; ++c)
                if (want[c] & !ok[c]) dest[c] = search_exact_row(cdf_row, Z, u53(clo[c], khi[c]), last);
        }
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            if (drive[c]) {
                if (!GROUPED) ++nd;  // (grouped: drivers = bucket size - stayers)
                if (TRAVEL) tt += travel_time_q16(a.dm, Z, a.T, a.t, z, dest[c], a.seed, static_cast<uint64_t>(a.car_begin) + id[c], a.step);
            }
        }
        CPM_STAMP(a, z, 4);
        if (GROUPED) {
            // stayers: one ticket per wave for all its CPT slots
            unsigned long long mS[CPT];
            uint32_t total = 0;
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                mS[c] = __ballot(valid[c] && !drive[c]);
                total += static_cast<uint32_t>(__popcll(mS[c]));
            }
            uint32_t bS = 0;
            if (lane == 0 && total) bS = atomicAdd(&s_nstay, total);
            bS = __shfl(bS, 0, 64);
            // drivers: CPT rank atomics in flight together
            uint32_t rank[CPT];
#pragma unroll
            for (int c = 0; c < CPT; ++c) rank[c] = drive[c] ? atomicAdd(&gb[dest[c] >> a.gshift], 1u) : 0u;
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                if (valid[c] && !drive[c]) stay_out[bS + static_cast<uint32_t>(__popcll(mS[c] & below))] = id[c];
                bS += static_cast<uint32_t>(__popcll(mS[c]));
            }
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                if (drive[c]) {
                    const uint32_t g = dest[c] >> a.gshift;
                    const uint32_t packed = id[c] | ((dest[c] & ((1u << a.gshift) - 1u)) << a.idbits);
                    if (rank[c] < static_cast<uint32_t>(kStage6)) stage[g * kStage6 + rank[c]] = packed;
                    else if (rank[c] < a.scap) runs[g * a.scap + rank[c]] = packed;
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < CPT; ++c)
                if (valid[c]) a.rec_out[b + tid + c * BLOCK] = dest[c] | (drive[c] ? kDriveBit : 0u);
        }
    } else if (tid == 0) {
        s_nstay = n;
    }
    for (uint32_t q0 = CPT * BLOCK; q0 < n && !CPM_ABL(a, 32); q0 += BLOCK) {  // buckets larger than CPT*BLOCK cars (wave-uniform trips)
        if (q0 + static_cast<uint32_t>(tid & ~63) >= n) continue;  // none of this wave's 64 slots holds a car (no barrier inside the loop)
        const uint32_t q = q0 + tid;
        const bool valid1 = q < n;
        const uint32_t idx = (q0 == CPT * BLOCK) ? id[CPT] : (valid1 ? a.ids[b + q] : 0u);
        const uint64_t car = static_cast<uint64_t>(a.car_begin) + idx;
        long long kb;
        uint32_t clo1[1], khi1[1], dest1[1];
        bool ok1[1], want1[1];
        car_draw_words(a.seed, car, a.step, kb, clo1[0], khi1[0]);
        const bool drive1 = valid1 && (kb <= thr);
        want1[0] = drive1 && last != 0.0;
        pack_search<1>(guide, hi, khi1, want1, sh, hi_last, a.Zq, dest1, ok1);
        if (!want1[0]) dest1[0] = z;
        else if (!ok1[0]) dest1[0] = search_exact_row(cdf_row, Z, u53(clo1[0], khi1[0]), last);
        if (drive1) {
            if (TRAVEL) tt += travel_time_q16(a.dm, Z, a.T, a.t, z, dest1[0], a.seed, car, a.step);
            if (!GROUPED) ++nd;
        }
        if (GROUPED) {
            const unsigned long long m1 = __ballot(valid1 && !drive1);
            uint32_t b1 = 0;
            if (lane == 0 && m1) b1 = atomicAdd(&s_nstay, static_cast<uint32_t>(__popcll(m1)));
            b1 = __shfl(b1, 0, 64);
            if (valid1 && !drive1) stay_out[b1 + static_cast<uint32_t>(__popcll(m1 & below))] = idx;
            if (drive1) {
                const uint32_t g = dest1[0] >> a.gshift;
                const uint32_t rank = atomicAdd(&gb[g], 1u);
                const uint32_t packed = idx | ((dest1[0] & ((1u << a.gshift) - 1u)) << a.idbits);
                if (rank < static_cast<uint32_t>(kStage6)) stage[g * kStage6 + rank] = packed;
                else if (rank < a.scap) runs[g * a.scap + rank] = packed;
            }
        } else {
            if (valid1) a.rec_out[b + q] = dest1[0] | (drive1 ? kDriveBit : 0u);
        }
    }
    CPM_STAMP(a, z, 5);
    if (!GROUPED) {
        for (int o = 32; o > 0; o >>= 1) nd += __shfl_down(nd, o, 64);
        if (lane == 0 && nd) atomicAdd(&s_ndrive, nd);
    }
    if (TRAVEL) {
        for (int o = 32; o > 0; o >>= 1) tt += __shfl_down(tt, o, 64);
        if (lane == 0 && tt) atomicAdd(&s_tt, static_cast<unsigned long long>(tt));
    }
    __syncthreads();  // ranks, staged drivers and counters are final
    CPM_STAMP(a, z, 6);
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
        a.driving_t[z] = GROUPED ? n - s_nstay : s_ndrive;  // every car of the bucket either stays or drives
        if (GROUPED) a.cnt_next[z] = s_nstay;  // k_zone6_place adds the arrivals
        if (TRAVEL && s_tt) atomicAdd(a.tt_sum, s_tt);
    }
    CPM_STAMP(a, z, 7);
}

// Drivers of destination group g -> their buckets.  blockIdx = j * kGroups6 + g: the blocks of a group share
// blockIdx % 8 (one XCD, one L2: all writes to a bucket merge there; speed only, never correctness).  Block
// (g, j) takes the group-g runs of the origin zones [j*zps, (j+1)*zps): 16 lanes per run, KDEEP entries per
// lane.  Run lengths and run contents sit at addresses known up front, so they are requested together.
constexpr int kPlace6Block = 1024;
constexpr int kPlace6Seg = kPlace6Block / 16;

// Travel times of an hour's drivers (src/resampling.jl:53-69), read back from the runs by a kernel of their own (one block per
// origin zone, 8 lanes per run) instead of riding in the sampler, which they slow to a third of its occupancy (138 VGPRs).
// The sum is an integer in 2^-16 s units: order-free, bit-exact.
constexpr int kTravelParts = 256;

struct TravelArgs {
    const double *dm;
    unsigned long long *tt_part;  // [kTravelParts] partial sums, zero between resamples
    int T, t, gshift;
    uint32_t step;
    int64_t car_begin;
    uint64_t seed;
};

__device__ __forceinline__ void zone6_travel_block(const uint32_t *__restrict__ D, const uint32_t *__restrict__ cntg, int Z, uint32_t scap,
                                                   uint32_t idbits, const TravelArgs &tr, int z)
{
    // The zone's drivers are dealt evenly over the threads whatever the run lengths are (popular destination groups
    // hold most of them): driver i of the zone sits in run g with prefix[g] <= i < prefix[g+1].
    __shared__ uint32_t prefix[kGroups6 + 1];
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid < 64) {
        const uint32_t c = (lane < kGroups6) ? min(cntg[static_cast<size_t>(z) * kGroups6 + lane], scap) : 0u;
        uint32_t incl = c;
        for (int o = 1; o < kGroups6; o <<= 1) {
            const uint32_t v = __shfl_up(incl, o, 64);
            if (lane >= o) incl += v;
        }
        if (lane < kGroups6) prefix[lane + 1] = incl;
        if (lane == 0) prefix[0] = 0;
    }
    __syncthreads();
    const uint32_t total = prefix[kGroups6];
    const uint32_t idmask = (idbits >= 32) ? 0xFFFFFFFFu : ((1u << idbits) - 1u);
    long long tt = 0;
    // Batches of kTravelBatch drivers per thread: their run entries, then their two datamatrix cells, are requested together
    // (a driver's chain entry -> cell -> mean, std -> draws is otherwise three exposed round trips).
    constexpr int kTravelBatch = 4;
    const size_t sd_off = static_cast<size_t>(Z) * Z * tr.T;
    for (uint32_t i0 = tid; i0 < total; i0 += kTravelBatch * blockDim.x) {
        uint32_t w[kTravelBatch], dest[kTravelBatch];
        bool live[kTravelBatch];
#pragma unroll
        for (int u = 0; u < kTravelBatch; ++u) {
            const uint32_t i = i0 + u * blockDim.x;
            live[u] = i < total;
            uint32_t g = 0;
#pragma unroll
            for (int step = kGroups6 / 2; step > 0; step >>= 1)
                if (prefix[g + step] <= i) g += step;
            g = live[u] ? g : 0u;
            w[u] = live[u] ? D[(static_cast<size_t>(z) * kGroups6 + g) * scap + (i - prefix[g])] : 0u;
            dest[u] = (g << tr.gshift) + (w[u] >> idbits);
        }
        double mean[kTravelBatch], sd[kTravelBatch];
#pragma unroll
        for (int u = 0; u < kTravelBatch; ++u) {
            const bool moving = live[u] && dest[u] != static_cast<uint32_t>(z);
            const size_t cell = static_cast<size_t>(z) + static_cast<size_t>(Z) * (dest[u] + static_cast<size_t>(Z) * tr.t);
            mean[u] = moving ? tr.dm[cell] : 0.0;
            sd[u] = moving ? tr.dm[cell + sd_off] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < kTravelBatch; ++u) {
            if (!live[u]) continue;
            if (dest[u] == static_cast<uint32_t>(z)) {  // same zone: 300 s (src/resampling.jl:58-60)
                tt += q16(300.0);
            } else {
                const double s1 = (sd[u] == 0) ? 0.1 * mean[u] : sd[u];  // :65-67
                tt += q16(truncnormal_pm10(tr.seed, static_cast<uint64_t>(tr.car_begin) + (w[u] & idmask), tr.step, 1, mean[u], s1));
            }
        }
    }
    // one global atomic per block, spread over kTravelParts words (atomics on ONE word are served one at a time at the memory
    // side: four per block on the sum itself made this kernel 119 us per hour); k_zone6_travel_finish adds the parts up
    __shared__ unsigned long long s_tt;
    if (tid == 0) s_tt = 0;
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) tt += __shfl_down(tt, o, 64);
    if (lane == 0 && tt) atomicAdd(&s_tt, static_cast<unsigned long long>(tt));
    __syncthreads();
    if (tid == 0 && s_tt) atomicAdd(&tr.tt_part[z % kTravelParts], s_tt);
}

__global__ __launch_bounds__(256) void k_zone6_travel_finish(unsigned long long *__restrict__ tt_part, unsigned long long *__restrict__ tt_sum)
{
    unsigned long long v = tt_part[threadIdx.x];
    tt_part[threadIdx.x] = 0;  // ready for the next resample
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(tt_sum, v);
}

// one 256-thread block per origin zone
__global__ __launch_bounds__(256) void k_zone6_travel(const uint32_t *__restrict__ D, const uint32_t *__restrict__ cntg, int Z, uint32_t scap,
                                                      uint32_t idbits, TravelArgs tr)
{
    zone6_travel_block(D, cntg, Z, scap, idbits, tr, blockIdx.x);
}

template <int KRUNS, int KDEEP>
__global__ __launch_bounds__(kPlace6Block) void k_zone6_place(const uint32_t *__restrict__ D, const uint32_t *__restrict__ cntg,
                                                              int zpg, int zps, int Z, uint32_t cap, uint32_t scap, uint32_t idbits,
                                                              uint32_t *__restrict__ cnt_next, uint32_t *__restrict__ ids_next,
                                                              unsigned long long *status)
{
    // bins: entries held in registers per destination zone, then this block's base inside the zone's bucket;
    // tbins: entries beyond 16 * KDEEP of their run (re-read in pass B), then the running position of those
    __shared__ uint32_t bins[kMaxZonesPerGroup6], tbins[kMaxZonesPerGroup6];
    const int tid = threadIdx.x;
    const int g = blockIdx.x % kGroups6, j = blockIdx.x / kGroups6;
    const int zg0 = g * zpg;
    const int nzl = max(0, min(zpg, Z - zg0));
    const int zs0 = j * zps, zs1 = min(Z, zs0 + zps);
    const int sub = tid >> 4, l16 = tid & 15;
    const uint32_t idmask = (idbits >= 32) ? 0xFFFFFFFFu : ((1u << idbits) - 1u);
    for (int k = tid; k < kMaxZonesPerGroup6; k += kPlace6Block) {
        bins[k] = 0;
        tbins[k] = 0;
    }
    if (zs0 >= zs1) return;  // (uniform per block)
    uint32_t c[KRUNS], v[KRUNS][KDEEP], r[KRUNS][KDEEP];
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
    // pass A: rank of every entry among the block's entries for the same destination zone (= the histogram, once all are in)
#pragma unroll
    for (int k = 0; k < KRUNS; ++k) {
#pragma unroll
        for (int d = 0; d < KDEEP; ++d) {
            r[k][d] = 0;
            if (static_cast<uint32_t>(l16 + 16 * d) < c[k]) r[k][d] = atomicAdd(&bins[v[k][d] >> idbits], 1u);
        }
    }
#pragma unroll
    for (int k = 0; k < KRUNS; ++k) {
        const int zc = min(zs0 + sub + k * kPlace6Seg, zs1 - 1);
        const size_t run = static_cast<size_t>(zc) * kGroups6 + g;
        for (uint32_t i = l16 + 16 * KDEEP; i < c[k]; i += 16) atomicAdd(&tbins[D[run * scap + i] >> idbits], 1u);
    }
    __syncthreads();
    if (tid < nzl) {  // ticket: this block's range inside each bucket of the group
        const uint32_t cr = bins[tid], ct = tbins[tid];
        uint32_t base = 0;
        if (cr + ct) {
            base = atomicAdd(&cnt_next[zg0 + tid], cr + ct);
            if (base + cr + ct > cap) atomicOr(status, 2ull);
        }
        bins[tid] = base;
        tbins[tid] = base + cr;
    }
    __syncthreads();
    // pass B: the ids move
#pragma unroll
    for (int k = 0; k < KRUNS; ++k) {
#pragma unroll
        for (int d = 0; d < KDEEP; ++d)
            if (static_cast<uint32_t>(l16 + 16 * d) < c[k]) {
                const uint32_t dl = v[k][d] >> idbits;
                const uint32_t p = bins[dl] + r[k][d];
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
            const uint32_t p = atomicAdd(&tbins[dl], 1u);
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

constexpr int kBlock6 = 256;  // measured at S4k: 512 threads x 2 cars: 31 us, 256 x 4: 28, 128 x 8: 44

template <bool TRAVEL, bool GROUPED, int CPT, int NQ>
inline void zone6_launch_nq(const Zone6Args &a, size_t lds, hipStream_t stream)
{
    if (lds > 48 * 1024) {  // LDS opt-in, once per device (contexts of several devices may live in one process)
        static bool attr_done[64] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 64 || !attr_done[dev]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_zone6_sample<TRAVEL, kBlock6, CPT, NQ, GROUPED>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            if (dev >= 0 && dev < 64) attr_done[dev] = true;
        }
    }
    hipLaunchKernelGGL((k_zone6_sample<TRAVEL, kBlock6, CPT, NQ, GROUPED>), dim3(a.Z), dim3(kBlock6), lds, stream, a);
}

template <bool TRAVEL, bool GROUPED, int CPT>
inline void zone6_launch_c(const Zone6Args &a, hipStream_t stream)
{
    const int words = pack_row_words(a.Zq, a.G);
    const size_t lds = sizeof(uint32_t) * static_cast<size_t>(words);
    const int need = (words / 4 + kBlock6 - 1) / kBlock6;
    if (need <= 1) zone6_launch_nq<TRAVEL, GROUPED, CPT, 1>(a, lds, stream);
    else if (need <= 2) zone6_launch_nq<TRAVEL, GROUPED, CPT, 2>(a, lds, stream);
    else if (need <= 3) zone6_launch_nq<TRAVEL, GROUPED, CPT, 3>(a, lds, stream);
    else if (need <= 4) zone6_launch_nq<TRAVEL, GROUPED, CPT, 4>(a, lds, stream);
    else if (need <= 5) zone6_launch_nq<TRAVEL, GROUPED, CPT, 5>(a, lds, stream);
    else if (need <= 6) zone6_launch_nq<TRAVEL, GROUPED, CPT, 6>(a, lds, stream);
    else if (need <= 8) zone6_launch_nq<TRAVEL, GROUPED, CPT, 8>(a, lds, stream);
    else if (need <= 12) zone6_launch_nq<TRAVEL, GROUPED, CPT, 12>(a, lds, stream);
    else if (need <= 20) zone6_launch_nq<TRAVEL, GROUPED, CPT, 20>(a, lds, stream);
    else zone6_launch_nq<TRAVEL, GROUPED, CPT, 40>(a, lds, stream);
}

// Cars per thread by the mean bucket size (cars of this GPU / zones): 256 x 4 slots for ~1000 cars per zone, 256 x 2 and 256 x 1
// for smaller buckets (every slot runs Philox whether a car sits in it or not); larger buckets take the overflow rounds.
template <bool TRAVEL, bool GROUPED>
inline void zone6_launch_t(const Zone6Args &a, int64_t mean, hipStream_t stream)
{
    if (mean <= 224) zone6_launch_c<TRAVEL, GROUPED, 1>(a, stream);
    else if (mean <= 560) zone6_launch_c<TRAVEL, GROUPED, 2>(a, stream);
    else zone6_launch_c<TRAVEL, GROUPED, 4>(a, stream);
}

template <bool GROUPED>
inline void zone6_launch(const Zone6Args &a, bool travel, int64_t mean, hipStream_t stream)
{
    if (travel) zone6_launch_t<true, GROUPED>(a, mean, stream);
    else zone6_launch_t<false, GROUPED>(a, mean, stream);
}

// a row pack must fit the 150 KiB of LDS a workgroup may ask for, and a guide entry is a u16
inline bool zone6_row_fits(int Z)
{
    if (Z < 2 || Z > 32768) return false;
    return sizeof(uint32_t) * static_cast<size_t>(pack_row_words(pack_zq(Z), pack_guide_bits(Z))) <= 150 * 1024;
}

// Diagnostic (cpm_debug_categorical): the categorical draw of the sampler for given 53-bit draws k against one
// installed row, through the same staging, search and exact-row code.  out[i] = destination (1-based), or 0 for a zero row.
__global__ __launch_bounds__(512) void k_zone6_search_debug(const uint32_t *__restrict__ pack_g, const double *__restrict__ last_p,
                                                            const double *__restrict__ cdf_row, int Z, int Zq, int G, int64_t n,
                                                            const uint64_t *__restrict__ k53, int64_t *__restrict__ out,
                                                            int *__restrict__ n_exact)
{
    extern __shared__ uint32_t pack[];
    const int tid = threadIdx.x;
    const int gw = pack_guide_words(G), pieces = pack_row_words(Zq, G) / 4, sh = 32 - G;
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
    const uint32_t hi_last = hi[Z - 1];
    for (int64_t i = tid; i < n; i += 512) {
        const uint64_t k = k53[i] & ((1ull << 53) - 1ull);
        if (last == 0.0) {
            out[i] = 0;
            continue;
        }
        uint32_t dest[1];
        bool ok[1];
        const uint32_t khi[1] = {static_cast<uint32_t>(k >> 21)};
        const bool want[1] = {true};
        pack_search<1>(guide, hi, khi, want, sh, hi_last, Zq, dest, ok);
        if (!ok[0]) {
            dest[0] = search_exact_row(cdf_row, Z, static_cast<double>(k) * 0x1.0p-53, last);
            atomicAdd(n_exact, 1);
        }
        out[i] = static_cast<int64_t>(dest[0]) + 1;
    }
}

}  // namespace cpm
