// cpm_zone3_kernels.h -- CPM_KERNEL_ZONE_STRIDED: the zone path with fixed-stride buckets.
//
// Same sampler as cpm_zone_kernels.h (k_zone_sample2).  What changes is the counting sort that
// regroups the cars by destination every hour.  In the exact layout a car's slot in next hour's
// bucket is offA[dest] + ticket, and offA is an exclusive scan over ALL zones' arrival counts,
// known only when every block has counted -- hence two launches (histogram+ticket, scan+scatter),
// a second read of the keys and a Z-wide scan in every scatter block.  Here bucket z owns the
// fixed region [z*cap, (z+1)*cap) of the id array, so a slot is dest*cap + ticket and nothing
// global has to be known: histogram, ticket and scatter are ONE kernel with block-level barriers
// only.  The bucket sizes (= next hour's parking histogram) are the ticket counters themselves.
//
// cap is 4x the mean bucket size (at least 1024).  A bucket that would outgrow it sets bit 1 of the
// status word (no out-of-range store is ever issued) and the caller repeats the step with the exact
// layout (cpm_resample does so by itself), so skewed tables stay correct, only slower.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <string>

#include "../../include/cpm.h"
#include "cpm_kernels.h"
#include "cpm_zone_kernels.h"

namespace cpm {

constexpr int kSort3Block = 1024;
constexpr int kSort3MaxZones = 64;  // zones per block of the strided-input form

// One kernel: LDS histogram of the destinations of this block's slots -> one batched round of global
// atomics reserves the block's range in every destination bucket -> ids move to dest*cap + position.
// Input slots: STRIDED_IN ? the buckets of zones [zb0, zb1) in the fixed-stride layout (sizes cnt_in[])
//                         : the linear range [i0, i1) of a car-indexed key array (id = index).
constexpr int kSort3MaxPass = 512;  // 1024-slot passes per block (zones_per_blk * ceil(cap / 1024) must fit)

template <bool STRIDED_IN>
__global__ __launch_bounds__(kSort3Block) void k_zone3_sort(const uint32_t *__restrict__ key, const uint32_t *__restrict__ ids,
                                                            const uint32_t *__restrict__ cnt_in, int zones_per_blk, int64_t n,
                                                            int64_t chunk, int Z, uint32_t cap, uint32_t *__restrict__ cnt_next,
                                                            uint32_t *__restrict__ ids_next, unsigned long long *status)
{
    extern __shared__ uint32_t bins[];  // Z: histogram, then running position inside each bucket
    // The block's input as a list of passes of up to 1024 consecutive slots: (first slot, number of slots).
    __shared__ unsigned long long pass_base[kSort3MaxPass];
    __shared__ uint32_t pass_len[kSort3MaxPass];
    __shared__ int s_npass;
    const int tid = threadIdx.x;
    for (int z = tid; z < Z; z += kSort3Block) bins[z] = 0;
    if (tid == 0) {
        int np = 0;
        if (STRIDED_IN) {
            const int zb0 = blockIdx.x * zones_per_blk;
            const int nz = max(0, min(zones_per_blk, Z - zb0));
            for (int k = 0; k < nz; ++k) {
                const uint32_t c = min(cnt_in[zb0 + k], cap);
                for (uint32_t o = 0; o < c && np < kSort3MaxPass; o += kSort3Block) {
                    pass_base[np] = static_cast<unsigned long long>(zb0 + k) * cap + o;
                    pass_len[np++] = min<uint32_t>(kSort3Block, c - o);
                }
            }
        } else {
            const int64_t i0 = static_cast<int64_t>(blockIdx.x) * chunk, i1 = min(i0 + chunk, n);
            for (int64_t o = i0; o < i1 && np < kSort3MaxPass; o += kSort3Block) {
                pass_base[np] = static_cast<unsigned long long>(o);
                pass_len[np++] = static_cast<uint32_t>(min<int64_t>(kSort3Block, i1 - o));
            }
        }
        s_npass = np;
    }
    __syncthreads();
    const int npass = s_npass;
    constexpr int kU = 8;  // passes in flight per thread: all their loads are issued before the first use
    // phase 1: histogram
    for (int p0 = 0; p0 < npass; p0 += kU) {
        uint32_t v[kU];
        bool ok[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            ok[u] = p0 + u < npass && static_cast<uint32_t>(tid) < pass_len[min(p0 + u, npass - 1)];
            if (ok[u]) v[u] = key[pass_base[p0 + u] + tid];
        }
#pragma unroll
        for (int u = 0; u < kU; ++u)
            if (ok[u]) atomicAdd(&bins[v[u] & kZoneMask], 1u);
    }
    __syncthreads();
    // phase 2: ticket.  bins[z] becomes this block's first position inside bucket z.
    for (int zb = 0; zb < Z; zb += kSort3Block * 4) {
        uint32_t r[4], c[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int z = zb + tid + k * kSort3Block;
            r[k] = 0;
            c[k] = 0;
            if (z < Z) {
                c[k] = bins[z];
                if (c[k]) r[k] = atomicAdd(&cnt_next[z], c[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int z = zb + tid + k * kSort3Block;
            if (z < Z) {
                bins[z] = r[k];
                if (r[k] + c[k] > cap) atomicOr(status, 2ull);  // bucket outgrew its region: step invalid
            }
        }
    }
    __syncthreads();
    // phase 3: scatter (keys and ids re-read: they are L2-resident from phase 1)
    for (int p0 = 0; p0 < npass; p0 += kU) {
        uint32_t v[kU], c[kU];
        bool ok[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            ok[u] = p0 + u < npass && static_cast<uint32_t>(tid) < pass_len[min(p0 + u, npass - 1)];
            if (ok[u]) {
                const unsigned long long s = pass_base[p0 + u] + tid;
                v[u] = key[s] & kZoneMask;
                c[u] = ids ? ids[s] : static_cast<uint32_t>(s);
            }
        }
#pragma unroll
        for (int u = 0; u < kU; ++u)
            if (ok[u]) {
                const uint32_t p = atomicAdd(&bins[v[u]], 1u);
                if (p < cap) ids_next[static_cast<size_t>(v[u]) * cap + p] = c[u];
            }
    }
}

// int64 parking counts are written by the sampler; this path keeps u32 bucket sizes per hour
struct Zone3Work {
    bool attrs_set = false;
    bool buckets0_valid = false;
    int64_t n = 0;
    int Z = 0, T = 0, nblk = 0, zones_per_blk = 0, nb0 = 0;
    uint32_t cap = 0;
    int cap_mult = 4;        // bucket region = cap_mult x the mean bucket size; doubled by the context after an overflow (up to kMaxCapMult)
    int cap_mult_alloc = 0;  // what the arrays below were sized for
    uint32_t *ids0 = nullptr, *idsA = nullptr, *idsB = nullptr, *dest = nullptr;  // [Z*cap]
    uint32_t *cnt0 = nullptr;                                                      // [Z] sizes of the cached initial buckets
    uint32_t *cnt = nullptr;                                                       // [T+1][Z] bucket sizes per hour
    ZoneWork sampler;  // launcher state of k_zone_sample2 (LDS opt-in flags)

    void release()
    {
        for (uint32_t **p : {&ids0, &idsA, &idsB, &dest, &cnt0, &cnt}) {
            if (*p) (void)hipFree(*p);
            *p = nullptr;
        }
        n = 0;
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
        const int64_t mean = (n + Z - 1) / Z;
        cap = static_cast<uint32_t>((std::max<int64_t>(cap_mult * mean, 1024) + 63) / 64 * 64);
        const int want = std::max(1, std::min(Z, 2 * cu_count));  // two 1024-thread blocks per CU overlap their phases
        zones_per_blk = std::min(kSort3MaxZones, (Z + want - 1) / want);
        while (static_cast<int64_t>(zones_per_blk) * ((cap + 1023) / 1024) > kSort3MaxPass && zones_per_blk > 1) --zones_per_blk;
        nblk = (Z + zones_per_blk - 1) / zones_per_blk;
        nb0 = static_cast<int>(std::max<int64_t>({int64_t(1), std::min<int64_t>(2 * cu_count, (n + 4095) / 4096),
                                                  (n + int64_t(kSort3MaxPass) * 1024 - 1) / (int64_t(kSort3MaxPass) * 1024)}));
        hipError_t e = hipSuccess;
        auto alloc = [&](uint32_t **p, size_t words) {
            if (e == hipSuccess) e = hipMalloc(p, sizeof(uint32_t) * std::max<size_t>(words, 1));
        };
        const size_t slots = static_cast<size_t>(Z) * cap;
        alloc(&ids0, slots);
        alloc(&idsA, slots);
        alloc(&idsB, slots);
        alloc(&dest, slots);
        alloc(&cnt0, Z);
        alloc(&cnt, static_cast<size_t>(T + 1) * Z);
        if (e != hipSuccess) release();
        return e;
    }
};

// true when the fixed-stride layout is worth its memory: Z*cap slots x 4 arrays x 4 B
constexpr int kMaxCapMult = 64;

inline bool zone3_path_fits(int Zp, int64_t n, int Z, int cap_mult = 4)
{
    if (!zone_path_fits(Zp) || n >= (int64_t(1) << 31)) return false;
    const int64_t mean = (n + Z - 1) / Z;
    const int64_t cap = std::max<int64_t>(cap_mult * mean, 1024);
    return static_cast<int64_t>(Z) * cap * 16 <= (int64_t(cap_mult <= 4 ? 8 : 32) << 30);  // <= 8 GiB of bucket arrays (32 GiB once grown)
}

// The T-hour resample from the state in d_zone0 (left unchanged).  On bucket overflow bit 1 of the status
// word (d_counts[2*T*Z+1]) is set and the counts are invalid.  (The IVP, which overwrites the state and so
// cannot simply be repeated, always runs on the exact layout of cpm_zone_kernels.h.)
template <typename F1, typename F2>
int32_t zone3_resample(Zone3Work &w, hipStream_t stream, const double *d_pdrive, const double *d_cdf, int Z, int Zp, int T,
                       int64_t n, int64_t car_begin, const uint32_t *d_zone0, uint64_t seed, bool travel,
                       const double *d_dm, int64_t *d_counts, int cu_count, F1 prof_begin, F2 prof_end, std::string &err)
{
    auto hip_fail = [&](hipError_t e, const char *what) {
        err = std::string(what) + ": " + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? CPM_ERR_NOMEM : CPM_ERR_HIP;
    };
    hipError_t e = w.ensure(n, Z, T, cu_count);
    if (e != hipSuccess) return hip_fail(e, "strided zone workspace");
    const size_t lds_bins = sizeof(uint32_t) * static_cast<size_t>(Z);
    if (!w.attrs_set) {
        if (lds_bins > 64 * 1024) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_zone3_sort<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      static_cast<int>(lds_bins));
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_zone3_sort<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      static_cast<int>(lds_bins));
        }
        w.attrs_set = true;
    }
    unsigned long long *parking = reinterpret_cast<unsigned long long *>(d_counts);
    unsigned long long *driving = parking + static_cast<size_t>(T) * Z;
    unsigned long long *tt_sum = parking + 2 * static_cast<size_t>(T) * Z;
    unsigned long long *status = tt_sum + 1;
    e = hipMemsetAsync(w.cnt, 0, sizeof(uint32_t) * static_cast<size_t>(T + 1) * Z, stream);
    if (e != hipSuccess) return hip_fail(e, "memset bucket sizes");
    if (!w.buckets0_valid) {  // bucket the car-indexed state once; reused until the state changes
        const int64_t chunk = (n + w.nb0 - 1) / w.nb0;
        if ((e = hipMemsetAsync(w.cnt0, 0, sizeof(uint32_t) * Z, stream)) != hipSuccess) return hip_fail(e, "memset cnt0");
        hipLaunchKernelGGL(k_zone3_sort<false>, dim3(w.nb0), dim3(kSort3Block), lds_bins, stream, d_zone0,
                           static_cast<const uint32_t *>(nullptr), static_cast<const uint32_t *>(nullptr), 0, n, chunk, Z, w.cap,
                           w.cnt0, w.ids0, status);
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "initial bucketing");
        w.buckets0_valid = true;
    }
    const uint32_t *ids = w.ids0, *cnt = w.cnt0;
    for (int t = 0; t < T; ++t) {
        const double *pd = d_pdrive + static_cast<size_t>(t) * Z;
        const double *cdf = d_cdf + static_cast<size_t>(t) * Z * Zp;
        const uint32_t step = static_cast<uint32_t>(T - 1 + t);
        prof_begin(t);
        launch_zone_sample(w.sampler, stream, travel, ids, cnt, pd, cdf, Z, Zp, car_begin, step, seed,
                           parking + static_cast<size_t>(t) * Z, driving + static_cast<size_t>(t) * Z, d_dm, T, t, tt_sum, 0,
                           w.dest, w.cap);
        prof_end(t);
        if (t + 1 < T) {  // hour T's transition is sampled but never applied (src/resampling.jl:81-83)
            uint32_t *cnt_next = w.cnt + static_cast<size_t>(t + 1) * Z;
            uint32_t *ids_next = (t & 1) ? w.idsB : w.idsA;
            hipLaunchKernelGGL(k_zone3_sort<true>, dim3(w.nblk), dim3(kSort3Block), lds_bins, stream, w.dest, ids, cnt,
                               w.zones_per_blk, n, int64_t(0), Z, w.cap, cnt_next, ids_next, status);
            ids = ids_next;
            cnt = cnt_next;
        }
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "strided zone hour launch");
    }
    return CPM_OK;
}

}  // namespace cpm
