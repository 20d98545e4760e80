// cpm_zone2_kernels.h -- CPM_KERNEL_ZONE_FUSED: the fused zone path (second generation of
// cpm_zone_kernels.h; same idea: cars bucketed by origin zone, the zone's CDF row streamed once
// per hour from HBM and searched in LDS).  What changed, and why (profiles/round1_notes.md):
//   * persistent workgroups own a contiguous range of zones; rows are requested TWO zones ahead
//     (two register sets), car ids one zone ahead, so HBM streaming overlaps Philox and the
//     search instead of alternating with them;
//   * the row lives in LDS as an implicit binary search tree in breadth-first (Eytzinger)
//     order, permuted while it is written: the top levels are broadcast reads, the rest spread
//     over the banks (the sorted layout's power-of-two probe strides spent 72 % of the LDS
//     cycles in bank conflicts);
//   * every thread carries CPT cars through straight-line code: independent Philox chains and
//     tree walks interleave, and the number of stores per zone is static, so the compiler's
//     vmcnt waits stay counted instead of draining the prefetch;
//   * cars that do not drive (about half) never leave their zone: they are compacted, coalesced,
//     into the zone's "stayer" segment S of the next hour.  Only drivers are moved;
//   * the counting sort of the drivers is fused in: LDS bins give each driver its rank inside
//     (workgroup, destination); one batched round of global atomics per workgroup reserves the
//     workgroup's range in every destination bucket (ticket).  The scatter kernel then needs no
//     atomics at all: position = offA[dest] + base[wg][dest] + rank.
// A zone's bucket at hour t is two segments: S (stayers, at the zone's slot of layout L) and
// A (arrivals, exclusive-scan layout offA); parking[t][z] = |S| + |A| (src/saveresults.jl:12).
//
// Per hour t:  k_zone2_sample  (sample + compact + rank + ticket)     [the dominant kernel]
//              k_zone2_scatter (scan of the arrival counts; drivers' ids -> next hour's A)
// The last hour's transition is sampled for the driving histogram but never applied
// (src/resampling.jl:81-83), so it runs the sampler alone.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <string>

#include "../../include/cpm.h"
#include "cpm_kernels.h"
#include "cpm_zone_kernels.h"

namespace cpm {

// Ticket counters are replicated kRep times on separate cache lines (workgroup b uses replica
// b % kRep): atomics on one line are served one after the other at the memory side (~50 ns each),
// so 683 workgroups ticketing the same 128 lines cost ~35 us; with 8 replicas ~4 us.
constexpr int kRep = 8;
constexpr int kMaxZonesPerWg = 64;
constexpr int kDestBits = 16;  // D key = dest | rank << 16 : needs Z <= 65536 and rank < 65536
constexpr uint32_t kDestMask = (1u << kDestBits) - 1u;

struct Zone2Args {
    // tables
    const double *pdrive_t;  // [Z]
    const double *cdf_t;     // [Z][Zp]
    const double *dm;        // travel tables or nullptr
    // bucket layout of this hour
    const uint32_t *S;       // stayer ids, zone z at [L[z], L[z] + nS[z])
    const uint32_t *A;       // arrival ids, zone z at [offA[z], offA[z+1])
    const uint32_t *L;       // [Z+1] layout of S (exclusive scan of the previous hour's zone sizes)
    const uint32_t *nS;      // [Z]
    const uint32_t *offA;    // [Z+1]
    // next hour
    uint32_t *S_next;        // stayers of zone z at [Ln[z], ...)
    const uint32_t *Ln;      // [Z+1] exclusive scan of THIS hour's zone sizes
    uint32_t *nS_next;       // [Z]
    uint2 *D;                // drivers (id, dest | rank << 16), workgroup b at [Ln[zb0], ...)
    uint32_t *nd;            // [gridDim.x] drivers per workgroup
    uint32_t *cursor;        // [kRep][Z] arrivals per zone of the next hour (ticket counters)
    uint32_t *base;          // [gridDim.x][Z] this workgroup's range inside each arrival bucket
    unsigned long long *dbg;     // diagnostic (ABL & 4): per-workgroup cycle sums of the kernel's segments
    unsigned long long *status;  // counts word 2*T*Z+1: bit 0 set when a rank does not fit 16 bits (host falls back)
    // results
    unsigned long long *parking_t, *driving_t, *tt_sum;
    int Z, Zp, H, T, t;      // H: tree height, 2^H >= Z
    int zones_per_wg;
    int64_t car_begin;
    uint32_t step;
    uint64_t seed;
};

// NP: 16-B row pieces per thread (>= Zp / 2 / BLOCK).  CPT: cars per thread in the straight-line
// part.  LAST: final hour (no next-hour state).  ABL (diagnostic, results wrong but every index
// stays in range): 1 = no tree search, 2 = cheap hash instead of Philox.
template <bool TRAVEL, bool LAST, int BLOCK, int NP, int CPT, int ABL>
__global__ __launch_bounds__(BLOCK) void k_zone2_sample(Zone2Args a)
{
    extern __shared__ double lds[];  // row tree: 2^H doubles, then bins: Z u32
    __shared__ uint32_t s_cntS, s_cntD, s_drv;
    __shared__ unsigned long long s_tt;
    __shared__ uint32_t mL[kMaxZonesPerWg + 1], mNS[kMaxZonesPerWg], mOffA[kMaxZonesPerWg + 1], mLn[kMaxZonesPerWg + 1];
    __shared__ long long mThr[kMaxZonesPerWg];
    const int Z = a.Z, Zp = a.Zp, H = a.H;
    const int P = 1 << H;
    double *row = lds;
    uint32_t *bins = reinterpret_cast<uint32_t *>(lds + P);
    const int tid = threadIdx.x, lane = tid & 63;
    const int zb0 = blockIdx.x * a.zones_per_wg;
    const int zb1 = min(zb0 + a.zones_per_wg, Z);
    if (zb0 >= zb1) {
        if (!LAST && tid == 0) a.nd[blockIdx.x] = 0;
        if (!LAST)
            for (int z = tid; z < Z; z += BLOCK) a.base[static_cast<size_t>(blockIdx.x) * Z + z] = 0;
        return;
    }
    // one-time LDS init: tree slots of ranks Z..2^H-1 are +inf (never written by the staging)
    for (int r = Z + tid; r < P; r += BLOCK) {
        int tz = __builtin_ctz(static_cast<unsigned>(r));
        row[(1u << (H - 1 - tz)) + (static_cast<unsigned>(r) >> (tz + 1))] = __builtin_huge_val();
    }
    if (!LAST)
        for (int z = tid; z < Z; z += BLOCK) bins[z] = 0;
    if (tid == 0) {
        s_cntD = 0;
        s_tt = 0;
    }
    const unsigned long long below = (1ull << lane) - 1ull;

    // Zone metadata of this workgroup's range -> LDS, once.  Read through LDS the per-zone
    // scalars cost no vector-memory operation, so nothing in the zone loop waits on vmcnt except
    // the one counted wait for the row registers.
    const int nzw = zb1 - zb0;
    for (int i = tid; i <= nzw; i += BLOCK) {
        mL[i] = a.L[zb0 + i];
        mOffA[i] = a.offA[zb0 + i];
        mLn[i] = LAST ? 0u : a.Ln[zb0 + i];
        if (i < nzw) {
            mNS[i] = a.nS[zb0 + i];
            mThr[i] = bernoulli_threshold(a.pdrive_t[zb0 + i]);
        }
    }
    __syncthreads();  // LDS init + metadata visible

    // Prefetch of zone (zb0 + zi): its CPT car ids per thread, then its row, two zones ahead.
    // Every load is unconditional and branch-free (indices clamped into range): the number of
    // vector-memory operations per step is then the same on every path, and the compiler's
    // in-order vmcnt wait for one zone's registers leaves the next zone's loads in flight.
    auto prefetch = [&](double2(&pc)[NP], uint32_t(&ids)[CPT], int zi) {
        zi = min(zi, nzw - 1);
        const uint32_t sBeg = mL[zi], nS = mNS[zi], aBeg = mOffA[zi];
        const uint32_t nz = nS + (mOffA[zi + 1] - aBeg);
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            const uint32_t q = min(static_cast<uint32_t>(tid + c * BLOCK), nz ? nz - 1 : 0u);
            const uint32_t *p = (q < nS) ? a.S + sBeg + q : a.A + (nz ? aBeg + (q - nS) : 0u);
            ids[c] = *p;
        }
        const double2 *src = reinterpret_cast<const double2 *>(a.cdf_t + static_cast<size_t>(zb0 + zi) * Zp);
#pragma unroll
        for (int m = 0; m < NP; ++m) pc[m] = src[min(tid + m * BLOCK, Zp / 2 - 1)];
    };
    double2 pieceA[NP], pieceB[NP];
    uint32_t idsA[CPT], idsB[CPT];
    prefetch(pieceA, idsA, 0);
    prefetch(pieceB, idsB, 1);

    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = (ABL & 4) ? clock64() : 0;
    auto stamp = [&](int k) {
        if (ABL & 4) {
            __builtin_amdgcn_sched_barrier(0);
            unsigned long long tn = clock64();
            __builtin_amdgcn_sched_barrier(0);
            seg[k] += tn - tprev;
            tprev = tn;
        }
    };
    const uint32_t dlo = mLn[0];
    auto step = [&](double2(&pc)[NP], uint32_t(&ids)[CPT], int z) {
        stamp(0);
        const int zi = z - zb0;
        const uint32_t sBeg = mL[zi], nS = mNS[zi];
        const uint32_t aBeg = mOffA[zi], nz = nS + (mOffA[zi + 1] - aBeg);
        const long long thr = mThr[zi];
        uint32_t cur[CPT];
        // registers -> LDS tree (breadth-first permutation); the barrier that closed the previous
        // zone guarantees nobody still searches the old tree
#pragma unroll
        for (int m = 0; m < NP; ++m) {
            int j = tid + m * BLOCK;
            if (2 * j < Zp) {
                uint32_t e = 2 * j;
                if (e < static_cast<uint32_t>(Z)) row[eytz_pos(e, Z, H)] = pc[m].x;
                if (e + 1 < static_cast<uint32_t>(Z)) row[eytz_pos(e + 1, Z, H)] = pc[m].y;
            }
        }
#pragma unroll
        for (int c = 0; c < CPT; ++c) cur[c] = ids[c];
        if (tid == 0) {
            s_cntS = 0;
            s_drv = 0;
        }
        stamp(1);  // wait for the row registers + tree write
        prefetch(pc, ids, zi + 2);
        __syncthreads();  // tree complete
        stamp(2);  // prefetch issue + barrier
        const double last = row[0];
        const uint32_t sOut = mLn[zi];
        long long tt = 0;
        uint32_t ndrv = 0;

        // ---- CPT cars per thread, straight line -------------------------------------------
        bool valid[CPT], drive[CPT];
        uint32_t dest[CPT];
        double ue[CPT];
        bool any_search = false;
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            valid[c] = static_cast<uint32_t>(tid + c * BLOCK) < nz;
            long long kb;
            double uc;
            if (ABL & 2) {
                kb = static_cast<long long>(cur[c] * 2654435761u) << 21;
                uc = ((cur[c] ^ a.step) * 2246822519u) * 0x1.0p-32;
            } else {
                car_draws(a.seed, static_cast<uint64_t>(a.car_begin) + cur[c], a.step, kb, uc);
            }
            drive[c] = valid[c] && (kb <= thr);
            dest[c] = z;
            ue[c] = clamp_u(uc, last);
            any_search |= drive[c] && last != 0.0;
        }
        if (ABL & 1) {
#pragma unroll
            for (int c = 0; c < CPT; ++c)
                if (drive[c]) dest[c] = min(static_cast<uint32_t>(ue[c] * Z), static_cast<uint32_t>(Z - 1));
        } else if (__any(any_search)) {
            uint32_t i[CPT];
#pragma unroll
            for (int c = 0; c < CPT; ++c) i[c] = 1;
            for (int l = 0; l < H; ++l) {  // CPT independent tree walks, interleaved
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
        stamp(3);  // Philox + Bernoulli + tree walks
        // compaction: one LDS reservation per wave for all CPT cars
        unsigned long long mD[CPT], mS[CPT];
        uint32_t totS = 0, totD = 0;
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            mD[c] = __ballot(drive[c]);
            mS[c] = __ballot(valid[c] && !drive[c]);
            totS += static_cast<uint32_t>(__popcll(mS[c]));
            totD += static_cast<uint32_t>(__popcll(mD[c]));
        }
        ndrv += totD;
        if (!LAST) {
            uint32_t bS = 0, bD = 0;
            if (lane == 0) {
                if (totS) bS = atomicAdd(&s_cntS, totS);
                if (totD) bD = atomicAdd(&s_cntD, totD);
            }
            bS = __shfl(bS, 0, 64);
            bD = __shfl(bD, 0, 64);
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                if (valid[c] && !drive[c]) a.S_next[sOut + bS + static_cast<uint32_t>(__popcll(mS[c] & below))] = cur[c];
                if (drive[c]) {
                    uint32_t rank = atomicAdd(&bins[dest[c]], 1u);
                    if (rank > kDestMask) atomicOr(a.status, 1ull);
                    a.D[dlo + bD + static_cast<uint32_t>(__popcll(mD[c] & below))] =
                        make_uint2(cur[c], dest[c] | (rank << kDestBits));
                }
                bS += static_cast<uint32_t>(__popcll(mS[c]));
                bD += static_cast<uint32_t>(__popcll(mD[c]));
            }
        }
        if (TRAVEL) {
#pragma unroll
            for (int c = 0; c < CPT; ++c)
                if (drive[c])
                    tt += travel_time_q16(a.dm, Z, a.T, a.t, z, dest[c], a.seed, static_cast<uint64_t>(a.car_begin) + cur[c],
                                          a.step);
        }
        stamp(4);  // compaction, rank, stores
        // ---- zones with more than CPT*BLOCK cars (wave-uniform trip count) ------------------
        for (uint32_t q0 = CPT * BLOCK; q0 < nz; q0 += BLOCK) {
            const uint32_t q = q0 + tid;
            const bool v = q < nz;
            const uint32_t id = v ? ((q < nS) ? a.S[sBeg + q] : a.A[aBeg + (q - nS)]) : 0u;
            const uint64_t car = static_cast<uint64_t>(a.car_begin) + id;
            long long kb;
            double uc;
            car_draws(a.seed, car, a.step, kb, uc);
            const bool drv = v && (kb <= thr);
            uint32_t d = z;
            if (drv && last != 0.0) {
                const double u1 = clamp_u(uc, last);
                uint32_t i = 1;
                for (int l = 0; l < H; ++l) i = 2 * i + (row[i] < u1 ? 1u : 0u);
                d = eytz_decode(i, Z, H);
            }
            const unsigned long long md = __ballot(drv), ms = __ballot(v && !drv);
            ndrv += static_cast<uint32_t>(__popcll(md));
            if (TRAVEL && drv) tt += travel_time_q16(a.dm, Z, a.T, a.t, z, d, a.seed, car, a.step);
            if (!LAST) {
                uint32_t bS = 0, bD = 0;
                if (lane == 0) {
                    if (ms) bS = atomicAdd(&s_cntS, static_cast<uint32_t>(__popcll(ms)));
                    if (md) bD = atomicAdd(&s_cntD, static_cast<uint32_t>(__popcll(md)));
                }
                bS = __shfl(bS, 0, 64);
                bD = __shfl(bD, 0, 64);
                if (v && !drv) a.S_next[sOut + bS + static_cast<uint32_t>(__popcll(ms & below))] = id;
                if (drv) {
                    uint32_t rank = atomicAdd(&bins[d], 1u);
                    if (rank > kDestMask) atomicOr(a.status, 1ull);
                    a.D[dlo + bD + static_cast<uint32_t>(__popcll(md & below))] = make_uint2(id, d | (rank << kDestBits));
                }
            }
        }
        if (lane == 0 && ndrv) atomicAdd(&s_drv, ndrv);
        if (TRAVEL) {
            for (int o = 32; o > 0; o >>= 1) tt += __shfl_down(tt, o, 64);
            if (lane == 0 && tt) atomicAdd(&s_tt, static_cast<unsigned long long>(tt));
        }
        stamp(5);  // overflow rounds + reductions
        __syncthreads();  // counters final; nobody still reads the tree
        stamp(6);  // closing barrier
        if (tid == 0) {
            a.parking_t[z] = nz;  // every car present, drivers included (Appendix A-14)
            a.driving_t[z] = s_drv;
            if (!LAST) a.nS_next[z] = nz - s_drv;
        }
    };
    for (int z = zb0; z < zb1; z += 2) {
        step(pieceA, idsA, z);
        if (z + 1 < zb1) step(pieceB, idsB, z + 1);
    }
    if (TRAVEL && tid == 0 && s_tt) atomicAdd(a.tt_sum, s_tt);
    auto flush_stamps = [&]() {
        if ((ABL & 4) && a.dbg && lane == 0)
            for (int k = 0; k < 8; ++k) atomicAdd(&a.dbg[k], seg[k]);
    };
    if (LAST) {
        flush_stamps();
        return;
    }
    if (tid == 0) a.nd[blockIdx.x] = s_cntD;
    // ticket: reserve this workgroup's range in every arrival bucket; all atomics of a thread
    // are issued before any result is used (one round trip instead of Z/BLOCK)
    uint32_t *mybase = a.base + static_cast<size_t>(blockIdx.x) * Z;
    constexpr int kBatch = 8;
    for (int z0 = 0; z0 < Z; z0 += BLOCK * kBatch) {
        uint32_t r[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            int z = z0 + tid + k * BLOCK;
            r[k] = 0;
            if (z < Z) {
                uint32_t c = bins[z];
                if (c) r[k] = atomicAdd(&a.cursor[static_cast<size_t>(blockIdx.x % kRep) * Z + z], c);
            }
        }
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            int z = z0 + tid + k * BLOCK;
            if (z < Z) mybase[z] = r[k];
        }
    }
    stamp(7);  // ticket
    flush_stamps();
}

// Scatter of the drivers: no atomics.  Block (r, j) moves the drivers of the sampler workgroups
// b with b % kRep == r.  Every block sums the kRep ticket replicas per zone, scans the Z arrival
// counts (-> offA of the next hour) and the Z zone sizes (-> layout L of the hour after); block 0
// publishes both.  position = offA[dest] + (replicas below r)[dest] + base[b][dest] + rank.
constexpr int kScatBlock = 1024;
constexpr int kScatPerMax = 16;  // zones per thread in the scan: Z <= 16384

__global__ __launch_bounds__(kScatBlock) void k_zone2_scatter(const uint2 *__restrict__ D, const uint32_t *__restrict__ Lcur,
                                                              const uint32_t *__restrict__ nd, int zones_per_wg, int nwg,
                                                              int wgs_per_blk, int Z, const uint32_t *__restrict__ cursor,
                                                              const uint32_t *__restrict__ nS_next,
                                                              const uint32_t *__restrict__ base, uint32_t *__restrict__ A_next,
                                                              uint32_t *__restrict__ offA_next, uint32_t *__restrict__ L_next)
{
    extern __shared__ uint32_t offR[];  // Z: start of replica r's range inside every arrival bucket
    __shared__ uint32_t wsumA[kScatBlock / 64], wsumT[kScatBlock / 64];
    const int tid = threadIdx.x;
    const int rep = blockIdx.x % kRep, jblk = blockIdx.x / kRep;
    const int per = (Z + kScatBlock - 1) / kScatBlock;
    const int z0 = tid * per;
    // all loads first (independent), then the arithmetic: one memory round trip, not per*kRep
    uint32_t tot[kScatPerMax], pre[kScatPerMax], ns[kScatPerMax];
#pragma unroll
    for (int k = 0; k < kScatPerMax; ++k) {
        tot[k] = 0;
        pre[k] = 0;
        ns[k] = (k < per && z0 + k < Z) ? nS_next[z0 + k] : 0u;
    }
#pragma unroll
    for (int r = 0; r < kRep; ++r) {
#pragma unroll
        for (int k = 0; k < kScatPerMax; ++k) {
            uint32_t c = (k < per && z0 + k < Z) ? cursor[static_cast<size_t>(r) * Z + z0 + k] : 0u;
            tot[k] += c;
            if (r < rep) pre[k] += c;
        }
    }
    uint32_t mineA = 0, mineT = 0;
#pragma unroll
    for (int k = 0; k < kScatPerMax; ++k) {
        mineA += tot[k];
        mineT += tot[k] + ns[k];
    }
    uint32_t inclA = mineA, inclT = mineT;
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t va = __shfl_up(inclA, o, 64), vt = __shfl_up(inclT, o, 64);
        if ((tid & 63) >= o) {
            inclA += va;
            inclT += vt;
        }
    }
    if ((tid & 63) == 63) {
        wsumA[tid >> 6] = inclA;
        wsumT[tid >> 6] = inclT;
    }
    __syncthreads();
    uint32_t wa = 0, wt = 0, totA = 0, totT = 0;
#pragma unroll
    for (int w = 0; w < kScatBlock / 64; ++w) {
        uint32_t sa = wsumA[w], st = wsumT[w];
        if (w < (tid >> 6)) {
            wa += sa;
            wt += st;
        }
        totA += sa;
        totT += st;
    }
    uint32_t runA = wa + inclA - mineA, runT = wt + inclT - mineT;
#pragma unroll
    for (int k = 0; k < kScatPerMax; ++k) {
        int z = z0 + k;
        if (k < per && z < Z) {
            offR[z] = runA + pre[k];
            if (blockIdx.x == 0) {
                offA_next[z] = runA;
                L_next[z] = runT;
            }
            runA += tot[k];
            runT += tot[k] + ns[k];
        }
    }
    if (blockIdx.x == 0 && tid == 0) {
        offA_next[Z] = totA;
        L_next[Z] = totT;
    }
    __syncthreads();
    // sampler workgroups of replica rep: b = rep + kRep * i, i in [jblk*wgs_per_blk, ...)
    for (int i = jblk * wgs_per_blk; i < (jblk + 1) * wgs_per_blk; ++i) {
        const int b = rep + kRep * i;
        if (b >= nwg) break;
        const uint32_t n = nd[b];
        const uint2 *src = D + Lcur[min(b * zones_per_wg, Z)];
        const uint32_t *bb = base + static_cast<size_t>(b) * Z;
        constexpr int kU = 4;
        for (uint32_t i0 = 0; i0 < n; i0 += kScatBlock * kU) {
            uint2 v[kU];
            uint32_t g[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                uint32_t q = i0 + tid + u * kScatBlock;
                if (q < n) v[u] = src[q];
            }
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                uint32_t q = i0 + tid + u * kScatBlock;
                if (q < n) g[u] = bb[v[u].y & kDestMask];
            }
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                uint32_t q = i0 + tid + u * kScatBlock;
                if (q < n) A_next[offR[v[u].y & kDestMask] + g[u] + (v[u].y >> kDestBits)] = v[u].x;
            }
        }
    }
}

struct Zone2Work {
    bool attrs_set = false;
    bool buckets0_valid = false;  // A0/offA0 describe the context's current car state
    int64_t n = 0;
    int Z = 0, T = 0, nwg = 0, zones_per_wg = 0, block = 512, cpt = 3, ablate = 0;
    uint32_t *S[2] = {nullptr, nullptr};  // [n]
    uint32_t *A = nullptr, *A0 = nullptr;  // [n]
    uint2 *D = nullptr;                    // [n]
    uint32_t *L[2] = {nullptr, nullptr};   // [Z+1]
    uint32_t *nS[2] = {nullptr, nullptr};  // [Z]
    uint32_t *nS0 = nullptr;               // [Z] zeros
    uint32_t *offA = nullptr, *offA0 = nullptr;  // [Z+1]
    uint32_t *cursor = nullptr;            // [T][kRep][Z] ticket counters, then [Z] for the initial bucketing
    uint32_t *base = nullptr;              // [nwg][Z]
    uint32_t *nd = nullptr;                // [nwg]
    uint32_t *initbase = nullptr;          // [nb0][Z] ticket bases of the initial bucketing
    unsigned long long *dbg = nullptr;     // [8] diagnostic cycle sums
    int nb0 = 0;

    void release()
    {
        uint32_t **ps[] = {&S[0], &S[1], &A, &A0, &L[0], &L[1], &nS[0], &nS[1], &nS0, &offA, &offA0, &cursor, &base, &nd, &initbase};
        for (uint32_t **p : ps) {
            if (*p) (void)hipFree(*p);
            *p = nullptr;
        }
        if (D) (void)hipFree(D);
        D = nullptr;
        if (dbg) (void)hipFree(dbg);
        dbg = nullptr;
        n = 0;
        buckets0_valid = false;
    }

    hipError_t ensure(int64_t n_, int Z_, int T_, int cu_count, int wg_per_cu)
    {
        int want_wg = std::max(1, std::min(Z_, cu_count * wg_per_cu));
        int zpw = (Z_ + want_wg - 1) / want_wg;
        int nwg_ = (Z_ + zpw - 1) / zpw;
        if (n_ == n && Z_ == Z && T_ == T && nwg_ == nwg && S[0]) return hipSuccess;
        release();
        n = n_;
        Z = Z_;
        T = T_;
        nwg = nwg_;
        zones_per_wg = zpw;
        nb0 = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(cu_count, (n + 4095) / 4096)));
        hipError_t e = hipSuccess;
        auto alloc = [&](uint32_t **p, size_t words) {
            if (e == hipSuccess) e = hipMalloc(p, sizeof(uint32_t) * std::max<size_t>(words, 1));
        };
        alloc(&S[0], n);
        alloc(&S[1], n);
        alloc(&A, n);
        alloc(&A0, n);
        alloc(&L[0], Z + 1);
        alloc(&L[1], Z + 1);
        alloc(&nS[0], Z);
        alloc(&nS[1], Z);
        alloc(&nS0, Z);
        alloc(&offA, Z + 1);
        alloc(&offA0, Z + 1);
        alloc(&cursor, (static_cast<size_t>(T) * kRep + 1) * Z);
        alloc(&base, static_cast<size_t>(nwg) * Z);
        alloc(&nd, nwg);
        alloc(&initbase, static_cast<size_t>(nb0) * Z);
        if (e == hipSuccess) e = hipMalloc(&D, sizeof(uint2) * std::max<int64_t>(n, 1));
        if (e == hipSuccess) e = hipMemset(nS0, 0, sizeof(uint32_t) * Z);
        if (e == hipSuccess) e = hipMalloc(&dbg, sizeof(unsigned long long) * 8);
        if (e == hipSuccess) e = hipMemset(dbg, 0, sizeof(unsigned long long) * 8);
        if (e != hipSuccess) release();
        return e;
    }
};

inline int zone2_tree_height(int Z) { return tree_height(Z); }

inline size_t zone2_sample_lds(int Z) { return sizeof(double) * (size_t(1) << zone2_tree_height(Z)) + sizeof(uint32_t) * Z; }
inline bool zone2_path_fits(int Z)
{
    return Z <= (1 << kDestBits) && Z <= kScatBlock * kScatPerMax && zone2_sample_lds(Z) + 256 <= 160 * 1024;
}

template <bool TRAVEL, bool LAST, int BLOCK, int NP, int CPT, int ABL>
inline void zone2_launch_one(const Zone2Args &a, int nwg, size_t lds, hipStream_t stream)
{
    static bool attr_done = false;  // per instantiation: opt in to > 64 KiB of dynamic LDS once
    if (!attr_done && lds > 64 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_zone2_sample<TRAVEL, LAST, BLOCK, NP, CPT, ABL>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    hipLaunchKernelGGL((k_zone2_sample<TRAVEL, LAST, BLOCK, NP, CPT, ABL>), dim3(nwg), dim3(BLOCK), lds, stream, a);
}

template <bool TRAVEL, bool LAST, int BLOCK, int CPT, int ABL>
inline void zone2_launch_np(const Zone2Args &a, int nwg, size_t lds, hipStream_t stream)
{
    int need = (a.Zp / 2 + BLOCK - 1) / BLOCK;
    if (need <= 2) zone2_launch_one<TRAVEL, LAST, BLOCK, 2, CPT, ABL>(a, nwg, lds, stream);
    else if (need <= 4) zone2_launch_one<TRAVEL, LAST, BLOCK, 4, CPT, ABL>(a, nwg, lds, stream);
    else if (need <= 8) zone2_launch_one<TRAVEL, LAST, BLOCK, 8, CPT, ABL>(a, nwg, lds, stream);
    else zone2_launch_one<TRAVEL, LAST, BLOCK, 16, CPT, ABL>(a, nwg, lds, stream);
}

template <int BLOCK, int CPT, int ABL>
inline void zone2_launch_tl(const Zone2Args &a, bool travel, bool last, int nwg, size_t lds, hipStream_t stream)
{
    if (travel) {
        if (last) zone2_launch_np<true, true, BLOCK, CPT, ABL>(a, nwg, lds, stream);
        else zone2_launch_np<true, false, BLOCK, CPT, ABL>(a, nwg, lds, stream);
    } else {
        if (last) zone2_launch_np<false, true, BLOCK, CPT, ABL>(a, nwg, lds, stream);
        else zone2_launch_np<false, false, BLOCK, CPT, ABL>(a, nwg, lds, stream);
    }
}

inline void zone2_launch(const Zone2Work &w, const Zone2Args &a, bool travel, bool last, size_t lds, hipStream_t stream)
{
    // diagnostic instantiations (ABL 1/2/3: no search / no Philox; 4: in-kernel stamps) are built only
    // with -DCPM_DIAGNOSTIC; see tools/kbench.py
#ifdef CPM_DIAGNOSTIC
    if (w.ablate & 7) {
        switch (w.ablate & 7) {
        case 3: zone2_launch_tl<512, 2, 3>(a, travel, last, w.nwg, lds, stream); break;
        default: zone2_launch_tl<512, 2, 4>(a, travel, last, w.nwg, lds, stream); break;
        }
        return;
    }
    if (w.cpt == 3) {
        zone2_launch_tl<512, 3, 0>(a, travel, last, w.nwg, lds, stream);
        return;
    }
#endif
    zone2_launch_tl<512, 2, 0>(a, travel, last, w.nwg, lds, stream);
}

// ---------------------------------------------------------------------------------------------
// host side: one fused resample on `stream`
// ---------------------------------------------------------------------------------------------
template <typename F1, typename F2>
int32_t zone2_resample(Zone2Work &w, hipStream_t stream, const double *d_pdrive, const double *d_cdf, int Z, int Zp, int T,
                       int64_t n, int64_t car_begin, const uint32_t *d_zone0, uint64_t seed, bool travel,
                       const double *d_dm, int64_t *d_counts, int cu_count, F1 prof_begin, F2 prof_end, std::string &err)
{
    auto hip_fail = [&](hipError_t e, const char *what) {
        err = std::string(what) + ": " + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? CPM_ERR_NOMEM : CPM_ERR_HIP;
    };
    if (!zone2_path_fits(Z)) {
        err = "zone path: a CDF row of this many zones does not fit in LDS (use CPM_KERNEL_CAR)";
        return CPM_ERR_ARG;
    }
    if (n >= (int64_t(1) << 32)) {
        err = "zone path: more than 2^32 cars per GPU";
        return CPM_ERR_ARG;
    }
    const size_t lds = zone2_sample_lds(Z) + 16, lds_bins = sizeof(uint32_t) * static_cast<size_t>(Z);
    const int blk = 512;
    int wg_per_cu = std::max<int>(1, static_cast<int>((160 * 1024 - 512) / (lds + 64)));
    wg_per_cu = std::min(wg_per_cu, 2048 / blk);
    hipError_t e = w.ensure(n, Z, T, cu_count, wg_per_cu);
    if (e != hipSuccess) return hip_fail(e, "zone workspace");
    if (w.zones_per_wg > kMaxZonesPerWg) {
        err = "zone path: too many zones per workgroup for this device";
        return CPM_ERR_ARG;
    }
    if (!w.attrs_set) {
        if (lds_bins > 64 * 1024) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_zone_hist<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      static_cast<int>(lds_bins));
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_zone_scatter<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      static_cast<int>(lds_bins));
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_zone2_scatter), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      static_cast<int>(lds_bins));
        }
        w.attrs_set = true;
    }
    e = hipMemsetAsync(w.cursor, 0, sizeof(uint32_t) * (static_cast<size_t>(T) * kRep + 1) * Z, stream);
    if (e != hipSuccess) return hip_fail(e, "memset cursor");
    if (!w.buckets0_valid) {  // bucket the car-indexed state once; reused until the state changes
        const int64_t chunk = ((n + w.nb0 - 1) / w.nb0 + 3) / 4 * 4;
        uint32_t *cur0 = w.cursor + static_cast<size_t>(T) * kRep * Z;
        hipLaunchKernelGGL(k_zone_hist<0>, dim3(w.nb0), dim3(kSortBlock), lds_bins, stream, d_zone0, n, Z, chunk, cur0, w.initbase);
        hipLaunchKernelGGL(k_zone_scatter<0>, dim3(w.nb0), dim3(kSortBlock), lds_bins, stream, d_zone0,
                           static_cast<const uint32_t *>(nullptr), n, Z, chunk, cur0, w.initbase, w.A0, w.offA0);
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "initial bucketing");
        w.buckets0_valid = true;
    }
    unsigned long long *parking = reinterpret_cast<unsigned long long *>(d_counts);
    unsigned long long *driving = parking + static_cast<size_t>(T) * Z;
    // scatter grid: kRep replicas x blocks per replica; block (r, j) takes wgs_per_blk workgroups of replica r
    const int wg_per_rep = (w.nwg + kRep - 1) / kRep;
    const int blk_per_rep = std::max(1, std::min(wg_per_rep, std::max(1, cu_count / kRep)));
    const int wgs_per_blk = (wg_per_rep + blk_per_rep - 1) / blk_per_rep;
    const int scat_grid = kRep * ((wg_per_rep + wgs_per_blk - 1) / wgs_per_blk);
    Zone2Args a;
    a.dm = d_dm;
    a.D = w.D;
    a.nd = w.nd;
    a.base = w.base;
    a.status = parking + 2 * static_cast<size_t>(T) * Z + 1;
    a.dbg = w.dbg;
    a.tt_sum = parking + 2 * static_cast<size_t>(T) * Z;
    a.Z = Z;
    a.Zp = Zp;
    a.H = zone2_tree_height(Z);
    a.T = T;
    a.zones_per_wg = w.zones_per_wg;
    a.car_begin = car_begin;
    a.seed = seed;
    for (int t = 0; t < T; ++t) {
        const bool last = (t + 1 == T);
        a.pdrive_t = d_pdrive + static_cast<size_t>(t) * Z;
        a.cdf_t = d_cdf + static_cast<size_t>(t) * Z * Zp;
        a.t = t;
        a.step = static_cast<uint32_t>(T - 1 + t);
        a.parking_t = parking + static_cast<size_t>(t) * Z;
        a.driving_t = driving + static_cast<size_t>(t) * Z;
        a.cursor = w.cursor + static_cast<size_t>(t) * kRep * Z;
        if (t == 0) {
            a.S = w.S[1];  // empty: nS0 is all zero
            a.nS = w.nS0;
            a.L = w.offA0;
            a.A = w.A0;
            a.offA = w.offA0;
            a.Ln = w.offA0;  // zone sizes of hour 0 = the arrival counts of the initial bucketing
        } else {
            a.S = w.S[(t - 1) & 1];
            a.nS = w.nS[(t - 1) & 1];
            a.L = (t == 1) ? w.offA0 : w.L[(t - 2) & 1];
            a.A = w.A;
            a.offA = w.offA;
            a.Ln = w.L[(t - 1) & 1];
        }
        a.S_next = w.S[t & 1];
        a.nS_next = w.nS[t & 1];
        prof_begin(t);
        zone2_launch(w, a, travel, last, lds, stream);
        prof_end(t);
        if (!last)
            hipLaunchKernelGGL(k_zone2_scatter, dim3(scat_grid), dim3(kScatBlock), lds_bins, stream,
                               w.D, a.Ln, w.nd, w.zones_per_wg, w.nwg, wgs_per_blk, Z, a.cursor, a.nS_next, w.base, w.A, w.offA,
                               w.L[t & 1]);
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "zone hour launch");
    }
    return CPM_OK;
}

}  // namespace cpm
