// cpm_dataset.h -- the tables of ONE dataset (main.jl:79-85: createdatamatrix -> createpdrive, createpdestin) for SPARSE datamatrices.
//
// Real Uber Movement datamatrices hold 8.68 % of their (origin, destination, hour) cells (README.md:302-310); the reference and the
// dense builders of cpm_tables.h / cpm_grouped.h sweep the Z x Z x T arrays five times per dataset all the same (weights, row sums,
// division, row tables, travel rows: ~16 GB moved at Melbourne's size for a 2.1 GB datamatrix, profiles/round3_notes.md).  Here the
// datamatrix is swept ONCE (k_ds_cells): every (origin, destination) pair's 24 hourly means and standard deviations are loaded, the
// pair's extrema over the day taken (src/createpdestin.jl:10-16), and every cell that holds anything -- a mean, a standard deviation or
// a weight -- goes into the compact row of its (hour, origin): {x = (m - min) / (max - min), mean, std, destination}.  Everything else
// works on those rows, 9 % of the data:
//   * k_ds_sort    : a row's cells by destination (they arrive in the order of the atomic tickets), and with them the travel rows
//                    the travel kernel stages (bitmap + cells in front, then (mean, sigma, window mass): cpm_grouped.h);
//   * k_ds_pdrive  : mean_sum[i,t] = mean over non-zero j of mean / dist (src/createpdrive.jl:14-20): the reference's loop skips
//                    the zero cells itself, so the sequential sum over the row's cells in destination order IS its sum;
//   * k_ds_pdest   : weights x^e_dest, the row sum, the division, the running sum (src/createpdestin.jl:19-46, src/resampling.jl:39)
//                    -- all sequential sums over the row's cells in destination order: adding the zeros in between changes
//                    nothing (x + 0.0 == x), so every value equals the dense builders', bit for bit -- and the row PACK of the
//                    grouped sampler in its sparse form (below), the row total, the normalised p of every cell (for ties);
//   * k_ds_dense_p : p_destin in the reference's layout, only when somebody asks for it (cpm_build_p_dest's out argument, a kernel
//                    family that searches f64 rows).
// A model-selection point with a new e_dest re-runs k_ds_pdest alone (0.1 ms instead of a createpdestin of 1.7).
//
// Sparse row pack (what the grouped sampler stages on these tables: a fifth of the dense pack's bytes at Melbourne's density).  A row's
// CDF only steps at destinations that hold weight, so the pack keeps those steps only: entry e = the e-th cell of the row (by
// destination): hi[e] = floor(cdf * 2^32) there, idx[e] = its destination.  "First j with hi[j] >= khi" over the dense row IS
// idx[first e with hi[e] >= khi]: between two cells the dense high words repeat the left one's.  The pack is a dense pack of the
// COMPACT row (guide over entries, high words padded with the row's last value up to the longest row of the table and with
// 0xFFFFFFFF behind it) plus the u16 map idx: pack_search walks it unchanged and maps the answer at the end; a tie (probability
// ~ n * 2^-32) repeats the reference's walk over the row's cells (search_exact_sparse).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cpm_grouped.h"
#include "cpm_tables.h"

namespace cpm {

// A cell of a dataset's compact rows
struct alignas(16) DsCell {
    double x, m, sd;  // (m - min_t m) / (max_t m - min_t m) of the pair (0 when max_t m <= 0: src/createpdestin.jl:19-28), mean, std
    uint32_t j, pad;  // destination (0-based)
};
constexpr uint32_t kDsCap = 512;    // cells a row can hold (22 % of Melbourne's 2,357 zones; the table takes the dense path when a row outgrows it)
constexpr int kDsJBc = 4;           // destinations per block of the sweep: one per wave
constexpr int kDsThreads = 64;      // one wave per row in the per-row kernels ...
constexpr int kDsRows = 4;          // ... and four rows per block: one-wave blocks are bound by the dispatch of workgroups (56,568 of them: ~0.2 ms
                                    // per kernel at Melbourne's size, whatever they did)
// the sweep's buffer loads address a (means or standard deviations) half of the datamatrix with 32-bit offsets
inline bool ds_fits(int64_t Z, int64_t T)
{
    return T == 24 && Z <= 65535 && T * Z * Z * 8 < (int64_t(1) << 32) && T * Z * static_cast<int64_t>(kDsCap) * 32 < (int64_t(1) << 32);
}
inline unsigned ds_grid(int64_t rows) { return static_cast<unsigned>((rows + kDsRows - 1) / kDsRows); }

// x^e as createpdestin raises it (cpm_tables.h): Float64^Int by squaring (main.jl:38: e_dest = 2), Float64^Float64 otherwise
__device__ __forceinline__ double ds_weight(double x, double e_dest, int e_is_integer)
{
    return e_is_integer ? pow_int(x, static_cast<long>(e_dest)) : pow_f64(x, e_dest);
}

// ONE sweep of the datamatrix: thread = (origin i on the lane: the reference's fastest index, every load a whole 512-B line;
// destination j: one per wave).  The pair's 24 hourly means are loaded once, all in flight, and stay in registers: extrema over the
// day (src/createpdestin.jl:10-16, NaN-propagating), then the cells in four turns of six hours -- ticket in the row of (hour, origin)
// and the standard deviation (loaded only where a cell is kept: 9 % of the cells, about half the sectors) asked for together, then
// the 32-byte stores.  A cell is kept when it holds a mean or a weight.  (A cell with a standard deviation alone -- no mean, weight 0
// -- is never a destination, src/createpdestin.jl:19-28, so no driver ever looks its travel time up; tests/test_gpu_parity.py plants one.)
// MEASURED (Melbourne's size, rocprofv3; profiles/round4_notes.md): 1.2 ms in four different shapes of this kernel (all values of a
// pair in registers, 200 VGPRs; the day's quarters in four waves with the extrema through LDS; the means loaded again per turn, 74
// VGPRs; this one, 136) and ~1.05 ms with tickets, deviations and stores compiled out of each: the time was the ONE statistics word at
// the end (below).  With that fixed: 0.89 ms = the sweep and its extrema 0.41 + tickets and deviations 0.23 + cell stores 0.25.
// stats[0] = longest row so far (atomicMax of ticket + 1), stats[1] = 1 when a row outgrew cap.
constexpr int kDsJB = kDsJBc;
__global__ __launch_bounds__(256) void k_ds_cells(const double *__restrict__ dm, DsCell *__restrict__ cells, uint32_t *__restrict__ cnt, int Z,
                                                  uint32_t cap, uint32_t *__restrict__ stats)
{
    constexpr int TT = 24, TQ = 6;
    const int lane = threadIdx.x & 63;
    // (Dealing the origin blocks to XCDs -- every row filled from ONE XCD, whose L2 gathers the row's 32-byte cells into whole lines --
    //  was measured: the stores went from 0.25 to 0.12 ms and the tickets, then all of a row's from one XCD, from 0.24 to 0.44.)
    const int bx = blockIdx.x, by = blockIdx.y;
    const int i = bx * 64 + lane;
    const int j = __builtin_amdgcn_readfirstlane(static_cast<int>(by * kDsJB + (threadIdx.x >> 6)));
    if (j >= Z) return;  // (wave-uniform)
    const bool live = i < Z;
    // (buffer loads: descriptor on this destination's column, the hour's slab a scalar offset, the lane's origin one VGPR -- with 64-bit
    //  addresses per load the 48 of them were 96 VGPRs of addresses.  T * Z * Z * 8 B < 4 GiB: ds_fits)
    const size_t slab = static_cast<size_t>(Z) * Z;
    const uint32_t slab_b = static_cast<uint32_t>(slab * 8);
    const __amdgpu_buffer_rsrc_t means = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(dm + static_cast<size_t>(j) * Z), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t sds = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(dm + slab * TT + static_cast<size_t>(j) * Z), 0, -1, 0x00020000);
    const uint32_t lane_off = static_cast<uint32_t>(live ? i : 0) << 3;
    // (tickets and cell stores likewise: the hour's part of the offset scalar, the lane's a 32-bit VGPR -- T * Z * cap * 32 B < 4 GiB: ds_fits)
    const __amdgpu_buffer_rsrc_t tickets = __builtin_amdgcn_make_buffer_rsrc(cnt, 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t out = __builtin_amdgcn_make_buffer_rsrc(cells, 0, -1, 0x00020000);
    const uint32_t row_b = static_cast<uint32_t>(Z) * cap * static_cast<uint32_t>(sizeof(DsCell));  // bytes of an hour's rows
    double v[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) v[t] = row_load(means, lane_off, static_cast<uint32_t>(t) * slab_b);
    __builtin_amdgcn_sched_barrier(0);  // (every load of the pair leaves before the first is used)
    double mx = v[0], mn = v[0];
#pragma unroll
    for (int t = 1; t < TT; ++t) {
        mx = jl_max(mx, v[t]);
        mn = jl_min(mn, v[t]);
    }
    const bool pos = mx > 0;
    const double den = mx - mn;
    uint32_t longest = 0;
#pragma unroll
    for (int q = 0; q < TT / TQ; ++q) {
        // six hours' cells take their tickets and ask for their standard deviations TOGETHER, then the stores.  (The turns are fenced
        // off from each other: scheduled as one, their quotients, addresses and tickets were all alive at once -- 138 VGPRs.  Loading
        // the means again per turn instead of holding them -- 74 VGPRs -- doubled the traffic: they no longer hit the L2.)
        __builtin_amdgcn_sched_barrier(0);
        uint32_t slot[TQ];
        double x[TQ], sd[TQ];
#pragma unroll
        for (int u = 0; u < TQ; ++u) {
            const int t = q * TQ + u;
            x[u] = pos ? (v[t] - mn) / den : 0.0;
            slot[u] = 0xFFFFFFFFu;
            sd[u] = 0.0;
            if (live && (x[u] != 0.0 || v[t] != 0.0)) {  // (NaN != 0: a pair constant over the day keeps its NaN, src/createpdestin.jl:24, Appendix A-5)
#ifdef CPM_DS_NOATOMIC  // (timing-only ablation builds, tools/build_variants.sh: never the product)
                slot[u] = static_cast<uint32_t>(j) & 511u;
#else
                slot[u] = static_cast<uint32_t>(__builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, tickets, lane_off >> 1, static_cast<uint32_t>(t) * static_cast<uint32_t>(Z) * 4u, 0));
#endif
#ifndef CPM_DS_NOSD
                sd[u] = row_load(sds, lane_off, static_cast<uint32_t>(t) * slab_b);
#endif
            }
        }
#pragma unroll
        for (int u = 0; u < TQ; ++u) {
            const int t = q * TQ + u;
            if (slot[u] != 0xFFFFFFFFu) {
                longest = max(longest, slot[u] + 1u);
#ifdef CPM_DS_NOSTORE
                if (slot[u] == 0xFFFFFFFEu) {
#else
                if (slot[u] < cap) {
#endif
                    // {x, m | sd, j, pad}: two 16-byte stores into the cell's 32-byte sector
                    const uint32_t off = (static_cast<uint32_t>(i) * cap + slot[u]) * static_cast<uint32_t>(sizeof(DsCell));
                    cpm_u32x4 lo4, hi4;
                    lo4.x = static_cast<uint32_t>(__double2loint(x[u]));
                    lo4.y = static_cast<uint32_t>(__double2hiint(x[u]));
                    lo4.z = static_cast<uint32_t>(__double2loint(v[t]));
                    lo4.w = static_cast<uint32_t>(__double2hiint(v[t]));
                    hi4.x = static_cast<uint32_t>(__double2loint(sd[u]));
                    hi4.y = static_cast<uint32_t>(__double2hiint(sd[u]));
                    hi4.z = static_cast<uint32_t>(j);
                    hi4.w = 0u;
                    __builtin_amdgcn_raw_buffer_store_b128(lo4, out, off, static_cast<uint32_t>(t) * row_b, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(hi4, out, off + 16u, static_cast<uint32_t>(t) * row_b, 0);
                }
            }
        }
    }
    // (the lanes' maxima meet in lane 0, which only then looks at the word: an atomicMax per wave on ONE address -- 86,800 of them, served
    //  one after the other at ~12 ns -- WAS this kernel: 1.05 ms whatever else it did; the running maximum settles within the first waves)
    for (int o = 32; o > 0; o >>= 1) longest = max(longest, static_cast<uint32_t>(__shfl_down(longest, o, 64)));
    if (lane == 0 && longest > __hip_atomic_load(&stats[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        atomicMax(&stats[0], longest);
        if (longest > cap) atomicOr(&stats[1], 1u);
    }
}

// A sequential f64 sum of n LDS values, eight loads in flight in front of their additions (one load per addition is a round trip
// through the LDS pipeline per term: the per-row kernels below are chains of such sums)
template <typename F>
__device__ __forceinline__ void ds_seq_walk(uint32_t n, const double *v, F &&step)
{
    uint32_t e = 0;
    for (; e + 8 <= n; e += 8) {
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = v[e + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) step(e + u, x[u]);
    }
    for (; e < n; ++e) step(e, v[e]);
}

// A row's cells in destination order, in place -- the rank of a cell is the number of cells with a smaller destination, which the
// row's bitmap knows (destinations are distinct): cells in front of its word + the bits below its own -- and the row as the travel
// kernel stages it: words[w] = (bitmap of destinations 32 w .. 32 w + 31 that hold a cell, cells in front of word w), cells (mean,
// sigma, mass of the window) with sigma = std, or a tenth of the mean where the data hold none (src/resampling.jl:65-67).
__global__ __launch_bounds__(kDsThreads * kDsRows) void k_ds_sort(DsCell *__restrict__ cells, const uint32_t *__restrict__ cnt, uint32_t cap, int64_t rows, int Z,
                                                                  uint2 *__restrict__ words, int W, TravelCell *__restrict__ tcells)
{
    extern __shared__ uint32_t ds_bits[];  // per wave: [W] bitmap, then [W] cells in front
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *bits = ds_bits + static_cast<size_t>(wave) * 2 * W;
    const size_t row = static_cast<size_t>(blockIdx.x) * kDsRows + wave;
    const bool live = row < static_cast<size_t>(rows);
    const uint32_t n = live ? min(cnt[row], cap) : 0u;
    DsCell *g = cells + (live ? row : 0) * cap;
    for (int w = lane; w < 2 * W; w += 64) bits[w] = 0;
    __syncthreads();
    constexpr int kMine = kDsCap / 64;  // cells a lane holds (in registers: the row is sorted in place)
    DsCell mine[kMine];
#pragma unroll
    for (int k = 0; k < kMine; ++k) {
        const uint32_t e = static_cast<uint32_t>(lane) + 64u * k;
        if (e < n) {
            mine[k] = g[e];
            atomicOr(&bits[mine[k].j >> 5], 1u << (mine[k].j & 31u));
        }
    }
    __syncthreads();  // (every cell of the row is in a register: the stores below may land anywhere in it)
    {   // cells in front of every word: each lane its stretch of words, the stretches' totals scanned across the wave
        const int per = (W + 63) / 64, w0 = lane * per, w1 = min(W, w0 + per);
        uint32_t sum = 0;
        for (int w = w0; w < w1; ++w) sum += static_cast<uint32_t>(__popc(bits[w]));
        uint32_t run = wave_incl_scan(sum) - sum;
        for (int w = w0; w < w1; ++w) {
            bits[W + w] = run;
            run += static_cast<uint32_t>(__popc(bits[w]));
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kMine; ++k) {
        const uint32_t e = static_cast<uint32_t>(lane) + 64u * k;
        if (e < n) {
            const DsCell c = mine[k];
            const uint32_t rank = bits[W + (c.j >> 5)] + static_cast<uint32_t>(__popc(bits[c.j >> 5] & ((1u << (c.j & 31u)) - 1u)));
            g[rank] = c;
            const double s1 = (c.sd == 0) ? 0.1 * c.m : c.sd;
            TravelCell tc;
            tc.mu = c.m;
            tc.sigma = s1;
            tc.mass = truncnormal_mass(c.m, s1);
            tcells[row * cap + rank] = tc;
        }
    }
    if (live)
        for (int w = lane; w < W; w += 64) words[row * W + w] = make_uint2(bits[w], bits[W + w]);
}

// mean_sum[i,t] of createpdrive (src/createpdrive.jl:10-21) from the sorted rows: the quotients in parallel, their sum in order (a cell
// without a mean adds 0.0: exact, the reference's loop skips it), the counter by a vote
__global__ __launch_bounds__(kDsThreads * kDsRows) void k_ds_pdrive(const DsCell *__restrict__ cells, const uint32_t *__restrict__ cnt, uint32_t cap, int64_t rows,
                                                                    int Z, const double *__restrict__ dist, double *__restrict__ mean_sum)
{
    __shared__ double q_all[kDsRows][kDsCap];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *q = q_all[wave];
    // rows in ORIGIN-major order: the 24 hours of an origin gather from the same column of dist (a line per destination, ~500 of them
    // hold data): side by side they find it in L2 (hour-major, every quotient was a line from memory: 0.25 ms at Melbourne's size)
    const size_t k = static_cast<size_t>(blockIdx.x) * kDsRows + wave;
    const bool live = k < static_cast<size_t>(rows);
    const int T = static_cast<int>(rows / Z);
    const int i = static_cast<int>(k / static_cast<size_t>(T));
    const size_t row = live ? static_cast<size_t>(k % static_cast<size_t>(T)) * Z + i : 0;
    const uint32_t n = live ? min(cnt[row], cap) : 0u;
    const DsCell *g = cells + row * cap;
    long long counter = 0;
    for (uint32_t e0 = 0; e0 < n; e0 += 64) {  // (wave-uniform trips: the vote below)
        const uint32_t e = e0 + lane;
        bool has = false;
        if (e < n) {
            const double m = g[e].m;
            has = m != 0;
            q[e] = has ? m / dist[i + static_cast<size_t>(g[e].j) * Z] : 0.0;
        }
        counter += __popcll(ballot64(has));
    }
    __syncthreads();
    if (lane == 0 && live) {
        double s = 0.0;
        ds_seq_walk(n, q, [&](uint32_t, double v) { s = s + v; });
        mean_sum[row] = s / static_cast<double>(counter);  // 0/0 = NaN (Appendix A-3)
    }
}

// geometry of the sparse pack of a table whose longest row holds nc cells
__host__ __device__ inline int sparse_pack_words(int Zq_c, int G_c) { return pack_row_words(Zq_c, G_c, 1); }

// createpdestin for one (hour, origin) from its sorted cells + everything the grouped sampler reads of the row.
// last_out[row] = the row total; sp / sj / scnt = normalised p, destination and number of the row's cells (what a tie walks: owned by the
// TABLE, so that it stays whole when the next dataset's cells arrive); pack: [guide u16][hi u32 x Zq_c][idx u16 x Zq_c].
__global__ __launch_bounds__(kDsThreads * kDsRows) void k_ds_pdest(const DsCell *__restrict__ cells, const uint32_t *__restrict__ cnt, uint32_t cap, int64_t rows, int Z,
                                                                   double e_dest, int e_is_integer, int nc, int Zq_c, int G_c, uint32_t *__restrict__ rp,
                                                                   double *__restrict__ last_out, double *__restrict__ sp, uint32_t *__restrict__ sj,
                                                                   uint32_t *__restrict__ scnt, int *err)
{
    __shared__ double w_all[kDsRows][kDsCap];  // weights, then normalised p, then the running sum
    __shared__ uint32_t hi_all[kDsRows][kDsCap + 64];
    __shared__ double s_nf[kDsRows];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *w = w_all[wave];
    uint32_t *hi = hi_all[wave];
    const size_t row = static_cast<size_t>(blockIdx.x) * kDsRows + wave;
    const bool live = row < static_cast<size_t>(rows);
    const uint32_t n = live ? min(cnt[row], cap) : 0u;
    const DsCell *g = cells + (live ? row : 0) * cap;
    for (uint32_t e = lane; e < n; e += 64) w[e] = ds_weight(g[e].x, e_dest, e_is_integer);
    __syncthreads();
    if (lane == 0) {  // nf = sum_j p[i,j,t], left to right (cpm_tables.h: k_pdest_rowsum)
        double nf = 0.0;
        ds_seq_walk(n, w, [&](uint32_t, double v) { nf = nf + v; });
        s_nf[wave] = nf;
    }
    __syncthreads();
    const double nf = s_nf[wave];
    bool bad = false;
    for (uint32_t e = lane; e < n; e += 64) {
        const double p = (nf > 0) ? w[e] / nf : w[e];  // src/createpdestin.jl:38-46
        bad |= !(p >= 0.0);
        w[e] = p;
        sp[row * cap + e] = p;
        sj[row * cap + e] = g[e].j;
    }
    if (lane == 0 && live) scnt[row] = n;
    if (bad) atomicOr(err, 1);  // NaN or negative entries: CPM_ERR_TABLE
    __syncthreads();
    if (lane == 0 && live) {  // range_up = range_up + distribution[j] (src/resampling.jl:39)
        double run = 0.0;
        ds_seq_walk(n, w, [&](uint32_t e, double v) {
            run = run + v;
            w[e] = run;
        });
        last_out[row] = run;
    }
    __syncthreads();
    uint32_t *pack = rp + (live ? row : 0) * static_cast<size_t>(sparse_pack_words(Zq_c, G_c));
    const int gw = pack_guide_words(G_c);
    uint32_t *hi_g = pack + gw;
    uint16_t *idx_g = reinterpret_cast<uint16_t *>(hi_g + Zq_c);
    const uint32_t v_last = n ? ((w[n - 1] < 1.0) ? static_cast<uint32_t>(floor(w[n - 1] * 0x1.0p32)) : kHiMax) : kHiMax;
    for (int e = lane; e < Zq_c; e += 64) {
        uint32_t h = kHiMax;
        if (e < static_cast<int>(n)) h = (w[e] < 1.0) ? static_cast<uint32_t>(floor(w[e] * 0x1.0p32)) : kHiMax;
        else if (e < nc) h = v_last;
        if (e < static_cast<int>(kDsCap) + 64) hi[e] = h;
        if (live) {
            hi_g[e] = h;
            idx_g[e] = static_cast<uint16_t>(e < static_cast<int>(n) ? g[e].j : (n ? g[n - 1].j : 0u));
        }
    }
    __syncthreads();
    // guide[m] = min(first e with hi[e] >= m << sh, nc - 1), m = 0 .. 2^G + 7 (the pad entries behind 2^G hold nc - 1)
    const int sh = 32 - G_c;
    uint16_t *guide = reinterpret_cast<uint16_t *>(pack);
    if (live)
        for (int m = lane; m < (1 << G_c) + 8; m += 64) {
            const unsigned long long key = static_cast<unsigned long long>(m) << sh;
            int lo = 0, len = nc;  // first e in [0, nc) with hi[e] >= key
            while (len > 0) {
                const int half = len >> 1;
                if (static_cast<unsigned long long>(hi[lo + half]) < key) {
                    lo += half + 1;
                    len -= half + 1;
                } else {
                    len = half;
                }
            }
            guide[m] = static_cast<uint16_t>(min(lo, nc - 1));
        }
}

// p_destin in the reference's layout from the rows' normalised cells (the array zeroed by the caller)
__global__ __launch_bounds__(kDsThreads * kDsRows) void k_ds_dense_p(const double *__restrict__ sp, const uint32_t *__restrict__ sj, const uint32_t *__restrict__ scnt,
                                                                     uint32_t cap, int64_t rows, int Z, double *__restrict__ p)
{
    const size_t row = static_cast<size_t>(blockIdx.x) * kDsRows + (threadIdx.x >> 6);
    if (row >= static_cast<size_t>(rows)) return;
    const size_t t = row / static_cast<size_t>(Z), i = row % static_cast<size_t>(Z);
    const uint32_t n = min(scnt[row], cap);
    for (uint32_t e = threadIdx.x & 63; e < n; e += 64) p[(t * Z + sj[row * cap + e]) * Z + i] = sp[row * cap + e];
}

}  // namespace cpm
