// cpm_day.h -- the hours of a resample (or of the initial-value problem) in ONE launch: k_grouped_day.
//
// src/resampling.jl:7-85 is a loop over hours with a loop over cars inside; cpm_grouped.h runs one hour per launch (sampler workgroups,
// then the placing blocks that move the hour's drivers into next hour's buckets).  What that costs per hour (profiles/round3_notes.md):
// a tail of ~1.6 placing-block lifetimes in which the vector units idle, and a launch boundary.  Here the grid is the whole chain:
//
//     segment of hour t  =  the placing blocks P(t-1) that move hour t-1's drivers into hour t's buckets,
//                           among the sampler workgroups S(t) of hour t                       (t = 0: no placing blocks)
//
// and a block only ever waits for blocks of LOWER index -- which the dispatcher has started before it -- so the chain cannot
// deadlock however the blocks are placed (every wait is bounded all the same; see "giving up" below):
//
//   * S(t) of zone z, FIRST HALF: the cars that stayed in z at hour t-1.  They were written by S(t-1) of the same zone, long done
//     when this block starts: it waits for that workgroup's flag (the stayer-count word, kCntValid), stages the zone's row pack,
//     draws for the stayers.  This is the work that runs while P(t-1) is still moving the drivers: the placing chain of an hour
//     (memory round trips, idle vector units) lies under the sampler issue of the next.
//   * S(t) of zone z, SECOND HALF: the cars that arrived.  Waits for the placing blocks of the zone's destination group
//     (pdone[g]: all chunks counted in), reads the arrivals from the top of the zone's region, draws for them.
//   * P(t-1) of (group g, chunk j of origin zones): waits for the chunk's sampler workgroups of hour t-1 (psdone[j]), as in
//     k_grouped_hour; counts itself in on pdone[g] when its stores have drained, as in k_grouped_hour_pf.
//
// Visibility (MI355X_MICROARCH.md, inter-workgroup visibility).  Runs and run lengths cross XCDs: stored sc1 (write-through),
// drained, barrier, agent-scope add; read with sc1 loads behind the poll + barrier -- the form k_grouped_hour uses.  A zone's
// bucket region never leaves its XCD: its stayers are written by the zone's own sampler workgroup, its arrivals by the placing
// blocks of its group, and the block order deals all of them to blockIdx % 8 = g % 8; they are stored plainly (the lines stay in
// that XCD's L2), drained before the flag / counter, and read past the L1 (sc1 loads are served by the L2).  Dispatch placement is
// not a HIP guarantee: every writer leaves its XCC_ID beside its flag / counter and a reader that finds another XCD's gives up.
//
// Giving up.  A wait that runs out raises status bit 2 (4), and the block reads NOTHING: it takes its input for empty, stores zero
// run lengths, and marks its own hand-offs (kCntAbort / kDoneAbort), so that whoever waits for it gives up at once as well: the grid
// drains in microseconds, no stale word is ever used as an id or a destination, and the context repeats the step with one launch per
// hour (cpm_api.hip: absorb_status).  tests/test_gpu_parity.py drives it with a spin limit of 0.
//
// Buffers: buckets ping-pong as before (S(t+1) of a zone overwrites what S(t) of the same zone read: it has waited for it; P(t+1)
// overwrites tops that S(t) read: it has waited for S(t+1) of its chunk, which waited for P(t), which waited for all of S(t)).
// Runs and run lengths rotate over THREE copies: S(t+2)'s first half only waits for S(t+1) of its own zone, and P(t) of another
// group may still be reading the zone's runs of hour t.
#pragma once
#include "cpm_grouped.h"

namespace cpm {

constexpr int kDayRunCopies = 3;

// ids in flight, then their wait: the registers travel through the wait as operands, so no use can move above it
template <int N>
__device__ __forceinline__ void wait_all(uint32_t &x, uint32_t (&id)[N])
{
    static_assert(N == 2 || N == 3 || N == 5, "written for CPT = 1, 2, 4");
    if constexpr (N == 5)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(x), "+v"(id[0]), "+v"(id[1]), "+v"(id[2]), "+v"(id[3]), "+v"(id[4])::"memory");
    else if constexpr (N == 3)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(x), "+v"(id[0]), "+v"(id[1]), "+v"(id[2])::"memory");
    else
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(x), "+v"(id[0]), "+v"(id[1])::"memory");
}

// One sampler workgroup of the day launch: zone z of destination group g, hour a.hour.
// Two ways through it.  MERGED: when the block starts, the placing blocks of its group are already done (every set of groups but the
// first of a segment: their placing blocks sit a whole window in front) -- the bucket is complete, and the block is the hourly
// sampler: all its cars in one pass, CPT per lane.  SPLIT: the placing blocks are still at work (the first set of a segment: they
// wait for the last sampler workgroups of the hour before) -- first the stayers, then the wait, then the arrivals.
template <int BLOCK, int CPT, int NQ, bool SPARSE>
__device__ __forceinline__ void day_sample_body(const GroupedArgs &a, const int z, const int g, uint32_t *pack, SampleLds &sl)
{
    uint32_t &s_nstay = sl.nstay, &s_bcast = sl.pad_, &s_ready = sl.ndrive;
    uint32_t(&gb)[kGroups] = sl.gb;
    uint32_t(&stage)[kGroups * kStage] = sl.stage;
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t cap = a.cap;
    const uint32_t b = static_cast<uint32_t>(z) * cap;
    const int gw = pack_guide_words(a.G), rw = pack_row_words(a.Zq, a.G, SPARSE ? 1 : 0), pieces = rw / 4, sh = 32 - a.G;
    const double last = a.last_t[z];
    const long long thr = a.thr_t[z];
    const uint32_t myx = xcc_id();
    const bool w0 = tid < 64;  // the wave that polls for the workgroup (its verdicts reach the others through LDS, behind a barrier)
    CPM_SSTAMP_DECL;
    CPM_SSTAMP(0);
    const uint32_t *pdone_g = a.pdone + static_cast<size_t>(g) * kDoneStride;
    const uint32_t need = static_cast<uint32_t>(a.pchunks);
    // ---- the stayers' count word (flags beside it when the zone's workgroup of the hour before belongs to this launch) and the
    // counter of the group's placing blocks: requested first, then this wave's pieces of the pack; with vmcnt <= NQ both are in their
    // registers while the pack is still landing
    {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        uint32_t w = 0;
        u32x2 seen = {0u, 0u};
        if (w0) {
            asm volatile("global_load_dword %0, %1, %2 sc1" : "=v"(w) : "v"(0), "s"(a.cnt_s + z) : "memory");
            if (need) asm volatile("global_load_dwordx2 %0, %1, %2 sc1" : "=v"(seen) : "v"(0), "s"(pdone_g) : "memory");
        }
        pack_dma<BLOCK, NQ>(pack, a.rp_t + static_cast<size_t>(z) * rw, pieces, tid);
        if (w0) {
            asm volatile("s_waitcnt vmcnt(%2)" : "+v"(w), "+v"(seen) : "n"(NQ) : "memory");
            uint32_t nsw = from_lane0(w);
            if (a.chained) {
                for (uint32_t spins = 0; !(nsw & kCntValid); ++spins) {  // (rare: that workgroup was dispatched a whole segment earlier)
                    if (spins >= a.spin_limit) break;
                    __builtin_amdgcn_s_sleep(32);
                    nsw = from_lane0(lane == 0 ? __hip_atomic_load(a.cnt_s + z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u);
                }
                const bool late = !(nsw & kCntValid), moved = ((nsw >> kCntXccShift) & 15u) != myx;
                if ((late || (!(nsw & kCntAbort) && moved)) && lane == 0) atomicOr(a.rare->status, 4ull);
                if (late || moved) nsw = kCntAbort;
            } else {
                nsw &= kCntMask;  // (a bucketing cached from an earlier run: its flags mean nothing here)
            }
            // the group's placing blocks: all counted in, none of them marked, all on this XCD -> the bucket is complete (merged pass)
            const uint32_t got = from_lane0(seen.x), where = from_lane0(seen.y);
            const bool ready = !need || ((got & kDoneCount) >= need && (got >> 16) == 0u && where == (1u << myx));
            if (tid == 0) {
                s_bcast = nsw;
                s_ready = ready ? 1u : 0u;
                s_nstay = 0;
            }
        }
        if (tid < kGroups) gb[tid] = 0;
    }
    lds_barrier();  // (LDS only: the pack stays in flight)
    CPM_SSTAMP(1);
    const uint32_t nsw = s_bcast;
    const bool merged = s_ready != 0u;
    bool dead = (nsw & kCntAbort) != 0u;
    const uint32_t ns = dead ? 0u : min(nsw & kCntMask, cap);
    uint32_t na = 0, n_all = ns;
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t *stay_out = a.ids_next + static_cast<size_t>(z) * cap;
    uint32_t *runs = a.D + static_cast<size_t>(z) * kGroups * a.scap;
    const uint16_t *guide = reinterpret_cast<const uint16_t *>(pack);
    const uint32_t *hi = pack + gw;
    const uint16_t *smap = SPARSE ? reinterpret_cast<const uint16_t *>(hi + a.Zq) : nullptr;
    uint32_t hi_last = 0;
    const CarIndex cars = a.cars;
    const uint32_t top4 = (cap - 1u) << 2;
    // the arrivals' count is known: the zone's parking count, the overflow check, heavy buckets
    auto bucket_known = [&](uint32_t na_raw) {
        if (tid == 0) {
            a.parking_t[z] = n_all;  // every car present at hour t, drivers included (src/saveresults.jl:12)
            if (static_cast<unsigned long long>(ns) + na_raw > cap) atomicOr(a.rare->status, 2ull);  // the two ends of the bucket met: step invalid
            if (n_all > a.heavy_x * CPT * BLOCK) {  // a heavy bucket: the context learns of it and leaves the one-launch form (k_grouped_sample_heavy)
                const GroupedRare *r = a.rare;
                const uint32_t items = (n_all + CPT * BLOCK - 1) / (CPT * BLOCK) - 1u;
                atomicMax(&r->maxn[0], n_all);
                const uint32_t idx = atomicAdd(r->nheavy + a.hour, items);
                atomicMax(&r->maxn[1], idx + items);
            }
        }
    };
#pragma unroll 1
    for (int ph = 0; ph < 2; ++ph) {
        // slot s of this pass: stayer s below ns_here, else arrival s - ns_here, which lies at position cap - 1 - (s - ns_here) -- a
        // position that does not depend on the arrivals' count, so count and ids are requested together
        uint32_t nseg, id[CPT + 1], nav = 0;
        const uint32_t ns_here = ph == 0 ? ns : 0u;
        if (ph == 1) {
            // ---- SPLIT: the arrivals, once the placing blocks of group g have all counted themselves in
            if (w0) {
                uint32_t bad = dead ? 1u : 0u;
                if (!dead) {
                    uint32_t got = 0, where = 0;
                    for (uint32_t spins = 0;; ++spins) {
                        got = from_lane0(lane == 0 ? __hip_atomic_load(pdone_g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u);
                        if ((got & kDoneCount) >= need || spins >= a.spin_limit) break;
                        __builtin_amdgcn_s_sleep(32);
                    }
                    // (the mask is complete once the count is: a block ORs its XCD in before it counts itself in)
                    where = from_lane0(lane == 0 ? __hip_atomic_load(pdone_g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u);
                    const bool late = (got & kDoneCount) < need, marked = (got >> 16) != 0u, moved = where != (1u << myx);
                    if ((late || (!marked && moved)) && lane == 0) atomicOr(a.rare->status, 4ull);
                    bad = (late || marked || moved) ? 1u : 0u;
                }
                if (tid == 0) s_bcast = bad;
            }
            lds_barrier();  // (the polling wave's loads come after its poll matched, the other waves' after this barrier)
            CPM_SSTAMP(4);
            dead = s_bcast != 0u;
        }
        const bool with_arrivals = (ph == 1 || merged) && !dead;
        {
            uint32_t s4 = static_cast<uint32_t>(tid) << 2;
            asm volatile("" : "+v"(s4));  // (opaque: hoisted out of the two-trip loop, the offsets of both passes lived across it -- in scratch)
            const uint32_t ns4 = ns_here << 2;
            if (with_arrivals) asm volatile("global_load_dword %0, %1, %2 sc1" : "=v"(nav) : "v"(0), "s"(a.cnt_a + z) : "memory");
#pragma unroll
            for (int c = 0; c <= CPT; ++c) {
                const uint32_t q4 = s4 + static_cast<uint32_t>(c * BLOCK * 4);
                const uint32_t off = q4 < ns4 ? q4 : top4 - min(q4 - ns4, top4);
                asm volatile("global_load_dword %0, %1, %2 sc1" : "=v"(id[c]) : "v"(off), "s"(a.ids + b) : "memory");
            }
            wait_all<CPT + 1>(nav, id);  // (the pack was requested earlier: this wave's pieces have landed too)
        }
        if (ph == 0) CPM_SSTAMP(2);
        else CPM_SSTAMP(5);
        if (with_arrivals) {
            const uint32_t na_raw = from_lane0(nav);
            na = min(na_raw, cap - ns);
            n_all = ns + na;
            bucket_known(na_raw);
        } else if (ph == 1) {  // (gave up: the bucket counts as empty)
            n_all = 0;
            bucket_known(0);
        }
        nseg = dead ? 0u : (ph == 0 ? ns + (merged ? na : 0u) : na);
        // The cars of this pass in registers: slots tid + c * BLOCK, so of a wave's CPT cars per lane the first k hold cars and the
        // rest are empty for the WHOLE wave; the work is written once for K live cars per lane (cpm_grouped.h: first_pass).
        auto pass = [&](auto kc) {
            constexpr int K = decltype(kc)::value;
            bool valid[K ? K : 1], drive[K ? K : 1], want[K ? K : 1], ok[K ? K : 1];
            uint32_t dest[K ? K : 1], clo[K ? K : 1], khi[K ? K : 1];
#pragma unroll
            for (int c = 0; c < K; ++c) {
                valid[c] = static_cast<uint32_t>(tid + c * BLOCK) < nseg;
                long long kb;
                car_draw_words(a.seed, cars.global(id[c]), a.step, kb, clo[c], khi[c]);
                drive[c] = valid[c] & (kb <= thr);   // u <= p_drive[origin,t] (src/resampling.jl:15) in integers
                want[c] = drive[c] & (last != 0.0);  // stays, or zero row: destination = origin (:35-36)
            }
            if (ph == 0) {  // every wave's pieces of the pack have landed (its own: waited for above)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                hi_last = hi[(SPARSE ? a.Zc : a.Z) - 1];
            }
            if constexpr (K > 0) {
                pack_search<K>(guide, hi, khi, want, sh, hi_last, a.Zq, dest, ok, smap);
                bool anyx = false;
#pragma unroll
                for (int c = 0; c < K; ++c) {
                    dest[c] = want[c] ? dest[c] : static_cast<uint32_t>(z);
                    anyx |= want[c] & !ok[c];
                }
                if (__builtin_expect(any64(anyx), 0)) {  // ties and draws above the row total: the table itself (wave-uniform, rare)
#pragma unroll
                    for (int c = 0; c < K; ++c)
                        if (want[c] & !ok[c]) dest[c] = search_exact_any<SPARSE>(a.rare, a.hour, z, u53(clo[c], khi[c]), last, khi[c] <= hi_last ? static_cast<int>(dest[c]) : -1);
                }
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
                uint32_t rank[K];
#pragma unroll
                for (int c = 0; c < K; ++c) rank[c] = drive[c] ? atomicAdd(&gb[gdiv_group(a.gdiv, dest[c])], 1u) : 0u;
#pragma unroll
                for (int c = 0; c < K; ++c) {
                    if (valid[c] & !drive[c]) put32(stay_out, bS + static_cast<uint32_t>(__popcll(mS[c] & below)), id[c]);
                    bS += static_cast<uint32_t>(__popcll(mS[c]));
                }
#pragma unroll
                for (int c = 0; c < K; ++c) {
                    if (drive[c]) {
                        const uint32_t gd = gdiv_group(a.gdiv, dest[c]);
                        const uint32_t packed = id[c] | (gdiv_local(a.gdiv, dest[c], gd) << a.idbits);
                        if (rank[c] < static_cast<uint32_t>(kStage)) stage[gd * kStage + rank[c]] = packed;
                        else if (rank[c] < a.scap) hand_store<true>(&runs[gd * a.scap + rank[c]], packed);
                    }
                }
            }
        };
        {
            const uint32_t w64 = from_lane0(static_cast<uint32_t>(tid) & ~63u);
            const uint32_t k = nseg > w64 ? min(static_cast<uint32_t>(CPT), (nseg - w64 + BLOCK - 1) / BLOCK) : 0u;
            if constexpr (CPT == 4) {
                switch (k) {
                case 0: pass(std::integral_constant<int, 0>{}); break;
                case 1: pass(std::integral_constant<int, 1>{}); break;
                case 2: pass(std::integral_constant<int, 2>{}); break;
                case 3: pass(std::integral_constant<int, 3>{}); break;
                default: pass(std::integral_constant<int, 4>{}); break;
                }
            } else if constexpr (CPT == 2) {
                switch (k) {
                case 0: pass(std::integral_constant<int, 0>{}); break;
                case 1: pass(std::integral_constant<int, 1>{}); break;
                default: pass(std::integral_constant<int, 2>{}); break;
                }
            } else {
                if (k == 0) pass(std::integral_constant<int, 0>{});
                else pass(std::integral_constant<int, 1>{});
            }
        }
        for (uint32_t q0 = CPT * BLOCK; q0 < nseg; q0 += BLOCK) {  // a pass of more than CPT * BLOCK cars (wave-uniform trips)
            if (q0 + static_cast<uint32_t>(tid & ~63) >= nseg) continue;
            const uint32_t q = q0 + tid;
            const bool valid1 = q < nseg;
            const uint32_t idx = (q0 == CPT * BLOCK) ? id[CPT] : (valid1 ? hand_load<true>(&a.ids[b + (q < ns_here ? q : cap - 1u - (q - ns_here))]) : 0u);
            long long kb;
            uint32_t clo1[1], khi1[1], dest1[1];
            bool ok1[1], want1[1];
            car_draw_words(a.seed, cars.global(idx), a.step, kb, clo1[0], khi1[0]);
            const bool drive1 = valid1 & (kb <= thr);
            want1[0] = drive1 & (last != 0.0);
            pack_search<1>(guide, hi, khi1, want1, sh, hi_last, a.Zq, dest1, ok1, smap);
            if (!want1[0]) dest1[0] = z;
            else if (!ok1[0]) dest1[0] = search_exact_any<SPARSE>(a.rare, a.hour, z, u53(clo1[0], khi1[0]), last, khi1[0] <= hi_last ? static_cast<int>(dest1[0]) : -1);
            const unsigned long long m1 = ballot64(valid1 & !drive1);
            uint32_t b1 = 0;
            if (lane == 0 && m1) b1 = atomicAdd(&s_nstay, static_cast<uint32_t>(__popcll(m1)));
            b1 = from_lane0(b1);
            if (valid1 & !drive1) stay_out[b1 + static_cast<uint32_t>(__popcll(m1 & below))] = idx;
            if (drive1) {
                const uint32_t gd = gdiv_group(a.gdiv, dest1[0]);
                const uint32_t rank = atomicAdd(&gb[gd], 1u);
                const uint32_t packed = idx | (gdiv_local(a.gdiv, dest1[0], gd) << a.idbits);
                if (rank < static_cast<uint32_t>(kStage)) stage[gd * kStage + rank] = packed;
                else if (rank < a.scap) hand_store<true>(&runs[gd * a.scap + rank], packed);
            }
        }
        if (ph == 0) CPM_SSTAMP(3);
        else CPM_SSTAMP(6);
        if (merged) break;  // (workgroup-uniform: the verdict came through LDS)
    }
    lds_barrier();  // ranks, staged drivers and counters (all in LDS) are final; the stayers' stores need not have landed
    {
        // staged drivers -> their runs: 8 lanes per group, 16 bytes per lane, written through (sc1: the placing blocks of other XCDs read them)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(runs, 0, static_cast<int>(kGroups * a.scap * 4u), 0x00020000);
        for (int i = tid; i < kGroups * (kStage / 4); i += BLOCK) {
            const int gd = i / (kStage / 4), ch = i % (kStage / 4);
            const uint32_t lim = min(gb[gd], static_cast<uint32_t>(kStage));
            if (static_cast<uint32_t>(4 * ch) < lim) {
                const uint4 q = *reinterpret_cast<const uint4 *>(&stage[gd * kStage + 4 * ch]);
                cpm_u32x4 qv;
                qv.x = q.x;
                qv.y = q.y;
                qv.z = q.z;
                qv.w = q.w;
                __builtin_amdgcn_raw_buffer_store_b128(qv, rs, (static_cast<uint32_t>(gd) * a.scap + 4u * ch) << 2, 0, 16);  // (aux 16: sc1)
            }
        }
        if (tid < kGroups) {
            const uint32_t c = gb[tid];
            hand_store<true>(&a.cntg[static_cast<size_t>(z) * kGroups + tid], min(c, a.scap));
            if (c > a.scap) atomicOr(a.rare->status, 2ull);  // a run outgrew its slot: the caller grows the regions and repeats
        }
    }
    const uint32_t nstay = s_nstay;
    if (tid == 0) a.driving_t[z] = n_all - nstay;  // every car of the bucket either stays or drives
    CPM_SSTAMP(7);
#if defined(CPM_DIAGNOSTIC) && defined(CPM_STAMP_BOTH)  // (tools/day_stamps.py: where the block ran and which way it took, in the low bits of two stamps)
    st_[0] = (st_[0] & ~0xFFull) | ((__builtin_amdgcn_s_getreg((4) | (8 << 6) | (7 << 11))) & 0xFFu);  // HW_REG_HW_ID bits 8..15: cu, sh, se
    st_[7] = (st_[7] & ~0x1Full) | myx | (merged ? 16u : 0u);
#endif
    CPM_SSTAMP_FLUSH;
    // the zone is handed over: every storing wave drains, the workgroup meets, ONE lane sets the zone's flag (for the zone's
    // workgroup of the next hour) and counts the workgroup in on its chunk's counter (for the placing blocks of the chunk)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (tid == 0) {
        __hip_atomic_store(&a.cnt_next[z], min(nstay, cap) | (myx << kCntXccShift) | kCntValid | (dead ? kCntAbort : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(a.sdone + static_cast<size_t>(z / kFusedChunk) * kDoneStride, dead ? 1u + kDoneAbort : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Block r of an hour's segment -> its role.  Per XCD (r % 8 = x, the groups g = 8 s + x, s = 0 .. 3 "sets"): the placing blocks of
// set 0, then windows of {the placing blocks of set s + 1, the sampler workgroups of set s}, then the sampler workgroups of the last
// set.  A sampler workgroup of set s waits for placing blocks of set s only -- a whole window in front of it, done when it starts
// (the merged pass) except in set 0, whose placing blocks wait for the last sampler workgroups of the hour before.
//   mix == 0: every placing block (set by set), then every sampler workgroup (set by set) -- k_grouped_hour_pf's order.
// MEASURED (S4k, one box, ms per resample; the hour as one launch, k_grouped_hour: 0.890): placing blocks spread EVENLY among the sampler
// workgroups of the window in front: 1.267 -- the last placing block of a set then sits right in front of the set's first sampler
// workgroups, and every window begins with all its workgroups waiting one placing-block lifetime; mix == 0 with the split pass only: 0.956.
struct DayRole {
    int z, g, j;  // sampler workgroup of zone z (j < 0), or placing block (g, j)
};
__device__ __forceinline__ DayRole day_role(int r, int zpg, int pc, int mix)
{
    DayRole o;
    const int x = r & 7, i = r >> 3;
    constexpr int kSets = kGroups / 8;
    int set, q = -1, j = -1;
    if (!mix) {
        if (i < kSets * pc) {
            set = i / pc;
            j = i - set * pc;
        } else {
            const int i2 = i - kSets * pc;
            set = i2 / zpg;
            q = i2 - set * zpg;
        }
    } else if (i < pc) {
        set = 0;
        j = i;
    } else {
        const int w = zpg + pc, i2 = i - pc;
        const int s = i2 / w, p = i2 - s * w;
        if (s < kSets - 1 && p < pc) {
            set = s + 1;
            j = p;
        } else {
            set = s;
            q = s < kSets - 1 ? p - pc : p;
        }
    }
    o.g = 8 * set + x;
    o.j = j;
    o.z = q >= 0 ? o.g * zpg + q : -1;
    return o;
}

// grid = hours x per_hour blocks, per_hour = kGroups * (zones per group + chunks); hour t's arguments in hours[t] (device memory,
// filled by k_grouped_zero from the run's bases: the same struct the hourly launches take by value)
template <int CPT, int NQ, bool SPARSE = false>
__global__ __launch_bounds__(kFusedThreads, CPM_WPS) CPM_SGPR_ATTR void k_grouped_day(const GroupedArgs *__restrict__ hours, int per_hour, int pc, int mix)
{
    extern __shared__ uint32_t dyn[];  // sampler: the zone's row pack; placing block: its sorted list
    __shared__ union {
        SampleLds s;
        PlaceLds<kFusedThreads, kFusedKruns, kFusedZpg> p;
    } u;
    const int t = blockIdx.x / per_hour, r = blockIdx.x - t * per_hour;
#if defined(CPM_DIAGNOSTIC) && defined(CPM_STAMP_BOTH)  // (tools/day_stamps.py: this launch's rows behind those the hourly launches of the run overwrite)
    if (threadIdx.x == 0 && g_place_stamps) g_place_stamps[(static_cast<size_t>(gridDim.x) + blockIdx.x) * 8] = __builtin_amdgcn_s_memtime();
#endif
    const GroupedArgs &a = hours[t];
    const int zpg = static_cast<int>(gdiv_zpg(a.gdiv));
    const DayRole role = day_role(r, zpg, pc, mix);
    if (role.j < 0) {
        if (role.g >= kGroups || role.z >= a.Z) return;
        day_sample_body<kFusedThreads, CPT, NQ, SPARSE>(a, role.z, role.g, dyn, u.s);
    } else {
        if (role.j >= a.pchunks) return;  // (hour 0: nothing to place; a.pchunks == pc otherwise)
        const uint32_t need = static_cast<uint32_t>(min(kFusedChunk, a.Z - role.j * kFusedChunk));
        grouped_place_body<kFusedThreads, kFusedKruns, kFusedKdeep, kFusedZpg, true, true>(
            role.g, role.j, u.p, dyn, a.pD, a.pcntg, zpg, kFusedChunk, a.Z, a.cap, a.scap, a.idbits, const_cast<uint32_t *>(a.cnt_a), const_cast<uint32_t *>(a.ids),
            a.rare->status, a.psdone + static_cast<size_t>(role.j) * kDoneStride, need, a.spin_limit, a.pdone + static_cast<size_t>(role.g) * kDoneStride);
    }
}

// What the hours of a day launch differ in, derived from the run's bases (k_grouped_zero fills hours[t] on the device: no
// host-to-device copy in the stream of every resample)
struct GroupedDay {
    GroupedArgs base;                                  // everything the hours share
    const uint32_t *ids0, *cnt0;                       // the cached initial bucketing: hour 0's buckets
    uint32_t *idsA, *idsB, *cnt;                       // ping-pong buckets; [T+1][2][Z] per-hour counts
    const uint32_t *rp;                                // tables: [T][Z][RW], [T][Z], [T][Z]
    const double *last;
    const long long *thr;
    uint32_t *Dq, *cntg;                               // runs / run lengths: `copies` of them, hour t in copy t % copies
    unsigned long long *parking, *driving;             // [T][Z]
    uint32_t *sdone0, *pdone0;                         // hand-off counters of hour 0 / segment 0
    size_t rw, run_words, len_words, sdone_stride, pdone_stride;
    int copies, nchunk, nhours;
    uint32_t step0;
};
__device__ __forceinline__ void day_fill_hour(const GroupedDay &d, int t, GroupedArgs *out)
{
    GroupedArgs a = d.base;
    const size_t Z = static_cast<size_t>(a.Z), ts = static_cast<size_t>(t);
    a.ids = t == 0 ? d.ids0 : (((t - 1) & 1) ? d.idsB : d.idsA);
    a.cnt_s = t == 0 ? d.cnt0 : d.cnt + ts * 2 * Z;
    a.cnt_a = a.cnt_s + Z;
    a.rp_t = d.rp + ts * Z * d.rw;
    a.last_t = d.last + ts * Z;
    a.thr_t = d.thr + ts * Z;
    a.ids_next = (t & 1) ? d.idsB : d.idsA;
    a.cnt_next = d.cnt + (ts + 1) * 2 * Z;
    a.D = d.Dq + d.run_words * static_cast<size_t>(t % d.copies);
    a.cntg = d.cntg + d.len_words * static_cast<size_t>(t % d.copies);
    a.parking_t = d.parking + ts * Z;
    a.driving_t = d.driving + ts * Z;
    a.hour = t;
    a.step = d.step0 + static_cast<uint32_t>(t);
    a.sdone = d.sdone0 + ts * d.sdone_stride;
    a.psdone = t ? d.sdone0 + (ts - 1) * d.sdone_stride : nullptr;
    a.pdone = d.pdone0 + ts * d.pdone_stride;
    a.pD = t ? d.Dq + d.run_words * static_cast<size_t>((t - 1) % d.copies) : nullptr;
    a.pcntg = t ? d.cntg + d.len_words * static_cast<size_t>((t - 1) % d.copies) : nullptr;
    a.pchunks = t ? d.nchunk : 0;
    a.chained = t ? 1u : 0u;
    *out = a;
}


template <int CPT, int NQ, bool SPARSE = false>
inline void grouped_launch_day_nq(const GroupedArgs *hours, int nhours, int Zq, int G, int smap, int zpg, int nchunk, int mix, hipStream_t stream)
{
    const size_t lds = fused_lds_bytes(Zq, G, smap);
    if (lds > 48 * 1024) {
        static bool attr_done[64] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 64 || !attr_done[dev]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_grouped_day<CPT, NQ, SPARSE>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            if (dev >= 0 && dev < 64) attr_done[dev] = true;
        }
    }
    const int per_hour = kGroups * (zpg + nchunk);
    launch(k_grouped_day<CPT, NQ, SPARSE>, dim3(static_cast<unsigned>(nhours) * static_cast<unsigned>(per_hour)), dim3(kFusedThreads), lds, stream, hours, per_hour, nchunk, mix);
}
template <int CPT>
inline void grouped_launch_day_c(const GroupedArgs *hours, int nhours, int Zq, int G, int smap, int zpg, int nchunk, int mix, hipStream_t stream)
{
    const int need = (pack_row_words(Zq, G, smap) / 4 + kSampleBlock - 1) / kSampleBlock;
#define CPM_DAY_ARGS hours, nhours, Zq, G, smap, zpg, nchunk, mix, stream
    if (smap) {
        grouped_launch_day_nq<CPT, 1, true>(CPM_DAY_ARGS);
        return;
    }
    if (need <= 1) grouped_launch_day_nq<CPT, 1>(CPM_DAY_ARGS);
    else if (need <= 2) grouped_launch_day_nq<CPT, 2>(CPM_DAY_ARGS);
    else if (need <= 3) grouped_launch_day_nq<CPT, 3>(CPM_DAY_ARGS);
    else if (need <= 4) grouped_launch_day_nq<CPT, 4>(CPM_DAY_ARGS);
    else if (need <= 5) grouped_launch_day_nq<CPT, 5>(CPM_DAY_ARGS);
    else if (need <= 6) grouped_launch_day_nq<CPT, 6>(CPM_DAY_ARGS);
    else if (need <= 8) grouped_launch_day_nq<CPT, 8>(CPM_DAY_ARGS);
    else grouped_launch_day_nq<CPT, 12>(CPM_DAY_ARGS);
#undef CPM_DAY_ARGS
}
inline void grouped_launch_day(const GroupedArgs *hours, int nhours, int Z, int Zq, int G, int smap, int zpg, int nchunk, int mix, int64_t mean, hipStream_t stream)
{
    (void)Z;
    switch (grouped_cpt(mean)) {
    case 1: grouped_launch_day_c<1>(hours, nhours, Zq, G, smap, zpg, nchunk, mix, stream); break;
    case 2: grouped_launch_day_c<2>(hours, nhours, Zq, G, smap, zpg, nchunk, mix, stream); break;
    default: grouped_launch_day_c<4>(hours, nhours, Zq, G, smap, zpg, nchunk, mix, stream); break;
    }
}

}  // namespace cpm
