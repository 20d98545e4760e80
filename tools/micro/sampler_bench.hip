// Development harness: k_zone6_sample alone on synthetic buffers of the S4k shape, timed back to back with hipEvents,
// for every ablation mask given on the command line (needs -DCPM_DIAGNOSTIC).  Results of ablated runs are meaningless.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../../carparkingmaps_amd/csrc/cpm_zone5_kernels.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv)
{
    using namespace cpm;
    const int Z = 4096, T = 24, cpz = 1000;
    const int Zp = Z, Zq = Z;
    const uint32_t cap = 4032, scap = zone6_scap(cap);
        const int G = pack_guide_bits(Z), RW = pack_row_words(Zq, G);
    std::vector<uint32_t> h_hi(static_cast<size_t>(Z) * RW), h_ids(static_cast<size_t>(Z) * cap), h_cnt(Z, cpz);
    for (int z = 0; z < Z; ++z) {
        uint32_t *pk = h_hi.data() + static_cast<size_t>(z) * RW;
        uint32_t *hi = pk + pack_guide_words(G);
        uint32_t x = 12345u + z;
        uint64_t acc = 0;
        std::vector<uint64_t> w(Z);
        for (int j = 0; j < Z; ++j) { x = x * 1664525u + 1013904223u; w[j] = (x >> 8) + 1; acc += w[j]; }
        uint64_t run = 0;
        for (int j = 0; j < Zq; ++j) {
            if (j < Z) { run += w[j]; hi[j] = static_cast<uint32_t>(std::min<unsigned __int128>((static_cast<unsigned __int128>(run) << 32) / acc, 0xFFFFFFFFull)); }
            else hi[j] = 0xFFFFFFFFu;
        }
        uint16_t *gd = reinterpret_cast<uint16_t *>(pk);
        int jj = 0;
        for (int m = 0; m < (1 << G) + 8; ++m) {
            if (m < (1 << G)) {
                const uint32_t edge = static_cast<uint32_t>(m) << (32 - G);
                while (jj < Z && hi[jj] < edge) ++jj;
            } else jj = Z - 1;
            gd[m] = static_cast<uint16_t>(std::min(jj, Z - 1));
        }
        for (uint32_t i = 0; i < cap; ++i) h_ids[static_cast<size_t>(z) * cap + i] = (z * cpz + i) % (Z * cpz);
    }
    std::vector<double> h_one(Z, 1.0), h_half(Z, 0.5);
    uint32_t *d_hi, *d_ids, *d_cnt, *d_ids_next, *d_cnt_next, *d_D, *d_cntg, *d_rec;
    double *d_last, *d_pd, *d_cdf;
    unsigned long long *d_counts;
    CK(hipMalloc(&d_hi, sizeof(uint32_t) * Z * RW * T));
    for (int t = 0; t < T; ++t) CK(hipMemcpy(d_hi + static_cast<size_t>(t) * Z * RW, h_hi.data(), sizeof(uint32_t) * Z * RW, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_ids, sizeof(uint32_t) * Z * cap));
    CK(hipMemcpy(d_ids, h_ids.data(), sizeof(uint32_t) * Z * cap, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_cnt, sizeof(uint32_t) * Z));
    CK(hipMemcpy(d_cnt, h_cnt.data(), sizeof(uint32_t) * Z, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_ids_next, sizeof(uint32_t) * Z * cap));
    CK(hipMalloc(&d_cnt_next, sizeof(uint32_t) * Z));
    CK(hipMalloc(&d_D, sizeof(uint32_t) * Z * kGroups6 * scap));
    CK(hipMalloc(&d_cntg, sizeof(uint32_t) * Z * kGroups6));
    CK(hipMalloc(&d_rec, sizeof(uint32_t) * Z * cap));
    CK(hipMalloc(&d_last, sizeof(double) * Z));
    CK(hipMalloc(&d_pd, sizeof(double) * Z));
    CK(hipMalloc(&d_cdf, sizeof(double) * Z * Zp));
    CK(hipMemset(d_cdf, 0, sizeof(double) * Z * Zp));
    CK(hipMemcpy(d_last, h_one.data(), sizeof(double) * Z, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_pd, h_half.data(), sizeof(double) * Z, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_counts, sizeof(unsigned long long) * (2 * Z + 2)));
    CK(hipMemset(d_counts, 0, sizeof(unsigned long long) * (2 * Z + 2)));
    Zone6Args a{};
    a.stamps = nullptr;
    a.ids = d_ids; a.cnt = d_cnt; a.thr_t = nullptr; a.last_t = d_last; a.pdrive_t = d_pd; a.cdf_t = d_cdf; a.dm = nullptr;
    a.ids_next = d_ids_next; a.cnt_next = d_cnt_next; a.D = d_D; a.cntg = d_cntg; a.rec_out = d_rec;
    a.parking_t = d_counts; a.driving_t = d_counts + Z; a.tt_sum = d_counts + 2 * Z; a.status = d_counts + 2 * Z + 1;
    a.Z = Z; a.Zp = Zp; a.Zq = Zq; a.G = G; a.T = T; a.t = 0;
    a.cap = cap; a.scap = scap; a.idbits = zone6_idbits(Z); a.step = 23; a.gshift = zone6_gshift(Z);
    a.car_begin = 0; a.seed = 0x5EEDCA125ull;
    unsigned long long *d_stamps;
    CK(hipMalloc(&d_stamps, sizeof(unsigned long long) * Z * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int reps = 120;
    for (int shape = 0; shape < 1; ++shape)
        for (int i = 1; i < argc; ++i) {
            a.abl = atoi(argv[i]);
            int hour = 0;
            auto go = [&] {
                a.rp_t = d_hi + static_cast<size_t>(hour++ % T) * Z * RW;
                zone6_launch<true>(a, false, 1000, 0);
            };
            for (int k = 0; k < 5; ++k) go();
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int k = 0; k < reps; ++k) go();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("shape %d abl %4d: %7.2f us per launch\n", shape, a.abl, ms * 1000.f / reps);
            if (a.abl == 0) {  // one stamped launch: mean phase lengths of wave 0 over the workgroups (s_memtime ticks = 100 MHz wall clock x ?)
                a.stamps = d_stamps;
                CK(hipMemset(d_stamps, 0, sizeof(unsigned long long) * Z * 8));
                go();
                CK(hipDeviceSynchronize());
                a.stamps = nullptr;
                std::vector<unsigned long long> st(static_cast<size_t>(Z) * 8);
                CK(hipMemcpy(st.data(), d_stamps, sizeof(unsigned long long) * Z * 8, hipMemcpyDeviceToHost));
                double sum[8] = {0};
                unsigned long long t0 = ~0ull, t1 = 0;
                for (int z = 0; z < Z; ++z) {
                    for (int k = 1; k < 8; ++k) sum[k] += static_cast<double>(st[z * 8 + k] - st[z * 8 + k - 1]);
                    t0 = std::min(t0, st[z * 8]);
                    t1 = std::max(t1, st[z * 8 + 7]);
                }
                printf("   ticks: kernel span %llu; per workgroup: issue loads %.0f | barrier1 (loads land) %.0f | philox %.0f | search %.0f | emit %.0f | reduce+barrier2 %.0f | flush %.0f ; lifetime %.0f\n",
                       t1 - t0, sum[1] / Z, sum[2] / Z, sum[3] / Z, sum[4] / Z, sum[5] / Z, sum[6] / Z, sum[7] / Z,
                       (sum[1] + sum[2] + sum[3] + sum[4] + sum[5] + sum[6] + sum[7]) / Z);
            }
        }
    return 0;
}
