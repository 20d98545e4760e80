// Development harness: k_zone6_sample alone on synthetic buffers of the S4k shape, timed back to back with hipEvents,
// for every ablation mask given on the command line (needs -DCPM_DIAGNOSTIC).  Results of ablated runs are meaningless.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../carparkingmaps_amd/csrc/cpm_zone5_kernels.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv)
{
    using namespace cpm;
    const int Z = 4096, T = 24, cpz = 1000;
    const int Zp = Z, Zq = Z;
    const uint32_t cap = 4032, scap = zone6_scap(cap);
    const int zpg = Z / kGroups6;
    const int H = tree_height(Z);
    std::vector<uint32_t> h_hi(static_cast<size_t>(Z) * Zq), h_ids(static_cast<size_t>(Z) * cap), h_cnt(Z, cpz);
    for (int z = 0; z < Z; ++z) {
        for (int j = 0; j < Zq; ++j) h_hi[static_cast<size_t>(z) * Zq + j] = static_cast<uint32_t>((static_cast<uint64_t>(j + 1) << 32) / Z - 1);
        for (uint32_t i = 0; i < cap; ++i) h_ids[static_cast<size_t>(z) * cap + i] = (z * cpz + i) % (Z * cpz);
    }
    std::vector<double> h_one(Z, 1.0), h_half(Z, 0.5);
    uint32_t *d_hi, *d_ids, *d_cnt, *d_ids_next, *d_cnt_next, *d_D, *d_cntg, *d_rec;
    double *d_last, *d_pd, *d_cdf;
    unsigned long long *d_counts;
    CK(hipMalloc(&d_hi, sizeof(uint32_t) * Z * Zq * T));
    for (int t = 0; t < T; ++t) CK(hipMemcpy(d_hi + static_cast<size_t>(t) * Z * Zq, h_hi.data(), sizeof(uint32_t) * Z * Zq, hipMemcpyHostToDevice));
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
    a.ids = d_ids; a.cnt = d_cnt; a.last_t = d_last; a.pdrive_t = d_pd; a.cdf_t = d_cdf; a.dm = nullptr;
    a.ids_next = d_ids_next; a.cnt_next = d_cnt_next; a.D = d_D; a.cntg = d_cntg; a.rec_out = d_rec;
    a.parking_t = d_counts; a.driving_t = d_counts + Z; a.tt_sum = d_counts + 2 * Z; a.status = d_counts + 2 * Z + 1;
    a.Z = Z; a.Zp = Zp; a.Zq = Zq; a.H = H; a.T = T; a.t = 0; a.zpg = zpg;
    a.cap = cap; a.scap = scap; a.idbits = zone6_idbits(zpg); a.step = 23; a.gmagic = (1u << 24) / zpg + 1u;
    a.car_begin = 0; a.seed = 0x5EEDCA125ull;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int reps = 120;
    for (int shape = 0; shape < 5; ++shape)
        for (int i = 1; i < argc; ++i) {
            a.abl = atoi(argv[i]);
            int hour = 0;
            auto go = [&] {
                a.hi_t = d_hi + static_cast<size_t>(hour++ % T) * Z * Zq;
                zone6_launch<true>(a, false, shape, 256, 0);
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
        }
    return 0;
}
