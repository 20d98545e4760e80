// cpm_tables.h -- device builders of the probability tables (src/createpdrive.jl,
// src/createpdestin.jl).  Input: datamatrix[o + Z*(d + Z*(t + T*k))] and dist[o + Z*d] in the
// reference's column-major layout; every kernel puts the origin on the lane so that all
// HBM accesses are coalesced (the reference walks the same arrays at stride Z or Z*Z).
// Sums run left to right over the destination in f64: this build's pinned order (the oracle's too).  createpdrive's mean is the same
// explicit sequential loop in the reference (src/createpdrive.jl:14-18); createpdestin's normaliser is NOT: the reference calls
// sum(p_dest[i, :, t]) (src/createpdestin.jl:33), Julia's pairwise sum, which can differ from the sequential sum by an ulp.  Julia
// parity of these tables is UNPINNED: nothing reference-held covers them, and x^0.5 is Julia's pow there, evaluated here as sqrt;
// other float exponents go through the device pow (a few ulp from glibc's / Julia's).  Statistically irrelevant; said so that no
// comment claims more than the tests show (device == oracle, bit for bit or to 4e-16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cpm {

constexpr int kTabBatch = 16;  // independent loads a lane keeps in flight in the sequential table sums

// Julia's maximum/minimum propagate NaN (Appendix A-3); fmax/fmin would not.
__device__ __forceinline__ double jl_max(double a, double b) { return (a != a || b != b) ? __builtin_nan("") : (a > b ? a : b); }
__device__ __forceinline__ double jl_min(double a, double b) { return (a != a || b != b) ? __builtin_nan("") : (a < b ? a : b); }

// mean_sum[i,t] = mean over non-zero j of datamatrix[i,j,t,1] / dist[i,j]   (createpdrive.jl:10-21)
__global__ void k_pdrive_mean(const double *__restrict__ dm, const double *__restrict__ dist,
                              double *__restrict__ mean_sum, int Z)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int t = blockIdx.y;
    if (i >= Z) return;
    const double *src = dm + static_cast<size_t>(t) * Z * Z + i;
    const double *dst = dist + i;
    double s = 0.0;
    long long counter = 0;
    // The sum runs left to right (src/createpdrive.jl:14-18 is the same explicit loop) but the loads do not depend on it: kTabBatch of them are requested
    // together, so that a lane has kTabBatch x 2 loads in flight instead of one round trip per destination.
    int j = 0;
    for (; j + kTabBatch <= Z; j += kTabBatch) {
        double m[kTabBatch], d[kTabBatch];
#pragma unroll
        for (int u = 0; u < kTabBatch; ++u) {
            m[u] = src[static_cast<size_t>(j + u) * Z];
            d[u] = dst[static_cast<size_t>(j + u) * Z];
        }
#pragma unroll
        for (int u = 0; u < kTabBatch; ++u)
            if (m[u] != 0) {
                s = s + m[u] / d[u];
                counter += 1;
            }
    }
    for (; j < Z; ++j) {
        double m = src[static_cast<size_t>(j) * Z];
        if (m != 0) {
            s = s + m / dst[static_cast<size_t>(j) * Z];
            counter += 1;
        }
    }
    mean_sum[i + static_cast<size_t>(t) * Z] = s / static_cast<double>(counter);  // 0/0 = NaN
}

// exponent modes: Float64^Float64 of the reference; 0.5, 1, 2 are evaluated exactly
__device__ __forceinline__ double pow_f64(double x, double e)
{
    if (e == 0.5) return sqrt(x);
    if (e == 1.0) return x;
    if (e == 2.0) return x * x;
    return pow(x, e);
}

__device__ __forceinline__ double pow_int(double x, long n)
{
    if (n == 0) return 1.0;
    if (n == 1) return x;
    if (n == 2) return x * x;
    if (n == 3) return x * x * x;
    long t = 0, m = n;  // Base.power_by_squaring order
    while ((m & 1) == 0) { m >>= 1; ++t; }
    double xx = x;
    for (long i = 0; i < t; ++i) xx *= xx;
    double yy = xx;
    m >>= 1;
    while (m > 0) {
        xx *= xx;
        if (m & 1) yy *= xx;
        m >>= 1;
    }
    return yy;
}

// p_drive[i,t] = p_min + (p_max - p_min) * ((ms - min) / (max - min))^e_drive   (createpdrive.jl:22-33), and the sampler's integer
// threshold of it (thr may be null).  One thread per (zone, hour): each finds the zone's extrema over the day itself (24 L2-resident
// loads) and raises ONE power -- a thread per zone walking the day did 24 pow_f64 one after the other, 19.6 us per grid point of a sweep.
__global__ void k_pdrive_final(const double *__restrict__ mean_sum, double *__restrict__ pdrive, long long *__restrict__ thr, int Z, int T,
                               double p_min, double p_max, double e_drive)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int t = blockIdx.y;
    if (i >= Z) return;
    double mx = mean_sum[i], mn = mean_sum[i];
    for (int h = 1; h < T; ++h) {
        double v = mean_sum[i + static_cast<size_t>(h) * Z];
        mx = jl_max(mx, v);
        mn = jl_min(mn, v);
    }
    double v = 0.0;
    if (mx > 0) v = p_min + (p_max - p_min) * pow_f64((mean_sum[i + static_cast<size_t>(t) * Z] - mn) / (mx - mn), e_drive);
    pdrive[i + static_cast<size_t>(t) * Z] = v;
    if (thr) thr[i + static_cast<size_t>(t) * Z] = bernoulli_threshold(v);
}

// unnormalised p_dest[i,j,t] = ((m - min_t m) / (max_t m - min_t m))^e_dest   (createpdestin.jl:10-28).  One thread per (origin,
// destination) pair, the origin on the lane.  TT > 0: the TT hourly means of the pair are loaded once, all in flight together, and
// stay in registers for the extrema and the weights (T = 24: main.jl:42); TT == 0: any T, the day is read twice.
template <int TT>
__global__ __launch_bounds__(256) void k_pdest_weights(const double *__restrict__ dm, double *__restrict__ p, int Z, int T, double e_dest,
                                                       int e_is_integer)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (i >= Z) return;
    const size_t base = i + static_cast<size_t>(j) * Z;
    const size_t slab = static_cast<size_t>(Z) * Z;
    if constexpr (TT > 0) {
        double v[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) v[t] = dm[base + t * slab];
        double mx = v[0], mn = v[0];
#pragma unroll
        for (int t = 1; t < TT; ++t) {
            mx = jl_max(mx, v[t]);
            mn = jl_min(mn, v[t]);
        }
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            double w = 0.0;
            if (mx > 0) {
                const double x = (v[t] - mn) / (mx - mn);
                w = e_is_integer ? pow_int(x, static_cast<long>(e_dest)) : pow_f64(x, e_dest);
            }
            p[base + t * slab] = w;
        }
    } else {
        double mx = dm[base], mn = dm[base];
        for (int t = 1; t < T; ++t) {
            const double v = dm[base + t * slab];
            mx = jl_max(mx, v);
            mn = jl_min(mn, v);
        }
        for (int t = 0; t < T; ++t) {
            double w = 0.0;
            if (mx > 0) {
                const double x = (dm[base + t * slab] - mn) / (mx - mn);
                w = e_is_integer ? pow_int(x, static_cast<long>(e_dest)) : pow_f64(x, e_dest);
            }
            p[base + t * slab] = w;
        }
    }
}

// normalise per (i,t): nf = sum_j p[i,j,t], left to right here (the reference: Julia's pairwise sum(), createpdestin.jl:33); divide if nf > 0
// (:38-46).  Two kernels: the sums, one lane per origin (32 loads in flight per lane, added in order: T x Z lanes is all the
// parallelism a sequential sum has), then the division over every entry at once -- as the second sweep of the summing lanes it ran
// with those 888 waves' parallelism (0.77 ms for both sweeps at Z = 2,357; the division alone is a plain streaming pass).
// (Dividing where the row tables are built instead -- inside the ONE lane per origin that forms the running sums -- was measured at
// 1.58 ms against 0.64 + the dividing sweep at Z = 2,357: thirteen f64 instructions per entry in front of every addition.)
constexpr int kSumBatch = 32;
__global__ __launch_bounds__(64) void k_pdest_rowsum(const double *__restrict__ p, double *__restrict__ nf_out, int Z)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int t = blockIdx.y;
    if (i >= Z) return;
    const double *row = p + static_cast<size_t>(t) * Z * Z + i;
    double nf = 0.0;
    int j = 0;
    for (; j + kSumBatch <= Z; j += kSumBatch) {  // loads in batches, sum in order (see k_pdrive_mean)
        double v[kSumBatch];
#pragma unroll
        for (int u = 0; u < kSumBatch; ++u) v[u] = row[static_cast<size_t>(j + u) * Z];
#pragma unroll
        for (int u = 0; u < kSumBatch; ++u) nf = nf + v[u];
    }
    for (; j < Z; ++j) nf = nf + row[static_cast<size_t>(j) * Z];
    nf_out[static_cast<size_t>(t) * Z + i] = nf;
}
// p[t][j][i] /= nf[t][i] where nf > 0: the origin on the lane (the reference's fastest index), kDivBatch destinations per thread
constexpr int kDivBatch = 16;
__global__ __launch_bounds__(256) void k_pdest_divide(double *__restrict__ p, const double *__restrict__ nf, int Z)
{
    const int t = blockIdx.z, j0 = blockIdx.y * kDivBatch;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Z) return;
    const double f = nf[static_cast<size_t>(t) * Z + i];
    if (!(f > 0)) return;
    double *q = p + (static_cast<size_t>(t) * Z + j0) * Z + i;
    double v[kDivBatch];
#pragma unroll
    for (int u = 0; u < kDivBatch; ++u) v[u] = (j0 + u < Z) ? q[static_cast<size_t>(u) * Z] : 0.0;
#pragma unroll
    for (int u = 0; u < kDivBatch; ++u)
        if (j0 + u < Z) q[static_cast<size_t>(u) * Z] = v[u] / f;
}

}  // namespace cpm
