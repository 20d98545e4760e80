// cpm_kernels.h -- HIP kernels of the sampler path for gfx950 (MI355X).
//
// Device data layout (all in HBM, owned by cpm_ctx):
//   pdrive [T][Z]        f64   == Julia's column-major Z x T, zone contiguous
//   cdf    [T][Z][Zp]    f64   canonical CDF rows, destination contiguous, row padded to
//                              Zp = roundup(Z,16) with +inf (a row is a whole number of
//                              128-B lines and starts on one); built on first need (k_build_rows,
//                              cpm_grouped.h): only the car and exact-layout kernels search f64 rows
//   zone0  [C]           u32   current (initial) zone of each local car, 0-based
//   rec    [T][C]        u32   per resampling hour: destination zone | drive flag << 31
//   counts [2][T][Z]     i64   parking | driving zone x hour histogram (+1 word: q16 time sum)
// Zone ids are 0-based on the device and 1-based on the host side of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "cpm_rng.h"

namespace cpm {

// Launch of an hourly kernel that may be timed (CPM_OPT_PROFILE): when the context has armed an event pair for the next launch, the
// launch carries it (hipExtLaunchKernelGGL stamps the pair with the dispatch's own begin and end).  hipEventRecord on either side of a
// launch instead times two extra barrier packets with it: ~3 us on a 26 us launch, and a bubble in the stream for each.
struct LaunchTimer {
    hipEvent_t start = nullptr, stop = nullptr;
};
inline LaunchTimer &launch_timer()
{
    static thread_local LaunchTimer t;
    return t;
}
template <typename... P, typename... A>
inline void launch(void (*kernel)(P...), dim3 grid, dim3 block, size_t lds, hipStream_t stream, A... args)
{
    static_assert(sizeof...(P) == sizeof...(A), "one argument per kernel parameter");
    LaunchTimer &t = launch_timer();
    if (t.start) {  // (the extended launch copies its arguments as they are: converted to the kernel's own parameter types here)
        hipExtLaunchKernelGGL(kernel, grid, block, static_cast<uint32_t>(lds), stream, t.start, t.stop, 0, static_cast<P>(args)...);
        t = LaunchTimer{};
    } else {
        hipLaunchKernelGGL(kernel, grid, block, static_cast<uint32_t>(lds), stream, static_cast<P>(args)...);
    }
}

constexpr uint32_t kDriveBit = 0x80000000u;
constexpr uint32_t kZoneMask = 0x7fffffffu;
constexpr double kTrueMin = 4.9406564584124654e-324;

// The cars a context simulates: local car i (0-based index into the context's state) is the GLOBAL car begin + i * stride of
// the whole fleet.  stride 1 = a contiguous shard; stride N = the interleaved deal over N GPUs (car g belongs to rank g mod N).
// Philox is keyed by the global id, so a car's draws do not depend on how the fleet is dealt.
struct CarIndex {
    int64_t begin;
    uint32_t stride;
    __host__ __device__ __forceinline__ uint64_t global(uint32_t local) const
    {
        return static_cast<uint64_t>(begin) + static_cast<uint64_t>(local) * stride;  // one v_mad_u64_u32
    }
};

// ---------------------------------------------------------------------------------------
// initializestates (src/initializestates.jl:11-16): global car g starts in zone g / cpz.
// ---------------------------------------------------------------------------------------
__global__ void k_init_states(uint32_t *zone0, CarIndex cars, int64_t n, int64_t cpz)
{
    int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) zone0[i] = static_cast<uint32_t>(cars.global(static_cast<uint32_t>(i)) / static_cast<uint64_t>(cpz));
}

__global__ void k_zones_from_i64(uint32_t *zone0, const int64_t *zones1, int64_t n, int64_t Z, int *err)
{
    int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) {
        int64_t z = zones1[i];
        if (z < 1 || z > Z) { atomicOr(err, 1); z = 1; }
        zone0[i] = static_cast<uint32_t>(z - 1);
    }
}

__global__ void k_zones_to_i64(int64_t *zones1, const uint32_t *zone0, int64_t n)
{
    int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) zones1[i] = static_cast<int64_t>(zone0[i] & kZoneMask) + 1;
}

// ---------------------------------------------------------------------------------------
// Synthetic tables of SURVEY.md 8(d) (bench / parity inputs), bit-identical to
// orc_synth_p_drive / orc_synth_p_dest_dense of the oracle.
// ---------------------------------------------------------------------------------------
__global__ void k_synth_p_drive(double *pdrive, int Z, int T, uint64_t table_seed)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Z * T) return;
    int z = i % Z, t = i / Z;
    double u = table_uniform(table_seed, z, t, 0, kStreamPDrive);
    pdrive[i] = 0.1 + 0.8 * u;
}

// Destination popularity of the skewed synthetic tables: weight 1 / (q + rank(d)), rank a fixed scrambling of the zone ids
// (Zipf-Mandelbrot; the shape of the reference's output_24_0.svg: a few destinations many times as likely as the mean).  q <= 0: flat.
__host__ __device__ inline double synth_dest_weight(double u, int o, int d, int Z, int64_t skew_q)
{
    if (o == d) return 0.0;  // Uber rows have origin != destination (README.md:244)
    double w = u * u;        // ((m - min) / (max - min))^2, src/createpdestin.jl:24
    if (skew_q > 0) w = w / static_cast<double>(skew_q + (static_cast<int64_t>(d) * 7919 + 13) % Z);
    return w;
}

// one thread per (origin, hour): sequential row sum, then the normalised row, written in
// the reference's layout (origin fastest -> coalesced across the wave)
__global__ void k_synth_p_dest(double *__restrict__ p, int Z, uint64_t table_seed, int64_t skew_q)
{
    int o = blockIdx.x * blockDim.x + threadIdx.x;
    int t = blockIdx.y;
    if (o >= Z) return;
    double nf = 0.0;
    for (int d = 0; d < Z; ++d) {
        double u = table_uniform(table_seed, o, d, t, kStreamPDest);
        nf = nf + synth_dest_weight(u, o, d, Z, skew_q);
    }
    double *dst = p + static_cast<size_t>(t) * Z * Z + o;
    for (int d = 0; d < Z; ++d) {
        double u = table_uniform(table_seed, o, d, t, kStreamPDest);
        double w = synth_dest_weight(u, o, d, Z, skew_q);
        if (nf > 0) w = w / nf;
        dst[static_cast<size_t>(d) * Z] = w;
    }
}

// Melbourne-shaped synthetic datamatrix + distance matrix (8.68 % of the (o, d, t) cells hold a mean of 300-2400 s
// and a standard deviation of 10-40 % of it; distances from random centroids), generated where it is used -- bench.py's per-dataset
// figures start from it without 2 GB crossing PCIe.  Bit-identical to orc_synth_datamatrix of the oracle (table streams 0x102-0x104).
constexpr uint32_t kStreamDataA = 0x102u, kStreamDataB = 0x103u, kStreamCentroid = 0x104u;
__global__ __launch_bounds__(256) void k_synth_datamatrix(double *__restrict__ dm, int Z, int T, uint64_t table_seed, double density)
{
    const int o = blockIdx.x * 256 + threadIdx.x;
    const int d = blockIdx.y, t = blockIdx.z;
    if (o >= Z) return;
    double ua, ub, uc, ud;
    car_uniforms(table_seed, (static_cast<uint64_t>(static_cast<uint32_t>(d)) << 32) | static_cast<uint32_t>(o), static_cast<uint32_t>(t), kStreamDataA, ua, ub);
    car_uniforms(table_seed, (static_cast<uint64_t>(static_cast<uint32_t>(d)) << 32) | static_cast<uint32_t>(o), static_cast<uint32_t>(t), kStreamDataB, uc, ud);
    double mean = 0, sd = 0;
    if (o != d && ua < density) {
        mean = 300.0 + 2100.0 * ub;
        sd = mean * (0.1 + 0.3 * uc);
    }
    const size_t cell = static_cast<size_t>(o) + static_cast<size_t>(Z) * (d + static_cast<size_t>(Z) * t);
    dm[cell] = mean;
    dm[cell + static_cast<size_t>(Z) * Z * T] = sd;
}

__global__ __launch_bounds__(256) void k_synth_dist(double *__restrict__ dist, int Z, uint64_t table_seed)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int j = blockIdx.y;
    if (i >= Z) return;
    double ui, vi, uj, vj;
    car_uniforms(table_seed, static_cast<uint64_t>(static_cast<uint32_t>(i)), 0u, kStreamCentroid, ui, vi);
    car_uniforms(table_seed, static_cast<uint64_t>(static_cast<uint32_t>(j)), 0u, kStreamCentroid, uj, vj);
    const double lat_i = -38.5 + 1.5 * ui, lon_i = 144.0 + 2.0 * vi, lat_j = -38.5 + 1.5 * uj, lon_j = 144.0 + 2.0 * vj;
    const double dlon = 0.79 * (lon_i - lon_j), dlat = lat_i - lat_j;  // equirectangular, cos(lat) frozen at 0.79 (src/processgeodata.jl:157)
    double v = (i == j) ? 1.0 : 111.3 * sqrt(dlon * dlon + dlat * dlat);
    if (v == 0) v = 1.0;
    dist[i + static_cast<size_t>(Z) * j] = v;
}

// ---------------------------------------------------------------------------------------
// Categorical draw on a canonical CDF row: first j with u <= cdf[j]  ==  the chained test
// range_low < u <= range_up of src/resampling.jl:38-45 for u > 0.  Deviation D1 (the
// reference leaves destination = 0 and crashes, Appendix A-7) is folded into the bounds:
// u == 0 -> first zone with p > 0 ; u > cdf[Z-1] -> last zone with p > 0.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double clamp_u(double u, double last)
{
    double ue = (u == 0.0) ? kTrueMin : u;
    return (ue > last) ? last : ue;
}

template <typename RowT>
__device__ __forceinline__ int lower_bound_row(RowT row, int n, double ue)
{
    int lo = 0;
    while (n > 0) {
        int half = n >> 1;
        double v = row[lo + half];
        if (v < ue) {
            lo += half + 1;
            n -= half + 1;
        } else {
            n = half;
        }
    }
    return lo;
}


// ---------------------------------------------------------------------------------------
// A CDF row as an implicit binary search tree in breadth-first (Eytzinger) order, the LDS
// layout of the zone kernels: the sorted layout's probes sit at power-of-two strides, which all
// map to the same LDS banks (72 % of the LDS cycles of the first zone kernel were bank
// conflicts); breadth-first, the top levels are broadcast reads and the rest spread evenly.
// Tree over elements 0..Z-2 (height H, 2^H >= Z); slot 0 holds element Z-1 (the row total);
// slots of ranks Z..2^H-1 hold +inf.
// ---------------------------------------------------------------------------------------
__host__ __device__ inline int tree_height(int Z)
{
    int h = 1;
    while ((1 << h) < Z) ++h;
    return h;
}

__device__ __forceinline__ uint32_t eytz_pos(uint32_t e, int Z, int H)
{
    if (e == static_cast<uint32_t>(Z - 1)) return 0u;
    uint32_t r = e + 1;
    int tz = __builtin_ctz(r);
    return (1u << (H - 1 - tz)) + (r >> (tz + 1));
}

// tree walk result -> sorted index of the first element >= ue (ue <= last guarantees one exists)
__device__ __forceinline__ uint32_t eytz_decode(uint32_t i, int Z, int H)
{
    i >>= __ffs(static_cast<int>(~i));
    if (i == 0) return static_cast<uint32_t>(Z - 1);
    int lv = 31 - __clz(static_cast<int>(i));
    uint32_t off = i - (1u << lv);
    return ((2 * off + 1) << (H - 1 - lv)) - 1;
}

// travel time of one driving car-hour (src/resampling.jl:57-69), q16 fixed point
__device__ __forceinline__ long long travel_time_q16(const double *__restrict__ dm, int Z, int T, int t,
                                                     uint32_t origin, uint32_t dest, uint64_t seed,
                                                     uint64_t car, uint32_t step)
{
    if (origin == dest) return q16(300.0);
    size_t cell = origin + static_cast<size_t>(Z) * (dest + static_cast<size_t>(Z) * t);
    double mean = dm[cell];
    double sd = dm[cell + static_cast<size_t>(Z) * Z * T];
    if (sd == 0) sd = 0.1 * mean;
    return q16(truncnormal_pm10(seed, car, step, 1, mean, sd));
}

// ---------------------------------------------------------------------------------------
// CPM_KERNEL_CAR: one hour, one thread per car.  Bernoulli (src/resampling.jl:11-22) and
// categorical (:26-49) draw, state update (:81-83) folded into the record written for the
// next hour.  The CDF row is searched where it lies (HBM / MALL / L2).
// ---------------------------------------------------------------------------------------
template <bool TRAVEL>
__global__ __launch_bounds__(256) void k_step_car(const uint32_t *__restrict__ zin, uint32_t *__restrict__ rec_out,
                                                  const double *__restrict__ pdrive_t,
                                                  const double *__restrict__ cdf_t, int Z, int Zp, int64_t n,
                                                  CarIndex cars, uint32_t step, uint64_t seed,
                                                  const double *__restrict__ dm, int T, int t,
                                                  unsigned long long *tt_sum)
{
    int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    long long tt = 0;
    if (i < n) {
        uint32_t zone = zin[i] & kZoneMask;
        uint64_t car = cars.global(static_cast<uint32_t>(i));
        double ub, uc;
        car_uniforms(seed, car, step, 0, ub, uc);
        bool drive = ub <= pdrive_t[zone];
        uint32_t dest = zone;
        if (drive) {
            const double *row = cdf_t + static_cast<size_t>(zone) * Zp;
            double last = row[Z - 1];
            if (last != 0.0) dest = static_cast<uint32_t>(lower_bound_row(row, Z, clamp_u(uc, last)));
            if (TRAVEL) tt = travel_time_q16(dm, Z, T, t, zone, dest, seed, car, step);
        }
        rec_out[i] = dest | (drive ? kDriveBit : 0u);
    }
    if (TRAVEL) {
        // wave-level sum, one atomic per wave
        for (int off = 32; off > 0; off >>= 1) tt += __shfl_down(tt, off, 64);
        if ((threadIdx.x & 63) == 0 && tt != 0) atomicAdd(tt_sum, static_cast<unsigned long long>(tt));
    }
}

// ---------------------------------------------------------------------------------------
// Zone x hour histogram of saveresults (src/saveresults.jl:8-17): parking[z,t] counts every
// car whose state at hour t is z (drivers included), driving[z,t] those that drive, binned by
// origin.  One (chunk, hour) per workgroup; bins are privatised in LDS (2 x Z u32) and
// flushed with contiguous global atomics, so HBM sees 8 B per car-hour and Z*8 B per block.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_histogram(const uint32_t *__restrict__ zone0,
                                                    const uint32_t *__restrict__ rec, int64_t n, int Z,
                                                    unsigned long long *__restrict__ parking,
                                                    unsigned long long *__restrict__ driving, int64_t chunk)
{
    extern __shared__ uint32_t bins[];
    uint32_t *lp = bins, *ld = bins + Z;
    const int t = blockIdx.y;
    for (int z = threadIdx.x; z < 2 * Z; z += blockDim.x) bins[z] = 0;
    __syncthreads();
    const uint32_t *zsrc = (t == 0) ? zone0 : rec + static_cast<size_t>(t - 1) * n;
    const uint32_t *fsrc = rec + static_cast<size_t>(t) * n;
    int64_t i0 = static_cast<int64_t>(blockIdx.x) * chunk;
    int64_t i1 = min(i0 + chunk, n);
    for (int64_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        uint32_t z = zsrc[i] & kZoneMask;
        uint32_t f = fsrc[i] >> 31;
        atomicAdd(&lp[z], 1u);
        if (f) atomicAdd(&ld[z], 1u);
    }
    __syncthreads();
    for (int z = threadIdx.x; z < Z; z += blockDim.x) {
        uint32_t a = lp[z], b = ld[z];
        if (a) atomicAdd(&parking[static_cast<size_t>(t) * Z + z], static_cast<unsigned long long>(a));
        if (b) atomicAdd(&driving[static_cast<size_t>(t) * Z + z], static_cast<unsigned long long>(b));
    }
}

// fallback for Z too large for LDS bins
__global__ void k_histogram_global(const uint32_t *__restrict__ zone0, const uint32_t *__restrict__ rec,
                                   int64_t n, int Z, unsigned long long *parking, unsigned long long *driving)
{
    const int t = blockIdx.y;
    int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t z = ((t == 0) ? zone0[i] : rec[static_cast<size_t>(t - 1) * n + i]) & kZoneMask;
    uint32_t f = rec[static_cast<size_t>(t) * n + i] >> 31;
    atomicAdd(&parking[static_cast<size_t>(t) * Z + z], 1ull);
    if (f) atomicAdd(&driving[static_cast<size_t>(t) * Z + z], 1ull);
}

// ---------------------------------------------------------------------------------------
// Compat export: one hour column of the reference's state_matrix / transition_matrix
// (src/initializestates.jl:6-7): state[:,t], trans[:,t,1..4].  Travel time / distance are
// re-derived from the same Philox streams, so nothing extra is stored per car.
// ---------------------------------------------------------------------------------------
__global__ void k_export_hour(const uint32_t *__restrict__ zsrc, const uint32_t *__restrict__ rec_t, int64_t n,
                              CarIndex cars, int64_t *__restrict__ state_col, double *__restrict__ drive_col,
                              double *__restrict__ dest_col, double *__restrict__ time_col,
                              double *__restrict__ dist_col, const double *__restrict__ dm,
                              const double *__restrict__ dist, int Z, int T, int t, uint32_t step, uint64_t seed)
{
    int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t zone = zsrc[i] & kZoneMask;
    uint32_t r = rec_t[i];
    uint32_t dest = r & kZoneMask;
    bool drive = (r & kDriveBit) != 0;
    state_col[i] = static_cast<int64_t>(zone) + 1;
    drive_col[i] = drive ? 1.0 : 0.0;
    dest_col[i] = static_cast<double>(dest) + 1.0;
    double tm = 0.0, ds = 0.0;
    if (drive && dm != nullptr) {
        if (zone == dest) {
            tm = 300.0;
            ds = 1.0;
        } else {
            uint64_t car = cars.global(static_cast<uint32_t>(i));
            size_t cell = zone + static_cast<size_t>(Z) * (dest + static_cast<size_t>(Z) * t);
            double mean = dm[cell];
            double sd = dm[cell + static_cast<size_t>(Z) * Z * T];
            if (sd == 0) sd = 0.1 * mean;
            tm = truncnormal_pm10(seed, car, step, 1, mean, sd);
            mean = dist[zone + static_cast<size_t>(Z) * dest];
            ds = truncnormal_pm10(seed, car, step, 2, mean, 0.1 * mean);
        }
    }
    time_col[i] = tm;
    dist_col[i] = ds;
}

}  // namespace cpm
