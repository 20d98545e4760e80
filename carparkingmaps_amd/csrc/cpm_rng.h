// cpm_rng.h -- device-side RNG and elementary functions of the sampler path (gfx950).
//
// Philox4x32-10 counter-based RNG (Salmon et al., SC'11; constants as published) replaces
// the reference's serial global rand() (src/resampling.jl:13,29): a car's draw depends only
// on (seed, global car id, step, stream), never on which lane/wave/GPU computes it.
//
// Everything here is compiled with -ffp-contract=off: each f64 operation is the IEEE
// operation the CPU twin performs, so comparisons against table entries agree bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cpm {

constexpr uint32_t kPhiloxM0 = 0xD2511F53u;
constexpr uint32_t kPhiloxM1 = 0xCD9E8D57u;
constexpr uint32_t kPhiloxW0 = 0x9E3779B9u;
constexpr uint32_t kPhiloxW1 = 0xBB67AE85u;

// table-generation streams (synthetic inputs only)
constexpr uint32_t kStreamPDrive = 0x100u;
constexpr uint32_t kStreamPDest = 0x101u;

struct U4 {
    uint32_t x, y, z, w;
};

__device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                            uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32x32->64 product per multiplier (v_mad_u64_u32) instead of a mul_hi + mul_lo pair
        uint64_t p0 = static_cast<uint64_t>(kPhiloxM0) * c0;
        uint64_t p1 = static_cast<uint64_t>(kPhiloxM1) * c2;
        uint32_t hi0 = static_cast<uint32_t>(p0 >> 32), lo0 = static_cast<uint32_t>(p0);
        uint32_t hi1 = static_cast<uint32_t>(p1 >> 32), lo1 = static_cast<uint32_t>(p1);
        // three-input xor in one instruction (v_bitop3_b32, truth table 0x96 = a ^ b ^ c; hipcc emits two v_xor otherwise)
        uint32_t n0 = __builtin_amdgcn_bitop3_b32(hi1, k0, c1, 0x96);
        uint32_t n2 = __builtin_amdgcn_bitop3_b32(hi0, k1, c3, 0x96);
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += kPhiloxW0; k1 += kPhiloxW1;
    }
    return U4{c0, c1, c2, c3};
}

// 53 high bits of (hi:lo) as a double in [0,1): the range of Julia's rand()
__device__ __forceinline__ double u53(uint32_t lo, uint32_t hi)
{
    uint64_t x = (static_cast<uint64_t>(hi) << 32) | lo;
    return static_cast<double>(x >> 11) * 0x1.0p-53;
}

// the two uniforms of (car, step, stream)
__device__ __forceinline__ void car_uniforms(uint64_t seed, uint64_t car, uint32_t step,
                                             uint32_t stream, double &u0, double &u1)
{
    U4 r = philox4x32_10(static_cast<uint32_t>(car), static_cast<uint32_t>(car >> 32), step, stream,
                         static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32));
    u0 = u53(r.x, r.y);
    u1 = u53(r.z, r.w);
}

// Integer form of the Bernoulli draw.  u = k * 2^-53 with k the 53 high bits, so
// u <= p  <=>  k <= floor(p * 2^53): the comparison of src/resampling.jl:15 without converting
// the draw to f64.  NaN -> never drives, p >= 1 -> always, p < 0 -> never (Appendix A-3, A-6).
__device__ __forceinline__ long long bernoulli_threshold(double p)
{
    if (!(p >= 0.0)) return -1;           // NaN or negative: k <= -1 is never true
    if (p >= 1.0) return 0x7fffffffffffffffLL;
    return static_cast<long long>(floor(p * 0x1.0p53));  // exact: scaling by a power of two
}

// (53-bit integer of the Bernoulli draw, f64 categorical uniform) of (car, step, stream 0)
__device__ __forceinline__ void car_draws(uint64_t seed, uint64_t car, uint32_t step, long long &kb, double &uc)
{
    U4 r = philox4x32_10(static_cast<uint32_t>(car), static_cast<uint32_t>(car >> 32), step, 0u,
                         static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32));
    kb = static_cast<long long>(((static_cast<uint64_t>(r.y) << 32) | r.x) >> 11);
    uc = u53(r.z, r.w);
}

__device__ __forceinline__ double table_uniform(uint64_t seed, uint32_t a, uint32_t b, uint32_t c,
                                                uint32_t stream)
{
    U4 r = philox4x32_10(a, b, c, stream, static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32));
    return u53(r.x, r.y);
}

// exp(-y), y >= 0, from +,-,*,/ and exact power-of-two scaling only: reproduces the CPU
// twin bit for bit (no libm on either side).
__device__ __forceinline__ double exp_neg(double y)
{
    if (!(y >= 0.0)) return 1.0;
    if (y > 745.0) return 0.0;
    const double LOG2E = 1.4426950408889634074;
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    double n = floor(y * LOG2E + 0.5);
    double r = (y - n * LN2_HI) - n * LN2_LO;
    double x = -r;
    double p = 1.0 / 6227020800.0;
    p = p * x + 1.0 / 479001600.0;
    p = p * x + 1.0 / 39916800.0;
    p = p * x + 1.0 / 3628800.0;
    p = p * x + 1.0 / 362880.0;
    p = p * x + 1.0 / 40320.0;
    p = p * x + 1.0 / 5040.0;
    p = p * x + 1.0 / 720.0;
    p = p * x + 1.0 / 120.0;
    p = p * x + 1.0 / 24.0;
    p = p * x + 1.0 / 6.0;
    p = p * x + 0.5;
    p = p * x + 1.0;
    p = p * x + 1.0;
    int ni = static_cast<int>(n);
    int h = ni / 2;
    double s1 = __longlong_as_double(static_cast<long long>(1023 - h) << 52);
    double s2 = __longlong_as_double(static_cast<long long>(1023 - (ni - h)) << 52);
    return (p * s1) * s2;
}

// Truncated normal on [0.9 mu, 1.1 mu] (src/resampling.jl:68,74).  The reference uses
// Distributions.jl (absent, version unpinned); this is the build's own pinned sampler:
// uniform proposal on the window, accept with exp(-(x-mu)^2 / (2 sigma^2)); attempt k draws
// Philox stream stream0 + 2k; after 4096 rejections mu.
// The acceptance test u2 <= exp_neg(y) is decided WITHOUT evaluating exp_neg wherever the alternating series brackets it:
// 1 - y + y^2/2 - y^3/6 <= e^-y <= 1 - y + y^2/2 for y >= 0.  A draw at least 2^-45 below the lower bound is accepted, one at least
// 2^-45 above the upper bound rejected -- 2^-45 is a hundred times the rounding error of either polynomial and of exp_neg (a few
// 1e-16), so the decision is the one the full evaluation would take, bit for bit (the oracle always evaluates exp_neg: results are
// compared exactly in tests/).  On the path y <= 0.5 (the window is +-0.1 mu, sigma >= 0.1 mu) and the undecided band is y^3/6 wide:
// a wave needs exp_neg -- sixty f64 operations, half of the travel kernel's arithmetic -- for a few per cent of its attempts.
// One attempt (Philox stream `stream`) for the window of mean mu and inv2s2 = 1 / (2 sigma^2): the proposal in x, true = accepted.
// Wave-level inside (the ballot): call it where the lanes that are still drawing are the active ones.
__device__ __forceinline__ bool truncnormal_attempt(uint64_t seed, uint64_t car, uint32_t step, uint32_t stream, double mu, double inv2s2, double &x)
{
    const double lo = 0.9 * mu, hi = 1.1 * mu;
    const double w = hi - lo;
    double u1, u2;
    car_uniforms(seed, car, step, stream, u1, u2);
    x = lo + w * u1;
    const double d = x - mu;
    const double y = (d * d) * inv2s2;
    bool decided = false, accept = false;
    if (y <= 1.0) {  // (false for NaN; y >= 0 otherwise)
        const double h = 0.5 * (y * y);
        const double upper = (1.0 - y) + h;
        const double lower = upper - (h * y) * (1.0 / 3.0);
        if (u2 <= lower - 0x1.0p-45) {
            decided = true;
            accept = true;
        } else if (u2 > upper + 0x1.0p-45) {
            decided = true;
        }
    }
    if (__builtin_amdgcn_ballot_w64(!decided) != 0ull) {  // (wave-uniform: most waves skip the evaluation altogether)
        if (!decided) accept = u2 <= exp_neg(y);
    }
    return accept;
}
constexpr uint32_t kTruncnormalAttempts = 4096;  // then mu

__device__ __forceinline__ double truncnormal_inv2s2(double sigma) { return 1.0 / (2.0 * sigma * sigma); }

__device__ __forceinline__ double truncnormal_pm10(uint64_t seed, uint64_t car, uint32_t step,
                                                   uint32_t stream0, double mu, double sigma)
{
    const double inv2s2 = truncnormal_inv2s2(sigma);
    for (uint32_t k = 0; k < kTruncnormalAttempts; ++k) {
        double x;
        if (truncnormal_attempt(seed, car, step, stream0 + 2 * k, mu, inv2s2, x)) return x;
    }
    return mu;
}

__device__ __forceinline__ long long q16(double seconds)
{
    return __double2ll_rn(seconds * 65536.0);
}

}  // namespace cpm
