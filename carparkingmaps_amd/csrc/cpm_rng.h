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

// ---- the deterministic f64 kit of the truncated-normal sampler: +,-,*,/ and bit manipulation only (no libm), the same statements in
// the same order as the CPU restatement under oracle/ (orc_log, orc_sqrt, orc_erf, orc_ppnd), so that every value is reproduced bit for bit
// ln(x), x > 0 normal: x = m * 2^e with m in [sqrt(1/2), sqrt(2)), ln m = 2 atanh((m-1)/(m+1)) as a series in s^2 (|s| < 0.1716)
__device__ __forceinline__ double det_log(double x)
{
    const unsigned long long b = static_cast<unsigned long long>(__double_as_longlong(x));
    int e = static_cast<int>((b >> 52) & 0x7FF) - 1023;
    double m = __longlong_as_double(static_cast<long long>((b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull));
    if (m > 1.4142135623730951) {
        m = m * 0.5;
        e += 1;
    }
    const double s = (m - 1.0) / (m + 1.0);
    const double s2 = s * s;
    double p = 1.0 / 23.0;
    p = p * s2 + 1.0 / 21.0;
    p = p * s2 + 1.0 / 19.0;
    p = p * s2 + 1.0 / 17.0;
    p = p * s2 + 1.0 / 15.0;
    p = p * s2 + 1.0 / 13.0;
    p = p * s2 + 1.0 / 11.0;
    p = p * s2 + 1.0 / 9.0;
    p = p * s2 + 1.0 / 7.0;
    p = p * s2 + 1.0 / 5.0;
    p = p * s2 + 1.0 / 3.0;
    p = p * s2 + 1.0;
    const double lm = (2.0 * s) * p;
    const double ef = static_cast<double>(e);
    return ef * 6.93147180369123816490e-01 + (lm + ef * 1.90821492927058770002e-10);
}
// sqrt(x), x > 0 normal: Newton from a bit-level first guess, five steps
__device__ __forceinline__ double det_sqrt(double x)
{
    double y = __longlong_as_double(static_cast<long long>((static_cast<unsigned long long>(__double_as_longlong(x)) >> 1) + 0x1FF8000000000000ull));
    y = 0.5 * (y + x / y);
    y = 0.5 * (y + x / y);
    y = 0.5 * (y + x / y);
    y = 0.5 * (y + x / y);
    y = 0.5 * (y + x / y);
    return y;
}
// erf(x), x >= 0: W. J. Cody's rational Chebyshev approximations (Math. Comp. 23 (1969) 631-637), exp(-x^2) by exp_neg
__device__ __forceinline__ double det_erf(double x)
{
    if (!(x >= 0.0)) return 0.0;
    if (x <= 0.46875) {
        const double y = x * x;
        double num = 1.85777706184603153e-1 * y;
        double den = y;
        num = (num + 3.16112374387056560e00) * y;
        den = (den + 2.36012909523441209e01) * y;
        num = (num + 1.13864154151050156e02) * y;
        den = (den + 2.44024637934444173e02) * y;
        num = (num + 3.77485237685302021e02) * y;
        den = (den + 1.28261652607737228e03) * y;
        return x * (num + 3.20937758913846947e03) / (den + 2.84423683343917062e03);
    }
    if (x <= 4.0) {
        double num = 2.15311535474403846e-8 * x;
        double den = x;
        num = (num + 5.64188496988670089e-1) * x;
        den = (den + 1.57449261107098347e01) * x;
        num = (num + 8.88314979438837594e00) * x;
        den = (den + 1.17693950891312499e02) * x;
        num = (num + 6.61191906371416295e01) * x;
        den = (den + 5.37181101862009858e02) * x;
        num = (num + 2.98635138197400131e02) * x;
        den = (den + 1.62138957456669019e03) * x;
        num = (num + 8.81952221241769090e02) * x;
        den = (den + 3.29079923573345963e03) * x;
        num = (num + 1.71204761263407058e03) * x;
        den = (den + 4.36261909014324716e03) * x;
        num = (num + 2.05107837782607147e03) * x;
        den = (den + 3.43936767414372164e03) * x;
        const double erfc = exp_neg(x * x) * ((num + 1.23033935479799725e03) / (den + 1.23033935480374942e03));
        return 1.0 - erfc;
    }
    if (x >= 6.0) return 1.0;  // erfc < 2.2e-17
    const double y = 1.0 / (x * x);
    double num = 1.63153871373020978e-2 * y;
    double den = y;
    num = (num + 3.05326634961232344e-1) * y;
    den = (den + 2.56852019228982242e00) * y;
    num = (num + 3.60344899949804439e-1) * y;
    den = (den + 1.87295284992346725e00) * y;
    num = (num + 1.25781726111229246e-1) * y;
    den = (den + 5.27905102951428412e-1) * y;
    num = (num + 1.60837851487422766e-2) * y;
    den = (den + 6.05183413124413191e-2) * y;
    const double r = y * (num + 6.58749161529837803e-4) / (den + 2.33520497626869185e-3);
    const double erfc = exp_neg(x * x) * ((5.6418958354775628695e-1 - r) / x);
    return 1.0 - erfc;
}
// Phi^-1(1/2 + q), |q| <= 1/2: Wichura's PPND16 (Algorithm AS 241, Appl. Statist. 37 (1988) 477-484), from q.  The tails (|q| > 0.425:
// only windows wider than +-1.44 sigma reach them) are out of line: a logarithm, a square root and a second rational.
__device__ __noinline__ double ppnd_tail(double q)
{
    const double aq = q < 0.0 ? -q : q;
    double r = 0.5 - aq;  // min(p, 1 - p)
    double val;
    if (!(r > 0.0)) {
        val = 9.0;  // beyond every window the sampler clamps to
    } else {
        r = det_sqrt(-det_log(r));
        if (r <= 5.0) {
            r = r - 1.6;
            double num = 7.74545014278341407640e-4;
            num = num * r + 2.27238449892691845833e-2;
            num = num * r + 2.41780725177450611770e-1;
            num = num * r + 1.27045825245236838258e0;
            num = num * r + 3.64784832476320460504e0;
            num = num * r + 5.76949722146069140550e0;
            num = num * r + 4.63033784615654529590e0;
            num = num * r + 1.42343711074968357734e0;
            double den = 1.05075007164441684324e-9;
            den = den * r + 5.47593808499534494600e-4;
            den = den * r + 1.51986665636164571966e-2;
            den = den * r + 1.48103976427480074590e-1;
            den = den * r + 6.89767334985100004550e-1;
            den = den * r + 1.67638483018380384940e0;
            den = den * r + 2.05319162663775882187e0;
            den = den * r + 1.0;
            val = num / den;
        } else {
            r = r - 5.0;
            double num = 2.01033439929228813265e-7;
            num = num * r + 2.71155556874348757815e-5;
            num = num * r + 1.24266094738807843860e-3;
            num = num * r + 2.65321895265761230930e-2;
            num = num * r + 2.96560571828504891230e-1;
            num = num * r + 1.78482653991729133580e0;
            num = num * r + 5.46378491116411436990e0;
            num = num * r + 6.65790464350110377720e0;
            double den = 2.04426310338993978564e-15;
            den = den * r + 1.42151175831644588870e-7;
            den = den * r + 1.84631831751005468180e-5;
            den = den * r + 7.86869131145613259100e-4;
            den = den * r + 1.48753612908506148525e-2;
            den = den * r + 1.36929880922735805310e-1;
            den = den * r + 5.99832206555887937690e-1;
            den = den * r + 1.0;
            val = num / den;
        }
    }
    return q < 0.0 ? -val : val;
}
__device__ __forceinline__ double ppnd(double q)
{
    const double aq = q < 0.0 ? -q : q;
    if (__builtin_expect(aq > 0.425, 0)) return ppnd_tail(q);
    const double r = 0.180625 - q * q;
    double num = 2.5090809287301226727e+3;
    num = num * r + 3.3430575583588128105e+4;
    num = num * r + 6.7265770927008700853e+4;
    num = num * r + 4.5921953931549871457e+4;
    num = num * r + 1.3731693765509461125e+4;
    num = num * r + 1.9715909503065514427e+3;
    num = num * r + 1.3314166789178437745e+2;
    num = num * r + 3.3871328727963666080e0;
    double den = 5.2264952788528545610e+3;
    den = den * r + 2.8729085735721942674e+4;
    den = den * r + 3.9307895800092710610e+4;
    den = den * r + 2.1213794301586595867e+4;
    den = den * r + 5.3941960214247511077e+3;
    den = den * r + 6.8718700749205790830e+2;
    den = den * r + 4.2313330701600911252e+1;
    den = den * r + 1.0;
    return q * num / den;
}

// Truncated normal on [0.9 mu, 1.1 mu] (src/resampling.jl:68,74).  The reference uses Distributions.jl (absent, version unpinned);
// this is the build's own pinned sampler: ONE draw by inversion (the restatement under oracle/: orc_truncnormal_draw).  With E = erf(a / sqrt 2),
// a = 0.1 mu / sigma, the mass of N(mu, sigma) inside the window, z = Phi^-1(1/2 + (u - 1/2) E) is a standard normal truncated to +-a and
// x = mu + sigma z; u = the first uniform of Philox stream `stream`.  E is a property of the cell: the travel rows carry it
// (TravelCell), the kernels that gather (mean, std) from the datamatrix compute it per driver.
// (Rounds 1-3 pinned a rejection sampler -- uniform proposal, accept with exp(-(x-mu)^2 / (2 sigma^2)), ~1.2 Philox calls and an f64
//  exp bracket per driver in a retry loop: what bound the travel kernel, profiles/round3_notes.md.)
struct TravelCell {
    double mu, sigma, mass;  // mean, sigma (std, or a tenth of the mean where the data hold none: src/resampling.jl:65-67), E
};
__device__ __forceinline__ double truncnormal_mass(double mu, double sigma)
{
    const double a = (0.1 * mu) / sigma;
    return det_erf(a * 7.0710678118654752440e-1);
}
__device__ __forceinline__ double truncnormal_draw(double u, double mu, double sigma, double mass)
{
    if (!(sigma > 0.0) || !(mass > 0.0)) return mu;
    const double z = ppnd((u - 0.5) * mass);
    double x = mu + sigma * z;
    const double lo = 0.9 * mu, hi = 1.1 * mu;
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return x;
}
__device__ __forceinline__ double truncnormal_pm10(uint64_t seed, uint64_t car, uint32_t step, uint32_t stream, double mu, double sigma)
{
    double u1, u2;
    car_uniforms(seed, car, step, stream, u1, u2);
    return truncnormal_draw(u1, mu, sigma, truncnormal_mass(mu, sigma));
}

__device__ __forceinline__ long long q16(double seconds)
{
    return __double2ll_rn(seconds * 65536.0);
}

}  // namespace cpm
