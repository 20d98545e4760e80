/*
 * cpm_oracle.c -- CPU restatement (ORACLE) of the CarParkingMaps sampler path.
 * TEST INFRASTRUCTURE ONLY; see cpm_oracle.h for the contract, the layout and
 * the "parity unpinned" statement.  Every function cites the reference lines
 * it restates (paths relative to /root/reference).
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile).
 * -ffp-contract=off matters: every f64 result here must be the IEEE result of
 * the individual +,-,*,/ the reference performs, with no fused multiply-add.
 */
#include "cpm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------ */
/* Philox4x32-10 (Random123; Salmon et al., SC'11).  Constants as published. */
/* ------------------------------------------------------------------------ */
#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
        uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += PHILOX_W0; k1 += PHILOX_W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static inline double u53(uint32_t lo, uint32_t hi)
{
    uint64_t x = ((uint64_t)hi << 32) | lo;
    return (double)(x >> 11) * 0x1.0p-53; /* exact: 53-bit integer times a power of two */
}

void orc_uniforms(uint64_t seed, uint64_t car, uint32_t step, uint32_t stream,
                  double *u0, double *u1)
{
    uint32_t ctr[4] = {(uint32_t)car, (uint32_t)(car >> 32), step, stream};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t w[4];
    orc_philox4x32_10(ctr, key, w);
    *u0 = u53(w[0], w[1]);
    *u1 = u53(w[2], w[3]);
}

/* table-generation streams (synthetic inputs only) */
#define TS_PDRIVE 0x100u
#define TS_PDEST 0x101u
#define TS_DATA_A 0x102u
#define TS_DATA_B 0x103u
#define TS_CENTROID 0x104u

static inline void table_uniforms(uint64_t seed, uint32_t a, uint32_t b, uint32_t c,
                                  uint32_t stream, double *u0, double *u1)
{
    uint32_t ctr[4] = {a, b, c, stream};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t w[4];
    orc_philox4x32_10(ctr, key, w);
    *u0 = u53(w[0], w[1]);
    *u1 = u53(w[2], w[3]);
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------ */
/* exp(-y) from +,-,*,/ and exact scaling only, so that a GPU twin compiled   */
/* with contraction off reproduces it bit for bit.  |rel err| < 1e-15.        */
/* ------------------------------------------------------------------------ */
double orc_exp_neg(double y)
{
    if (!(y >= 0.0)) return 1.0;
    if (y > 745.0) return 0.0;
    const double LOG2E = 1.4426950408889634074;
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    double n = floor(y * LOG2E + 0.5);
    double r = (y - n * LN2_HI) - n * LN2_LO; /* r in about [-0.35, 0.35]; want exp(-r) */
    double x = -r;
    /* Taylor to x^13 / 13!, Horner, unfused */
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
    /* multiply by 2^-n in two exact-or-correctly-rounded steps (n <= 1075) */
    int ni = (int)n;
    int h = ni / 2;
    uint64_t b1 = (uint64_t)(1023 - h) << 52, b2 = (uint64_t)(1023 - (ni - h)) << 52;
    double s1, s2;
    memcpy(&s1, &b1, 8);
    memcpy(&s2, &b2, 8);
    return (p * s1) * s2;
}

/* ------------------------------------------------------------------------ */
/* Table builders                                                             */
/* ------------------------------------------------------------------------ */

/* Julia's maximum/minimum propagate NaN (Appendix A-3); C fmax/fmin do not. */
static inline double jl_max(double a, double b) { return (a != a || b != b) ? NAN : (a > b ? a : b); }
static inline double jl_min(double a, double b) { return (a != a || b != b) ? NAN : (a < b ? a : b); }

/* x^e as the reference writes it: Float64^Float64 -> pow; Float64^Int -> integer power
 * by repeated multiplication (Julia's power_by_squaring gives x*x for 2). */
static inline double jl_pow(double x, double e, int e_is_integer)
{
    if (e_is_integer) {
        long n = (long)e;
        if (n == 0) return 1.0;
        if (n == 1) return x;
        if (n == 2) return x * x;
        if (n == 3) return x * x * x;
        /* power by squaring, Base.power_by_squaring order */
        long t = 0;
        long m = n;
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
    return pow(x, e);
}

/* src/createpdrive.jl:3-38 */
int orc_createpdrive(const double *dm, const double *dist, int64_t Z, int64_t T, double p_min,
                     double p_max, double e_drive, double *p_drive)
{
    if (!dm || !dist || !p_drive || Z <= 0 || T <= 0) return ORC_ERR_BADARG;
    double *mean_sum = (double *)malloc(sizeof(double) * (size_t)T);
    for (int64_t i = 0; i < Z; ++i) {
        double max_t = 0, min_t = 0;
        for (int64_t t = 0; t < T; ++t) { /* :11-21 */
            double s = 0;
            int64_t counter = 0;
            for (int64_t j = 0; j < Z; ++j) { /* :14-19 */
                double m = dm[i + Z * (j + Z * t)];
                if (m != 0) {
                    s = s + m / dist[i + Z * j];
                    counter += 1;
                }
            }
            mean_sum[t] = s / (double)counter; /* :20 ; 0/0 = NaN when no data */
            if (t == 0) { max_t = mean_sum[t]; min_t = mean_sum[t]; }
            else { max_t = jl_max(max_t, mean_sum[t]); min_t = jl_min(min_t, mean_sum[t]); } /* :22-23 */
        }
        for (int64_t t = 0; t < T; ++t) { /* :27-33 */
            if (max_t > 0) {
                p_drive[i + Z * t] =
                    p_min + (p_max - p_min) * pow((mean_sum[t] - min_t) / (max_t - min_t), e_drive);
            } else {
                p_drive[i + Z * t] = 0; /* zeros() initial value :4 */
            }
        }
    }
    free(mean_sum);
    return ORC_OK;
}

/* src/createpdestin.jl:3-50 */
int orc_createpdestin(const double *dm, int64_t Z, int64_t T, double e_dest, int e_is_integer,
                      double *p_dest)
{
    if (!dm || !p_dest || Z <= 0 || T <= 0) return ORC_ERR_BADARG;
    memset(p_dest, 0, sizeof(double) * (size_t)(Z * Z * T));
    /* :10-28 min/max over ALL T hours of the zero-filled array, then unnormalised weight */
    for (int64_t i = 0; i < Z; ++i) {
        for (int64_t j = 0; j < Z; ++j) {
            double mx = dm[i + Z * j], mn = dm[i + Z * j];
            for (int64_t t = 1; t < T; ++t) {
                double v = dm[i + Z * (j + Z * t)];
                mx = jl_max(mx, v);
                mn = jl_min(mn, v);
            }
            if (mx > 0) { /* :22 */
                for (int64_t t = 0; t < T; ++t) {
                    double mean = dm[i + Z * (j + Z * t)];
                    p_dest[i + Z * (j + Z * t)] = jl_pow((mean - mn) / (mx - mn), e_dest, e_is_integer);
                }
            }
        }
    }
    /* :31-46 normalise per (i,t); sum over j left to right (Julia's sum() is pairwise for
     * n >= 16 blocks of 1024; here plain sequential -- the divisor may differ from Julia's
     * in the last bits, one more reason parity with the Julia program is statistical). */
    for (int64_t i = 0; i < Z; ++i) {
        for (int64_t t = 0; t < T; ++t) {
            double nf = 0;
            for (int64_t j = 0; j < Z; ++j) nf = nf + p_dest[i + Z * (j + Z * t)];
            if (nf > 0) {
                for (int64_t j = 0; j < Z; ++j)
                    p_dest[i + Z * (j + Z * t)] = p_dest[i + Z * (j + Z * t)] / nf;
            }
        }
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------ */
/* Sampler, faithful form                                                     */
/* ------------------------------------------------------------------------ */

/* src/initializestates.jl:4-22 */
int orc_initializestates(int64_t C, int64_t cpz, int64_t T, int64_t car_offset, int64_t *state,
                         double *trans)
{
    if (C < 0 || cpz <= 0 || T <= 0 || !state) return ORC_ERR_BADARG;
    memset(state, 0, sizeof(int64_t) * (size_t)(C * T));
    if (trans) memset(trans, 0, sizeof(double) * (size_t)(C * T * 4));
    /* :11-16 car (1-based, global) g sits in zone ceil(g / cpz) */
    for (int64_t i = 0; i < C; ++i) state[i] = (car_offset + i) / cpz + 1;
    return ORC_OK;
}

/* One categorical draw exactly as src/resampling.jl:29-47 walks it, plus deviation D1
 *  where the reference would leave destination = 0 and crash (A-7).
 * distribution = copy of p_dest[origin,:,t] (a stride-Z gather, :34). */
static int64_t categorical_faithful(const double *p_dest, int64_t Z, int64_t origin1, int64_t t,
                                    double u, double *distribution)
{
    const double *row = p_dest + (origin1 - 1) + Z * Z * t;
    double sum = 0;
    for (int64_t j = 0; j < Z; ++j) { /* :34 row copy, :35 sum */
        distribution[j] = row[Z * j];
        sum += distribution[j];
    }
    if (sum == 0) return origin1; /* :35-36 */
    double range_up = 0, range_low = 0;
    int64_t destination = 0;
    for (int64_t j = 0; j < Z; ++j) { /* :38-45 */
        range_up = range_up + distribution[j];
        if (range_low < u && u <= range_up) {
            destination = j + 1;
            break;
        }
        range_low = range_up;
    }
    if (destination == 0) { /* D1 */
        if (u == 0.0) {
            for (int64_t j = 0; j < Z; ++j)
                if (distribution[j] > 0) { destination = j + 1; break; }
        } else {
            for (int64_t j = Z - 1; j >= 0; --j)
                if (distribution[j] > 0) { destination = j + 1; break; }
        }
    }
    return destination; /* 0 only for a row with NaN and no positive entry */
}

/* ------------------------------------------------------------------------ */
/* Deterministic f64 kit of the truncated-normal sampler: +,-,*,/ and bit      */
/* manipulation only (no libm), so that the GPU twin compiled with contraction */
/* off reproduces every value bit for bit (csrc/cpm_rng.h holds the same       */
/* statements in the same order).                                              */
/* ------------------------------------------------------------------------ */
static double bits_to_double(uint64_t b)
{
    double d;
    memcpy(&d, &b, 8);
    return d;
}
static uint64_t double_to_bits(double d)
{
    uint64_t b;
    memcpy(&b, &d, 8);
    return b;
}

/* ln(x), x > 0 normal: x = m * 2^e with m in [sqrt(1/2), sqrt(2)), ln m = 2 atanh((m-1)/(m+1)) as a series in s^2 (|s| < 0.1716) */
double orc_log(double x)
{
    uint64_t b = double_to_bits(x);
    int e = (int)((b >> 52) & 0x7FF) - 1023;
    double m = bits_to_double((b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull); /* [1, 2) */
    if (m > 1.4142135623730951) {
        m = m * 0.5;
        e += 1;
    }
    double s = (m - 1.0) / (m + 1.0);
    double s2 = s * s;
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
    double lm = (2.0 * s) * p;
    double ef = (double)e;
    return ef * 6.93147180369123816490e-01 + (lm + ef * 1.90821492927058770002e-10);
}

/* sqrt(x), x > 0 normal: Newton from a bit-level first guess (five steps: to an ulp or two; what matters is that both sides take them) */
double orc_sqrt(double x)
{
    double y = bits_to_double((double_to_bits(x) >> 1) + 0x1FF8000000000000ull);
    y = 0.5 * (y + x / y);
    y = 0.5 * (y + x / y);
    y = 0.5 * (y + x / y);
    y = 0.5 * (y + x / y);
    y = 0.5 * (y + x / y);
    return y;
}

/* erf(x), x >= 0: W. J. Cody's rational Chebyshev approximations (Math. Comp. 23 (1969) 631-637), exp(-x^2) by orc_exp_neg */
double orc_erf(double x)
{
    if (!(x >= 0.0)) return 0.0;
    if (x <= 0.46875) {
        double y = x * x;
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
        double erfc = orc_exp_neg(x * x) * ((num + 1.23033935479799725e03) / (den + 1.23033935480374942e03));
        return 1.0 - erfc;
    }
    if (x >= 6.0) return 1.0; /* erfc < 2.2e-17 */
    {
        double y = 1.0 / (x * x);
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
        double r = y * (num + 6.58749161529837803e-4) / (den + 2.33520497626869185e-3);
        double erfc = orc_exp_neg(x * x) * ((5.6418958354775628695e-1 - r) / x);
        return 1.0 - erfc;
    }
}

/* Phi^-1(1/2 + q), |q| <= 1/2: Wichura's PPND16 (Algorithm AS 241, Appl. Statist. 37 (1988) 477-484), from q */
double orc_ppnd(double q)
{
    double aq = q < 0.0 ? -q : q;
    if (aq <= 0.425) {
        double r = 0.180625 - q * q;
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
    double r = 0.5 - aq; /* min(p, 1 - p) */
    double val;
    if (!(r > 0.0)) {
        val = 9.0; /* beyond every window the sampler clamps to */
    } else {
        r = orc_sqrt(-orc_log(r));
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

/* E = erf(a / sqrt 2), a = 0.1 mu / sigma: the probability mass of N(mu, sigma) inside the window [0.9 mu, 1.1 mu] */
double orc_truncnormal_mass(double mu, double sigma)
{
    double a = (0.1 * mu) / sigma;
    return orc_erf(a * 7.0710678118654752440e-1);
}

/* Truncated normal on [0.9 mu, 1.1 mu] (src/resampling.jl:68,74).  The reference calls
 * Distributions.jl (version unpinned, absent) -> parity with Julia is distributional only.
 * This restatement, the build's own pinned sampler: ONE draw by inversion.  With E the mass of
 * N(mu, sigma) inside the window, z = Phi^-1(1/2 + (u - 1/2) E) is a standard normal truncated
 * to +-0.1 mu / sigma, and x = mu + sigma z; u = the first uniform of Philox stream `stream`.
 * (Rounds 1-3 pinned a rejection sampler -- uniform proposal, accept with exp(-(x-mu)^2/(2 sigma^2)):
 * its retry loop and f64 exp were what bound the GPU's travel kernel.) */
double orc_truncnormal_draw(double u, double mu, double sigma, double E)
{
    if (!(sigma > 0.0) || !(E > 0.0)) return mu;
    double z = orc_ppnd((u - 0.5) * E);
    double x = mu + sigma * z;
    double lo = 0.9 * mu, hi = 1.1 * mu;
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return x;
}

static double truncnormal_pm10(uint64_t seed, uint64_t car, uint32_t step, uint32_t stream,
                               double mu, double sigma)
{
    double u1, u2;
    orc_uniforms(seed, car, step, stream, &u1, &u2);
    return orc_truncnormal_draw(u1, mu, sigma, orc_truncnormal_mass(mu, sigma));
}

/* passes 1+2 of one hour: src/resampling.jl:11-49 == src/solveinitialvalueproblem.jl:11-49 */
static int hour_passes_1_2(int64_t *state, double *trans, const double *p_drive,
                           const double *p_dest, int64_t C, int64_t Z, int64_t T, int64_t t,
                           uint32_t step, uint64_t seed, int64_t car_offset, double *distribution)
{
    /* pass 1 (:11-22) */
    for (int64_t i = 0; i < C; ++i) {
        int64_t origin = state[i + C * t];
        if (origin < 1 || origin > Z) return ORC_ERR_FALLTHROUGH; /* Julia: BoundsError */
        double ub, uc;
        orc_uniforms(seed, (uint64_t)(car_offset + i), step, 0, &ub, &uc);
        double driving_probability = p_drive[(origin - 1) + Z * t];
        double drive;
        if (ub <= driving_probability) {
            drive = 1;
        } else {
            drive = 0;
            trans[i + C * (t + T * 1)] = (double)origin;
        }
        trans[i + C * (t + T * 0)] = drive;
    }
    /* pass 2 (:26-49) */
    for (int64_t i = 0; i < C; ++i) {
        if (trans[i + C * (t + T * 0)] == 1) {
            double ub, uc;
            orc_uniforms(seed, (uint64_t)(car_offset + i), step, 0, &ub, &uc);
            int64_t origin = state[i + C * t];
            int64_t destination = categorical_faithful(p_dest, Z, origin, t, uc, distribution);
            if (destination == 0) return ORC_ERR_FALLTHROUGH;
            trans[i + C * (t + T * 1)] = (double)destination;
        }
    }
    return ORC_OK;
}

/* src/solveinitialvalueproblem.jl:4-62 */
int orc_solveinitialvalueproblem(int64_t *state, double *trans, const double *p_drive,
                                 const double *p_dest, int64_t C, int64_t Z, int64_t T,
                                 uint64_t seed, int64_t car_offset, int64_t *initial_state)
{
    if (!state || !trans || !p_drive || !p_dest || !initial_state) return ORC_ERR_BADARG;
    double *distribution = (double *)malloc(sizeof(double) * (size_t)Z);
    int rc = ORC_OK;
    for (int64_t t = 0; t < T - 1 && rc == ORC_OK; ++t) { /* :8 t = 1:(T-1) */
        rc = hour_passes_1_2(state, trans, p_drive, p_dest, C, Z, T, t, (uint32_t)t, seed,
                             car_offset, distribution);
        if (rc != ORC_OK) break;
        for (int64_t i = 0; i < C; ++i) /* :53 unconditional state update */
            state[i + C * (t + 1)] = (int64_t)llround(trans[i + C * (t + T * 1)]);
    }
    free(distribution);
    if (rc != ORC_OK) return rc;
    for (int64_t i = 0; i < C; ++i) initial_state[i] = state[i + C * (T - 1)]; /* :57-58 */
    return ORC_OK;
}

/* src/resampling.jl:3-89 */
int orc_resampling(int64_t *state, double *trans, int64_t C, int64_t Z, int64_t T,
                   const double *p_drive, const double *p_dest, const double *dm,
                   const double *dist, uint64_t seed, int64_t car_offset)
{
    if (!state || !trans || !p_drive || !p_dest) return ORC_ERR_BADARG;
    double *distribution = (double *)malloc(sizeof(double) * (size_t)Z);
    int rc = ORC_OK;
    for (int64_t t = 0; t < T && rc == ORC_OK; ++t) { /* :7 */
        uint32_t step = (uint32_t)(T - 1 + t);
        rc = hour_passes_1_2(state, trans, p_drive, p_dest, C, Z, T, t, step, seed, car_offset,
                             distribution);
        if (rc != ORC_OK) break;
        /* pass 3 (:53-78) */
        if (dm && dist) {
            for (int64_t i = 0; i < C; ++i) {
                if (trans[i + C * (t + T * 0)] == 1) {
                    int64_t origin = state[i + C * t];
                    int64_t destination = (int64_t)llround(trans[i + C * (t + T * 1)]); /* :57 */
                    if (origin == destination) { /* :58-60 */
                        trans[i + C * (t + T * 2)] = 5 * 60;
                        trans[i + C * (t + T * 3)] = 1;
                    } else {
                        uint64_t car = (uint64_t)(car_offset + i);
                        double mean = dm[(origin - 1) + Z * ((destination - 1) + Z * (t + T * 0))];
                        double sd = dm[(origin - 1) + Z * ((destination - 1) + Z * (t + T * 1))];
                        if (sd == 0) sd = 0.1 * mean; /* :65-67 */
                        trans[i + C * (t + T * 2)] = truncnormal_pm10(seed, car, step, 1, mean, sd);
                        mean = dist[(origin - 1) + Z * (destination - 1)]; /* :72 */
                        sd = 0.1 * mean;
                        trans[i + C * (t + T * 3)] = truncnormal_pm10(seed, car, step, 2, mean, sd);
                    }
                }
            }
        }
        if (t < T - 1) /* :81-83 */
            for (int64_t i = 0; i < C; ++i)
                state[i + C * (t + 1)] = (int64_t)llround(trans[i + C * (t + T * 1)]);
    }
    free(distribution);
    return rc;
}

/* ------------------------------------------------------------------------ */
/* Reductions                                                                 */
/* ------------------------------------------------------------------------ */

/* src/saveresults.jl:6-20 */
int orc_histogram(int64_t Z, int64_t T, const int64_t *state, const double *trans, int64_t C,
                  double *parking, double *driving, double *density, double C_norm)
{
    if (!state || !trans || !parking || !driving) return ORC_ERR_BADARG;
    memset(parking, 0, sizeof(double) * (size_t)(Z * T));
    memset(driving, 0, sizeof(double) * (size_t)(Z * T));
    for (int64_t i = 0; i < C; ++i) {
        for (int64_t t = 0; t < T; ++t) {
            int64_t index = state[i + C * t];
            if (index < 1 || index > Z) return ORC_ERR_BADARG;
            double act = trans[i + C * (t + T * 0)];
            parking[(index - 1) + Z * t] += 1; /* :12 every car, drivers included (A-14) */
            if (act == 1) driving[(index - 1) + Z * t] += 1; /* :13-15 binned by origin */
        }
    }
    if (density)
        for (int64_t k = 0; k < Z * T; ++k) density[k] = parking[k] / C_norm; /* :20 */
    return ORC_OK;
}

/* src/saveresults.jl:23-28 */
int orc_trafficactivity(int64_t Z, int64_t T, const double *driving, double *activity)
{
    double mn = 0, mx = 0;
    for (int64_t t = 0; t < T; ++t) {
        double s = 0;
        for (int64_t z = 0; z < Z; ++z) s += driving[z + Z * t];
        activity[t] = s;
        if (t == 0) { mn = s; mx = s; }
        else { mn = jl_min(mn, s); mx = jl_max(mx, s); }
    }
    for (int64_t t = 0; t < T; ++t) activity[t] = (activity[t] - mn) / (mx - mn); /* NaN if flat */
    return ORC_OK;
}

/* src/averagedrivingtime.jl:3-12 */
double orc_averagedrivingtime(int64_t C, int64_t T, double A_drive, const double *trans)
{
    double sum_driving_time = 0;
    for (int64_t t = 0; t < T; ++t) {
        double s = 0;
        for (int64_t i = 0; i < C; ++i) s += trans[i + C * (t + T * 2)];
        sum_driving_time = sum_driving_time + s;
    }
    return A_drive + sum_driving_time / ((double)C * (double)T * 60 * 60);
}

static inline int64_t q16(double seconds) { return (int64_t)llrint(seconds * 65536.0); }

int64_t orc_sum_travel_time_q16(int64_t C, int64_t T, const double *trans)
{
    int64_t s = 0;
    for (int64_t t = 0; t < T; ++t)
        for (int64_t i = 0; i < C; ++i) s += q16(trans[i + C * (t + T * 2)]);
    return s;
}

/* src/correctparameters.jl:3-22 (note the else-nesting, A-11) */
void orc_correctparameters(double p_min_next, double p_max_next, double p_min, double p_max,
                           double *p_min_out, double *p_max_out)
{
    if (p_min_next < 0) {
        p_min_next = 0;
    } else {
        if (p_min_next > p_max) p_min_next = p_max;
    }
    if (p_max_next > 1) {
        p_max_next = 1;
    } else {
        if (p_max_next < p_min) p_max_next = p_min;
    }
    *p_min_out = p_min_next;
    *p_max_out = p_max_next;
}

/* ------------------------------------------------------------------------ */
/* Fast twin                                                                  */
/* ------------------------------------------------------------------------ */

int orc_build_cdf(const double *p_dest, int64_t Z, int64_t T, double *cdf)
{
    if (!p_dest || !cdf) return ORC_ERR_BADARG;
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t t = 0; t < T; ++t) {
        for (int64_t o = 0; o < Z; ++o) {
            double range_up = 0;
            double *out = cdf + (t * Z + o) * Z;
            const double *row = p_dest + o + Z * Z * t;
            for (int64_t d = 0; d < Z; ++d) {
                range_up = range_up + row[Z * d]; /* src/resampling.jl:39 */
                out[d] = range_up;
            }
        }
    }
    return ORC_OK;
}

/* first j with u_eff <= cdf[j]; u_eff folds deviation D1 into the search bounds */
static inline int64_t categorical_cdf(const double *row, int64_t Z, int64_t origin1, double u)
{
    double last = row[Z - 1];
    if (last == 0) return origin1;            /* zero row (src/resampling.jl:35-36) */
    double ue = u;
    if (ue == 0.0) ue = 4.9406564584124654e-324; /* D1: first zone with p > 0 */
    if (ue > last) ue = last;                     /* D1: last zone with p > 0 */
    int64_t lo = 0, n = Z;
    while (n > 0) { /* lower_bound */
        int64_t half = n >> 1;
        if (row[lo + half] < ue) { lo = lo + half + 1; n -= half + 1; }
        else n = half;
    }
    if (lo >= Z) return 0; /* NaN row */
    return lo + 1;
}

int orc_fast_run(const double *p_drive, const double *cdf, int64_t Z, int64_t T, int64_t C,
                 int64_t car_offset, int64_t car_stride, uint64_t seed, int do_ivp, int64_t *zone0,
                 int64_t *parking, int64_t *driving, int64_t *state_out, const double *dm,
                 const double *dist, int64_t *sum_tt_q16, int nthreads)
{
    if (!p_drive || !cdf || !zone0 || !parking || !driving) return ORC_ERR_BADARG;
    memset(parking, 0, sizeof(int64_t) * (size_t)(Z * T));
    memset(driving, 0, sizeof(int64_t) * (size_t)(Z * T));
    int err = 0;
    int64_t tt_total = 0;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
#pragma omp parallel num_threads(nthreads) reduction(+ : tt_total)
    {
        int64_t *lp = (int64_t *)calloc((size_t)(2 * Z * T), sizeof(int64_t));
        int64_t *ld = lp + Z * T;
#pragma omp for schedule(static)
        for (int64_t i = 0; i < C; ++i) {
            uint64_t car = (uint64_t)(car_offset + i * car_stride);  /* global car id: contiguous shard (stride 1) or interleaved deal */
            int64_t zone = zone0[i];
            if (zone < 1 || zone > Z) { err = 1; continue; }
            int bad = 0;
            if (do_ivp) {
                for (int64_t t = 0; t < T - 1 && !bad; ++t) {
                    double ub, uc;
                    orc_uniforms(seed, car, (uint32_t)t, 0, &ub, &uc);
                    if (ub <= p_drive[(zone - 1) + Z * t]) {
                        int64_t d = categorical_cdf(cdf + (t * Z + (zone - 1)) * Z, Z, zone, uc);
                        if (d == 0) bad = 1; else zone = d;
                    }
                }
                zone0[i] = zone;
            }
            for (int64_t t = 0; t < T && !bad; ++t) {
                uint32_t step = (uint32_t)(T - 1 + t);
                double ub, uc;
                orc_uniforms(seed, car, step, 0, &ub, &uc);
                lp[(zone - 1) + Z * t] += 1;
                if (state_out) state_out[i + C * t] = zone;
                if (ub <= p_drive[(zone - 1) + Z * t]) {
                    ld[(zone - 1) + Z * t] += 1;
                    int64_t d = categorical_cdf(cdf + (t * Z + (zone - 1)) * Z, Z, zone, uc);
                    if (d == 0) { bad = 1; break; }
                    if (dm && dist) {
                        double tt;
                        if (d == zone) tt = 5 * 60;
                        else {
                            double mean = dm[(zone - 1) + Z * ((d - 1) + Z * (t + T * 0))];
                            double sd = dm[(zone - 1) + Z * ((d - 1) + Z * (t + T * 1))];
                            if (sd == 0) sd = 0.1 * mean;
                            tt = truncnormal_pm10(seed, car, step, 1, mean, sd);
                        }
                        tt_total += q16(tt);
                    }
                    if (t < T - 1) zone = d;
                }
            }
            if (bad) err = 1;
        }
#pragma omp critical
        {
            for (int64_t k = 0; k < Z * T; ++k) { parking[k] += lp[k]; driving[k] += ld[k]; }
        }
        free(lp);
    }
    if (sum_tt_q16) *sum_tt_q16 = tt_total;
    return err ? ORC_ERR_FALLTHROUGH : ORC_OK;
}

/* ------------------------------------------------------------------------ */
/* Procedural synthetic inputs.  The HIP library carries its   */
/* own generators for the bench; tests check the two produce identical bits.  */
/* ------------------------------------------------------------------------ */

int orc_synth_p_drive(int64_t Z, int64_t T, uint64_t table_seed, double *p_drive)
{
    for (int64_t t = 0; t < T; ++t)
        for (int64_t z = 0; z < Z; ++z) {
            double u, v;
            table_uniforms(table_seed, (uint32_t)z, (uint32_t)t, 0, TS_PDRIVE, &u, &v);
            p_drive[z + Z * t] = 0.1 + 0.8 * u; /* range of main.jl:39-40 */
        }
    return ORC_OK;
}

int orc_synth_p_dest_dense(int64_t Z, int64_t T, uint64_t table_seed, double *p_dest)
{
    return orc_synth_p_dest_skewed(Z, T, table_seed, 0, p_dest);
}

/* skew_q > 0: destination popularity 1 / (skew_q + rank(d)), rank(d) = (7919 d + 13) mod Z (Zipf-Mandelbrot) */
int orc_synth_p_dest_skewed(int64_t Z, int64_t T, uint64_t table_seed, int64_t skew_q, double *p_dest)
{
    if (skew_q < 0 || (skew_q > 0 && Z % 7919 == 0)) return ORC_ERR_BADARG;
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t t = 0; t < T; ++t)
        for (int64_t o = 0; o < Z; ++o) {
            double nf = 0;
            for (int64_t d = 0; d < Z; ++d) {
                double u, v;
                table_uniforms(table_seed, (uint32_t)o, (uint32_t)d, (uint32_t)t, TS_PDEST, &u, &v);
                double w = (o == d) ? 0.0 : u * u; /* ((m-min)/(max-min))^2, createpdestin.jl:24 */
                if (skew_q > 0 && o != d) w = w / (double)(skew_q + (d * 7919 + 13) % Z);
                p_dest[o + Z * (d + Z * t)] = w;
                nf = nf + w; /* createpdestin.jl:31-35 */
            }
            if (nf > 0)
                for (int64_t d = 0; d < Z; ++d)
                    p_dest[o + Z * (d + Z * t)] = p_dest[o + Z * (d + Z * t)] / nf;
        }
    return ORC_OK;
}

int orc_synth_datamatrix(int64_t Z, int64_t T, uint64_t table_seed, double density,
                         double *dm, double *dist)
{
    double *lat = (double *)malloc(sizeof(double) * (size_t)Z * 2), *lon = lat + Z;
    for (int64_t z = 0; z < Z; ++z) {
        double u, v;
        table_uniforms(table_seed, (uint32_t)z, 0, 0, TS_CENTROID, &u, &v);
        lat[z] = -38.5 + 1.5 * u;
        lon[z] = 144.0 + 2.0 * v;
    }
    for (int64_t i = 0; i < Z; ++i)
        for (int64_t j = 0; j < Z; ++j) {
            double dlon = 0.79 * (lon[i] - lon[j]), dlat = lat[i] - lat[j];
            /* equirectangular, as src/processgeodata.jl:157 with cos(lat) frozen at 0.79 */
            dist[i + Z * j] = (i == j) ? 1.0 : 111.3 * sqrt(dlon * dlon + dlat * dlat);
            if (dist[i + Z * j] == 0) dist[i + Z * j] = 1.0;
        }
    free(lat);
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t t = 0; t < T; ++t)
        for (int64_t d = 0; d < Z; ++d)
            for (int64_t o = 0; o < Z; ++o) {
                double ua, ub, uc, ud;
                table_uniforms(table_seed, (uint32_t)o, (uint32_t)d, (uint32_t)t, TS_DATA_A, &ua, &ub);
                table_uniforms(table_seed, (uint32_t)o, (uint32_t)d, (uint32_t)t, TS_DATA_B, &uc, &ud);
                double mean = 0, sd = 0;
                if (o != d && ua < density) {
                    mean = 300.0 + 2100.0 * ub;
                    sd = mean * (0.1 + 0.3 * uc);
                }
                dm[o + Z * (d + Z * (t + T * 0))] = mean;
                dm[o + Z * (d + Z * (t + T * 1))] = sd;
            }
    return ORC_OK;
}

/* =======================================================================================
 * Data formats feeding the tables
 * ======================================================================================= */

/* src/createdatamatrix.jl:3-27 (after `rawdata = convert(Matrix, rawdata[:,1:5])`, :5) */
int orc_createdatamatrix(const double *rawdata, int64_t n, int64_t Z, int64_t T, double *datamatrix)
{
    if (!datamatrix || Z < 1 || T < 1 || n < 0 || (n > 0 && !rawdata)) return ORC_ERR_BADARG;
    memset(datamatrix, 0, sizeof(double) * (size_t)Z * (size_t)Z * (size_t)T * 2); /* :7 */
    for (int64_t i = 0; i < n; ++i) {                                               /* :8 */
        double r1 = rawdata[i], r2 = rawdata[n + i], r3 = rawdata[2 * n + i];
        if (r1 == 0) r1 = (double)Z; /* :9-11 */
        if (r2 == 0) r2 = (double)Z; /* :12-14 */
        if (r3 == 0) r3 = 24;        /* :15-17 */
        /* convert(Int64, x) (:18-20): InexactError unless integral; then BoundsError unless in range */
        if (r1 != floor(r1) || r2 != floor(r2) || r3 != floor(r3)) return ORC_ERR_BADARG;
        if (r1 < 1 || r1 > (double)Z || r2 < 1 || r2 > (double)Z || r3 < 1 || r3 > (double)T) return ORC_ERR_BADARG;
        const size_t i1 = (size_t)r1 - 1, i2 = (size_t)r2 - 1, i3 = (size_t)r3 - 1;
        const size_t cell = i1 + (size_t)Z * (i2 + (size_t)Z * i3);
        datamatrix[cell] = rawdata[3 * n + i];                                   /* :21 */
        datamatrix[cell + (size_t)Z * (size_t)Z * (size_t)T] = rawdata[4 * n + i]; /* :22 */
    }
    return ORC_OK;
}

/* src/processgeodata.jl:99-146.  Index j below is the reference's 1-based j; M(i,j) = matrix[i, j]. */
int orc_centroids(double *lon, double *lat, int64_t Z, int64_t width, int64_t scan,
                  double *centroid_lat, double *centroid_long, double *area)
{
    if (!lon || !lat || !centroid_lat || !centroid_long || !area || Z < 1 || scan + 1 > width) return ORC_ERR_BADARG;
#define LON(i, j) lon[(size_t)(i) * (size_t)width + (size_t)((j)-1)]
#define LAT(i, j) lat[(size_t)(i) * (size_t)width + (size_t)((j)-1)]
    for (int64_t i = 0; i < Z; ++i) {
        centroid_lat[i] = 0;
        centroid_long[i] = 0;
        area[i] = 0;
    }
    /* area of each polygon (:104-124) */
    for (int64_t i = 0; i < Z; ++i) {
        if (LON(i, 1) != 0) {
            double summation_term = 0;
            for (int64_t j = 1; j <= scan; ++j) {
                if (LON(i, j) == 0) {
                    if (j < 2) return ORC_ERR_BADARG; /* cannot happen: LON(i,1) != 0 */
                    /* correction of summation term (:108) */
                    summation_term = summation_term - (LAT(i, j - 1) * LON(i, j) - LAT(i, j) * LON(i, j - 1));
                    /* addition of first vertex as last one (:111-112) */
                    LON(i, j) = LON(i, 1);
                    LAT(i, j) = LAT(i, 1);
                    /* recalculation of last summation term (:115) */
                    summation_term = summation_term + (LAT(i, j - 1) * LON(i, j) - LAT(i, j) * LON(i, j - 1));
                    break;
                } else {
                    summation_term = summation_term + (LAT(i, j) * LON(i, j + 1) - LAT(i, j + 1) * LON(i, j)); /* :118 */
                }
            }
            area[i] = summation_term / 2; /* :121 */
        }
    }
    /* centroid coordinates (:127-146): no break in the zero branch */
    for (int64_t i = 0; i < Z; ++i) {
        if (LON(i, 1) != 0) {
            double summation_long = 0, summation_lat = 0;
            for (int64_t j = 1; j <= scan; ++j) {
                if (LON(i, j) == 0) {
                    if (j < 2) return ORC_ERR_BADARG;
                    summation_long = summation_long - (LON(i, j - 1) + LON(i, j)) * (LON(i, j - 1) * LAT(i, j) - LON(i, j) * LAT(i, j - 1));
                    summation_lat = summation_lat - (LAT(i, j - 1) + LAT(i, j)) * (LON(i, j - 1) * LAT(i, j) - LON(i, j) * LAT(i, j - 1));
                } else {
                    summation_long = summation_long + (LON(i, j) + LON(i, j + 1)) * (LON(i, j) * LAT(i, j + 1) - LON(i, j + 1) * LAT(i, j));
                    summation_lat = summation_lat + (LAT(i, j) + LAT(i, j + 1)) * (LON(i, j) * LAT(i, j + 1) - LON(i, j + 1) * LAT(i, j));
                }
            }
            centroid_long[i] = -summation_long / (6 * area[i]); /* :143 */
            centroid_lat[i] = -summation_lat / (6 * area[i]);   /* :144 */
        }
    }
#undef LON
#undef LAT
    return ORC_OK;
}

/* src/processgeodata.jl:148-166 */
int orc_distance_matrix(const double *centroid_lat, const double *centroid_long, int64_t Z, double *dist)
{
    if (!centroid_lat || !centroid_long || !dist || Z < 1) return ORC_ERR_BADARG;
    const double c_lat_long = 111.3;     /* :148 */
    const double conv_deg_rad = 0.01745; /* :149 */
    memset(dist, 0, sizeof(double) * (size_t)Z * (size_t)Z); /* :150 */
    for (int64_t i = 0; i < Z - 1; ++i) {                    /* :151 i = 1:(number_zones-1) */
        const double long1 = centroid_long[i], lat1 = centroid_lat[i];
        for (int64_t j = i; j < Z; ++j) {                    /* :154 j = i:number_zones */
            const double long2 = centroid_long[j], lat2 = centroid_lat[j];
            const double c = cos((lat1 + lat2) / 2 * conv_deg_rad);
            const double distance_km = c_lat_long * sqrt((c * c) * ((long1 - long2) * (long1 - long2)) + (lat1 - lat2) * (lat1 - lat2)); /* :157 */
            dist[i + (size_t)Z * j] = distance_km; /* :158 */
            dist[j + (size_t)Z * i] = distance_km; /* :159 */
        }
    }
    for (int64_t i = 0; i < Z; ++i) dist[i + (size_t)Z * i] = 1; /* :164-166 */
    return ORC_OK;
}
