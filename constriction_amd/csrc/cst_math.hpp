// cst_math.hpp -- bit-exact f64 special functions and the LeakyQuantizer arithmetic, for gfx950.
//
// The reference computes the quantized-Gaussian cumulative in f64 through third-party crates
// (probability 0.20.3 -> special 0.10.3 -> libm 0.2.16, see Cargo.lock); the call sites are
// src/stream/model/quantize.rs:546,558.  libm's `erf`/`exp` are the msun algorithms as arranged by
// musl.  This header evaluates the same expression trees on the GPU.  It MUST be compiled with
// -ffp-contract=off: every operation below has to round exactly once, in the order written
// (v_fma_f64 fusion would change low bits and with them floor(free_weight * cdf)).
// f64 add/mul/div on gfx950 are IEEE correctly rounded and keep subnormals, so the results are
// identical to a scalar CPU evaluation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cst {

__device__ __forceinline__ uint32_t f64_hi(double x) { return (uint32_t)__double2hiint(x); }
__device__ __forceinline__ uint32_t f64_lo(double x) { return (uint32_t)__double2loint(x); }
__device__ __forceinline__ double f64_from(uint32_t lo, uint32_t hi) { return __hiloint2double((int)hi, (int)lo); }
__device__ __forceinline__ double f64_clear_lo(double x) { return __hiloint2double(__double2hiint(x), 0); }
__device__ __forceinline__ double f64_pow2(int n) { return __hiloint2double((0x3ff + n) << 20, 0); }

// scalbn as musl writes it; only |n| < 2100 can occur.
__device__ inline double scalbn_exact(double x, int n) {
    double y = x;
    if (n > 1023) {
        y *= 0x1p1023; n -= 1023;
        if (n > 1023) { y *= 0x1p1023; n -= 1023; if (n > 1023) n = 1023; }
    } else if (n < -1022) {
        y *= 0x1p-1022 * 0x1p53; n += 1022 - 53;
        if (n < -1022) { y *= 0x1p-1022 * 0x1p53; n += 1022 - 53; if (n < -1022) n = -1022; }
    }
    return y * f64_pow2(n);
}

__device__ inline double exp_exact(double x) {
    constexpr double ln2hi = 6.93147180369123816490e-01, ln2lo = 1.90821492927058770002e-10,
                     invln2 = 1.44269504088896338700e+00, P1 = 1.66666666666666019037e-01,
                     P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                     P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    uint32_t hx = f64_hi(x);
    const int sign = (int)(hx >> 31);
    hx &= 0x7fffffffu;
    if (hx >= 0x4086232bu) {
        if (x != x) return x;
        if (x > 709.782712893383973096) return x * 0x1p1023;
        if (x < -745.13321910194110842) return 0.0;
    }
    double hi, lo;
    int k;
    if (hx > 0x3fd62e42u) {
        if (hx >= 0x3ff0a2b2u) k = (int)(invln2 * x + (sign ? -0.5 : 0.5));
        else k = 1 - sign - sign;
        hi = x - (double)k * ln2hi;
        lo = (double)k * ln2lo;
        x = hi - lo;
    } else if (hx > 0x3e300000u) {
        k = 0; hi = x; lo = 0.0;
    } else {
        return 1.0 + x;
    }
    const double xx = x * x;
    const double c = x - xx * (P1 + xx * (P2 + xx * (P3 + xx * (P4 + xx * P5))));
    const double y = 1.0 + (x * c / (2.0 - c) - lo + hi);
    return k == 0 ? y : scalbn_exact(y, k);
}

__device__ inline double erf_exact(double x) {
    constexpr double erx = 8.45062911510467529297e-01, efx8 = 1.02703333676410069053e+00,
        pp0 = 1.28379167095512558561e-01, pp1 = -3.25042107247001499370e-01, pp2 = -2.84817495755985104766e-02,
        pp3 = -5.77027029648944159157e-03, pp4 = -2.37630166566501626084e-05,
        qq1 = 3.97917223959155352819e-01, qq2 = 6.50222499887672944485e-02, qq3 = 5.08130628187576562776e-03,
        qq4 = 1.32494738004321644526e-04, qq5 = -3.96022827877536812320e-06,
        pa0 = -2.36211856075265944077e-03, pa1 = 4.14856118683748331666e-01, pa2 = -3.72207876035701323847e-01,
        pa3 = 3.18346619901161753674e-01, pa4 = -1.10894694282396677476e-01, pa5 = 3.54783043256182359371e-02,
        pa6 = -2.16637559486879084300e-03,
        qa1 = 1.06420880400844228286e-01, qa2 = 5.40397917702171048937e-01, qa3 = 7.18286544141962662868e-02,
        qa4 = 1.26171219808761642112e-01, qa5 = 1.36370839120290507362e-02, qa6 = 1.19844998467991074170e-02,
        ra0 = -9.86494403484714822705e-03, ra1 = -6.93858572707181764372e-01, ra2 = -1.05586262253232909814e+01,
        ra3 = -6.23753324503260060396e+01, ra4 = -1.62396669462573470355e+02, ra5 = -1.84605092906711035994e+02,
        ra6 = -8.12874355063065934246e+01, ra7 = -9.81432934416914548592e+00,
        sa1 = 1.96512716674392571292e+01, sa2 = 1.37657754143519042600e+02, sa3 = 4.34565877475229228821e+02,
        sa4 = 6.45387271733267880336e+02, sa5 = 4.29008140027567833386e+02, sa6 = 1.08635005541779435134e+02,
        sa7 = 6.57024977031928170135e+00, sa8 = -6.04244152148580987438e-02,
        rb0 = -9.86494292470009928597e-03, rb1 = -7.99283237680523006574e-01, rb2 = -1.77579549177547519889e+01,
        rb3 = -1.60636384855821916062e+02, rb4 = -6.37566443368389627722e+02, rb5 = -1.02509513161107724954e+03,
        rb6 = -4.83519191608651397019e+02,
        sb1 = 3.03380607434824582924e+01, sb2 = 3.25792512996573918826e+02, sb3 = 1.53672958608443695994e+03,
        sb4 = 3.19985821950859553908e+03, sb5 = 2.55305040643316442583e+03, sb6 = 4.74528541206955367215e+02,
        sb7 = -2.24409524465858183362e+01;

    uint32_t ix = f64_hi(x);
    const int sign = (int)(ix >> 31);
    ix &= 0x7fffffffu;
    if (ix >= 0x7ff00000u) return (double)(1 - 2 * sign) + 1.0 / x;
    if (ix < 0x3feb0000u) { // |x| < 0.84375
        if (ix < 0x3e300000u) return 0.125 * (8.0 * x + efx8 * x);
        const double z = x * x;
        const double r = pp0 + z * (pp1 + z * (pp2 + z * (pp3 + z * pp4)));
        const double s = 1.0 + z * (qq1 + z * (qq2 + z * (qq3 + z * (qq4 + z * qq5))));
        const double y = r / s;
        return x + x * y;
    }
    double y;
    if (ix < 0x40180000u) { // 0.84375 <= |x| < 6: y = 1 - erfc(|x|)
        const double ax = fabs(x);
        double erfc_val;
        if (ix < 0x3ff40000u) { // |x| < 1.25
            const double s = ax - 1.0;
            const double P = pa0 + s * (pa1 + s * (pa2 + s * (pa3 + s * (pa4 + s * (pa5 + s * pa6)))));
            const double Q = 1.0 + s * (qa1 + s * (qa2 + s * (qa3 + s * (qa4 + s * (qa5 + s * qa6)))));
            erfc_val = 1.0 - erx - P / Q;
        } else {
            const double s = 1.0 / (ax * ax);
            double R, Sv;
            if (ix < 0x4006db6du) { // |x| < 1/.35
                R = ra0 + s * (ra1 + s * (ra2 + s * (ra3 + s * (ra4 + s * (ra5 + s * (ra6 + s * ra7))))));
                Sv = 1.0 + s * (sa1 + s * (sa2 + s * (sa3 + s * (sa4 + s * (sa5 + s * (sa6 + s * (sa7 + s * sa8)))))));
            } else {
                R = rb0 + s * (rb1 + s * (rb2 + s * (rb3 + s * (rb4 + s * (rb5 + s * rb6)))));
                Sv = 1.0 + s * (sb1 + s * (sb2 + s * (sb3 + s * (sb4 + s * (sb5 + s * (sb6 + s * sb7))))));
            }
            const double z = f64_clear_lo(ax);
            erfc_val = exp_exact(-z * z - 0.5625) * exp_exact((z - ax) * (z + ax) + R / Sv) / ax;
        }
        y = 1.0 - erfc_val;
    } else {
        y = 1.0 - 0x1p-1022;
    }
    return sign ? -y : y;
}

// ------------------------------------------------------------------------------------------------
// The same erf for kernels whose lanes sit in DIFFERENT branches of it (one model per lane): a wave pays for every
// branch some lane takes, and the four rational approximations above (98 multiply-adds, 120 f64 constants that do
// not fit the scalar registers) are the bulk.  Here every lane runs ONE Horner recurrence of degree 8 over its own
// branch's coefficients, read per lane from a 576-byte LDS table; shorter polynomials are padded with zero
// coefficients at the high end, which leaves every partial result unchanged (c + s * 0 == c exactly).  The two exp of
// the erfc branches run straight-line (their arguments are bounded there).  Same operations in the same order as
// erf_exact on every path, hence the same bits; tests/test_gpu_ans_batch.py compares both with the CPU oracle.

constexpr int kErfExactEntries = 4 * 9;           // [branch][k] -> {numerator[k], denominator[k]}
// ... followed by the polynomials of the FAST evaluation (erf_fast_poly below): kErfPolyRows rows of 8 coefficients, one row
// every 80 bytes (the four 16-byte reads of a lane then start in one of 16 bank groups instead of 4)
constexpr int kErfPolyRows = 96, kErfPolyRowD2 = 5;
constexpr int kErfTabEntries = kErfExactEntries + kErfPolyRows * kErfPolyRowD2;      // double2 entries of the LDS image (8256 bytes)
constexpr size_t kErfTabBytes = 9 * 1024;         // what a kernel with dynamic LDS sets aside for it

__device__ const double kErfPolyInit[kErfPolyRows * 8] = {
#include "cst_erf_poly.inc"
};

__device__ const double kErfTabInit[kErfExactEntries * 2] = {
    // |x| < 0.84375: pp / qq in z = x * x
    1.28379167095512558561e-01, 1.0, -3.25042107247001499370e-01, 3.97917223959155352819e-01,
    -2.84817495755985104766e-02, 6.50222499887672944485e-02, -5.77027029648944159157e-03, 5.08130628187576562776e-03,
    -2.37630166566501626084e-05, 1.32494738004321644526e-04, 0.0, -3.96022827877536812320e-06, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0,
    // 0.84375 <= |x| < 1.25: pa / qa in s = |x| - 1
    -2.36211856075265944077e-03, 1.0, 4.14856118683748331666e-01, 1.06420880400844228286e-01,
    -3.72207876035701323847e-01, 5.40397917702171048937e-01, 3.18346619901161753674e-01, 7.18286544141962662868e-02,
    -1.10894694282396677476e-01, 1.26171219808761642112e-01, 3.54783043256182359371e-02, 1.36370839120290507362e-02,
    -2.16637559486879084300e-03, 1.19844998467991074170e-02, 0.0, 0.0, 0.0, 0.0,
    // 1.25 <= |x| < 1/0.35: ra / sa in s = 1 / x^2
    -9.86494403484714822705e-03, 1.0, -6.93858572707181764372e-01, 1.96512716674392571292e+01,
    -1.05586262253232909814e+01, 1.37657754143519042600e+02, -6.23753324503260060396e+01, 4.34565877475229228821e+02,
    -1.62396669462573470355e+02, 6.45387271733267880336e+02, -1.84605092906711035994e+02, 4.29008140027567833386e+02,
    -8.12874355063065934246e+01, 1.08635005541779435134e+02, -9.81432934416914548592e+00, 6.57024977031928170135e+00,
    0.0, -6.04244152148580987438e-02,
    // 1/0.35 <= |x| < 6: rb / sb in s = 1 / x^2
    -9.86494292470009928597e-03, 1.0, -7.99283237680523006574e-01, 3.03380607434824582924e+01,
    -1.77579549177547519889e+01, 3.25792512996573918826e+02, -1.60636384855821916062e+02, 1.53672958608443695994e+03,
    -6.37566443368389627722e+02, 3.19985821950859553908e+03, -1.02509513161107724954e+03, 2.55305040643316442583e+03,
    -4.83519191608651397019e+02, 4.74528541206955367215e+02, 0.0, -2.24409524465858183362e+01, 0.0, 0.0,
};

// every thread of the block calls this once (then __syncthreads, or a wave fence if `tab` is private to the wave)
__device__ __forceinline__ void erf_tab_fill(double2* tab, int tid, int n_threads) {
    for (int i = tid; i < kErfExactEntries; i += n_threads) tab[i] = make_double2(kErfTabInit[2 * i], kErfTabInit[2 * i + 1]);
    double* rows = reinterpret_cast<double*>(tab + kErfExactEntries);
    for (int i = tid; i < kErfPolyRows * 8; i += n_threads) rows[(i >> 3) * (2 * kErfPolyRowD2) + (i & 7)] = kErfPolyInit[i];
}

// exp_exact for 2^-28 < |x| < 700, without branches: k == 0 multiplies by 2^0, lo == 0.0 subtracts nothing
__device__ __forceinline__ double exp_exact_bounded(double x) {
    constexpr double ln2hi = 6.93147180369123816490e-01, ln2lo = 1.90821492927058770002e-10,
                     invln2 = 1.44269504088896338700e+00, P1 = 1.66666666666666019037e-01,
                     P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                     P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    const uint32_t hx = f64_hi(x) & 0x7fffffffu;
    const bool neg = (f64_hi(x) >> 31) != 0;
    const bool reduce = hx > 0x3fd62e42u;
    const int kg = (int)(invln2 * x + (neg ? -0.5 : 0.5));
    const int k = !reduce ? 0 : hx >= 0x3ff0a2b2u ? kg : neg ? -1 : 1;
    const double kd = (double)k;
    const double hi = reduce ? x - kd * ln2hi : x;
    const double lo = reduce ? kd * ln2lo : 0.0;
    const double r = reduce ? hi - lo : x;
    const double xx = r * r;
    const double c = r - xx * (P1 + xx * (P2 + xx * (P3 + xx * (P4 + xx * P5))));
    const double y = 1.0 + (r * c / (2.0 - c) - lo + hi);
    const double scaled = y * f64_pow2(k);
    return hx > 0x3e300000u ? scaled : 1.0 + x;
}

__device__ inline double erf_exact_tab(double x, const double2* tab) {
    constexpr double erx = 8.45062911510467529297e-01, efx8 = 1.02703333676410069053e+00;
    const uint32_t hx = f64_hi(x), ix = hx & 0x7fffffffu;
    const bool neg = (hx >> 31) != 0;
    const double ax = fabs(x);
    const uint32_t branch = (ix >= 0x3feb0000u ? 1u : 0u) + (ix >= 0x3ff40000u ? 1u : 0u) + (ix >= 0x4006db6du ? 1u : 0u);
    const bool tail = ix >= 0x3ff40000u && ix < 0x40180000u;        // 1.25 <= |x| < 6: exp(-x^2 + R / S) / x
    // all nine LDS reads are requested before the division below, which covers their latency
    const double2* c = tab + branch * 9;
    double2 t[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) t[k] = c[k];
    double s = branch == 0 ? x * x : ax - 1.0;
    if (tail) s = 1.0 / (ax * ax);
    double num = t[8].x, den = t[8].y;
#pragma unroll
    for (int k = 7; k >= 0; --k) { num = t[k].x + s * num; den = t[k].y + s * den; }
    const double quot = num / den;
    double y = 1.0 - ((1.0 - erx) - quot);                          // 0.84375 <= |x| < 1.25
    if (tail) {
        const double z = f64_clear_lo(ax);
        y = 1.0 - exp_exact_bounded(-z * z - 0.5625) * exp_exact_bounded((z - ax) * (z + ax) + quot) / ax;
    }
    if (ix >= 0x40180000u) y = 1.0 - 0x1p-1022;                     // (and +-inf: (1 - 2 sign) + 1 / x)
    y = neg ? -y : y;
    if (ix < 0x3feb0000u) y = x + x * quot;
    if (ix < 0x3e300000u) y = 0.125 * (8.0 * x + efx8 * x);
    if (x != x) y = x;
    return y;
}

// ------------------------------------------------------------------------------------------------
// FAST erf with an exact fallback.  What the coders need from the Gaussian CDF is ONE integer, trunc(free_weight * cdf):
// the reference's value of it changes only when free_weight * cdf crosses an integer.  So the per-symbol kernels evaluate
// erf the cheap way first -- a polynomial of degree 7 per interval of width 1/16 (scripts/gen_erf_poly.py; 15 VALU
// instructions and four 16-byte LDS reads; within 2^-52 of erf, tests/test_models_cpu.py replays it on the CPU and
// tests/test_gpu_ans_batch.py::test_fast_erf_error_bound measures the device) -- and fall back to erf_exact only where
// free_weight * cdf lands within kLeftGuard of an integer, i.e. where the deviation could change the truncation (about two
// evaluations in a million).  The result is the reference's integer in every case; the cost drops from ~290 (exact) to ~30
// instructions per evaluation.  (Round 3's first fast path kept msun's rational approximations with fused operations,
// Newton reciprocals and one exp: ~100 instructions, and the per-symbol encoder was bound by them.)

constexpr double kErfFastBound = 0x1p-46;    // assumed (and tested) bound on |erf_fast_poly - erf_exact_tab|
constexpr double kLeftGuard = 0x1p-20;       // >= 2^24 * kErfFastBound / 2 + rounding of the two products, with room to spare
constexpr int kQuickMaxPrecision = 24;       // the guard is sound for free_weight <= 2^24 only: callers of the *_quick functions keep P <= 24
                                             // (check_model_args, config_supported, and the debug hook below enforce it)

// 1 / b to 2^-48 (v_rcp_f64 is good to 2^-24.4 on gfx950, scripts/microbench/rcp_f64_error.hip; one Newton step): enough wherever the consumer has slack of its own -- the argument of the fast erf (an
// argument off by 2^-48 moves erf by < 2^-49), quotients that are corrected by their exact remainder
__device__ __forceinline__ double fast_rcp1(double b) {
    const double r = __builtin_amdgcn_rcp(b);
    return __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
}

// Rust's `f64 as u32` in one instruction: v_cvt_u32_f64 truncates, saturates at both ends and turns a NaN into 0
__device__ __forceinline__ uint32_t f64_as_u32_hw(double v) {
    uint32_t r;
    asm("v_cvt_u32_f64 %0, %1" : "=v"(r) : "v"(v));
    return r;
}

// erf(x) to 2^-52 (absolute) for every finite x; |x| >= 6 evaluates the last polynomial at its upper end (1 - 2e-17).
// A NaN comes back as +-1: callers that care test the argument (leaky_left_value_quick does).
// In two halves, so that a caller with several arguments can have ALL their coefficient reads in flight before the first
// multiply-add (erf_poly_fetch for each, a scheduling barrier, erf_poly_eval for each): left to itself the compiler reads,
// waits and evaluates one argument after the other, and a lone wave then pays an LDS round trip per argument.
struct ErfPoly { double u; double2 c01, c23, c45, c67; };
__device__ __forceinline__ ErfPoly erf_poly_fetch(double x, const double2* tab) {
    const double s = __builtin_fmin(__builtin_fabs(x) * 16.0, 0x1.7ffffffffffffp+6);     // interval number + position in it, < 96
    const int idx = (int)s;
    const double2* row = tab + kErfExactEntries + idx * kErfPolyRowD2;
    return ErfPoly{__builtin_fma(__builtin_amdgcn_fract(s), 2.0, -1.0), row[0], row[1], row[2], row[3]};
}
__device__ __forceinline__ double erf_poly_eval(const ErfPoly& e, double x) {
    double y = __builtin_fma(e.c67.y, e.u, e.c67.x);
    y = __builtin_fma(y, e.u, e.c45.y);
    y = __builtin_fma(y, e.u, e.c45.x);
    y = __builtin_fma(y, e.u, e.c23.y);
    y = __builtin_fma(y, e.u, e.c23.x);
    y = __builtin_fma(y, e.u, e.c01.y);
    y = __builtin_fma(y, e.u, e.c01.x);
    return __builtin_copysign(y, x);
}
__device__ __forceinline__ double erf_fast_poly(double x, const double2* tab) { return erf_poly_eval(erf_poly_fetch(x, tab), x); }

// probability::distribution::Gaussian::distribution (third-party; used at quantize.rs:546,558)
// (TAB: erf_exact_tab with the LDS table `tab` instead of the branchy erf_exact -- same bits either way.  A template
// flag, not a null test: address 0 is a valid LDS address, so the compiler would keep both.)
template <bool TAB = false>
__device__ __forceinline__ double gaussian_cdf_exact(double x, double mu, double sigma, const double2* tab = nullptr) {
    constexpr double sqrt2 = 1.41421356237309504880168872420969808;
    const double arg = (x - mu) / (sigma * sqrt2);
    if constexpr (TAB) return (1.0 + erf_exact_tab(arg, tab)) / 2.0;
    else return (1.0 + erf_exact(arg)) / 2.0;
}

// Rust `f64 as u32`
__device__ __forceinline__ uint32_t f64_as_u32_sat(double v) {
    if (!(v > 0.0)) return 0u;
    if (v >= 4294967296.0) return 0xffffffffu;
    return (uint32_t)v;
}

// LeakilyQuantizedDistribution::left_cumulative_and_probability, quantize.rs:525-568, for
// Symbol=i32 and Probability=u{prob_bits}.  Returns false if sym is outside [lo, hi].
// `left` and `prob` are wrapped to prob_bits bits; prob == 0 signals an invalid distribution.
template <bool TAB = false>
__device__ __forceinline__ bool leaky_gaussian_lcp(int32_t sym, int32_t lo, int32_t hi, int P, int prob_bits, double mu,
                                                   double sigma, uint32_t& left, uint32_t& prob, const double2* tab = nullptr) {
    if (sym < lo || sym > hi) return false;
    const uint32_t pmask = prob_bits >= 32 ? 0xffffffffu : ((1u << prob_bits) - 1u);
    const uint32_t max_prob = pmask >> (prob_bits - P);
    const double free_weight = (double)(max_prob - ((uint32_t)hi - (uint32_t)lo)); // quantize.rs:297-301
    const uint32_t slack = ((uint32_t)sym - (uint32_t)lo) & pmask;               // quantize.rs:475-486
    uint32_t l, r;
    if (sym == lo) l = 0u;
    else l = (f64_as_u32_sat(free_weight * gaussian_cdf_exact<TAB>((double)sym - 0.5, mu, sigma, tab)) + slack) & pmask;
    if (sym == hi) r = (P >= 32 ? 0u : (1u << P)) & pmask;
    else r = (f64_as_u32_sat(free_weight * gaussian_cdf_exact<TAB>((double)sym + 0.5, mu, sigma, tab)) + slack + 1u) & pmask;
    left = l;
    prob = (r - l) & pmask;
    return true;
}

// left cumulative of symbol index i in [0, n]  (i == n gives 2^P); used to tabulate a model and for
// decode-side searches.  Bit-identical to the `left` of leaky_gaussian_lcp(lo + i).
template <bool TAB = false>
__device__ __forceinline__ uint32_t leaky_gaussian_left(int32_t i, int32_t lo, int32_t n, int P, int prob_bits, double mu,
                                                        double sigma, const double2* tab = nullptr) {
    const uint32_t pmask = prob_bits >= 32 ? 0xffffffffu : ((1u << prob_bits) - 1u);
    if (i <= 0) return 0u;
    if (i >= n) return (P >= 32 ? 0u : (1u << P)) & pmask;
    const uint32_t max_prob = pmask >> (prob_bits - P);
    const double free_weight = (double)(max_prob - (uint32_t)(n - 1));
    const double x = (double)(int32_t)((uint32_t)lo + (uint32_t)i) - 0.5;
    return (f64_as_u32_sat(free_weight * gaussian_cdf_exact<TAB>(x, mu, sigma, tab)) + (uint32_t)i) & pmask;
}

// The exact evaluation of one left cumulative's f64 value, out of line: the quick paths call it for about two evaluations in
// a million, and inlined into each of them it was most of their code
static __device__ __attribute__((noinline)) double leaky_left_f64_exact(double x, double mu, double sigma, double free_weight, const double2* tab) {
    constexpr double sqrt2 = 1.41421356237309504880168872420969808;
    return free_weight * ((1.0 + erf_exact_tab((x - mu) / (sigma * sqrt2), tab)) / 2.0);
}

// The integer trunc(free_weight * cdf(x)) + slack through the fast erf, with the exact evaluation wherever the truncation
// could depend on the difference (see erf_fast_poly).  `inv_d` ~ 1 / (sigma sqrt 2): the fast path's argument
// (x - mu) inv_d is within a few ulp of the reference's (x - mu) / (sigma sqrt 2) -- part of the same error budget; the
// fallback forms the reference's argument with the reference's three operations.  `n_exact` (optional) counts fallbacks.
__device__ __forceinline__ uint32_t leaky_left_value_quick(double x, double mu, double sigma, double inv_d, double free_weight,
                                                           const double2* tab, uint32_t* n_exact) {
    constexpr double sqrt2 = 1.41421356237309504880168872420969808;
    const double arg = (x - mu) * inv_d;
    const double half = 0.5 * free_weight;
    double y = __builtin_fma(erf_fast_poly(arg, tab), half, half);             // free_weight (1 + erf) / 2, one rounding
    const double fr = __builtin_amdgcn_fract(y);                               // (y >= 0)
    // |arg| >= 6: erf is +-1 in both evaluations (the exact one rounds to +-1 from 5.87 on), no doubt there although
    // free_weight * cdf is then an exact integer.  A NaN argument (sigma sqrt 2 overflowed or is subnormal: the reciprocal
    // broke down) is always in doubt: it fails both tests.
    const bool same = __builtin_fabs(arg) >= 6.0;
    const bool sure = __builtin_fabs(fr - 0.5) < 0.5 - kLeftGuard;
    const bool unsure = !(same || sure);
    if (__builtin_amdgcn_ballot_w64(unsure) != 0ull) {
        if (unsure) {
            y = leaky_left_f64_exact(x, mu, sigma, free_weight, tab);
            if (n_exact) ++*n_exact;
        }
    }
    return f64_as_u32_hw(y);
}

// leaky_gaussian_left, quick.  Bit-identical to leaky_gaussian_left<true> for every input.  INNER: the caller guarantees
// 0 < i < n (a decoder's probes), which spares two divergent early exits per evaluation.
template <bool INNER = false>
__device__ __forceinline__ uint32_t leaky_gaussian_left_quick(int32_t i, int32_t lo, int32_t n, int P, int prob_bits, double mu,
                                                              double sigma, const double2* tab, uint32_t* n_exact = nullptr) {
    constexpr double sqrt2 = 1.41421356237309504880168872420969808;
    const uint32_t pmask = prob_bits >= 32 ? 0xffffffffu : ((1u << prob_bits) - 1u);
    if constexpr (!INNER) {
        if (i <= 0) return 0u;
        if (i >= n) return (P >= 32 ? 0u : (1u << P)) & pmask;
    }
    const uint32_t max_prob = pmask >> (prob_bits - P);
    const double free_weight = (double)(max_prob - (uint32_t)(n - 1));
    const double x = (double)(int32_t)((uint32_t)lo + (uint32_t)i) - 0.5;
    return (leaky_left_value_quick(x, mu, sigma, fast_rcp1(sigma * sqrt2), free_weight, tab, n_exact) + (uint32_t)i) & pmask;
}

// Three consecutive left cumulatives at once, indices g - 1, g, g + 1 with 1 <= g <= n - 1 (a decoder's first look around
// its guess): one reciprocal, twelve LDS reads in flight together, one test for the exact evaluation.  Bit-identical to
// leaky_gaussian_left<true> at the three indices.
__device__ __forceinline__ void leaky_gaussian_left3_quick(uint32_t g, int32_t lo, uint32_t n, int P, double mu, double sigma,
                                                           const double2* tab, uint32_t (&v)[3]) {
    constexpr double sqrt2 = 1.41421356237309504880168872420969808;
    const uint32_t total = P >= 32 ? 0u : (1u << P);
    const double free_weight = (double)((total - 1u) - (n - 1u));
    const double half = 0.5 * free_weight;
    const double inv_d = fast_rcp1(sigma * sqrt2);
    const double x1 = (double)(int32_t)((uint32_t)lo + g) - 0.5;
    const double x[3] = {x1 - 1.0, x1, x1 + 1.0};
    double y[3], arg[3];
    bool unsure[3];
    ErfPoly e[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        arg[k] = (x[k] - mu) * inv_d;
        e[k] = erf_poly_fetch(arg[k], tab);
    }
    __builtin_amdgcn_sched_barrier(0);               // (all twelve reads are under way before the first of them is needed)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        y[k] = __builtin_fma(erf_poly_eval(e[k], arg[k]), half, half);
        unsure[k] = !(__builtin_fabs(arg[k]) >= 6.0 || __builtin_fabs(__builtin_amdgcn_fract(y[k]) - 0.5) < 0.5 - kLeftGuard);
    }
    unsure[0] = unsure[0] && g > 1u;                 // (index 0 and index n are not evaluated: 0 and 2^P)
    unsure[2] = unsure[2] && g + 1u < n;
    if (__builtin_amdgcn_ballot_w64(unsure[0] || unsure[1] || unsure[2]) != 0ull) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (unsure[k]) y[k] = leaky_left_f64_exact(x[k], mu, sigma, free_weight, tab);
    }
    v[0] = g > 1u ? f64_as_u32_hw(y[0]) + (g - 1u) : 0u;
    v[1] = f64_as_u32_hw(y[1]) + g;
    v[2] = g + 1u < n ? f64_as_u32_hw(y[2]) + (g + 1u) : total;
}

// left_cumulative_and_probability through the quick evaluation (same contract as leaky_gaussian_lcp), BOTH ends of the bin in
// one straight line: one reciprocal, the eight LDS reads of the two polynomials in flight together, one test for the rare
// exact evaluation.  A symbol outside [lo, hi] is evaluated as `lo` and reported by the return value (no divergent early exit:
// one model per lane).
__device__ __forceinline__ bool leaky_gaussian_lcp_quick(int32_t sym, int32_t lo, int32_t hi, int P, int prob_bits, double mu, double sigma,
                                                         uint32_t& left, uint32_t& prob, const double2* tab) {
    constexpr double sqrt2 = 1.41421356237309504880168872420969808;
    const bool inside = sym >= lo && sym <= hi;
    const int32_t sc = inside ? sym : lo;
    const uint32_t pmask = prob_bits >= 32 ? 0xffffffffu : ((1u << prob_bits) - 1u);
    const uint32_t max_prob = pmask >> (prob_bits - P);
    const double free_weight = (double)(max_prob - ((uint32_t)hi - (uint32_t)lo));
    const double half = 0.5 * free_weight;
    const uint32_t slack = ((uint32_t)sc - (uint32_t)lo) & pmask;
    const double inv_d = fast_rcp1(sigma * sqrt2);
    const double xl = (double)sc - 0.5, xr = (double)sc + 0.5;
    const double al = (xl - mu) * inv_d, ar = (xr - mu) * inv_d;
    const ErfPoly el = erf_poly_fetch(al, tab), er = erf_poly_fetch(ar, tab);
    __builtin_amdgcn_sched_barrier(0);               // (all eight reads are under way before the first of them is needed)
    double yl = __builtin_fma(erf_poly_eval(el, al), half, half), yr = __builtin_fma(erf_poly_eval(er, ar), half, half);
    // in doubt (see leaky_left_value_quick): not saturated and within kLeftGuard of an integer -- or a NaN argument
    const bool unsure_l = !(__builtin_fabs(al) >= 6.0 || __builtin_fabs(__builtin_amdgcn_fract(yl) - 0.5) < 0.5 - kLeftGuard) && sc != lo;
    const bool unsure_r = !(__builtin_fabs(ar) >= 6.0 || __builtin_fabs(__builtin_amdgcn_fract(yr) - 0.5) < 0.5 - kLeftGuard) && sc != hi;
    if (__builtin_amdgcn_ballot_w64(unsure_l || unsure_r) != 0ull) {
#pragma unroll 1
        for (int side = 0; side < 2; ++side) {
            const bool u = side ? unsure_r : unsure_l;
            if (__builtin_amdgcn_ballot_w64(u) == 0ull) continue;
            if (u) {
                const double y = leaky_left_f64_exact(side ? xr : xl, mu, sigma, free_weight, tab);
                if (side) yr = y; else yl = y;
            }
        }
    }
    const uint32_t l = sc == lo ? 0u : (f64_as_u32_hw(yl) + slack) & pmask;
    const uint32_t r = sc == hi ? ((P >= 32 ? 0u : (1u << P)) & pmask) : (f64_as_u32_hw(yr) + slack + 1u) & pmask;
    left = l;
    prob = (r - l) & pmask;
    return inside;
}

} // namespace cst
