// cst_families.hip -- the model families of the Python API that are not the quantized Gaussian:
// LeakyQuantizer<f64, i32, u32, P> over Laplace / Cauchy / Binomial, tabulated ON THE DEVICE one thread per table entry
// (constriction.stream.model.QuantizedLaplace / QuantizedCauchy / Binomial, src/pybindings/stream/model.rs:736-966;
// quantizer arithmetic src/stream/model/quantize.rs:284-308, 525-568), and `perfectly_quantized_probabilities`
// (src/stream/model/categorical.rs:56-177), which is a sequential greedy search and runs on the host.
//
// The continuous distributions live in the un-vendored `probability` crate (0.20.3 -> special 0.10.3 -> libm 0.2.16,
// Cargo.lock); what is evaluated here are its published formulas over the libm-crate (musl / FreeBSD msun)
// elementary functions: log / log1p / atan / lgamma_r below, exp in cst_math.hpp.  categorical.rs:11 imports
// `libm::log1p` explicitly.  Like everything in cst_math.hpp this file needs -ffp-contract=off: each operation
// rounds once, in the order written, so that the CPU checker under tests (a separate C restatement)
// and the GPU agree bit for bit.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>
#include <vector>

#include "cst_common.hpp"
#include "cst_math.hpp"

#define CST_HD __host__ __device__ __forceinline__

namespace cst {

CST_HD uint64_t bits_of(double x) { return __builtin_bit_cast(uint64_t, x); }
CST_HD double from_bits(uint64_t u) { return __builtin_bit_cast(double, u); }
CST_HD uint32_t top_word(double x) { return (uint32_t)(bits_of(x) >> 32); }
CST_HD double replace_top(double x, uint32_t hi) { return from_bits(((uint64_t)hi << 32) | (bits_of(x) & 0xffffffffull)); }

// The shared tail of msun's log and log1p: x = 2^k (1 + f) with sqrt(2)/2 <= 1 + f < sqrt(2),
// log(x) = k ln2 + f - f^2/2 + s (f^2/2 + R(s^2)), s = f / (2 + f); `corr` is log1p's correction term (0 for log,
// where `dk * ln2_lo + 0.0` is the same number)
CST_HD double log_tail(double f, int k, double corr) {
    constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    constexpr double L1 = 6.666666666666735130e-01, L2 = 3.999999999940941908e-01, L3 = 2.857142874366239149e-01,
                     L4 = 2.222219843214978396e-01, L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01,
                     L7 = 1.479819860511658591e-01;
    const double hfsq = 0.5 * f * f;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double even = w * (L2 + w * (L4 + w * L6));
    const double odd = z * (L1 + w * (L3 + w * (L5 + w * L7)));
    const double R = odd + even;
    const double dk = (double)k;
    return s * (hfsq + R) + (dk * ln2_lo + corr) - hfsq + f + dk * ln2_hi;
}

// libm `log` (FreeBSD e_log.c)
CST_HD double log_exact(double x) {
    uint32_t hx = top_word(x);
    int k = 0;
    if (hx < 0x00100000u || (hx >> 31)) {
        if ((bits_of(x) << 1) == 0) return -1.0 / (x * x);
        if (hx >> 31) return (x - x) / 0.0;
        k = -54;                                     // subnormal: scale up
        x *= 0x1p54;
        hx = top_word(x);
    } else if (hx >= 0x7ff00000u) {
        return x;
    } else if (bits_of(x) == 0x3ff0000000000000ull) {
        return 0.0;
    }
    hx += 0x3ff00000u - 0x3fe6a09eu;                 // into [sqrt(2)/2, sqrt(2))
    k += (int)(hx >> 20) - 0x3ff;
    const double m = replace_top(x, (hx & 0x000fffffu) + 0x3fe6a09eu);
    return log_tail(m - 1.0, k, 0.0);
}

// libm `log1p` (FreeBSD s_log1p.c as arranged by musl)
CST_HD double log1p_exact(double x) {
    const uint32_t hx = top_word(x);
    if (hx < 0x3fda827au || (hx >> 31)) {            // 1 + x < sqrt(2)
        if (hx >= 0xbff00000u) return x == -1.0 ? x / 0.0 : (x - x) / 0.0;
        if ((hx << 1) < (0x3ca00000u << 1)) return x;                        // |x| < 2^-53
        if (hx <= 0xbfd2bec4u) return log_tail(x, 0, 0.0);                   // sqrt(2)/2 <= 1 + x: no reduction
    } else if (hx >= 0x7ff00000u) {
        return x;
    }
    const double u = 1.0 + x;
    uint32_t hu = top_word(u) + (0x3ff00000u - 0x3fe6a09eu);
    const int k = (int)(hu >> 20) - 0x3ff;
    double corr = 0.0;                               // log(1 + x) - log(u), from the rounding error of 1 + x
    if (k < 54) corr = (k >= 2 ? 1.0 - (u - x) : x - (u - 1.0)) / u;
    const double m = replace_top(u, (hu & 0x000fffffu) + 0x3fe6a09eu);
    return log_tail(m - 1.0, k, corr);
}

// libm `atan` (FreeBSD s_atan.c): argument reduction against atan(0.5), atan(1), atan(1.5), atan(inf)
__device__ inline double atan_exact(double x) {
    constexpr double hi_part[4] = {4.63647609000806093515e-01, 7.85398163397448278999e-01, 9.82793723247329054082e-01,
                                   1.57079632679489655800e+00};
    constexpr double lo_part[4] = {2.26987774529616870924e-17, 3.06161699786838301793e-17, 1.39033110312309984516e-17,
                                   6.12323399573676603587e-17};
    constexpr double T0 = 3.33333333333329318027e-01, T1 = -1.99999999998764832476e-01, T2 = 1.42857142725034663711e-01,
                     T3 = -1.11111104054623557880e-01, T4 = 9.09088713343650656196e-02, T5 = -7.69187620504482999495e-02,
                     T6 = 6.66107313738753120669e-02, T7 = -5.83357013379057348645e-02, T8 = 4.97687799461593236017e-02,
                     T9 = -3.65315727442169155270e-02, T10 = 1.62858201153657823623e-02;
    const uint32_t hx = top_word(x), ix = hx & 0x7fffffffu;
    const bool neg = (hx >> 31) != 0;
    if (ix >= 0x44100000u) {                         // |x| >= 2^66
        if (x != x) return x;
        const double z = hi_part[3] + (double)0x1p-120f;
        return neg ? -z : z;
    }
    int id = -1;
    double r = x;
    if (ix >= 0x3fdc0000u) {                         // |x| >= 0.4375
        const double ax = fabs(x);
        if (ix < 0x3fe60000u) { id = 0; r = (2.0 * ax - 1.0) / (2.0 + ax); }
        else if (ix < 0x3ff30000u) { id = 1; r = (ax - 1.0) / (ax + 1.0); }
        else if (ix < 0x40038000u) { id = 2; r = (ax - 1.5) / (1.0 + 1.5 * ax); }
        else { id = 3; r = -1.0 / ax; }
    } else if (ix < 0x3e400000u) {                   // |x| < 2^-27
        return x;
    }
    const double z = r * r, w = z * z;
    const double s1 = z * (T0 + w * (T2 + w * (T4 + w * (T6 + w * (T8 + w * T10)))));
    const double s2 = w * (T1 + w * (T3 + w * (T5 + w * (T7 + w * T9))));
    if (id < 0) return r - r * (s1 + s2);
    const double y = hi_part[id] - (r * (s1 + s2) - lo_part[id] - r);
    return neg ? -y : y;
}

// libm `lgamma_r` for x > 0 (FreeBSD e_lgamma_r.c); the Binomial CDF only ever asks for positive counts
__device__ inline double lgamma_pos_exact(double x) {
    constexpr double
        a0 = 7.72156649015328655494e-02, a1 = 3.22467033424113591611e-01, a2 = 6.73523010531292681824e-02,
        a3 = 2.05808084325167332806e-02, a4 = 7.38555086081402883957e-03, a5 = 2.89051383673415629091e-03,
        a6 = 1.19270763183362067845e-03, a7 = 5.10069792153511336608e-04, a8 = 2.20862790713908385557e-04,
        a9 = 1.08011567247583939954e-04, a10 = 2.52144565451257326939e-05, a11 = 4.48640949618915160150e-05,
        tc = 1.46163214496836224576e+00, tf = -1.21486290535849611461e-01, tt = -3.63867699703950536541e-18,
        t0 = 4.83836122723810047042e-01, t1 = -1.47587722994593911752e-01, t2 = 6.46249402391333854778e-02,
        t3 = -3.27885410759859649565e-02, t4 = 1.79706750811820387126e-02, t5 = -1.03142241298341437450e-02,
        t6 = 6.10053870246291332635e-03, t7 = -3.68452016781138256760e-03, t8 = 2.25964780900612472250e-03,
        t9 = -1.40346469989232843813e-03, t10 = 8.81081882437654011382e-04, t11 = -5.38595305356740546715e-04,
        t12 = 3.15632070903625950361e-04, t13 = -3.12754168375120860518e-04, t14 = 3.35529192635519073543e-04,
        u0 = -7.72156649015328655494e-02, u1 = 6.32827064025093366517e-01, u2 = 1.45492250137234768737e+00,
        u3 = 9.77717527963372745603e-01, u4 = 2.28963728064692451092e-01, u5 = 1.33810918536787660377e-02,
        v1 = 2.45597793713041134822e+00, v2 = 2.12848976379893395361e+00, v3 = 7.69285150456672783825e-01,
        v4 = 1.04222645593369134254e-01, v5 = 3.21709242282423911810e-03,
        s0 = -7.72156649015328655494e-02, s1 = 2.14982415960608852501e-01, s2 = 3.25778796408930981787e-01,
        s3 = 1.46350472652464452805e-01, s4 = 2.66422703033638609560e-02, s5 = 1.84028451407337715652e-03,
        s6 = 3.19475326584100867617e-05,
        r1 = 1.39200533467621045958e+00, r2 = 7.21935547567138069525e-01, r3 = 1.71933865632803078993e-01,
        r4 = 1.86459191715652901344e-02, r5 = 7.77942496381893596434e-04, r6 = 7.32668430744625636189e-06,
        w0 = 4.18938533204672725052e-01, w1 = 8.33333333333329678849e-02, w2 = -2.77777777728775536470e-03,
        w3 = 7.93650558643019558500e-04, w4 = -5.95187557450339963135e-04, w5 = 8.36339918996282139126e-04,
        w6 = -1.63092934096575273989e-03;
    const uint32_t hx = top_word(x), ix = hx & 0x7fffffffu;
    if (ix >= 0x7ff00000u) return x * x;
    if (hx >> 31) return (x - x) / 0.0;
    if (ix < ((0x3ffu - 70u) << 20)) return -log_exact(x);
    if ((ix == 0x3ff00000u || ix == 0x40000000u) && (uint32_t)bits_of(x) == 0u) return 0.0;     // lgamma(1) = lgamma(2) = 0
    if (ix < 0x40000000u) {                          // x < 2: three polynomial pieces, around 1 (or 2), tc, and 0 (or 1)
        double r, y;
        int piece;
        if (ix <= 0x3fecccccu) {                     // lgamma(x) = lgamma(x + 1) - log(x)
            r = -log_exact(x);
            if (ix >= 0x3fe76944u) { y = 1.0 - x; piece = 0; }
            else if (ix >= 0x3fcda661u) { y = x - (tc - 1.0); piece = 1; }
            else { y = x; piece = 2; }
        } else {
            r = 0.0;
            if (ix >= 0x3ffbb4c3u) { y = 2.0 - x; piece = 0; }
            else if (ix >= 0x3ff3b4c4u) { y = x - tc; piece = 1; }
            else { y = x - 1.0; piece = 2; }
        }
        if (piece == 0) {
            const double z = y * y;
            const double p1 = a0 + z * (a2 + z * (a4 + z * (a6 + z * (a8 + z * a10))));
            const double p2 = z * (a1 + z * (a3 + z * (a5 + z * (a7 + z * (a9 + z * a11)))));
            const double p = y * p1 + p2;
            return r + (p - 0.5 * y);
        }
        if (piece == 1) {
            const double z = y * y, w = z * y;
            const double p1 = t0 + w * (t3 + w * (t6 + w * (t9 + w * t12)));
            const double p2 = t1 + w * (t4 + w * (t7 + w * (t10 + w * t13)));
            const double p3 = t2 + w * (t5 + w * (t8 + w * (t11 + w * t14)));
            const double p = z * p1 - (tt - w * (p2 + y * p3));
            return r + (tf + p);
        }
        const double p1 = y * (u0 + y * (u1 + y * (u2 + y * (u3 + y * (u4 + y * u5)))));
        const double p2 = 1.0 + y * (v1 + y * (v2 + y * (v3 + y * (v4 + y * v5))));
        return r + (-0.5 * y + p1 / p2);
    }
    if (ix < 0x40200000u) {                          // 2 <= x < 8: lgamma(2 + y) times the product (y + 2) ... (y + i - 1)
        const int i = (int)x;
        const double y = x - (double)i;
        const double p = y * (s0 + y * (s1 + y * (s2 + y * (s3 + y * (s4 + y * (s5 + y * s6))))));
        const double q = 1.0 + y * (r1 + y * (r2 + y * (r3 + y * (r4 + y * (r5 + y * r6)))));
        double r = 0.5 * y + p / q;
        if (i >= 3) {
            double z = 1.0;
            for (int j = i; j >= 3; --j) z *= y + (double)(j - 1);
            r += log_exact(z);
        }
        return r;
    }
    if (ix < 0x43900000u) {                          // 8 <= x < 2^58: Stirling
        const double t = log_exact(x), z = 1.0 / x, y = z * z;
        const double w = w0 + z * (w1 + y * (w2 + y * (w3 + y * (w4 + y * (w5 + y * w6)))));
        return (x - 0.5) * (t - 1.0) + w;
    }
    return x * (log_exact(x) - 1.0);
}

// ---- probability 0.20.3 `distribution` (the CDFs) ----

__device__ inline double laplace_cdf_exact(double x, double mu, double b) {
    return x <= mu ? 0.5 * exp_exact((x - mu) / b) : 1.0 - 0.5 * exp_exact((mu - x) / b);
}

__device__ inline double cauchy_cdf_exact(double x, double x0, double gamma) {
    constexpr double pi = 3.14159265358979323846264338327950288;
    return atan_exact((x - x0) / gamma) / pi + 0.5;
}

// special::Beta::inc_beta: Algorithm AS 63 (Soper's reduction formulae) with remark AS R19 / algorithm AS 109
__device__ inline double inc_beta_exact(double x, double p, double q, double ln_beta) {
    constexpr double acu = 0.1e-14;
    if (x <= 0.0) return 0.0;
    if (x >= 1.0) return 1.0;
    double psq = p + q;
    double xx = x, cx = 1.0 - x, pp = p, qq = q;
    const bool tail = p < psq * x;                   // work on the other tail and return the complement
    if (tail) { xx = cx; cx = x; pp = q; qq = p; }
    double term = 1.0, ai = 1.0, value = 1.0;
    int ns = (int)(qq + cx * psq);
    double rx = ns == 0 ? xx : xx / cx;
    double temp = qq - ai;
    for (;;) {
        term = term * temp * rx / (pp + ai);
        value += term;
        temp = fabs(term);
        if (temp <= acu && temp <= acu * value) break;
        ai += 1.0;
        --ns;
        if (ns >= 0) {
            temp = qq - ai;
            if (ns == 0) rx = xx;
        } else {
            temp = psq;
            psq += 1.0;
        }
    }
    value = value * exp_exact(pp * log_exact(xx) + (qq - 1.0) * log_exact(cx) - ln_beta) / pp;
    return tail ? 1.0 - value : value;
}

// f64::powi (compiler-rt's __powidf2: square and multiply from the low bit up)
__device__ inline double powi_exact(double a, int b) {
    const bool recip = b < 0;
    double r = 1.0;
    for (;;) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return recip ? 1.0 / r : r;
}

// Binomial(n, p)::distribution: P[X <= floor(x)] = I_{1-p}(n - k, k + 1)
__device__ inline double binomial_cdf_exact(double x, int32_t n, double p) {
    if (x < 0.0) return 0.0;
    if (x >= (double)n) return 1.0;
    const int32_t k = (int32_t)x;
    const double q = 1.0 - p;
    if (k == 0) return powi_exact(q, n);
    const double a = (double)(n - k), b = (double)(k + 1);
    return inc_beta_exact(q, a, b, lgamma_pos_exact(a) + lgamma_pos_exact(b) - lgamma_pos_exact(a + b));
}

// rows[row][i] = left cumulative of symbol index i under LeakyQuantizer(lo ..= hi_row) x family(a[row], b[row]);
// hi_row = lo + n_per_row[row] for the Binomial family form with its own `n` per row (entries past the row's own
// 2^P repeat 2^P: zero-width bins the decoders never select), hi otherwise.  One thread per entry.
__global__ void family_rows_kernel(int family, int P, int32_t lo, int32_t hi, const double* __restrict__ a, const double* __restrict__ b,
                                   const int32_t* __restrict__ n_per_row, size_t n_rows, uint32_t* __restrict__ rows) {
    const size_t per = (size_t)((int64_t)hi - lo + 2);
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n_rows * per) return;
    const size_t row = gid / per;
    const int64_t i = (int64_t)(gid - row * per);
    const int32_t hi_row = n_per_row ? lo + n_per_row[row] : hi;
    const int64_t n = (int64_t)hi_row - lo + 1;
    const uint32_t total = P >= 32 ? 0u : (1u << P);
    uint32_t v;
    if (i == 0) v = 0u;
    else if (i >= n) v = total;
    else {
        const uint32_t max_prob = 0xffffffffu >> (32 - P);
        const double free_weight = (double)(max_prob - ((uint32_t)hi_row - (uint32_t)lo));
        const double x = (double)(int32_t)(lo + i) - 0.5;
        const double pa = a[row], pb = b ? b[row] : 0.0;
        double c;
        if (family == CST_FAMILY_LAPLACE) c = laplace_cdf_exact(x, pa, pb);
        else if (family == CST_FAMILY_CAUCHY) c = cauchy_cdf_exact(x, pa, pb);
        else c = binomial_cdf_exact(x, hi_row - lo, pa);
        v = f64_as_u32_sat(free_weight * c) + (uint32_t)i;
    }
    rows[gid] = v;
}

// bad[row] = 1 if the row is not strictly increasing up to its own 2^P (quantize.rs:560-566 panics there)
__global__ void family_rows_check_kernel(int P, int32_t lo, int32_t hi, const int32_t* __restrict__ n_per_row, size_t n_rows,
                                         const uint32_t* __restrict__ rows, int32_t* __restrict__ bad) {
    const size_t per = (size_t)((int64_t)hi - lo + 2);
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n_rows * per) return;
    const size_t row = gid / per;
    const int64_t i = (int64_t)(gid - row * per);
    const int64_t n = n_per_row ? (int64_t)n_per_row[row] + 1 : (int64_t)hi - lo + 1;
    if (i < n && rows[gid + 1] <= rows[gid]) atomicOr(&bad[row], 1);
}

__global__ void debug_fn_kernel(int which, const double* __restrict__ x, double* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    out[i] = which == 0 ? log_exact(v) : which == 1 ? log1p_exact(v) : which == 2 ? atan_exact(v) : which == 3 ? lgamma_pos_exact(v) : exp_exact(v);
}

} // namespace cst

using namespace cst;

extern "C" {

cst_status cst_family_cdf_rows(int32_t family, int32_t precision, int32_t min_symbol, int32_t max_symbol, const double* d_a,
                                       const double* d_b, const int32_t* d_n_per_row, size_t n_rows, uint32_t* d_rows, int32_t* d_bad,
                                       void* stream) {
    if (family != CST_FAMILY_LAPLACE && family != CST_FAMILY_CAUCHY && family != CST_FAMILY_BINOMIAL) return CST_ERR_INVALID_ARGUMENT;
    if (precision < 1 || precision > 32 || max_symbol <= min_symbol || !d_a || !d_rows) return CST_ERR_INVALID_ARGUMENT;
    if (family != CST_FAMILY_BINOMIAL && !d_b) return CST_ERR_INVALID_ARGUMENT;
    if (family == CST_FAMILY_BINOMIAL && min_symbol != 0) return CST_ERR_INVALID_ARGUMENT;
    if (family != CST_FAMILY_BINOMIAL && d_n_per_row) return CST_ERR_INVALID_ARGUMENT;
    const uint64_t span = (uint64_t)((int64_t)max_symbol - min_symbol);
    if (precision < 32 && span >= (1ull << precision)) return CST_ERR_MODEL;        // LeakyQuantizer::new asserts this
    if (n_rows == 0) return CST_OK;
    hipStream_t hs = (hipStream_t)stream;
    const size_t total = n_rows * (size_t)(span + 2);
    const unsigned blocks = (unsigned)((total + 255) / 256);
    family_rows_kernel<<<blocks, 256, 0, hs>>>(family, precision, min_symbol, max_symbol, d_a, d_b, d_n_per_row, n_rows, d_rows);
    if (d_bad) {
        CST_HIP_TRY(hipMemsetAsync(d_bad, 0, sizeof(int32_t) * n_rows, hs));
        family_rows_check_kernel<<<blocks, 256, 0, hs>>>(precision, min_symbol, max_symbol, d_n_per_row, n_rows, d_rows, d_bad);
    }
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

// perfectly_quantized_probabilities (categorical.rs:56-177) + the cumulation of contiguous.rs:301-313; host code: the
// search moves one unit of weight at a time and every move depends on the previous one.
cst_status cst_categorical_perfect_cdf(const double* h_probs, size_t n, int32_t precision, uint32_t* h_cdf) {
    if (!h_probs || !h_cdf || precision < 1 || precision > 32) return CST_ERR_INVALID_ARGUMENT;
    if (n < 2 || n > 0xffffffffull || (precision < 32 && n > (1ull << precision))) return CST_ERR_MODEL;
    const uint32_t total = precision >= 32 ? 0u : (1u << precision);
    struct Slot { double prob, win, loss; uint32_t weight; };
    std::vector<Slot> slot(n);
    double norm = 0.0;
    for (size_t i = 0; i < n; ++i) norm += h_probs[i];
    if (!(std::isnormal(norm) && norm > 0.0)) return CST_ERR_MODEL;
    uint32_t left_over = total - (uint32_t)n;
    const double scale = (double)left_over / norm;
    const double inf = std::numeric_limits<double>::infinity();
    auto gain = [](double prob, uint32_t weight) { return prob * log1p_exact(1.0 / (double)weight); };
    auto cost = [&](double prob, uint32_t weight) { return weight == 1u ? inf : -prob * log1p_exact(-1.0 / (double)weight); };
    for (size_t i = 0; i < n; ++i) {
        const double prob = h_probs[i];
        if (prob < 0.0) return CST_ERR_MODEL;
        const double share = prob * scale;
        const uint32_t extra = !(share > 0.0) ? 0u : share >= 4294967296.0 ? 0xffffffffu : (uint32_t)share;
        if (extra > left_over) return CST_ERR_MODEL;
        left_over -= extra;
        slot[i] = {prob, gain(prob, extra + 1u), cost(prob, extra + 1u), extra + 1u};
    }
    // the weight that truncation left over goes to the symbols with the largest gains, at most one unit each per round;
    // `order` is the reference's `slots` vector, which keeps its order from round to round (stable sort)
    std::vector<uint32_t> order(n);
    std::iota(order.begin(), order.end(), 0u);
    while (left_over != 0u) {
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return slot[x].win > slot[y].win; });
        const size_t batch = std::min<size_t>(left_over, n);
        for (size_t j = 0; j < batch; ++j) {
            Slot& s = slot[order[j]];
            s.weight += 1u;
            s.win = gain(s.prob, s.weight);
            s.loss = -s.prob * log1p_exact(-1.0 / (double)s.weight);
        }
        left_over -= (uint32_t)batch;
    }
    // then single units move from the cheapest seller to the best buyer while that lowers the cross entropy
    for (;;) {
        size_t buyer = 0, seller = 0;                // positions in `order`: max_by keeps the last maximum, min_by the first minimum
        for (size_t j = 1; j < n; ++j) {
            if (!(slot[order[j]].win < slot[order[buyer]].win)) buyer = j;
            if (slot[order[j]].loss < slot[order[seller]].loss) seller = j;
        }
        if (buyer == seller) break;
        Slot& bs = slot[order[buyer]];
        Slot& ss = slot[order[seller]];
        if (bs.win <= ss.loss) break;
        ss.weight -= 1u;
        ss.win = -inf;                               // a weight that went down never goes up again, and vice versa
        ss.loss = cost(ss.prob, ss.weight);
        bs.weight += 1u;
        bs.loss = inf;
        bs.win = gain(bs.prob, bs.weight);
    }
    uint32_t acc = 0u;
    for (size_t i = 0; i < n; ++i) { h_cdf[i] = acc; acc += slot[i].weight; }
    h_cdf[n] = acc;
    return acc == total ? CST_OK : CST_ERR_MODEL;
}

// test hooks: the elementary functions on the device (which: 0 log, 1 log1p, 2 atan, 3 lgamma (x > 0), 4 exp) and
// log1p on the host (the perfect quantizer's)
cst_status cst_debug_family_fn(int32_t which, const double* d_x, double* d_out, size_t n, void* stream) {
    if (which < 0 || which > 4 || !d_x || !d_out) return CST_ERR_INVALID_ARGUMENT;
    if (n == 0) return CST_OK;
    debug_fn_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(which, d_x, d_out, n);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

double cst_debug_host_log1p(double x) { return log1p_exact(x); }

} // extern "C"
