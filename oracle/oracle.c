/*
 * oracle.c -- CPU restatement of constriction's stream-coder hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (constriction_amd/)
 * never links, imports or calls anything in this directory.
 *
 * Parity status: PINNED against the reference's own golden vectors (tests/golden/
 * reference_vectors.json, transcribed from the reference's tests/python/ files and the
 * doc-tests under its src/stream/ tree; see tests/test_oracle_golden.py).  The
 * reference itself (a Rust crate) cannot be built here (no rustc/cargo), so there is no
 * oracle/_ref.  The Gaussian CDF lives in third-party crates that are NOT vendored under
 * /root/reference: probability 0.20.3 -> special 0.10.3 -> libm 0.2.16 (Cargo.lock).  Their
 * published algorithm (the Sun/FreeBSD msun erf as re-expressed by musl, which the Rust
 * `libm` crate transliterates, plus musl's pre-2019 exp) is restated below; the golden
 * vectors pin it to 24-bit granularity, its last-ulp behaviour is defined by this file.
 *
 * Each function cites the reference file:line it follows (paths relative to
 * /root/reference/).  Plain C99, no dependencies, compile with -ffp-contract=off.
 */
#define _GNU_SOURCE            /* pthread_setaffinity_np (cpu_baseline leg only) */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <sched.h>
#include <unistd.h>

#define API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * f64 special functions (third-party in the reference: libm 0.2.16 `exp`, `erf`).
 * ---------------------------------------------------------------------------------------- */

static inline uint64_t f64_bits(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline double bits_f64(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }
static inline uint32_t hi_word(double x) { return (uint32_t)(f64_bits(x) >> 32); }

/* scalbn restricted to what exp needs (|n| < 2100); follows musl scalbn.c */
static double o_scalbn(double x, int n)
{
    double y = x;
    if (n > 1023) {
        y *= 0x1p1023; n -= 1023;
        if (n > 1023) { y *= 0x1p1023; n -= 1023; if (n > 1023) n = 1023; }
    } else if (n < -1022) {
        y *= 0x1p-1022 * 0x1p53; n += 1022 - 53;
        if (n < -1022) { y *= 0x1p-1022 * 0x1p53; n += 1022 - 53; if (n < -1022) n = -1022; }
    }
    return y * bits_f64((uint64_t)(0x3ff + n) << 52);
}

/* exp(x): musl exp.c (origin FreeBSD e_exp.c) == Rust libm 0.2 `exp` */
API double cst_oracle_exp(double x)
{
    static const double half[2] = {0.5, -0.5};
    static const double ln2hi = 6.93147180369123816490e-01, ln2lo = 1.90821492927058770002e-10,
                        invln2 = 1.44269504088896338700e+00, P1 = 1.66666666666666019037e-01,
                        P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                        P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    double hi, lo, c, xx, y;
    int k, sign;
    uint32_t hx = hi_word(x);
    sign = (int)(hx >> 31);
    hx &= 0x7fffffff;
    if (hx >= 0x4086232b) { /* |x| >= 708.39 or nan */
        if (x != x) return x;
        if (x > 709.782712893383973096) return x * 0x1p1023;
        if (x < -708.39641853226410622) {
            if (x < -745.13321910194110842) return 0.0;
        }
    }
    if (hx > 0x3fd62e42) { /* |x| > 0.5 ln2 */
        if (hx >= 0x3ff0a2b2) /* |x| >= 1.5 ln2 */
            k = (int)(invln2 * x + half[sign]);
        else
            k = 1 - sign - sign;
        hi = x - k * ln2hi; /* k*ln2hi is exact here */
        lo = k * ln2lo;
        x = hi - lo;
    } else if (hx > 0x3e300000) { /* |x| > 2**-28 */
        k = 0; hi = x; lo = 0;
    } else {
        return 1 + x;
    }
    xx = x * x;
    c = x - xx * (P1 + xx * (P2 + xx * (P3 + xx * (P4 + xx * P5))));
    y = 1 + (x * c / (2 - c) - lo + hi);
    if (k == 0) return y;
    return o_scalbn(y, k);
}

static const double
erx  = 8.45062911510467529297e-01, efx8 = 1.02703333676410069053e+00,
pp0  = 1.28379167095512558561e-01, pp1  = -3.25042107247001499370e-01,
pp2  = -2.84817495755985104766e-02, pp3  = -5.77027029648944159157e-03,
pp4  = -2.37630166566501626084e-05,
qq1  = 3.97917223959155352819e-01, qq2  = 6.50222499887672944485e-02,
qq3  = 5.08130628187576562776e-03, qq4  = 1.32494738004321644526e-04,
qq5  = -3.96022827877536812320e-06,
pa0  = -2.36211856075265944077e-03, pa1  = 4.14856118683748331666e-01,
pa2  = -3.72207876035701323847e-01, pa3  = 3.18346619901161753674e-01,
pa4  = -1.10894694282396677476e-01, pa5  = 3.54783043256182359371e-02,
pa6  = -2.16637559486879084300e-03,
qa1  = 1.06420880400844228286e-01, qa2  = 5.40397917702171048937e-01,
qa3  = 7.18286544141962662868e-02, qa4  = 1.26171219808761642112e-01,
qa5  = 1.36370839120290507362e-02, qa6  = 1.19844998467991074170e-02,
ra0  = -9.86494403484714822705e-03, ra1  = -6.93858572707181764372e-01,
ra2  = -1.05586262253232909814e+01, ra3  = -6.23753324503260060396e+01,
ra4  = -1.62396669462573470355e+02, ra5  = -1.84605092906711035994e+02,
ra6  = -8.12874355063065934246e+01, ra7  = -9.81432934416914548592e+00,
sa1  = 1.96512716674392571292e+01, sa2  = 1.37657754143519042600e+02,
sa3  = 4.34565877475229228821e+02, sa4  = 6.45387271733267880336e+02,
sa5  = 4.29008140027567833386e+02, sa6  = 1.08635005541779435134e+02,
sa7  = 6.57024977031928170135e+00, sa8  = -6.04244152148580987438e-02,
rb0  = -9.86494292470009928597e-03, rb1  = -7.99283237680523006574e-01,
rb2  = -1.77579549177547519889e+01, rb3  = -1.60636384855821916062e+02,
rb4  = -6.37566443368389627722e+02, rb5  = -1.02509513161107724954e+03,
rb6  = -4.83519191608651397019e+02,
sb1  = 3.03380607434824582924e+01, sb2  = 3.25792512996573918826e+02,
sb3  = 1.53672958608443695994e+03, sb4  = 3.19985821950859553908e+03,
sb5  = 2.55305040643316442583e+03, sb6  = 4.74528541206955367215e+02,
sb7  = -2.24409524465858183362e+01;

static double o_fabs(double x) { return bits_f64(f64_bits(x) & 0x7fffffffffffffffull); }

static double erfc1(double x)
{
    double s, P, Q;
    s = o_fabs(x) - 1;
    P = pa0 + s * (pa1 + s * (pa2 + s * (pa3 + s * (pa4 + s * (pa5 + s * pa6)))));
    Q = 1 + s * (qa1 + s * (qa2 + s * (qa3 + s * (qa4 + s * (qa5 + s * qa6)))));
    return 1 - erx - P / Q;
}

static double erfc2(uint32_t ix, double x)
{
    double s, R, S, z;
    if (ix < 0x3ff40000) /* |x| < 1.25 */
        return erfc1(x);
    x = o_fabs(x);
    s = 1 / (x * x);
    if (ix < 0x4006db6d) { /* |x| < 1/.35 ~ 2.85714 */
        R = ra0 + s * (ra1 + s * (ra2 + s * (ra3 + s * (ra4 + s * (ra5 + s * (ra6 + s * ra7))))));
        S = 1.0 + s * (sa1 + s * (sa2 + s * (sa3 + s * (sa4 + s * (sa5 + s * (sa6 + s * (sa7 + s * sa8)))))));
    } else { /* |x| > 1/.35 */
        R = rb0 + s * (rb1 + s * (rb2 + s * (rb3 + s * (rb4 + s * (rb5 + s * rb6)))));
        S = 1.0 + s * (sb1 + s * (sb2 + s * (sb3 + s * (sb4 + s * (sb5 + s * (sb6 + s * sb7))))));
    }
    z = bits_f64(f64_bits(x) & 0xffffffff00000000ull);
    return cst_oracle_exp(-z * z - 0.5625) * cst_oracle_exp((z - x) * (z + x) + R / S) / x;
}

/* erf(x): musl erf.c (origin FreeBSD s_erf.c) == Rust libm 0.2 `erf`; reached from the
 * reference through special::Error::error (call sites src/stream/model/quantize.rs:546,558). */
API double cst_oracle_erf(double x)
{
    double r, s, z, y;
    uint32_t ix = hi_word(x);
    int sign = (int)(ix >> 31);
    ix &= 0x7fffffff;
    if (ix >= 0x7ff00000) /* erf(nan)=nan, erf(+-inf)=+-1 */
        return 1 - 2 * sign + 1 / x;
    if (ix < 0x3feb0000) { /* |x| < 0.84375 */
        if (ix < 0x3e300000) /* |x| < 2**-28 */
            return 0.125 * (8 * x + efx8 * x);
        z = x * x;
        r = pp0 + z * (pp1 + z * (pp2 + z * (pp3 + z * pp4)));
        s = 1.0 + z * (qq1 + z * (qq2 + z * (qq3 + z * (qq4 + z * qq5))));
        y = r / s;
        return x + x * y;
    }
    if (ix < 0x40180000) /* 0.84375 <= |x| < 6 */
        y = 1 - erfc2(ix, x);
    else
        y = 1 - 0x1p-1022;
    return sign ? -y : y;
}

/* Gaussian CDF: probability 0.20.3 `Gaussian::distribution`
 *   (1 + erf((x - mu) / (sigma * SQRT_2))) / 2
 * constructed at src/pybindings/stream/model.rs:658. */
API double cst_oracle_gaussian_cdf(double x, double mu, double sigma)
{
    static const double sqrt2 = 1.41421356237309504880168872420969808;
    return (1.0 + cst_oracle_erf((x - mu) / (sigma * sqrt2))) / 2.0;
}

/* ------------------------------------------------------------------------------------------
 * LeakyQuantizer (src/stream/model/quantize.rs)
 * ---------------------------------------------------------------------------------------- */

/* Rust `f64 as u32`: truncate toward zero, saturate, NaN -> 0 */
static inline uint32_t f64_as_u32(double v)
{
    if (!(v > 0.0)) return 0; /* also NaN and negatives */
    if (v >= 4294967296.0) return 0xffffffffu;
    return (uint32_t)v;
}

static inline uint32_t wrapping_pow2_u32(int e) { return e >= 32 ? 0u : (1u << e); } /* src/lib.rs:733-739 */

/* free_weight, quantize.rs:284-308 (Probability::max_value() >> (BITS - P)) - (hi-lo) */
static inline double leaky_free_weight(int32_t lo, int32_t hi, int P, int prob_bits)
{
    uint32_t max_prob = (prob_bits == 32 ? 0xffffffffu : ((1u << prob_bits) - 1u)) >> (prob_bits - P);
    uint32_t span = (uint32_t)hi - (uint32_t)lo;
    return (double)(max_prob - span);
}

/* left_cumulative_and_probability, quantize.rs:525-568, for D = Gaussian(mu, sigma),
 * Symbol = i32, Probability = u{prob_bits} held in u32 (wrapping at prob_bits).
 * Returns 0 on success, 1 if the symbol is outside [lo,hi] (-> ImpossibleSymbol),
 * 2 if the resulting probability is zero (reference panics). */
API int cst_oracle_leaky_gaussian_lcp(int32_t sym, int32_t lo, int32_t hi, int P, int prob_bits,
                                      double mu, double sigma, uint32_t *left, uint32_t *prob)
{
    uint32_t pmask = prob_bits == 32 ? 0xffffffffu : ((1u << prob_bits) - 1u);
    double fw = leaky_free_weight(lo, hi, P, prob_bits);
    if (sym < lo || sym > hi) return 1;
    uint32_t slack = ((uint32_t)sym - (uint32_t)lo) & pmask; /* quantize.rs:475-486 */
    uint32_t l, r;
    if (sym == lo)
        l = 0;
    else
        l = (f64_as_u32(fw * cst_oracle_gaussian_cdf((double)sym - 0.5, mu, sigma)) + slack) & pmask;
    if (sym == hi)
        r = wrapping_pow2_u32(P) & pmask;
    else
        r = (f64_as_u32(fw * cst_oracle_gaussian_cdf((double)sym + 0.5, mu, sigma)) + slack + 1u) & pmask;
    uint32_t p = (r - l) & pmask;
    *left = l; *prob = p;
    return p == 0 ? 2 : 0;
}

/* Tabulate left cumulatives L[0..n] (L[n] = 2^P wrapping) from left_cumulative_and_probability.
 * (NOT from symbol_table(), see SURVEY.md hazard 1.)  Returns 0, or 2 if any prob is zero. */
API int cst_oracle_leaky_gaussian_cdf_table(int32_t lo, int32_t hi, int P, int prob_bits, double mu,
                                            double sigma, uint32_t *cdf)
{
    int n = (int)((int64_t)hi - lo + 1), bad = 0;
    for (int i = 0; i < n; i++) {
        uint32_t l, p;
        int rc = cst_oracle_leaky_gaussian_lcp((int32_t)(lo + i), lo, hi, P, prob_bits, mu, sigma, &l, &p);
        if (rc) bad = 2;
        cdf[i] = l;
    }
    cdf[n] = wrapping_pow2_u32(P) & (prob_bits == 32 ? 0xffffffffu : ((1u << prob_bits) - 1u));
    return bad;
}

/* quantile_function semantics of quantize.rs:580-779: the unique symbol with
 * left(sym) <= q < right(sym).  right(s) is bit-identical to left(s+1), so a bisection over
 * left() returns the same triple as the reference's hinted search. */
API int cst_oracle_leaky_gaussian_quantile(uint32_t q, int32_t lo, int32_t hi, int P, int prob_bits,
                                           double mu, double sigma, int32_t *sym, uint32_t *left,
                                           uint32_t *prob)
{
    int64_t a = lo, b = hi; /* invariant: left(a) <= q */
    while (a < b) {
        int64_t m = a + (b - a + 1) / 2;
        uint32_t l, p;
        cst_oracle_leaky_gaussian_lcp((int32_t)m, lo, hi, P, prob_bits, mu, sigma, &l, &p);
        if (l <= q) a = m; else b = m - 1;
    }
    *sym = (int32_t)a;
    return cst_oracle_leaky_gaussian_lcp((int32_t)a, lo, hi, P, prob_bits, mu, sigma, left, prob);
}

/* ------------------------------------------------------------------------------------------
 * Categorical "fast" tables (src/stream/model/categorical.rs:16-54,
 * contiguous.rs:203-214) and the lookup table (lookup_contiguous.rs:297-331, 611-636)
 * ---------------------------------------------------------------------------------------- */

API int cst_oracle_categorical_fast_cdf_f64(const double *probs, int n, int P, uint32_t *cdf)
{
    if (n < 2 || (uint64_t)n >= (((uint64_t)1 << P) - 1)) return 1;
    uint32_t free_weight = wrapping_pow2_u32(P) - (uint32_t)n;
    double norm = 0.0;
    for (int i = 0; i < n; i++) norm = norm + probs[i];
    if (!(norm > 0.0) || norm != norm || norm > 1.7976931348623157e308 || norm < 2.2250738585072014e-308) return 1;
    double scale = (double)free_weight / norm;
    double cum = 0.0;
    for (int i = 0; i < n; i++) {
        cdf[i] = f64_as_u32(cum * scale) + (uint32_t)i;
        cum = cum + probs[i];
    }
    cdf[n] = wrapping_pow2_u32(P);
    return 0;
}

API int cst_oracle_categorical_fast_cdf_f32(const float *probs, int n, int P, uint32_t *cdf)
{
    if (n < 2 || (uint64_t)n >= (((uint64_t)1 << P) - 1)) return 1;
    uint32_t free_weight = wrapping_pow2_u32(P) - (uint32_t)n;
    float norm = 0.0f;
    for (int i = 0; i < n; i++) norm = norm + probs[i];
    if (!(norm > 0.0f) || norm != norm || norm > 3.4028234e38f || norm < 1.17549435e-38f) return 1;
    float scale = (float)free_weight / norm;
    float cum = 0.0f;
    for (int i = 0; i < n; i++) {
        float v = cum * scale;
        uint32_t t = !(v > 0.0f) ? 0u : (v >= 4294967296.0f ? 0xffffffffu : (uint32_t)v);
        cdf[i] = t + (uint32_t)i;
        cum = cum + probs[i];
    }
    cdf[n] = wrapping_pow2_u32(P);
    return 0;
}

/* lookup_table[q] = index i with cdf[i] <= q < cdf[i+1]  (lookup_contiguous.rs:611-636) */
API void cst_oracle_lookup_from_cdf(const uint32_t *cdf, int n, int P, uint16_t *lookup)
{
    uint32_t total = 1u << P;
    int i = 0;
    for (uint32_t q = 0; q < total; q++) {
        while (i + 1 < n && cdf[i + 1] <= q) i++;
        lookup[q] = (uint16_t)i;
    }
}

/* ------------------------------------------------------------------------------------------
 * ANS coder (src/stream/stack.rs), generic over (W, S) at run time; words are stored in
 * uint32_t slots even when W = 16.
 * ---------------------------------------------------------------------------------------- */

typedef struct {
    uint64_t state;
    uint32_t *bulk;
    size_t len, cap;
    int W, S;
} oans_t;

static void bulk_push(oans_t *c, uint32_t w)
{
    if (c->len == c->cap) {
        c->cap = c->cap ? c->cap * 2 : 64;
        c->bulk = (uint32_t *)realloc(c->bulk, c->cap * sizeof(uint32_t));
    }
    c->bulk[c->len++] = w;
}

API oans_t *cst_oracle_ans_new(int W, int S) /* stack.rs:249-276: state = 0, empty bulk */
{
    oans_t *c = (oans_t *)calloc(1, sizeof(oans_t));
    c->W = W; c->S = S;
    return c;
}
API void cst_oracle_ans_free(oans_t *c) { if (c) { free(c->bulk); free(c); } }
API void cst_oracle_ans_clear(oans_t *c) { c->len = 0; c->state = 0; } /* stack.rs:711-714 */
API uint64_t cst_oracle_ans_state(const oans_t *c) { return c->state; }
API size_t cst_oracle_ans_bulk_len(const oans_t *c) { return c->len; }
API int cst_oracle_ans_is_empty(const oans_t *c) { return c->len == 0 && c->state == 0; } /* stack.rs:481-487 */

static inline uint64_t word_mask(int W) { return W == 64 ? ~0ull : (((uint64_t)1 << W) - 1); }
static inline int bit_len(uint64_t x) { return x ? 64 - __builtin_clzll(x) : 0; }

/* from_compressed (stack.rs:299-318) + read_initial_state (stack.rs:440-462).
 * Returns NULL if the last word is zero. */
API oans_t *cst_oracle_ans_from_compressed(int W, int S, const uint32_t *words, size_t n)
{
    oans_t *c = cst_oracle_ans_new(W, S);
    for (size_t i = 0; i < n; i++) bulk_push(c, words[i]);
    if (c->len > 0) {
        uint32_t first = c->bulk[--c->len];
        if (first == 0) { cst_oracle_ans_free(c); return NULL; }
        uint64_t st = first;
        while (c->len > 0) {
            st = (st << W) | c->bulk[--c->len];
            if (st >= ((uint64_t)1 << (S - W))) break;
        }
        c->state = st;
    }
    return c;
}

/* from_binary (stack.rs:341-360): seeds state = 1 */
API oans_t *cst_oracle_ans_from_binary(int W, int S, const uint32_t *words, size_t n)
{
    oans_t *c = cst_oracle_ans_new(W, S);
    for (size_t i = 0; i < n; i++) bulk_push(c, words[i]);
    uint64_t st = 1;
    while (st < ((uint64_t)1 << (S - W))) {
        if (c->len == 0) break;
        st = (st << W) | c->bulk[--c->len];
    }
    c->state = st;
    return c;
}

/* number of words the state serialises to: bit_array_to_chunks_truncated (lib.rs:719-731) */
static inline int state_chunks(uint64_t state, int W) { return (bit_len(state) + W - 1) / W; }

API size_t cst_oracle_ans_num_words(const oans_t *c) { return c->len + (size_t)state_chunks(c->state, c->W); } /* stack.rs:609-614 */
API size_t cst_oracle_ans_num_valid_bits(const oans_t *c) /* stack.rs:623-630 */
{
    int b = bit_len(c->state);
    return (size_t)c->W * c->len + (size_t)(b > 1 ? b : 1) - 1;
}

/* get_compressed / into_compressed (stack.rs:537-547, 891-895, guard 1164-1195):
 * bulk ++ state words, least significant first, zero high words dropped. */
API size_t cst_oracle_ans_get_compressed(const oans_t *c, uint32_t *out)
{
    size_t n = c->len;
    if (out) memcpy(out, c->bulk, n * sizeof(uint32_t));
    int k = state_chunks(c->state, c->W);
    for (int i = 0; i < k; i++) {
        if (out) out[n] = (uint32_t)((c->state >> (i * c->W)) & word_mask(c->W));
        n++;
    }
    return n;
}

/* encode_symbol (stack.rs:1014-1048) given the model's (left cumulative, probability) */
API void cst_oracle_ans_encode_cp(oans_t *c, uint32_t left, uint32_t prob, int P)
{
    uint64_t st = c->state;
    if ((st >> (c->S - P)) >= prob) {
        bulk_push(c, (uint32_t)(st & word_mask(c->W)));
        st >>= c->W;
    }
    uint64_t rem = st % prob, prefix = st / prob;
    c->state = (prefix << P) | ((uint64_t)left + rem);
}

/* decode_symbol part 1 (stack.rs:1084): the quantile to look up */
API uint32_t cst_oracle_ans_peek_quantile(const oans_t *c, int P) { return (uint32_t)(c->state & (((uint64_t)1 << P) - 1)); }

/* decode_symbol part 2 (stack.rs:1086-1097) given the model's answer for that quantile */
API void cst_oracle_ans_decode_advance(oans_t *c, uint32_t left, uint32_t prob, int P)
{
    uint64_t q = c->state & (((uint64_t)1 << P) - 1);
    uint64_t st = (c->state >> P) * prob + (q - left);
    if (c->S < 64) st &= (((uint64_t)1 << c->S) - 1);
    if (st < ((uint64_t)1 << (c->S - c->W))) {
        if (c->len > 0) st = (st << c->W) | c->bulk[--c->len];
    }
    c->state = st;
}

/* Pos/Seek (stack.rs:1107-1139, backends.rs:537-555: seek on Vec = truncate) */
API int cst_oracle_ans_seek(oans_t *c, size_t pos, uint64_t state)
{
    if (pos > c->len) return 1;
    c->len = pos; c->state = state;
    return 0;
}

/* --- whole-array loops (src/stream/mod.rs:592-606, 1284-1291; stack.rs:784-849) --- */

/* encode_iid_symbols_reverse with a tabulated model: symbol -> index = sym - lo,
 * (c,p) = (cdf[i], cdf[i+1]-cdf[i]) (contiguous.rs:673-700).  Returns 0, or 1 + index of
 * the first impossible symbol *in encoding order* (nothing after it is encoded; what was
 * encoded before stays, as in the reference's loop). */
API int64_t cst_oracle_ans_encode_iid_table_reverse(oans_t *c, const int32_t *sym, size_t n, int32_t lo,
                                                    const uint32_t *cdf, int n_sym, int P)
{
    uint32_t pmask = P == 32 ? 0xffffffffu : ((1u << P) - 1u);
    (void)pmask;
    for (size_t t = n; t-- > 0;) {
        int64_t i = (int64_t)sym[t] - lo;
        if (i < 0 || i >= n_sym) return (int64_t)t + 1;
        cst_oracle_ans_encode_cp(c, cdf[i], cdf[i + 1] - cdf[i], P);
    }
    return 0;
}

API void cst_oracle_ans_decode_iid_table(oans_t *c, int32_t *sym, size_t n, int32_t lo, const uint32_t *cdf,
                                         int n_sym, int P)
{
    for (size_t t = 0; t < n; t++) {
        uint32_t q = cst_oracle_ans_peek_quantile(c, P);
        int a = 0, b = n_sym - 1; /* largest i with cdf[i] <= q */
        while (a < b) { int m = a + (b - a + 1) / 2; if (cdf[m] <= q) a = m; else b = m - 1; }
        sym[t] = lo + a;
        cst_oracle_ans_decode_advance(c, cdf[a], cdf[a + 1] - cdf[a], P);
    }
}

/* encode_symbols_reverse with per-symbol quantized Gaussians: the Python call
 * encode_reverse(symbols, QuantizedGaussian(lo,hi), means, stds)
 * (src/pybindings/stream/stack.rs:567-588; model ctor src/pybindings/stream/model.rs:649-660) */
API int64_t cst_oracle_ans_encode_gaussian_reverse(oans_t *c, const int32_t *sym, size_t n, int32_t lo, int32_t hi,
                                                   const double *mu, const double *sigma, int iid, int P,
                                                   int prob_bits)
{
    for (size_t t = n; t-- > 0;) {
        uint32_t l, p;
        double m = iid ? mu[0] : mu[t], s = iid ? sigma[0] : sigma[t];
        int rc = cst_oracle_leaky_gaussian_lcp(sym[t], lo, hi, P, prob_bits, m, s, &l, &p);
        if (rc) return (int64_t)t + 1;
        cst_oracle_ans_encode_cp(c, l, p, P);
    }
    return 0;
}

API void cst_oracle_ans_decode_gaussian(oans_t *c, int32_t *sym, size_t n, int32_t lo, int32_t hi,
                                        const double *mu, const double *sigma, int iid, int P, int prob_bits)
{
    for (size_t t = 0; t < n; t++) {
        uint32_t l, p;
        double m = iid ? mu[0] : mu[t], s = iid ? sigma[0] : sigma[t];
        uint32_t q = cst_oracle_ans_peek_quantile(c, P);
        cst_oracle_leaky_gaussian_quantile(q, lo, hi, P, prob_bits, m, s, &sym[t], &l, &p);
        cst_oracle_ans_decode_advance(c, l, p, P);
    }
}

/* ------------------------------------------------------------------------------------------
 * Range coder (src/stream/queue.rs), generic over (W, S) at run time, S in {32, 64}
 * ---------------------------------------------------------------------------------------- */

typedef struct {
    uint64_t lower, range;
    uint32_t *bulk;
    size_t len, cap;
    int W, S;
    /* EncoderSituation (queue.rs:126-142): inverted_n == 0 <=> Normal */
    size_t inverted_n;
    uint32_t inverted_first;
} orc_enc_t;

static inline uint64_t smask(int S) { return S == 64 ? ~0ull : (((uint64_t)1 << S) - 1); }

static void rc_push(orc_enc_t *e, uint32_t w)
{
    if (e->len == e->cap) {
        e->cap = e->cap ? e->cap * 2 : 64;
        e->bulk = (uint32_t *)realloc(e->bulk, e->cap * sizeof(uint32_t));
    }
    e->bulk[e->len++] = w;
}

API orc_enc_t *cst_oracle_rc_encoder_new(int W, int S) /* queue.rs:96-104: lower 0, range MAX */
{
    orc_enc_t *e = (orc_enc_t *)calloc(1, sizeof(orc_enc_t));
    e->W = W; e->S = S; e->lower = 0; e->range = smask(S);
    return e;
}
API void cst_oracle_rc_encoder_free(orc_enc_t *e) { if (e) { free(e->bulk); free(e); } }

/* Pos for RangeEncoder (queue.rs:182-196): bulk.pos() + num_inverted, and the coder state (lower, range) */
API size_t cst_oracle_rc_encoder_pos(const orc_enc_t *e, uint64_t *lower, uint64_t *range)
{
    *lower = e->lower; *range = e->range;
    return e->len + e->inverted_n;
}

/* encode_symbol (queue.rs:612-705).  Returns 0, or 1 if range would become zero. */
API int cst_oracle_rc_encode_cp(orc_enc_t *e, uint32_t left, uint32_t prob, int P)
{
    const int W = e->W, S = e->S;
    const uint64_t M = smask(S), wmask = word_mask(W);
    uint64_t scale = e->range >> P;
    uint64_t new_range = (scale * prob) & M;
    if (new_range == 0) return 1;
    e->range = new_range;
    uint64_t new_lower = (e->lower + scale * left) & M;

    if (e->inverted_n) {
        if (((new_lower + e->range) & M) > new_lower) {
            /* transition inverted -> normal */
            uint32_t first_word, consecutive;
            if (new_lower < e->lower) { first_word = (uint32_t)((e->inverted_first + 1u) & wmask); consecutive = 0; }
            else { first_word = e->inverted_first; consecutive = (uint32_t)wmask; }
            rc_push(e, first_word);
            for (size_t i = 1; i < e->inverted_n; i++) rc_push(e, consecutive);
            e->inverted_n = 0;
        }
    }
    e->lower = new_lower;

    if (e->range < ((uint64_t)1 << (S - W))) {
        e->range = (e->range << W) & M;
        uint32_t lower_word = (uint32_t)((e->lower >> (S - W)) & wmask);
        e->lower = (e->lower << W) & M;
        if (e->inverted_n) {
            e->inverted_n += 1;
        } else if (((e->lower + e->range) & M) > e->lower) {
            rc_push(e, lower_word);
        } else {
            e->inverted_n = 1; e->inverted_first = lower_word;
        }
    }
    return 0;
}

/* get_compressed = bulk ++ iter_seal (queue.rs:458-523, 552-556) */
API size_t cst_oracle_rc_get_compressed(const orc_enc_t *e, uint32_t *out)
{
    const int W = e->W, S = e->S;
    const uint64_t M = smask(S), wmask = word_mask(W);
    size_t n = e->len;
    if (out) memcpy(out, e->bulk, n * sizeof(uint32_t));
    if (e->range == M) return n; /* nothing encoded yet: no seal words */
    uint64_t point = (e->lower + (((uint64_t)1 << (S - W)) - 1)) & M;
    if (e->inverted_n) {
        uint32_t first, cons;
        if (point >= e->lower) { first = e->inverted_first; cons = (uint32_t)wmask; }
        else { first = (uint32_t)((e->inverted_first + 1u) & wmask); cons = 0; }
        if (out) out[n] = first;
        n++;
        for (size_t i = 1; i < e->inverted_n; i++) { if (out) out[n] = cons; n++; }
    }
    uint32_t point_word = (uint32_t)((point >> (S - W)) & wmask);
    uint32_t upper_word = (uint32_t)((((e->lower + e->range) & M) >> (S - W)) & wmask);
    if (out) out[n] = point_word;
    n++;
    if (upper_word == point_word) { if (out) out[n] = 0; n++; }
    return n;
}

typedef struct {
    uint64_t lower, range, point;
    const uint32_t *words;
    size_t pos, n;
    int W, S;
} orc_dec_t;

/* from_compressed + read_point (queue.rs:776-790, 847-868).  `words` must outlive the decoder. */
API orc_dec_t *cst_oracle_rc_decoder_new(int W, int S, const uint32_t *words, size_t n)
{
    orc_dec_t *d = (orc_dec_t *)calloc(1, sizeof(orc_dec_t));
    d->W = W; d->S = S; d->words = words; d->n = n; d->pos = 0;
    d->lower = 0; d->range = smask(S);
    int num_read = 0;
    uint64_t point = 0;
    while (d->pos < n) {
        point = ((point << W) | words[d->pos++]) & smask(S);
        if (++num_read == S / W) break;
    }
    if (num_read < S / W && num_read != 0) point = (point << (S - num_read * W)) & smask(S);
    d->point = point;
    return d;
}
API void cst_oracle_rc_decoder_free(orc_dec_t *d) { free(d); }

/* decode_symbol part 1 (queue.rs:988-993): returns the quantile or 0xffffffff on InvalidData */
API uint32_t cst_oracle_rc_peek_quantile(const orc_dec_t *d, int P)
{
    uint64_t scale = d->range >> P;
    uint64_t q = ((d->point - d->lower) & smask(d->S)) / scale;
    if (q >= ((uint64_t)1 << P)) return 0xffffffffu;
    return (uint32_t)q;
}

/* decode_symbol part 2 (queue.rs:998-1030) */
API void cst_oracle_rc_decode_advance(orc_dec_t *d, uint32_t left, uint32_t prob, int P)
{
    const int W = d->W, S = d->S;
    const uint64_t M = smask(S);
    uint64_t scale = d->range >> P;
    d->lower = (d->lower + scale * left) & M;
    d->range = (scale * prob) & M;
    if (d->range < ((uint64_t)1 << (S - W))) {
        d->lower = (d->lower << W) & M;
        d->range = (d->range << W) & M;
        d->point = (d->point << W) & M;
        if (d->pos < d->n) d->point |= d->words[d->pos++];
    }
}

API int cst_oracle_rc_maybe_exhausted(const orc_dec_t *d) /* queue.rs:870-891 semantics, simplified */
{
    return d->pos >= d->n;
}

/* ------------------------------------------------------------------------------------------
 * Batched drivers: many independent streams with one shared table (config C2) or one table
 * per stream (config C3), multi-threaded over disjoint stream ranges.  These are the CPU
 * baseline that bench.py times ("kind": "port"); the inner loops are the specialised,
 * allocation-free form of the generic functions above for (W,S) = (32,64) and (16,32).
 * ---------------------------------------------------------------------------------------- */

typedef struct {
    int W, S, P;
    const int32_t *symbols;      /* [n_streams][n_per_stream], stream-major */
    int32_t *decoded;            /* same shape (decode) */
    size_t n_streams, n_per_stream;
    int32_t lo; int n_sym;
    const uint32_t *cdf;         /* [n_sym+1] shared, or [n_streams][n_sym+1] */
    const uint16_t *lookup;      /* [2^P] shared, or [n_streams][2^P]; may be NULL (bisection) */
    int per_stream_tables;
    uint32_t *words;             /* [n_streams][stride] slabs */
    size_t stride;
    uint32_t *n_words;           /* [n_streams] */
    int32_t *status;             /* [n_streams] */
    size_t s_begin, s_end;
    int thread_index;            /* -1: the caller's own thread */
    int32_t *sink;               /* decode diagnostics: write every stream's symbols HERE instead of into `decoded` */
} batch_job_t;

static void encode_one_stream(const batch_job_t *j, size_t s)
{
    const int W = j->W, S = j->S, P = j->P;
    const uint64_t wm = word_mask(W);
    const int32_t *x = j->symbols + s * j->n_per_stream;
    const uint32_t *cdf = j->per_stream_tables ? j->cdf + s * (size_t)(j->n_sym + 1) : j->cdf;
    uint32_t *out = j->words + s * j->stride;
    size_t len = 0;
    uint64_t st = 0;
    j->status[s] = 0;
    for (size_t t = j->n_per_stream; t-- > 0;) {
        int64_t i = (int64_t)x[t] - j->lo;
        if (i < 0 || i >= j->n_sym) { j->status[s] = 1; j->n_words[s] = 0; return; }
        uint32_t c = cdf[i], p = cdf[i + 1] - c;
        if ((st >> (S - P)) >= p) {
            if (len >= j->stride) { j->status[s] = 2; j->n_words[s] = 0; return; }
            out[len++] = (uint32_t)(st & wm);
            st >>= W;
        }
        st = ((st / p) << P) | (c + st % p);
    }
    int k = state_chunks(st, W);
    if (len + (size_t)k > j->stride) { j->status[s] = 2; j->n_words[s] = 0; return; }
    for (int i = 0; i < k; i++) out[len++] = (uint32_t)((st >> (i * W)) & wm);
    j->n_words[s] = (uint32_t)len;
}

static void decode_one_stream(const batch_job_t *j, size_t s)
{
    const int W = j->W, S = j->S, P = j->P;
    const uint32_t *cdf = j->per_stream_tables ? j->cdf + s * (size_t)(j->n_sym + 1) : j->cdf;
    const uint16_t *lut = j->lookup ? (j->per_stream_tables ? j->lookup + (s << P) : j->lookup) : NULL;
    const uint32_t *in = j->words + s * j->stride;
    size_t len = j->n_words[s];
    int32_t *y = j->decoded + s * j->n_per_stream;
    if (j->sink) y = j->sink;                  /* (scaling diagnostics only: every stream of a thread into the same 16 KiB) */
    uint64_t st = 0;
    j->status[s] = 0;
    if (len > 0) {
        uint32_t first = in[--len];
        if (first == 0) { j->status[s] = 3; return; }
        st = first;
        while (len > 0) { st = (st << W) | in[--len]; if (st >= ((uint64_t)1 << (S - W))) break; }
    }
    const uint64_t qmask = ((uint64_t)1 << P) - 1, thresh = (uint64_t)1 << (S - W);
    for (size_t t = 0; t < j->n_per_stream; t++) {
        uint32_t q = (uint32_t)(st & qmask);
        int a;
        if (lut) a = lut[q];
        else { int lo_ = 0, hi_ = j->n_sym - 1; while (lo_ < hi_) { int m = lo_ + (hi_ - lo_ + 1) / 2; if (cdf[m] <= q) lo_ = m; else hi_ = m - 1; } a = lo_; }
        uint32_t c = cdf[a], p = cdf[a + 1] - c;
        y[t] = j->lo + a;
        st = (st >> P) * p + (q - c);
        if (st < thresh && len > 0) st = (st << W) | in[--len];
    }
}

static void pin_self(int i);
static void *encode_worker(void *arg) { batch_job_t *j = (batch_job_t *)arg; pin_self(j->thread_index); for (size_t s = j->s_begin; s < j->s_end; s++) encode_one_stream(j, s); return NULL; }
static void *decode_worker(void *arg)
{
    batch_job_t *j = (batch_job_t *)arg;
    pin_self(j->thread_index);
    /* CST_ORACLE_DECODE_SINK=1 (scripts/cpu_scaling.py): the decoder WITHOUT its gigabyte of output -- what is left is the coder */
    const char *e = getenv("CST_ORACLE_DECODE_SINK");
    if (e && *e == '1') j->sink = (int32_t *)malloc(sizeof(int32_t) * (j->n_per_stream ? j->n_per_stream : 1));
    for (size_t s = j->s_begin; s < j->s_end; s++) decode_one_stream(j, s);
    if (j->sink) { free(j->sink); j->sink = NULL; }
    return NULL;
}

/* the box's store bandwidth with the same threads and pinning: every thread fills its own block (scripts/cpu_scaling.py) */
typedef struct { unsigned char *p; size_t n; int idx; } fill_job_t;
static void *fill_worker(void *arg) { fill_job_t *f = (fill_job_t *)arg; pin_self(f->idx); memset(f->p, f->idx & 0xff, f->n); return NULL; }
API void cst_oracle_fill_threads(unsigned char *dst, size_t bytes, int n_threads)
{
    if (n_threads < 1) n_threads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * n_threads);
    fill_job_t *jobs = (fill_job_t *)malloc(sizeof(fill_job_t) * n_threads);
    size_t per = (bytes / n_threads) & ~(size_t)4095;
    for (int i = 0; i < n_threads; i++) {
        jobs[i].p = dst + (size_t)i * per; jobs[i].n = per; jobs[i].idx = n_threads == 1 ? -1 : i;
        if (n_threads == 1) fill_worker(&jobs[i]); else pthread_create(&th[i], NULL, fill_worker, &jobs[i]);
    }
    if (n_threads > 1) for (int i = 0; i < n_threads; i++) pthread_join(th[i], NULL);
    free(th); free(jobs);
}

/* CST_ORACLE_PIN=1 (bench.py's cpu_baseline leg): thread i stays on logical CPU i mod (online CPUs), so that a thread keeps the
 * pages it touched first on its own NUMA node and is not migrated in the middle of a timed pass. */
static void pin_self(int i)
{
    const char *e = getenv("CST_ORACLE_PIN");
    if (!e || *e != '1' || i < 0) return;
    long n = sysconf(_SC_NPROCESSORS_ONLN);
    if (n < 1) return;
    cpu_set_t set; CPU_ZERO(&set); CPU_SET((int)(i % n), &set);
    (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set);
}

static void run_threads(batch_job_t *proto, int n_threads, void *(*fn)(void *))
{
    if (n_threads < 1) n_threads = 1;
    if ((size_t)n_threads > proto->n_streams) n_threads = proto->n_streams ? (int)proto->n_streams : 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * n_threads);
    batch_job_t *jobs = (batch_job_t *)malloc(sizeof(batch_job_t) * n_threads);
    size_t per = (proto->n_streams + n_threads - 1) / n_threads;
    for (int i = 0; i < n_threads; i++) {
        jobs[i] = *proto;
        jobs[i].s_begin = (size_t)i * per < proto->n_streams ? (size_t)i * per : proto->n_streams;
        jobs[i].s_end = jobs[i].s_begin + per < proto->n_streams ? jobs[i].s_begin + per : proto->n_streams;
        jobs[i].thread_index = n_threads == 1 ? -1 : i;
        if (n_threads == 1) fn(&jobs[i]); else pthread_create(&th[i], NULL, fn, &jobs[i]);
    }
    if (n_threads > 1) for (int i = 0; i < n_threads; i++) pthread_join(th[i], NULL);
    free(th); free(jobs);
}

API void cst_oracle_ans_encode_batch(int W, int S, int P, const int32_t *symbols, size_t n_streams,
                                     size_t n_per_stream, int32_t lo, int n_sym, const uint32_t *cdf,
                                     int per_stream_tables, uint32_t *words, size_t stride, uint32_t *n_words,
                                     int32_t *status, int n_threads)
{
    batch_job_t j; memset(&j, 0, sizeof j);
    j.W = W; j.S = S; j.P = P; j.symbols = symbols; j.n_streams = n_streams; j.n_per_stream = n_per_stream;
    j.lo = lo; j.n_sym = n_sym; j.cdf = cdf; j.per_stream_tables = per_stream_tables; j.words = words;
    j.stride = stride; j.n_words = n_words; j.status = status;
    run_threads(&j, n_threads, encode_worker);
}

API void cst_oracle_ans_decode_batch(int W, int S, int P, int32_t *decoded, size_t n_streams, size_t n_per_stream,
                                     int32_t lo, int n_sym, const uint32_t *cdf, const uint16_t *lookup,
                                     int per_stream_tables, const uint32_t *words, size_t stride,
                                     const uint32_t *n_words, int32_t *status, int n_threads)
{
    batch_job_t j; memset(&j, 0, sizeof j);
    j.W = W; j.S = S; j.P = P; j.decoded = decoded; j.n_streams = n_streams; j.n_per_stream = n_per_stream;
    j.lo = lo; j.n_sym = n_sym; j.cdf = cdf; j.lookup = lookup; j.per_stream_tables = per_stream_tables;
    j.words = (uint32_t *)words; j.stride = stride; j.n_words = (uint32_t *)n_words; j.status = status;
    run_threads(&j, n_threads, decode_worker);
}

/* splitmix64 (the benchmark's own generator; no reference code) and the synthetic symbol recipe of
 * q = splitmix64(seed ^ stream).next() >> (64-P); sym = quantile(q). */
static inline uint64_t splitmix64_next(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

API void cst_oracle_synth_symbols(uint64_t seed, size_t stream_begin, size_t n_streams, size_t n_per_stream,
                                  int32_t lo, int n_sym, const uint32_t *cdf, int per_stream_tables, int P,
                                  int32_t *symbols)
{
    for (size_t s = 0; s < n_streams; s++) {
        uint64_t st = seed ^ (uint64_t)(stream_begin + s);
        const uint32_t *c = per_stream_tables ? cdf + s * (size_t)(n_sym + 1) : cdf;
        for (size_t t = 0; t < n_per_stream; t++) {
            uint32_t q = (uint32_t)(splitmix64_next(&st) >> (64 - P));
            int a = 0, b = n_sym - 1;
            while (a < b) { int m = a + (b - a + 1) / 2; if (c[m] <= q) a = m; else b = m - 1; }
            symbols[s * n_per_stream + t] = lo + a;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Batched range coder drivers (config C4): one RangeEncoder / RangeDecoder per stream, shared table.
 * ---------------------------------------------------------------------------------------- */
API void cst_oracle_rc_encode_batch(int W, int S, int P, const int32_t *symbols, size_t n_streams, size_t n_per_stream,
                                    int32_t lo, int n_sym, const uint32_t *cdf, uint32_t *words, size_t stride,
                                    uint32_t *n_words, int32_t *status)
{
    for (size_t s = 0; s < n_streams; s++) {
        orc_enc_t *e = cst_oracle_rc_encoder_new(W, S);
        status[s] = 0;
        for (size_t t = 0; t < n_per_stream; t++) {
            int64_t i = (int64_t)symbols[s * n_per_stream + t] - lo;
            if (i < 0 || i >= n_sym) { status[s] = 1; break; }
            cst_oracle_rc_encode_cp(e, cdf[i], cdf[i + 1] - cdf[i], P);
        }
        size_t n = cst_oracle_rc_get_compressed(e, NULL);
        if (status[s] == 0 && n > stride) status[s] = 2;
        if (status[s] == 0) { cst_oracle_rc_get_compressed(e, words + s * stride); n_words[s] = (uint32_t)n; }
        else n_words[s] = 0;
        cst_oracle_rc_encoder_free(e);
    }
}

API void cst_oracle_rc_decode_batch(int W, int S, int P, int32_t *decoded, size_t n_streams, size_t n_per_stream, int32_t lo,
                                    int n_sym, const uint32_t *cdf, const uint32_t *words, size_t stride,
                                    const uint32_t *n_words, int32_t *status)
{
    for (size_t s = 0; s < n_streams; s++) {
        orc_dec_t *d = cst_oracle_rc_decoder_new(W, S, words + s * stride, n_words[s]);
        status[s] = 0;
        for (size_t t = 0; t < n_per_stream; t++) {
            uint32_t q = cst_oracle_rc_peek_quantile(d, P);
            if (q == 0xffffffffu) { status[s] = 3; break; }
            int a = 0, b = n_sym - 1;
            while (a < b) { int m = a + (b - a + 1) / 2; if (cdf[m] <= q) a = m; else b = m - 1; }
            decoded[s * n_per_stream + t] = lo + a;
            cst_oracle_rc_decode_advance(d, cdf[a], cdf[a + 1] - cdf[a], P);
        }
        cst_oracle_rc_decoder_free(d);
    }
}

/* ------------------------------------------------------------------------------------------
 * Jump tables: the reference's Pos / Seek for both coders.
 *   AnsCoder::pos() = (bulk.len(), state)                      stack.rs:1130-1139;  seek: truncate the bulk to `pos`,
 *                                                              set the state                         stack.rs:1117-1128
 *   RangeEncoder::pos() = (bulk.len() + num_inverted, state)   queue.rs:182-196;    RangeDecoder::seek: continue reading at
 *                                                              `pos`, read_point, set (lower, range) queue.rs:911-926
 * These drivers note a jump point in front of every chunk of `interval` symbols (chunk j = symbols [j * interval, ...)),
 * the way the reference's own tests build their jump tables (stack.rs:1470-1500, queue.rs:1345-1356), and decode single
 * chunks from them.  Shared table or one table per stream, like the batched drivers above.
 * ---------------------------------------------------------------------------------------- */
API void cst_oracle_ans_jump_table(int W, int S, int P, const int32_t *symbols, size_t n_streams, size_t n_per_stream, int32_t lo,
                                   int n_sym, const uint32_t *cdf, int per_stream_tables, size_t interval, uint32_t *pos,
                                   uint64_t *state)
{
    const size_t n_chunks = (n_per_stream + interval - 1) / interval;
    for (size_t s = 0; s < n_streams; s++) {
        const uint32_t *row = per_stream_tables ? cdf + s * (size_t)(n_sym + 1) : cdf;
        const int32_t *x = symbols + s * n_per_stream;
        size_t len = 0;
        uint64_t st = 0;
        for (size_t t = n_per_stream; t-- > 0;) {           /* encode_symbol, stack.rs:1014-1048 (symbols last to first) */
            const int64_t i = (int64_t)x[t] - lo;
            if (i < 0 || i >= n_sym) break;                  /* an impossible symbol: the stream's remaining jump points stay as they are */
            const uint32_t c = row[i], p = row[i + 1] - c;
            if ((st >> (S - P)) >= p) { len++; st >>= W; }
            st = ((st / p) << P) | (c + st % p);
            if (t % interval == 0) { pos[s * n_chunks + t / interval] = (uint32_t)len; state[s * n_chunks + t / interval] = st; }
        }
    }
}

/* AnsCoder::seek(pos, state) on the words of one stream, then `n` symbols (stack.rs:1117-1128, 1070-1100) */
API void cst_oracle_ans_decode_from(int W, int S, int P, const uint32_t *words, size_t pos, uint64_t state, int32_t *decoded, size_t n,
                                    int32_t lo, int n_sym, const uint32_t *cdf)
{
    size_t len = pos;
    uint64_t st = state;
    const uint64_t qmask = ((uint64_t)1 << P) - 1, thresh = (uint64_t)1 << (S - W);
    for (size_t t = 0; t < n; t++) {
        const uint32_t q = (uint32_t)(st & qmask);
        int a = 0, b = n_sym - 1;
        while (a < b) { const int m = a + (b - a + 1) / 2; if (cdf[m] <= q) a = m; else b = m - 1; }
        decoded[t] = lo + a;
        st = (st >> P) * (cdf[a + 1] - cdf[a]) + (q - cdf[a]);
        if (st < thresh && len > 0) st = (st << W) | words[--len];
    }
}

API void cst_oracle_rc_jump_table(int W, int S, int P, const int32_t *symbols, size_t n_streams, size_t n_per_stream, int32_t lo,
                                  int n_sym, const uint32_t *cdf, size_t interval, uint32_t *pos, uint64_t *lower, uint64_t *range)
{
    const size_t n_chunks = (n_per_stream + interval - 1) / interval;
    (void)n_sym;                                              /* (symbols are taken as valid: the callers encoded them before) */
    for (size_t s = 0; s < n_streams; s++) {
        orc_enc_t *e = cst_oracle_rc_encoder_new(W, S);
        for (size_t t = 0; t < n_per_stream; t++) {
            if (t % interval == 0) {                          /* RangeEncoder::pos() in front of the chunk, queue.rs:188-195 */
                const size_t k = s * n_chunks + t / interval;
                pos[k] = (uint32_t)(e->len + e->inverted_n); lower[k] = e->lower; range[k] = e->range;
            }
            const int64_t i = (int64_t)symbols[s * n_per_stream + t] - lo;
            cst_oracle_rc_encode_cp(e, cdf[i], cdf[i + 1] - cdf[i], P);
        }
        cst_oracle_rc_encoder_free(e);
    }
}

/* RangeDecoder::seek((pos, (lower, range))) on the words of one stream, then `n` symbols (queue.rs:911-926, 968-1033).
 * Returns 0, or 3 on DecoderFrontendError::InvalidData. */
API int cst_oracle_rc_decode_from(int W, int S, int P, const uint32_t *words, size_t n_words, size_t pos, uint64_t lower, uint64_t range,
                                  int32_t *decoded, size_t n, int32_t lo, int n_sym, const uint32_t *cdf)
{
    orc_dec_t *d = cst_oracle_rc_decoder_new(W, S, words + (pos < n_words ? pos : n_words), pos < n_words ? n_words - pos : 0);   /* bulk.seek(pos) + read_point */
    d->lower = lower; d->range = range;
    int status = 0;
    for (size_t t = 0; t < n; t++) {
        const uint32_t q = cst_oracle_rc_peek_quantile(d, P);
        if (q == 0xffffffffu) { status = 3; break; }
        int a = 0, b = n_sym - 1;
        while (a < b) { const int m = a + (b - a + 1) / 2; if (cdf[m] <= q) a = m; else b = m - 1; }
        decoded[t] = lo + a;
        cst_oracle_rc_decode_advance(d, cdf[a], cdf[a + 1] - cdf[a], P);
    }
    cst_oracle_rc_decoder_free(d);
    return status;
}
