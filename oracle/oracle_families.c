/*
 * oracle_families.c -- CPU restatement of the model families of SURVEY 8(f) row 2 that are NOT the quantized
 * Gaussian: perfectly quantized categorical tables, the lazy categorical model, and the LeakyQuantizer over the
 * Laplace / Cauchy / Binomial distributions of the `probability` crate.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE (see the header of oracle.c): only tests/, smoke() and bench.py's
 * cpu_baseline leg may load it.  It is the INDEPENDENT second implementation the product's table builders
 * (constriction_amd/csrc/cst_families.hip) are compared with.
 *
 * What pins what:
 *   - lazy categorical: src/stream/model/categorical/lazy_contiguous.rs:228-331 is restated line by line and pinned by
 *     the reference's own vectors (tests/python/test_lazy_f32.py, test_lazy_f64.py -> tests/golden/lazy_vectors.json).
 *   - perfect categorical: src/stream/model/categorical.rs:56-177 restated; `log1p` is the reference's EXPLICIT
 *     dependency `libm::log1p` (libm 0.2.16, Cargo.lock; == musl log1p.c), restated below.  Pinned (round 4) by the
 *     known answers of the reference's own unit tests (contiguous.rs:709-731 the 37-entry fixed point at P = 32,
 *     :735-833 KL(perfect) < KL(fast) for f64 and f32 inputs, :836-873 the issue-#20 inputs): tests/golden/
 *     perfect_categorical.json, tests/test_model_families_cpu.py::test_perfect_quantisation_reference_known_answers.
 *     The reference holds no COMPRESSED vector coded with a perfect table.
 *   - Laplace / Cauchy / Binomial: the distributions live in the un-vendored crate probability 0.20.3 (-> special
 *     0.10.3 -> libm 0.2.16); its published formulas are restated (Laplace::distribution, Cauchy::distribution,
 *     Binomial::distribution = regularised incomplete beta by Algorithm AS 63 + ln_beta from libm::lgamma_r).  The
 *     crate calls f64::exp / ln / atan of the Rust standard library, i.e. the platform's libm; this file fixes them
 *     to the libm-crate (musl / FreeBSD msun) algorithms, the only arithmetic the lock file pins.  The reference
 *     tests hold no compressed vector for these families (round trips only): PARITY UNPINNED, the last ulp of every
 *     table entry is DEFINED by this file.
 *
 * Plain C99, compile with -ffp-contract=off.
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define API __attribute__((visibility("default")))

double cst_oracle_exp(double x); /* oracle.c: musl exp.c == libm 0.2 `exp` */

static inline uint64_t f64_bits(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline double bits_f64(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }
static inline uint32_t hi_word(double x) { return (uint32_t)(f64_bits(x) >> 32); }
static inline uint32_t lo_word(double x) { return (uint32_t)f64_bits(x); }
static inline double with_hi(double x, uint32_t hi) { return bits_f64(((uint64_t)hi << 32) | (f64_bits(x) & 0xffffffffull)); }
static double o_fabs(double x) { return bits_f64(f64_bits(x) & 0x7fffffffffffffffull); }

static const double
ln2_hi = 6.93147180369123816490e-01, /* 3fe62e42 fee00000 */
ln2_lo = 1.90821492927058770002e-10, /* 3dea39ef 35793c76 */
Lg1 = 6.666666666666735130e-01,      /* 3FE55555 55555593 */
Lg2 = 3.999999999940941908e-01,      /* 3FD99999 9997FA04 */
Lg3 = 2.857142874366239149e-01,      /* 3FD24924 94229359 */
Lg4 = 2.222219843214978396e-01,      /* 3FCC71C5 1D8E78AF */
Lg5 = 1.818357216161805012e-01,      /* 3FC74664 96CB03DE */
Lg6 = 1.531383769920937332e-01,      /* 3FC39A09 D078C69F */
Lg7 = 1.479819860511658591e-01;      /* 3FC2F112 DF3E5244 */

/* log(x): FreeBSD e_log.c as arranged by musl (<= 1.1.19) == libm 0.2 `log` */
API double cst_oracle_log(double x)
{
    double hfsq, f, s, z, R, w, t1, t2, dk;
    uint32_t hx = hi_word(x);
    int k = 0;
    if (hx < 0x00100000 || hx >> 31) {
        if ((f64_bits(x) << 1) == 0) return -1 / (x * x);
        if (hx >> 31) return (x - x) / 0.0;
        k -= 54;
        x *= 0x1p54;
        hx = hi_word(x);
    } else if (hx >= 0x7ff00000) {
        return x;
    } else if (hx == 0x3ff00000 && lo_word(x) == 0)
        return 0;
    hx += 0x3ff00000 - 0x3fe6a09e;
    k += (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffff) + 0x3fe6a09e;
    x = with_hi(x, hx);
    f = x - 1.0;
    hfsq = 0.5 * f * f;
    s = f / (2.0 + f);
    z = s * s;
    w = z * z;
    t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    R = t2 + t1;
    dk = k;
    return s * (hfsq + R) + dk * ln2_lo - hfsq + f + dk * ln2_hi;
}

/* log1p(x): musl log1p.c (origin FreeBSD s_log1p.c) == libm 0.2 `log1p`, the function categorical.rs:11 imports */
API double cst_oracle_log1p(double x)
{
    double hfsq, f = 0, c = 0, s, z, R, w, t1, t2, dk, uf;
    uint32_t hx = hi_word(x), hu;
    int k = 1;
    if (hx < 0x3fda827a || hx >> 31) { /* 1+x < sqrt(2)+ */
        if (hx >= 0xbff00000) {        /* x <= -1.0 */
            if (x == -1) return x / 0.0;
            return (x - x) / 0.0;
        }
        if (hx << 1 < 0x3ca00000u << 1) /* |x| < 2**-53 */
            return x;
        if (hx <= 0xbfd2bec4) { /* sqrt(2)/2- <= 1+x < sqrt(2)+ */
            k = 0;
            c = 0;
            f = x;
        }
    } else if (hx >= 0x7ff00000)
        return x;
    if (k) {
        uf = 1 + x;
        hu = hi_word(uf);
        hu += 0x3ff00000 - 0x3fe6a09e;
        k = (int)(hu >> 20) - 0x3ff;
        if (k < 54) { /* correction term ~ log(1+x)-log(u) */
            c = k >= 2 ? 1 - (uf - x) : x - (uf - 1);
            c /= uf;
        } else
            c = 0;
        hu = (hu & 0x000fffff) + 0x3fe6a09e;
        uf = with_hi(uf, hu);
        f = uf - 1;
    }
    hfsq = 0.5 * f * f;
    s = f / (2.0 + f);
    z = s * s;
    w = z * z;
    t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    R = t2 + t1;
    dk = k;
    return s * (hfsq + R) + (dk * ln2_lo + c) - hfsq + f + dk * ln2_hi;
}

/* atan(x): musl atan.c (origin FreeBSD s_atan.c) == libm 0.2 `atan` */
API double cst_oracle_atan(double x)
{
    static const double atanhi[] = {
        4.63647609000806093515e-01, /* atan(0.5)hi 0x3FDDAC67, 0x0561BB4F */
        7.85398163397448278999e-01, /* atan(1.0)hi 0x3FE921FB, 0x54442D18 */
        9.82793723247329054082e-01, /* atan(1.5)hi 0x3FEF730B, 0xD281F69B */
        1.57079632679489655800e+00, /* atan(inf)hi 0x3FF921FB, 0x54442D18 */
    };
    static const double atanlo[] = {
        2.26987774529616870924e-17, /* atan(0.5)lo 0x3C7A2B7F, 0x222F65E2 */
        3.06161699786838301793e-17, /* atan(1.0)lo 0x3C81A626, 0x33145C07 */
        1.39033110312309984516e-17, /* atan(1.5)lo 0x3C700788, 0x7AF0CBBD */
        6.12323399573676603587e-17, /* atan(inf)lo 0x3C91A626, 0x33145C07 */
    };
    static const double aT[] = {
        3.33333333333329318027e-01,  /* 0x3FD55555, 0x5555550D */
        -1.99999999998764832476e-01, /* 0xBFC99999, 0x9998EBC4 */
        1.42857142725034663711e-01,  /* 0x3FC24924, 0x920083FF */
        -1.11111104054623557880e-01, /* 0xBFBC71C6, 0xFE231671 */
        9.09088713343650656196e-02,  /* 0x3FB745CD, 0xC54C206E */
        -7.69187620504482999495e-02, /* 0xBFB3B0F2, 0xAF749A6D */
        6.66107313738753120669e-02,  /* 0x3FB10D66, 0xA0D03D51 */
        -5.83357013379057348645e-02, /* 0xBFADDE2D, 0x52DEFD9A */
        4.97687799461593236017e-02,  /* 0x3FA97B4B, 0x24760DEB */
        -3.65315727442169155270e-02, /* 0xBFA2B444, 0x2C6A6C2F */
        1.62858201153657823623e-02,  /* 0x3F90AD3A, 0xE322DA11 */
    };
    double w, s1, s2, z;
    uint32_t ix = hi_word(x), sign = ix >> 31;
    int id;
    ix &= 0x7fffffff;
    if (ix >= 0x44100000) { /* |x| >= 2^66 */
        if (x != x) return x;
        z = atanhi[3] + 0x1p-120f;
        return sign ? -z : z;
    }
    if (ix < 0x3fdc0000) {     /* |x| < 0.4375 */
        if (ix < 0x3e400000)   /* |x| < 2^-27 */
            return x;
        id = -1;
    } else {
        x = o_fabs(x);
        if (ix < 0x3ff30000) {     /* |x| < 1.1875 */
            if (ix < 0x3fe60000) { /* 7/16 <= |x| < 11/16 */
                id = 0;
                x = (2.0 * x - 1.0) / (2.0 + x);
            } else { /* 11/16 <= |x| < 19/16 */
                id = 1;
                x = (x - 1.0) / (x + 1.0);
            }
        } else {
            if (ix < 0x40038000) { /* |x| < 2.4375 */
                id = 2;
                x = (x - 1.5) / (1.0 + 1.5 * x);
            } else { /* 2.4375 <= |x| < 2^66 */
                id = 3;
                x = -1.0 / x;
            }
        }
    }
    z = x * x;
    w = z * z;
    s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    z = atanhi[id] - (x * (s1 + s2) - atanlo[id] - x);
    return sign ? -z : z;
}

/* lgamma_r(x) for x > 0: musl lgamma_r.c (origin FreeBSD e_lgamma_r.c) == libm 0.2 `lgamma_r`; reached from the
 * reference through special::Gamma::ln_gamma (Binomial::distribution -> ln_beta).  Negative arguments never occur
 * on that path (ln_beta of positive counts) and return NaN here. */
static const double
a0 = 7.72156649015328655494e-02,   /* 0x3FB3C467, 0xE37DB0C8 */
a1 = 3.22467033424113591611e-01,   /* 0x3FD4A34C, 0xC4A60FAD */
a2 = 6.73523010531292681824e-02,   /* 0x3FB13E00, 0x1A5562A7 */
a3 = 2.05808084325167332806e-02,   /* 0x3F951322, 0xAC92547B */
a4 = 7.38555086081402883957e-03,   /* 0x3F7E404F, 0xB68FEFE8 */
a5 = 2.89051383673415629091e-03,   /* 0x3F67ADD8, 0xCCB7926B */
a6 = 1.19270763183362067845e-03,   /* 0x3F538A94, 0x116F3F5D */
a7 = 5.10069792153511336608e-04,   /* 0x3F40B6C6, 0x89B99C00 */
a8 = 2.20862790713908385557e-04,   /* 0x3F2CF2EC, 0xED10E54D */
a9 = 1.08011567247583939954e-04,   /* 0x3F1C5088, 0x987DFB07 */
a10 = 2.52144565451257326939e-05,  /* 0x3EFA7074, 0x428CFA52 */
a11 = 4.48640949618915160150e-05,  /* 0x3F07858E, 0x90A45837 */
tc = 1.46163214496836224576e+00,   /* 0x3FF762D8, 0x6356BE3F */
tf = -1.21486290535849611461e-01,  /* 0xBFBF19B9, 0xBCC38A42 */
tt = -3.63867699703950536541e-18,  /* 0xBC50C7CA, 0xA48A971F */
t0 = 4.83836122723810047042e-01,   /* 0x3FDEF72B, 0xC8EE38A2 */
t1_ = -1.47587722994593911752e-01, /* 0xBFC2E427, 0x8DC6C509 */
t2_ = 6.46249402391333854778e-02,  /* 0x3FB08B42, 0x94D5419B */
t3 = -3.27885410759859649565e-02,  /* 0xBFA0C9A8, 0xDF35B713 */
t4 = 1.79706750811820387126e-02,   /* 0x3F9266E7, 0x970AF9EC */
t5 = -1.03142241298341437450e-02,  /* 0xBF851F9F, 0xBA91EC6A */
t6 = 6.10053870246291332635e-03,   /* 0x3F78FCE0, 0xE370E344 */
t7 = -3.68452016781138256760e-03,  /* 0xBF6E2EFF, 0xB3E914D7 */
t8 = 2.25964780900612472250e-03,   /* 0x3F6282D3, 0x2E15C915 */
t9 = -1.40346469989232843813e-03,  /* 0xBF56FE8E, 0xBF2D1AF1 */
t10 = 8.81081882437654011382e-04,  /* 0x3F4CDF0C, 0xEF61A8E9 */
t11 = -5.38595305356740546715e-04, /* 0xBF41A610, 0x9C73E0EC */
t12 = 3.15632070903625950361e-04,  /* 0x3F34AF6D, 0x6C0EBBF7 */
t13 = -3.12754168375120860518e-04, /* 0xBF347F24, 0xECC38C38 */
t14 = 3.35529192635519073543e-04,  /* 0x3F35FD3E, 0xE8C2D3F4 */
u0 = -7.72156649015328655494e-02,  /* 0xBFB3C467, 0xE37DB0C8 */
u1 = 6.32827064025093366517e-01,   /* 0x3FE4401E, 0x8B005DFF */
u2 = 1.45492250137234768737e+00,   /* 0x3FF7475C, 0xD119BD6F */
u3 = 9.77717527963372745603e-01,   /* 0x3FEF4976, 0x44EA8450 */
u4 = 2.28963728064692451092e-01,   /* 0x3FCD4EAE, 0xF6010924 */
u5 = 1.33810918536787660377e-02,   /* 0x3F8B678B, 0xBF2BAB09 */
v1 = 2.45597793713041134822e+00,   /* 0x4003A5D7, 0xC2BD619C */
v2 = 2.12848976379893395361e+00,   /* 0x40010725, 0xA42B18F5 */
v3 = 7.69285150456672783825e-01,   /* 0x3FE89DFB, 0xE45050AF */
v4 = 1.04222645593369134254e-01,   /* 0x3FBAAE55, 0xD6537C88 */
v5 = 3.21709242282423911810e-03,   /* 0x3F6A5ABB, 0x57D0CF61 */
s0 = -7.72156649015328655494e-02,  /* 0xBFB3C467, 0xE37DB0C8 */
s1_ = 2.14982415960608852501e-01,  /* 0x3FCB848B, 0x36E20878 */
s2_ = 3.25778796408930981787e-01,  /* 0x3FD4D98F, 0x4F139F59 */
s3 = 1.46350472652464452805e-01,   /* 0x3FC2BB9C, 0xBEE5F2F7 */
s4 = 2.66422703033638609560e-02,   /* 0x3F9B481C, 0x7E939961 */
s5 = 1.84028451407337715652e-03,   /* 0x3F5E26B6, 0x7368F239 */
s6 = 3.19475326584100867617e-05,   /* 0x3F00BFEC, 0xDD17E945 */
r1 = 1.39200533467621045958e+00,   /* 0x3FF645A7, 0x62C4AB74 */
r2 = 7.21935547567138069525e-01,   /* 0x3FE71A18, 0x93D3DCDC */
r3 = 1.71933865632803078993e-01,   /* 0x3FC601ED, 0xCCFBDF27 */
r4 = 1.86459191715652901344e-02,   /* 0x3F9317EA, 0x742ED475 */
r5 = 7.77942496381893596434e-04,   /* 0x3F497DDA, 0xCA41A95B */
r6 = 7.32668430744625636189e-06,   /* 0x3EDEBAF7, 0xA5B38140 */
w0 = 4.18938533204672725052e-01,   /* 0x3FDACFE3, 0x90C97D69 */
w1 = 8.33333333333329678849e-02,   /* 0x3FB55555, 0x5555553B */
w2 = -2.77777777728775536470e-03,  /* 0xBF66C16C, 0x16B02E5C */
w3 = 7.93650558643019558500e-04,   /* 0x3F4A019F, 0x98CF38B6 */
w4 = -5.95187557450339963135e-04,  /* 0xBF4380CB, 0x8C0FE741 */
w5 = 8.36339918996282139126e-04,   /* 0x3F4B67BA, 0x4CDAD5D1 */
w6 = -1.63092934096575273989e-03;  /* 0xBF5AB89D, 0x0B9E43E4 */

API double cst_oracle_lgamma(double x)
{
    double t, y, z, nadj = 0, p, p1, p2, p3, q, r, w;
    uint32_t ix = hi_word(x);
    int i, sign = (int)(ix >> 31);
    ix &= 0x7fffffff;
    if (ix >= 0x7ff00000) return x * x;
    if (sign) return (x - x) / 0.0;
    if (ix < (0x3ffu - 70) << 20) return -cst_oracle_log(x);
    (void)nadj;
    if ((ix == 0x3ff00000 || ix == 0x40000000) && lo_word(x) == 0)
        r = 0;
    else if (ix < 0x40000000) { /* x < 2.0 */
        if (ix <= 0x3feccccc) { /* lgamma(x) = lgamma(x+1)-log(x) */
            r = -cst_oracle_log(x);
            if (ix >= 0x3FE76944) { y = 1.0 - x; i = 0; }
            else if (ix >= 0x3FCDA661) { y = x - (tc - 1.0); i = 1; }
            else { y = x; i = 2; }
        } else {
            r = 0.0;
            if (ix >= 0x3FFBB4C3) { y = 2.0 - x; i = 0; }      /* [1.7316,2] */
            else if (ix >= 0x3FF3B4C4) { y = x - tc; i = 1; }  /* [1.23,1.73] */
            else { y = x - 1.0; i = 2; }
        }
        switch (i) {
        case 0:
            z = y * y;
            p1 = a0 + z * (a2 + z * (a4 + z * (a6 + z * (a8 + z * a10))));
            p2 = z * (a1 + z * (a3 + z * (a5 + z * (a7 + z * (a9 + z * a11)))));
            p = y * p1 + p2;
            r += (p - 0.5 * y);
            break;
        case 1:
            z = y * y;
            w = z * y;
            p1 = t0 + w * (t3 + w * (t6 + w * (t9 + w * t12)));
            p2 = t1_ + w * (t4 + w * (t7 + w * (t10 + w * t13)));
            p3 = t2_ + w * (t5 + w * (t8 + w * (t11 + w * t14)));
            p = z * p1 - (tt - w * (p2 + y * p3));
            r += tf + p;
            break;
        default:
            p1 = y * (u0 + y * (u1 + y * (u2 + y * (u3 + y * (u4 + y * u5)))));
            p2 = 1.0 + y * (v1 + y * (v2 + y * (v3 + y * (v4 + y * v5))));
            r += -0.5 * y + p1 / p2;
        }
    } else if (ix < 0x40200000) { /* x < 8.0 */
        i = (int)x;
        y = x - (double)i;
        p = y * (s0 + y * (s1_ + y * (s2_ + y * (s3 + y * (s4 + y * (s5 + y * s6))))));
        q = 1.0 + y * (r1 + y * (r2 + y * (r3 + y * (r4 + y * (r5 + y * r6)))));
        r = 0.5 * y + p / q;
        z = 1.0; /* lgamma(1+s) = log(s) + lgamma(s) */
        switch (i) {
        case 7: z *= y + 6.0; /* FALLTHRU */
        case 6: z *= y + 5.0; /* FALLTHRU */
        case 5: z *= y + 4.0; /* FALLTHRU */
        case 4: z *= y + 3.0; /* FALLTHRU */
        case 3: z *= y + 2.0;
            r += cst_oracle_log(z);
            break;
        }
    } else if (ix < 0x43900000) { /* 8.0 <= x < 2**58 */
        t = cst_oracle_log(x);
        z = 1.0 / x;
        y = z * z;
        w = w0 + z * (w1 + y * (w2 + y * (w3 + y * (w4 + y * (w5 + y * w6)))));
        r = (x - 0.5) * (t - 1.0) + w;
    } else /* 2**58 <= x <= inf */
        r = x * (cst_oracle_log(x) - 1.0);
    return r;
}

/* ------------------------------------------------------------------------------------------
 * probability 0.20.3: Laplace / Cauchy / Binomial `distribution` (the CDFs)
 * ---------------------------------------------------------------------------------------- */

/* Laplace::distribution: x <= mu ? exp((x - mu) / b) / 2 : 1 - exp((mu - x) / b) / 2 */
API double cst_oracle_laplace_cdf(double x, double mu, double b)
{
    if (x <= mu) return 0.5 * cst_oracle_exp((x - mu) / b);
    return 1.0 - 0.5 * cst_oracle_exp((mu - x) / b);
}

/* Cauchy::distribution: atan((x - x0) / gamma) / PI + 0.5 */
API double cst_oracle_cauchy_cdf(double x, double x0, double gamma)
{
    static const double pi = 3.14159265358979323846264338327950288;
    return cst_oracle_atan((x - x0) / gamma) / pi + 0.5;
}

/* f64::powi (compiler-rt __powidf2): square and multiply, the order LLVM's runtime uses */
static double o_powi(double a, int b)
{
    const int recip = b < 0;
    double r = 1;
    while (1) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return recip ? 1 / r : r;
}

/* special::Beta::inc_beta(x; p, q, ln_beta): Algorithm AS 63 (Majumder & Bhattacharjee 1973) with the remarks
 * AS R19 / AS 109 -- the regularised incomplete beta function as the `special` crate evaluates it */
API double cst_oracle_inc_beta(double x, double p, double q, double ln_beta)
{
    const double ACU = 0.1e-14;
    if (x <= 0.0) return 0.0;
    if (x >= 1.0) return 1.0;
    double psq = p + q, pbase, qbase, temp, rx;
    int flip = p < psq * x;
    if (flip) { pbase = 1.0 - x; qbase = x; temp = q; q = p; p = temp; }
    else { pbase = x; qbase = 1.0 - x; }
    double term = 1.0, ai = 1.0, alpha = 1.0;
    int ns = (int)(q + qbase * psq);
    rx = pbase / qbase;
    temp = q - ai;
    if (ns == 0) rx = pbase;
    for (;;) {
        term = term * temp * rx / (p + ai);
        alpha += term;
        temp = o_fabs(term);
        if (temp <= ACU && temp <= ACU * alpha) break;
        ai += 1.0;
        ns -= 1;
        if (0 < ns) {
            temp = q - ai;
        } else if (ns == 0) {
            temp = q - ai;
            rx = pbase;
        } else {
            temp = psq;
            psq += 1.0;
        }
    }
    alpha = alpha * cst_oracle_exp(p * cst_oracle_log(pbase) + (q - 1.0) * cst_oracle_log(qbase) - ln_beta) / p;
    return flip ? 1.0 - alpha : alpha;
}

/* Binomial::distribution(x) of Binomial(n, p): x < 0 -> 0; k = floor(x); k >= n -> 1; k == 0 -> q^n;
 * else I_q(n - k, k + 1) with ln_beta = lgamma(n - k) + lgamma(k + 1) - lgamma(n + 1) */
API double cst_oracle_binomial_cdf(double x, int32_t n, double p)
{
    if (x < 0.0) return 0.0;
    const double qq = 1.0 - p;
    if (x >= (double)n) return 1.0;
    int32_t k = (int32_t)x;
    if (k == 0) return o_powi(qq, n);
    const double a = (double)(n - k), b = (double)(k + 1);
    const double lb = cst_oracle_lgamma(a) + cst_oracle_lgamma(b) - cst_oracle_lgamma(a + b);
    return cst_oracle_inc_beta(qq, a, b, lb);
}

/* ------------------------------------------------------------------------------------------
 * LeakyQuantizer<f64, i32, u32, P> over these CDFs (src/stream/model/quantize.rs:284-308, 525-568), tabulated:
 * L[0] = 0, L[i] = (f64 as u32)(free_weight * cdf(lo + i - 0.5)) + i, L[n] = 2^P.
 * family: 1 Laplace(mean, scale), 2 Cauchy(loc, scale), 3 Binomial(n = hi, p = a) on the support 0..=n.
 * Returns 0, or 2 if some probability is zero (the reference panics, quantize.rs:560-566).
 * ---------------------------------------------------------------------------------------- */
static inline uint32_t f64_as_u32(double v)
{
    if (!(v > 0.0)) return 0;
    if (v >= 4294967296.0) return 0xffffffffu;
    return (uint32_t)v;
}

API int cst_oracle_leaky_family_cdf_table(int family, int32_t lo, int32_t hi, int P, double a, double b, uint32_t *cdf)
{
    const int64_t n = (int64_t)hi - lo + 1;
    const uint32_t max_prob = 0xffffffffu >> (32 - P);
    const double fw = (double)(max_prob - ((uint32_t)hi - (uint32_t)lo));
    cdf[0] = 0;
    for (int64_t i = 1; i < n; i++) {
        const double x = (double)(int32_t)(lo + i) - 0.5;
        double c;
        if (family == 1) c = cst_oracle_laplace_cdf(x, a, b);
        else if (family == 2) c = cst_oracle_cauchy_cdf(x, a, b);
        else c = cst_oracle_binomial_cdf(x, hi, a);
        cdf[i] = f64_as_u32(fw * c) + (uint32_t)i;
    }
    cdf[n] = P >= 32 ? 0u : (1u << P);
    int bad = 0;
    for (int64_t i = 0; i < n; i++)
        if ((uint32_t)(cdf[i + 1] - cdf[i]) == 0 || (P < 32 && cdf[i + 1] < cdf[i])) bad = 2;
    return bad;
}

/* ------------------------------------------------------------------------------------------
 * perfectly_quantized_probabilities (src/stream/model/categorical.rs:56-177) + the cumulation of
 * contiguous.rs:301-313, for Probability = u32.  `probs` are the f64 values of the caller's F (f32 inputs widened:
 * `F: Into<f64>`).  Returns 0 and cdf[0..n], or 1 (Err).
 * ---------------------------------------------------------------------------------------- */
typedef struct { int64_t original_index; double prob; uint32_t weight; double win, loss; } slot_t;

/* slots.sort_by(|a, b| b.win.partial_cmp(&a.win).unwrap()): a STABLE sort, descending by win */
static void stable_sort_desc_win(slot_t *s, int64_t n, slot_t *tmp)
{
    for (int64_t width = 1; width < n; width *= 2) {
        for (int64_t lo = 0; lo < n; lo += 2 * width) {
            int64_t mid = lo + width < n ? lo + width : n, hi = lo + 2 * width < n ? lo + 2 * width : n;
            int64_t i = lo, j = mid, k = lo;
            while (i < mid && j < hi) tmp[k++] = (s[j].win > s[i].win) ? s[j++] : s[i++]; /* ties keep the left one first */
            while (i < mid) tmp[k++] = s[i++];
            while (j < hi) tmp[k++] = s[j++];
        }
        memcpy(s, tmp, (size_t)n * sizeof(slot_t));
    }
}

API int cst_oracle_categorical_perfect_cdf(const double *probs, int64_t n, int P, uint32_t *cdf)
{
    const double inf = 1.0 / 0.0;
    if (n < 2 || (uint64_t)n > 0xffffffffull) return 1;
    uint32_t remaining = (P >= 32 ? 0u : (1u << P)) - (uint32_t)n;
    if ((uint64_t)n > ((uint64_t)1 << P)) return 1; /* (the reference's subtraction would wrap and panic later) */
    double norm = 0.0;
    for (int64_t i = 0; i < n; i++) norm += probs[i];
    if (!(norm >= 2.2250738585072014e-308) || norm > 1.7976931348623157e308) return 1; /* is_normal && positive */
    const double scale = (double)remaining / norm;
    slot_t *s = (slot_t *)malloc((size_t)n * sizeof(slot_t)), *tmp = (slot_t *)malloc((size_t)n * sizeof(slot_t));
    int rc = 0;
    for (int64_t i = 0; i < n && !rc; i++) {
        const double prob = probs[i];
        if (prob < 0.0) { rc = 1; break; }
        const uint32_t cur = f64_as_u32(prob * scale);
        if (cur > remaining) { rc = 1; break; } /* (debug builds of the reference panic on the underflow) */
        remaining -= cur;
        const uint32_t weight = cur + 1u;
        s[i].original_index = i;
        s[i].prob = prob;
        s[i].weight = weight;
        s[i].win = prob * cst_oracle_log1p(1.0 / (double)weight);
        s[i].loss = weight == 1u ? inf : -prob * cst_oracle_log1p(-1.0 / (double)weight);
    }
    while (!rc && remaining != 0) {
        stable_sort_desc_win(s, n, tmp);
        const int64_t batch = (int64_t)remaining < n ? (int64_t)remaining : n;
        for (int64_t k = 0; k < batch; k++) {
            s[k].weight += 1u;
            s[k].win = s[k].prob * cst_oracle_log1p(1.0 / (double)s[k].weight);
            s[k].loss = -s[k].prob * cst_oracle_log1p(-1.0 / (double)s[k].weight);
        }
        remaining -= (uint32_t)batch;
    }
    while (!rc) {
        int64_t buyer = 0, seller = 0;
        for (int64_t k = 1; k < n; k++) { /* Iterator::max_by returns the LAST maximum, min_by the FIRST minimum */
            if (s[k].win >= s[buyer].win) buyer = k;
            if (s[k].loss < s[seller].loss) seller = k;
        }
        if (buyer == seller) break;
        if (s[buyer].win <= s[seller].loss) break;
        s[seller].weight -= 1u;
        s[seller].win = -inf;
        s[seller].loss = s[seller].weight == 1u ? inf : -s[seller].prob * cst_oracle_log1p(-1.0 / (double)s[seller].weight);
        s[buyer].weight += 1u;
        s[buyer].loss = inf;
        s[buyer].win = s[buyer].prob * cst_oracle_log1p(1.0 / (double)s[buyer].weight);
    }
    if (!rc) {
        uint32_t *wt = (uint32_t *)malloc((size_t)n * sizeof(uint32_t));
        for (int64_t k = 0; k < n; k++) wt[s[k].original_index] = s[k].weight; /* sort_unstable_by_key(original_index) */
        uint32_t acc = 0;
        for (int64_t i = 0; i < n; i++) { cdf[i] = acc; acc += wt[i]; }
        cdf[n] = acc;
        free(wt);
        if (acc != (P >= 32 ? 0u : (1u << P))) rc = 1;
    }
    free(s); free(tmp);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * LazyContiguousCategoricalEntropyModel<u32, F, _, P> (src/stream/model/categorical/lazy_contiguous.rs:132-331),
 * F = f64 (is_f32 == 0) or f32 (is_f32 != 0; `probs` then points at floats).
 * ---------------------------------------------------------------------------------------- */
static inline uint32_t f32_as_u32(float v)
{
    if (!(v > 0.0f)) return 0;
    if (v >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)v;
}

/* from_floating_point_probabilities_fast, :132-166: scale = F(2^P - n) / sum(pmf).  Returns 0 or 1 (Err). */
static int lazy_scale(const void *probs, int64_t n, int is_f32, int P, double *scale64, float *scale32)
{
    if (n < 2 || (uint64_t)n >= (((uint64_t)1 << P) - 1)) return 1;
    const uint32_t free_weight = (1u << P) - (uint32_t)n;
    if (is_f32) {
        const float *p = (const float *)probs;
        float norm = 0.0f;
        for (int64_t i = 0; i < n; i++) norm = norm + p[i];
        if (!(norm >= 1.17549435e-38f) || norm > 3.4028234e38f) return 1;
        *scale32 = (float)free_weight / norm;
    } else {
        const double *p = (const double *)probs;
        double norm = 0.0;
        for (int64_t i = 0; i < n; i++) norm = norm + p[i];
        if (!(norm >= 2.2250738585072014e-308) || norm > 1.7976931348623157e308) return 1;
        *scale64 = (double)free_weight / norm;
    }
    return 0;
}

/* left_cumulative_and_probability, :228-262.  Returns 0, 1 (symbol out of range -> None) or 3 (model Err) */
API int cst_oracle_lazy_categorical_lcp(const void *probs, int64_t n, int is_f32, int P, int64_t symbol, uint32_t *left,
                                        uint32_t *prob)
{
    double s64 = 0; float s32 = 0;
    if (lazy_scale(probs, n, is_f32, P, &s64, &s32)) return 3;
    if (symbol < 0 || symbol >= n) return 1;
    uint32_t l, r;
    if (is_f32) {
        const float *p = (const float *)probs;
        float lc = 0.0f;
        for (int64_t i = 0; i < symbol; i++) lc = lc + p[i];
        l = f32_as_u32(lc * s32) + (uint32_t)symbol;
        const float rc = lc + p[symbol];
        r = symbol == n - 1 ? (1u << P) : f32_as_u32(rc * s32) + (uint32_t)symbol + 1u;
    } else {
        const double *p = (const double *)probs;
        double lc = 0.0;
        for (int64_t i = 0; i < symbol; i++) lc = lc + p[i];
        l = f64_as_u32(lc * s64) + (uint32_t)symbol;
        const double rc = lc + p[symbol];
        r = symbol == n - 1 ? (1u << P) : f64_as_u32(rc * s64) + (uint32_t)symbol + 1u;
    }
    *left = l;
    *prob = r - l;
    return 0;
}

/* quantile_function, :264-331: skip ahead on the float cumulative (over-estimated scale), then walk with the same
 * float-to-int conversions as the encoder side; the last symbol takes everything up to 2^P */
#define LAZY_QUANTILE(F, AS_U32, EPS)                                                                                  \
    do {                                                                                                               \
        const F *p = (const F *)probs;                                                                                 \
        const F scale = (F)SCALE;                                                                                      \
        F left_f = (F)0, right_f = (F)0;                                                                               \
        const F enlarged = ((F)1 + EPS + EPS) * scale;                                                                 \
        const uint32_t qs = quantile > (uint32_t)n ? quantile - (uint32_t)n : 0u; /* saturating_sub */                 \
        const F lower_bound = (F)qs / enlarged;                                                                        \
        int64_t it = 0;                                                                                                \
        uint32_t next_symbol = 0;                                                                                      \
        while (it < n) {                                                                                               \
            const F np_ = p[it++];                                                                                     \
            next_symbol += 1u;                                                                                         \
            left_f = right_f;                                                                                          \
            right_f = right_f + np_;                                                                                   \
            if (right_f >= lower_bound) break;                                                                         \
        }                                                                                                              \
        uint32_t left_c = AS_U32(left_f * scale) + (next_symbol - 1u);                                                 \
        while (it < n) {                                                                                               \
            const F np_ = p[it++];                                                                                     \
            const uint32_t right_c = AS_U32(right_f * scale) + next_symbol;                                            \
            if (right_c > quantile) { *symbol = (int64_t)next_symbol - 1; *left = left_c; *prob = right_c - left_c; return 0; } \
            left_c = right_c;                                                                                          \
            right_f = right_f + np_;                                                                                   \
            next_symbol += 1u;                                                                                         \
        }                                                                                                              \
        *symbol = (int64_t)next_symbol - 1; *left = left_c; *prob = (1u << P) - left_c;                                \
        return 0;                                                                                                      \
    } while (0)

API int cst_oracle_lazy_categorical_quantile(const void *probs, int64_t n, int is_f32, int P, uint32_t quantile,
                                             int64_t *symbol, uint32_t *left, uint32_t *prob)
{
    double s64 = 0; float s32 = 0;
    if (lazy_scale(probs, n, is_f32, P, &s64, &s32)) return 3;
    if (is_f32) {
#define SCALE s32
        LAZY_QUANTILE(float, f32_as_u32, 1.1920929e-07f);
#undef SCALE
    } else {
#define SCALE s64
        LAZY_QUANTILE(double, f64_as_u32, 2.220446049250313e-16);
#undef SCALE
    }
}
