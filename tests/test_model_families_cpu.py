"""CPU-only: the oracle's restatement of the model families that are not the quantized Gaussian (oracle/oracle_families.c)
-- its elementary functions against mpmath, its lazy categorical model against the fast tables and against the
reference's own vectors (tests/golden: L*), its perfect quantisation against the PRODUCT's separate host implementation
(cst_categorical_perfect_cdf: a host function of the HIP library, callable without a GPU), and its Laplace / Cauchy /
Binomial tables against tables built from glibc / scipy (a third, unrelated evaluation of the same formulas)."""
import math
import struct

import numpy as np
import pytest

from oracle import oracle as O


def _ulp_err(got, exact_mpf):
    import mpmath
    if exact_mpf == 0:
        return 0.0 if got == 0 else float("inf")
    e = int(mpmath.floor(mpmath.log(abs(exact_mpf), 2)))
    return float(abs(mpmath.mpf(got) - exact_mpf) / mpmath.mpf(2) ** (e - 52))


@pytest.mark.parametrize("name,bound,draw", [
    ("log", 1.0, lambda r: math.exp(r.uniform(-700, 700))),
    ("log1p", 1.0, lambda r: r.choice([-1, 1]) * math.exp(r.uniform(-40, 0)) * 0.999),
    ("log1p", 1.0, lambda r: math.exp(r.uniform(-3, 40))),
    ("atan", 1.0, lambda r: r.choice([-1, 1]) * math.exp(r.uniform(-30, 45))),
    ("lgamma", 2.0, lambda r: math.exp(r.uniform(-5, 20))),
])
def test_oracle_libm_functions_against_mpmath(name, bound, draw):
    """the libm-crate algorithms as restated: faithfully rounded (< 1 ulp; lgamma's Stirling branch < 2 ulp)"""
    mpmath = pytest.importorskip("mpmath")
    import random
    mpmath.mp.prec = 200
    ref = {"log": mpmath.log, "log1p": mpmath.log1p, "atan": mpmath.atan, "lgamma": mpmath.loggamma}[name]
    fn = getattr(O.load(), "cst_oracle_" + name)
    r = random.Random(hash(name) & 0xffff)
    worst = 0.0
    for _ in range(1500):
        x = draw(r)
        if name == "lgamma" and abs(x - 1) < 0.05 or name == "lgamma" and abs(x - 2) < 0.05:
            continue            # lgamma's zeros: msun is accurate in absolute, not relative terms there
        worst = max(worst, _ulp_err(fn(x), ref(mpmath.mpf(x))))
    assert worst < bound, worst


def test_oracle_libm_special_values():
    lib = O.load()
    assert lib.cst_oracle_log(1.0) == 0.0 and lib.cst_oracle_log1p(0.0) == 0.0 and lib.cst_oracle_atan(0.0) == 0.0
    assert lib.cst_oracle_lgamma(1.0) == 0.0 and lib.cst_oracle_lgamma(2.0) == 0.0
    assert lib.cst_oracle_log(0.0) == -math.inf and lib.cst_oracle_log1p(-1.0) == -math.inf
    assert math.isnan(lib.cst_oracle_log(-1.0)) and math.isnan(lib.cst_oracle_log1p(-2.0))
    assert lib.cst_oracle_atan(1e300) == math.pi / 2 and lib.cst_oracle_atan(-math.inf) == -math.pi / 2
    assert lib.cst_oracle_lgamma(171.0) == pytest.approx(math.lgamma(171.0), rel=1e-15)
    assert lib.cst_oracle_log1p(1e-20) == 1e-20 and lib.cst_oracle_atan(1e-10) == 1e-10


def test_lazy_categorical_equals_fast_tables():
    """test_lazy_f32.py:217 and the Python docs: `lazy=True` and `perfect=False` give the same code.  Here: the lazy
    model's left_cumulative_and_probability / quantile_function (lazy_contiguous.rs:228-331) against the tabulated fast
    cdf (categorical.rs:16-54), symbol by symbol and quantile by quantile, f32 and f64."""
    rng = np.random.default_rng(1)
    for trial in range(120):
        n = int(rng.integers(2, 200))
        P = int(rng.choice([12, 24]))
        p = rng.random(n) if trial % 2 else np.exp(rng.normal(0, 3, n))
        if trial % 3 == 0:
            p = p.astype(np.float32)
        fast = O.categorical_fast_cdf(p, P)
        lz = O.LazyCategoricalModel(p, P)
        for s in range(n):
            l, pr = lz.lcp(s)
            assert l == fast[s] and l + pr == fast[s + 1]
        with pytest.raises(KeyError):
            lz.lcp(n)
        tm = O.TableModel(fast, 0, P)
        qs = list(rng.integers(0, 1 << P, 40)) + [0, (1 << P) - 1] + [int(x) for x in fast[:n]] + [int(x) - 1 for x in fast[1:]]
        for q in qs:
            assert lz.quantile(q) == tm.quantile(q)


def _product_perfect(probs, P):
    from constriction_amd import _native as N
    lib = N.load_library()               # host function of the HIP library: no GPU needed
    p64 = np.ascontiguousarray(probs, dtype=np.float64)
    out = np.zeros(len(p64) + 1, dtype=np.uint32)
    rc = lib.cst_categorical_perfect_cdf(p64.ctypes.data, len(p64), P, out.ctypes.data)
    return rc, out


def test_perfect_quantisation_product_vs_oracle():
    """two separate implementations of categorical.rs:56-177 (library host C++ / oracle C), each over its own restatement
    of libm::log1p: every table identical, including ties (equal probabilities), zeros, skewed and f32 inputs"""
    rng = np.random.default_rng(0)
    for trial in range(300):
        n = int(rng.integers(2, 400))
        kind = trial % 5
        p = (rng.random(n) if kind == 0 else rng.dirichlet(np.ones(n) * 0.1) if kind == 1 else np.exp(rng.normal(0, 4, n)) if kind == 2
             else (rng.random(n) < 0.3) * rng.random(n) + 1e-12 if kind == 3 else np.repeat(rng.random(max(1, n // 7)), 7)[:n] + 0.0)
        if len(p) < 2:
            continue
        if trial % 7 == 0:
            p = p.astype(np.float32)
        for P in (12, 24):
            if len(p) >= (1 << P):
                continue
            want = O.categorical_perfect_cdf(p, P)
            rc, got = _product_perfect(p, P)
            assert rc == 0 and np.array_equal(want, got), (trial, P)
    lib_log1p = __import__("constriction_amd._native", fromlist=["x"]).load_library().cst_debug_host_log1p
    ol = O.load()
    xs = np.concatenate([rng.uniform(-0.999, 5, 50000), -1 / rng.integers(2, 1 << 24, 50000), 1 / rng.integers(1, 1 << 24, 50000)])
    assert all(struct.pack("<d", lib_log1p(float(x))) == struct.pack("<d", ol.cst_oracle_log1p(float(x))) for x in xs)
    for bad in ([0.5, -0.1], [float("nan"), 0.5], [float("inf"), 1.0], [0.0, 0.0]):
        assert _product_perfect(bad, 24)[0] == -4
        with pytest.raises(ValueError):
            O.categorical_perfect_cdf(bad, 24)


def test_family_tables_against_glibc_and_scipy():
    """The reference's `probability` crate calls the platform libm (glibc in its Linux wheels); the oracle fixes the
    libm-crate algorithms.  Both are faithfully rounded, so a table entry could differ by one unit where free_weight * cdf
    lies within an ulp of an integer.  Measured here on ~ 4e5 entries: none does (asserted <= 2 so that an unlucky
    draw after a change of seed does not read as a bug; a restatement error shows up as thousands)."""
    scipy_stats = pytest.importorskip("scipy.stats")
    rng = np.random.default_rng(1)

    def leaky(cdf, lo, hi):
        fw = float(((1 << 24) - 1) - (hi - lo))
        out = np.zeros(hi - lo + 2, dtype=np.uint32)
        for i in range(1, hi - lo + 1):
            v = fw * float(cdf(lo + i - 0.5))
            out[i] = (0 if not v > 0 else min(int(v), 0xFFFFFFFF)) + i
        out[-1] = 1 << 24
        return out

    total = differ = 0
    for trial in range(150):
        lo, hi = int(rng.integers(-300, 0)), int(rng.integers(1, 300))
        mu, sc = rng.uniform(-50, 50), math.exp(rng.uniform(-3, 5))
        for fam, cdf in ((O.FAMILY_LAPLACE, lambda x: 0.5 * math.exp((x - mu) / sc) if x <= mu else 1.0 - 0.5 * math.exp(-(x - mu) / sc)),
                         (O.FAMILY_CAUCHY, lambda x: math.atan((x - mu) / sc) / math.pi + 0.5)):
            try:
                want = O.leaky_family_cdf(fam, lo, hi, mu, sc)
            except ArithmeticError:
                continue
            got = leaky(cdf, lo, hi)
            total += len(want)
            differ += int((want != got).sum())
    for trial in range(80):
        n = int(rng.integers(1, 1500))
        p = rng.random() if trial % 5 else float(rng.choice([0.0, 1.0, 1e-9, 0.5]))
        want = O.leaky_family_cdf(O.FAMILY_BINOMIAL, 0, n, p)
        got = leaky(lambda x: float(scipy_stats.binom.cdf(x, n, p)), 0, n)
        total += len(want)
        differ += int((want != got).sum())
    assert total > 100000 and differ <= 2, (total, differ)


def test_family_cdf_formulas():
    lib = O.load()
    assert lib.cst_oracle_laplace_cdf(1.5, 1.5, 3.0) == 0.5 and lib.cst_oracle_cauchy_cdf(-2.0, -2.0, 1.5) == 0.5
    assert lib.cst_oracle_binomial_cdf(-0.5, 10, 0.3) == 0.0 and lib.cst_oracle_binomial_cdf(10.0, 10, 0.3) == 1.0
    assert lib.cst_oracle_binomial_cdf(0.5, 10, 0.5) == 0.5 ** 10
    assert lib.cst_oracle_binomial_cdf(4.5, 10, 0.5) == pytest.approx(386 / 1024, rel=1e-14)
    row = O.leaky_family_cdf(O.FAMILY_BINOMIAL, 0, 20, 0.3).astype(np.int64)
    assert row[0] == 0 and row[-1] == 1 << 24 and (np.diff(row) >= 1).all() and np.argmax(np.diff(row)) == 6
    # the leak keeps every symbol at >= 1 unit even 10^6 scales out in the tail
    far = O.leaky_family_cdf(O.FAMILY_LAPLACE, 0, 1 << 16, 0.0, 1e-3).astype(np.int64)
    assert (np.diff(far) >= 1).all() and far[-1] == 1 << 24


# ---- the reference's own known answers for `perfectly_quantized_probabilities` (tests/golden/perfect_categorical.json) ----
def _perfect_vectors():
    import json
    from pathlib import Path
    return json.loads((Path(__file__).parent / "golden" / "perfect_categorical.json").read_text())["vectors"]


def _weights(cdf, P):
    """symbol_table()'s probabilities: differences of the cumulatives, the last one against 2^P (which wraps to 0 in a
    u32 at P = 32, contiguous.rs:301-313)"""
    c = cdf.astype(np.int64).copy()
    c[-1] = 1 << P
    return np.diff(c)


def _verify_iterable_entropy_model(weights, hist, P, tol):
    """src/stream/model.rs:1016-1060 in numpy: the weights sum to 2^P, none is zero, sorting by weight is compatible with
    sorting by the histogram, and the KL divergence (model.rs:689-708, f64) is below `tol`; returns the KL divergence"""
    hist = np.asarray(hist, dtype=np.float64)
    assert len(weights) == len(hist) and int(weights.sum()) == 1 << P and (weights > 0).all()
    order = np.lexsort((hist, weights))              # sort_unstable_by on (weight, hist) pairs
    assert (np.diff(hist[order]) >= 0).all()
    p = hist / hist.sum()
    nz = p > 0
    kl = float(np.sum(p[nz] * (np.log2(p[nz]) - np.log2(weights[nz].astype(np.float64)))) + P)
    assert kl < tol, kl
    return kl


@pytest.mark.parametrize("impl", ["oracle", "product"])
@pytest.mark.parametrize("vec", _perfect_vectors(), ids=lambda v: v["id"])
def test_perfect_quantisation_reference_known_answers(vec, impl):
    """contiguous.rs:709-731 (a 37-entry histogram that already sums to 2^32 comes back unchanged), :735-833 (the perfect
    quantisation has a smaller KL divergence than the fast one, f64 and f32 inputs, P = 32 and the default 24), :836-873
    (the two inputs of issue #20 converge) -- through the oracle AND through the library's host function."""
    dtype = np.float32 if vec["dtype"] == "f32" else np.float64

    def perfect(p, P):
        if impl == "oracle":
            return O.categorical_perfect_cdf(p, P)
        rc, out = _product_perfect(p, P)             # F: Into<f64>: the f32 values, widened
        assert rc == 0
        return out

    if vec["expect"] == "weights_equal_hist":
        hist = np.array(vec["hist"], dtype=np.uint64)
        assert int(hist.sum()) == 1 << 32
        cdf = perfect(hist.astype(dtype), vec["precision"])
        assert _weights(cdf, vec["precision"]).tolist() == hist.tolist()
    elif vec["expect"] == "kl_perfect_below_kl_fast":
        hist = np.array(vec["hist"], dtype=np.uint64)
        assert int(hist.sum()) != 1 << 32
        probs = hist.astype(dtype)
        for P in vec["precisions"]:
            kl_fast = _verify_iterable_entropy_model(_weights(O.categorical_fast_cdf(probs, P), P), hist, P, vec["kl_tolerance"])
            kl_perfect = _verify_iterable_entropy_model(_weights(perfect(probs, P), P), hist, P, vec["kl_tolerance"])
            assert kl_perfect < kl_fast, (P, kl_perfect, kl_fast)
    else:
        probs = np.array(vec["probs"], dtype=dtype)
        P = vec["precision"]
        w = _weights(perfect(probs, P), P)
        if vec["expect"].endswith("within_1"):
            assert -1 <= int(w[0]) - int(w[2]) <= 1
        _verify_iterable_entropy_model(w, probs, P, vec["kl_tolerance"])
