"""GPU tests of the remaining `constriction.stream.model` families through the single-coder drop-in
(src/pybindings/stream/model.rs:455-1055): Categorical(perfect / lazy), Bernoulli, Uniform, QuantizedLaplace,
QuantizedCauchy, Binomial -- concrete and with per-symbol parameters, ANS and range coder.  Every expected table comes
from the ORACLE's own restatement of the family (oracle/oracle_families.c: perfect quantisation over libm::log1p, the lazy
categorical model, Laplace / Cauchy / Binomial CDFs) and the expected words from the oracle coder over those tables -- two
implementations, never the product's table builder on both sides.  Uniform tables are rebuilt here from uniform.rs's
integer formula; Binomial is additionally exercised exactly as the reference's own test does
(tests/python/test_constriction.py:192-226: round trips)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def constriction():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    import constriction_amd
    from constriction_amd import stream  # noqa: F401
    return constriction_amd


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def oracle_words(O, coder, symbols, rows, lo=0):
    """one oracle coder, one tabulated model per symbol (rows: [n_symbols][n+1] or a single cdf)"""
    rows = np.asarray(rows)
    models = [O.TableModel(rows if rows.ndim == 1 else rows[t], lo, 24) for t in range(len(symbols))]
    if coder == "ans":
        c = O.AnsCoder()
        c.encode_reverse(np.asarray(symbols, dtype=np.int32), models, 24)
    else:
        c = O.RangeEncoder()
        c.encode(np.asarray(symbols, dtype=np.int32), models, 24)
    return c.get_compressed()


def roundtrip(constriction, coder, symbols, model, params, n=None):
    stack, queue = constriction.stream.stack, constriction.stream.queue
    symbols = np.asarray(symbols, dtype=np.int32)
    if coder == "ans":
        enc = stack.AnsCoder()
        enc.encode_reverse(symbols, model, *params)
        words = enc.get_compressed()
        dec = stack.AnsCoder(words)
    else:
        enc = queue.RangeEncoder()
        enc.encode(symbols, model, *params)
        words = enc.get_compressed()
        dec = queue.RangeDecoder(words)
    got = dec.decode(model, *params) if params else dec.decode(model, len(symbols))
    assert np.array_equal(got, symbols)
    return words


@pytest.mark.parametrize("coder", ["ans", "range"])
def test_uniform(constriction, O, coder):
    mod = constriction.stream.model
    rng = np.random.default_rng(3)
    size = 1000
    sym = rng.integers(0, size, 200).astype(np.int32)
    sym[:2] = [0, size - 1]
    cdf = np.concatenate([np.arange(size, dtype=np.int64) * ((1 << 24) // size), [1 << 24]]).astype(np.uint32)   # uniform.rs:120-137
    words = roundtrip(constriction, coder, sym, mod.Uniform(size), ())
    assert words.tolist() == oracle_words(O, coder, sym, cdf).tolist()
    sizes = rng.integers(2, 300, 150).astype(np.int32)
    sym2 = (rng.random(150) * sizes).astype(np.int32)
    rows = np.stack([np.concatenate([np.arange(z, dtype=np.int64) * ((1 << 24) // z), np.full(300 - z + 1, 1 << 24)]) for z in sizes]).astype(np.uint32)
    words2 = roundtrip(constriction, coder, sym2, mod.Uniform(), (sizes,))
    assert words2.tolist() == oracle_words(O, coder, sym2, rows).tolist()
    with pytest.raises(KeyError):
        constriction.stream.stack.AnsCoder().encode_reverse(np.array([size], dtype=np.int32), mod.Uniform(size))


@pytest.mark.parametrize("coder", ["ans", "range"])
@pytest.mark.parametrize("flags", [{"perfect": True}, {"perfect": False}, {"lazy": True}, {}])
def test_categorical_quantisations(constriction, O, coder, flags):
    mod = constriction.stream.model
    rng = np.random.default_rng(len(flags) * 7 + (coder == "ans"))
    probs = rng.dirichlet(np.ones(12) * 0.4)
    sym = rng.choice(12, 300, p=probs).astype(np.int32)
    model = mod.Categorical(probs, **flags)
    perfect = flags.get("perfect", not flags)              # (no flag at all = the legacy default = perfect)
    quantise = O.categorical_perfect_cdf if perfect else O.categorical_fast_cdf
    want_cdf = quantise(probs)
    assert model.cdf.tolist() == want_cdf.tolist()
    if flags.get("lazy"):                                  # the lazy model proper, symbol by symbol (lazy_contiguous.rs:228-262)
        lz = O.LazyCategoricalModel(probs)
        assert [lz.lcp(i)[0] for i in range(12)] == want_cdf[:12].tolist()
    words = roundtrip(constriction, coder, sym, model, ())
    assert words.tolist() == oracle_words(O, coder, sym, want_cdf).tolist()
    # family form: one probability row per symbol
    mat = rng.dirichlet(np.ones(5), size=40)
    sym2 = np.array([rng.choice(5, p=row) for row in mat], dtype=np.int32)
    fam = mod.Categorical(**flags)
    rows = np.stack([quantise(row) for row in mat])
    words2 = roundtrip(constriction, coder, sym2, fam, (mat,))
    assert words2.tolist() == oracle_words(O, coder, sym2, rows).tolist()


@pytest.mark.parametrize("coder", ["ans", "range"])
@pytest.mark.parametrize("perfect", [True, False, None])
def test_bernoulli(constriction, O, coder, perfect):
    mod = constriction.stream.model
    rng = np.random.default_rng(11)
    sym = (rng.random(400) < 0.2).astype(np.int32)
    model = mod.Bernoulli(0.2, perfect=perfect)
    q = O.categorical_perfect_cdf if perfect in (True, None) else O.categorical_fast_cdf
    words = roundtrip(constriction, coder, sym, model, ())
    assert words.tolist() == oracle_words(O, coder, sym, q(np.array([0.8, 0.2]))).tolist()
    ps = rng.uniform(0.01, 0.99, 100)
    sym2 = (rng.random(100) < ps).astype(np.int32)
    words2 = roundtrip(constriction, coder, sym2, mod.Bernoulli(perfect=perfect), (ps,))
    assert words2.tolist() == oracle_words(O, coder, sym2, np.stack([q(np.array([1.0 - p, p])) for p in ps])).tolist()


@pytest.mark.parametrize("coder", ["ans", "range"])
def test_laplace_cauchy(constriction, O, coder):
    mod = constriction.stream.model
    rng = np.random.default_rng(5)
    for fam, cls, draw in ((O.FAMILY_LAPLACE, mod.QuantizedLaplace, lambda n: rng.laplace(2.5, 4.0, n)),
                           (O.FAMILY_CAUCHY, mod.QuantizedCauchy, lambda n: 2.5 + 4.0 * rng.standard_cauchy(n))):
        sym = np.clip(np.rint(draw(300)), -100, 100).astype(np.int32)
        model = cls(-100, 100, 2.5, 4.0)
        table = O.leaky_family_cdf(fam, -100, 100, 2.5, 4.0)
        assert model.cdf_table().tolist() == table.tolist()
        words = roundtrip(constriction, coder, sym, model, ())
        assert words.tolist() == oracle_words(O, coder, sym, table, lo=-100).tolist()
        locs, scales = rng.uniform(-20, 20, 60), rng.uniform(0.5, 10, 60)
        sym2 = np.clip(np.rint(locs + scales * rng.standard_normal(60)), -100, 100).astype(np.int32)
        rows = np.stack([O.leaky_family_cdf(fam, -100, 100, a, b) for a, b in zip(locs, scales)])
        words2 = roundtrip(constriction, coder, sym2, cls(-100, 100), (locs, scales))
        assert words2.tolist() == oracle_words(O, coder, sym2, rows, lo=-100).tolist()
        with pytest.raises(ValueError):
            cls(-100, 100, 0.0, -1.0)


def test_device_family_tables_bit_exact(constriction, O):
    """cst_family_cdf_rows (one GPU thread per table entry) against the oracle's separate C restatement: Laplace and
    Cauchy over narrow and wide supports with needle-thin to huge scales, Binomial from n = 1 to 5000 with p from 0 to 1 --
    every entry of every row."""
    from constriction_amd import batched as B
    rng = np.random.default_rng(17)
    n_entries = 0
    for fam in (O.FAMILY_LAPLACE, O.FAMILY_CAUCHY):
        for lo, hi in ((-100, 100), (-5, 3), (0, 1), (-3000, 4000), (1000, 1300)):
            a = np.concatenate([rng.uniform(lo - 50, hi + 50, 24), [lo - 0.5, hi + 0.5, 0.0, 0.25]])
            b = np.concatenate([np.exp(rng.uniform(-6, 9, 24)), [1e-3, 1e3, 1.0, 0.5]])
            keep = []
            for x, y in zip(a, b):
                try:
                    keep.append(O.leaky_family_cdf(fam, lo, hi, x, y))
                except ArithmeticError:        # a zero probability: the device flags the same rows
                    keep.append(None)
            ok = [i for i, r in enumerate(keep) if r is not None]
            got = B.family_cdf_rows(fam, lo, hi, a[ok], b[ok])
            assert np.array_equal(got, np.stack([keep[i] for i in ok])), (fam, lo, hi)
            n_entries += got.size
            for i, r in enumerate(keep):
                if r is None:
                    with pytest.raises(ValueError):
                        B.family_cdf_rows(fam, lo, hi, a[i:i + 1], b[i:i + 1])
    ns = np.concatenate([[1, 2, 3, 7, 8, 9, 40, 100, 5000], rng.integers(1, 1500, 40)]).astype(np.int32)
    ps = np.concatenate([[0.5, 0.0, 1.0, 1e-12, 0.999999, 0.3, 0.5, 0.01, 0.37], rng.random(40)])
    got = B.family_cdf_rows(O.FAMILY_BINOMIAL, 0, int(ns.max()), ps, None, n_per_row=ns)
    for r, (n, p) in enumerate(zip(ns, ps)):
        want = O.leaky_family_cdf(O.FAMILY_BINOMIAL, 0, int(n), p)
        assert np.array_equal(got[r, :n + 2], want), (n, p)
        assert (got[r, n + 1:] == 1 << 24).all()
        n_entries += int(n) + 2
    one = B.family_cdf_rows(O.FAMILY_BINOMIAL, 0, 40, [0.5], None)
    assert np.array_equal(one[0], O.leaky_family_cdf(O.FAMILY_BINOMIAL, 0, 40, 0.5))
    assert n_entries > 300000


def test_device_elementary_functions_bit_exact(constriction, O):
    """log / log1p / atan / lgamma / exp as the device evaluates them (the libm-crate algorithms) vs the oracle's"""
    from constriction_amd import _native as N
    lib, ol = N.lib(), O.load()
    rng = np.random.default_rng(23)
    n = 200000
    pos = np.exp(rng.uniform(-700, 700, n))
    cases = {0: (ol.cst_oracle_log, np.concatenate([pos, [1.0, 0.5, 2.0, 5e-324, 1e-310, 1.7e308]])),
             1: (ol.cst_oracle_log1p, np.concatenate([rng.uniform(-0.9999, 3.0, n), np.exp(rng.uniform(-60, 40, n)), -np.exp(rng.uniform(-60, 0, n)) * 0.99,
                                                      [0.0, -0.0, 1e-17, -1e-17, 0.41421356237309503, -0.2928932188134525, 1e300]])),
             2: (ol.cst_oracle_atan, np.concatenate([rng.standard_cauchy(n), np.exp(rng.uniform(-70, 70, n)) * rng.choice([-1.0, 1.0], n),
                                                     [0.0, 0.4375, 0.6875, 1.1875, 2.4375, -0.4375, 1e20, -1e20, 7.4e19]])),
             3: (ol.cst_oracle_lgamma, np.concatenate([np.exp(rng.uniform(-30, 40, n)), np.arange(1, 3000, dtype=np.float64), [0.23, 0.73, 0.9, 1.23, 1.46, 1.73, 2.0, 7.999, 8.0]])),
             4: (ol.cst_oracle_exp, np.concatenate([rng.uniform(-750, 710, n), rng.uniform(-1, 1, n)]))}
    for which, (fn, x) in cases.items():
        x = np.ascontiguousarray(x, dtype=np.float64)
        dx = torch.from_numpy(x).cuda()
        out = torch.empty_like(dx)
        N.check(lib.cst_debug_family_fn(which, dx.data_ptr(), out.data_ptr(), x.size, None), "cst_debug_family_fn")
        torch.cuda.synchronize()
        want = np.array([fn(float(v)) for v in x])
        assert np.array_equal(out.cpu().numpy().view(np.uint64), want.view(np.uint64)), which


@pytest.mark.parametrize("coder", ["ans", "range"])
def test_binomial_vs_oracle(constriction, O, coder):
    mod = constriction.stream.model
    rng = np.random.default_rng(29)
    sym = rng.binomial(40, 0.5, 300).astype(np.int32)
    words = roundtrip(constriction, coder, sym, mod.Binomial(40, 0.5), ())
    assert words.tolist() == oracle_words(O, coder, sym, O.leaky_family_cdf(O.FAMILY_BINOMIAL, 0, 40, 0.5)).tolist()
    ns = rng.integers(1, 200, 80).astype(np.int32)
    ps = rng.random(80)
    sym2 = rng.binomial(ns, ps).astype(np.int32)
    width = int(ns.max())
    rows = np.stack([np.concatenate([O.leaky_family_cdf(O.FAMILY_BINOMIAL, 0, int(n), p), np.full(width - n, 1 << 24, np.uint32)])
                     for n, p in zip(ns, ps)])
    words2 = roundtrip(constriction, coder, sym2, mod.Binomial(), (ns, ps))
    assert words2.tolist() == oracle_words(O, coder, sym2, rows).tolist()
    sym3 = rng.binomial(100, ps).astype(np.int32)
    rows3 = np.stack([O.leaky_family_cdf(O.FAMILY_BINOMIAL, 0, 100, p) for p in ps])
    words3 = roundtrip(constriction, coder, sym3, mod.Binomial(100), (ps,))
    assert words3.tolist() == oracle_words(O, coder, sym3, rows3).tolist()


def test_binomial_like_the_reference_test(constriction):
    """tests/python/test_constriction.py:192-226"""
    mod, queue = constriction.stream.model, constriction.stream.queue
    symbols = np.array([15, 33, 22], dtype=np.int32)
    ns = np.array([20, 53, 42], dtype=np.int32)
    ps = np.array([0.6, 0.7, 0.5], dtype=np.float64)
    for model, params in ((mod.Binomial(), (ns, ps)), (mod.Binomial(100), (ps,)), (mod.Binomial(40, 0.5), ())):
        encoder = queue.RangeEncoder()
        encoder.encode(symbols, model, *params)
        decoder = queue.RangeDecoder(encoder.get_compressed())
        decoded = decoder.decode(model, *params) if params else decoder.decode(model, 3)
        assert np.all(decoded == symbols)
        roundtrip(constriction, "ans", symbols, model, params)
