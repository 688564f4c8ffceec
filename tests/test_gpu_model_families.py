"""GPU tests of the remaining `constriction.stream.model` families through the single-coder drop-in
(src/pybindings/stream/model.rs:455-1055): Categorical(perfect / lazy), Bernoulli, Uniform, QuantizedLaplace,
QuantizedCauchy, Binomial -- concrete and with per-symbol parameters, ANS and range coder.  Compressed words are compared
with the CPU oracle coding the same tables; Uniform tables are rebuilt here from uniform.rs's integer formula; Binomial is
exercised exactly as the reference's own test does (tests/python/test_constriction.py:192-226: round trips)."""
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
    want_cdf = mod.perfect_quantized_cdf(probs) if perfect else mod.fast_quantized_cdf(probs)
    assert model.cdf.tolist() == want_cdf.tolist()
    words = roundtrip(constriction, coder, sym, model, ())
    assert words.tolist() == oracle_words(O, coder, sym, want_cdf).tolist()
    # family form: one probability row per symbol
    mat = rng.dirichlet(np.ones(5), size=40)
    sym2 = np.array([rng.choice(5, p=row) for row in mat], dtype=np.int32)
    fam = mod.Categorical(**flags)
    rows = np.stack([(mod.perfect_quantized_cdf if perfect else mod.fast_quantized_cdf)(row) for row in mat])
    words2 = roundtrip(constriction, coder, sym2, fam, (mat,))
    assert words2.tolist() == oracle_words(O, coder, sym2, rows).tolist()


@pytest.mark.parametrize("coder", ["ans", "range"])
@pytest.mark.parametrize("perfect", [True, False, None])
def test_bernoulli(constriction, O, coder, perfect):
    mod = constriction.stream.model
    rng = np.random.default_rng(11)
    sym = (rng.random(400) < 0.2).astype(np.int32)
    model = mod.Bernoulli(0.2, perfect=perfect)
    q = mod.perfect_quantized_cdf if perfect in (True, None) else mod.fast_quantized_cdf
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
    for cls, draw in ((mod.QuantizedLaplace, lambda n: rng.laplace(2.5, 4.0, n)), (mod.QuantizedCauchy, lambda n: 2.5 + 4.0 * rng.standard_cauchy(n))):
        sym = np.clip(np.rint(draw(300)), -100, 100).astype(np.int32)
        model = cls(-100, 100, 2.5, 4.0)
        words = roundtrip(constriction, coder, sym, model, ())
        table = mod.leaky_cdf_table(cls._cdf, -100, 100, (2.5, 4.0))
        assert words.tolist() == oracle_words(O, coder, sym, table, lo=-100).tolist()
        locs, scales = rng.uniform(-20, 20, 60), rng.uniform(0.5, 10, 60)
        sym2 = np.clip(np.rint(locs + scales * rng.standard_normal(60)), -100, 100).astype(np.int32)
        roundtrip(constriction, coder, sym2, cls(-100, 100), (locs, scales))
        with pytest.raises(ValueError):
            cls(-100, 100, 0.0, -1.0)


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
