"""GPU tests of the single-coder drop-in (constriction_amd.stream) -- written like the reference's own Python
tests (tests/python/test_constriction.py, test_docexamples*.py): same calls, same golden numbers."""
import numpy as np
import pytest

from conftest import golden_vectors

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def constriction():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    import constriction_amd
    from constriction_amd import stream  # noqa: F401
    return constriction_amd


def _dtype(d):
    return np.float32 if d == "f32" else np.float64


def _call_args(constriction, step):
    """(model, extra positional parameters) exactly as the reference test would pass them."""
    m = step["model"]
    mod = constriction.stream.model
    if m["kind"] == "gaussian":
        if "means" in m:
            dt = _dtype(m.get("dtype", "f64"))
            return mod.QuantizedGaussian(m["lo"], m["hi"]), (np.array(m["means"], dtype=dt), np.array(m["stds"], dtype=dt))
        return mod.QuantizedGaussian(m["lo"], m["hi"], m["mean"], m["std"]), ()
    if m["kind"] == "categorical_fast":
        return mod.Categorical(np.array(m["probs"], dtype=_dtype(m["dtype"])), perfect=False), ()
    if m["kind"] == "categorical_fast_rows":
        return mod.Categorical(perfect=False), (np.array(m["probs"], dtype=_dtype(m["dtype"])),)
    if m["kind"] == "categorical_lazy":      # tests/python/test_lazy_f64.py:130-131
        return mod.Categorical(np.array(m["probs"], dtype=_dtype(m["dtype"])), lazy=True), ()
    if m["kind"] == "categorical_lazy_rows":  # tests/python/test_lazy_f64.py:157-161
        return mod.Categorical(lazy=True), (np.array(m["probs"], dtype=_dtype(m["dtype"])),)
    if m["kind"] == "scipy_norm":            # tests/python/test_constriction.py:233-235
        import scipy.stats
        model_py = scipy.stats.norm(m["loc"], m["scale"])
        return mod.CustomModel(model_py.cdf, model_py.ppf, m["lo"], m["hi"]), ()
    if m["kind"] == "scipy_norm_family":     # tests/python/test_constriction.py:241-244
        import scipy.stats
        model = mod.CustomModel(lambda x, loc, scale: scipy.stats.norm.cdf(x, loc, scale), scipy.stats.norm.ppf, m["lo"], m["hi"])
        return model, (np.array(m["locs"], dtype=np.float64), np.array(m["scales"], dtype=np.float64))
    raise ValueError(m["kind"])


@pytest.mark.parametrize("vec", [v for v in golden_vectors() if (v["W"], v["S"], v["P"]) == (32, 64, 24)], ids=lambda v: v["id"])
def test_golden_vector_through_dropin(constriction, vec):
    is_ans = vec["coder"] == "ans"
    stack, queue = constriction.stream.stack, constriction.stream.queue
    enc_steps = [s for s in vec["steps"] if s["op"] == "encode"]
    dec_steps = [s for s in vec["steps"] if s["op"] == "decode"]
    if enc_steps:
        coder = stack.AnsCoder() if is_ans else queue.RangeEncoder()
        for st in enc_steps:
            model, extra = _call_args(constriction, st)
            symbols = np.array(st["symbols"], dtype=np.int32)
            if is_ans:
                coder.encode_reverse(symbols, model, *extra)
            else:
                coder.encode(symbols, model, *extra)
        compressed = coder.get_compressed()
        assert compressed.dtype == np.uint32
        assert compressed.tolist() == vec["expect_compressed"]
        if "expect_num_bits" in vec:
            assert coder.num_bits() == vec["expect_num_bits"]
        if "expect_num_valid_bits" in vec:
            assert coder.num_valid_bits() == vec["expect_num_valid_bits"]
        if vec.get("roundtrip"):
            dec = stack.AnsCoder(compressed) if is_ans else queue.RangeDecoder(compressed)
            for st in (reversed(enc_steps) if is_ans else enc_steps):
                model, extra = _call_args(constriction, st)
                got = dec.decode(model, *extra) if extra else dec.decode(model, len(st["symbols"]))
                assert got.dtype == np.int32 and got.tolist() == st["symbols"]
            if is_ans:
                assert dec.is_empty()
            else:
                assert dec.maybe_exhausted()
            if is_ans:   # decoding from the encoder itself (test_constriction.py:52-55)
                for st in reversed(enc_steps):
                    model, extra = _call_args(constriction, st)
                    got = coder.decode(model, *extra) if extra else coder.decode(model, len(st["symbols"]))
                    assert got.tolist() == st["symbols"]
                assert coder.is_empty()
    init = vec.get("init")
    if init is not None:
        words = np.array(init["compressed"], dtype=np.uint32)
        dec = stack.AnsCoder(words, init.get("seal", False)) if is_ans else queue.RangeDecoder(words)
        for st in dec_steps:
            model, extra = _call_args(constriction, st)
            n = st.get("n", len(st["expect"]))
            if extra:
                got = dec.decode(model, *extra)
            elif len(st["expect"]) == 1 and "n" not in st:
                got = np.array([dec.decode(model)])        # scalar form: `coder.decode(model)` returns an int
            else:
                got = dec.decode(model, n)
            assert np.asarray(got).tolist() == st["expect"]
        if vec.get("expect_empty_after"):
            assert dec.is_empty()


def test_sizes_compress_few(constriction, golden):
    """src/stream/stack.rs:1249-1291."""
    sz = golden["sizes"]
    model = constriction.stream.model.QuantizedGaussian(sz["lo"], sz["hi"], sz["mean"], sz["std"])
    for case in sz["cases"]:
        symbols = case.get("symbols")
        if symbols is None:
            a, b = case["symbols_range"]
            symbols = list(range(a, b))
        coder = constriction.stream.stack.AnsCoder()
        if symbols:
            coder.encode_reverse(np.array(symbols[::-1], dtype=np.int32), model)
        assert len(coder.get_compressed()) == case["expect_num_words"] == coder.num_words()


def test_errors_like_the_reference(constriction):
    stack, model = constriction.stream.stack, constriction.stream.model
    with pytest.raises(ValueError):
        stack.AnsCoder(np.array([5, 0], dtype=np.uint32))            # trailing zero word
    with pytest.raises(ValueError):
        stack.AnsCoder(None, True)                                   # seal without data
    coder = stack.AnsCoder()
    m = model.QuantizedGaussian(-10, 10, 0.0, 3.0)
    with pytest.raises(KeyError):
        coder.encode_reverse(np.array([1, 11, 2], dtype=np.int32), m)   # symbol outside the support
    assert coder.is_empty()
    with pytest.raises(ValueError):
        coder.encode_reverse(np.array([1, 2], dtype=np.int32), model.QuantizedGaussian(-10, 10), np.array([0.0]), np.array([1.0]))
    with pytest.raises(ValueError):
        coder.encode_reverse(3, model.QuantizedGaussian(-10, 10), np.array([0.0]), np.array([1.0]))
    with pytest.raises(ValueError):
        model.QuantizedGaussian(-10, 10, 0.0, -1.0)
    with pytest.raises(AssertionError):
        stack.AnsCoder().get_compressed(unseal=True)
    # seal / unseal round trip (src/stream/stack.rs:913-940)
    data = np.array([0x89ABCDEF, 0x01234567], dtype=np.uint32)
    c = stack.AnsCoder(data, seal=True)
    assert c.get_compressed().tolist() == [0x89ABCDEF, 0x01234567, 1]
    assert c.get_compressed(unseal=True).tolist() == data.tolist()


def test_seek_like_docexample(constriction):
    """tests/python/test_docexamples.py:403-427 (ANS) and 619-641 (range)."""
    stack, queue, model = constriction.stream.stack, constriction.stream.queue, constriction.stream.model
    m = model.Categorical(np.array([0.2, 0.4, 0.1, 0.3], dtype=np.float64), perfect=False)
    part1 = np.array([1, 2, 0, 3, 2, 3, 0], dtype=np.int32)
    part2 = np.array([2, 2, 0, 1, 3], dtype=np.int32)
    coder = stack.AnsCoder()
    coder.encode_reverse(part2, m)
    position, state = coder.pos()
    coder.encode_reverse(part1, m)
    assert coder.decode(m) == 1
    coder.seek(position, state)
    assert coder.decode(m, 5).tolist() == part2.tolist()
    enc = queue.RangeEncoder()
    enc.encode(part1, m)
    position, state = enc.pos()
    enc.encode(part2, m)
    dec = queue.RangeDecoder(enc.get_compressed())
    assert dec.decode(m) == 1
    dec.seek(position, state)
    assert dec.decode(m, 5).tolist() == part2.tolist()


def test_long_message_vs_oracle(constriction):
    """A long per-symbol-parameter message (the README workload) against the CPU oracle, both coders."""
    from oracle import oracle as O
    rng = np.random.default_rng(42)
    n = 20000
    means = rng.uniform(-30, 30, n)
    stds = np.exp(rng.uniform(np.log(0.2), np.log(40.0), n))
    symbols = np.clip(np.round(rng.normal(means, stds)), -100, 100).astype(np.int32)
    fam = constriction.stream.model.QuantizedGaussian(-100, 100)
    coder = constriction.stream.stack.AnsCoder()
    coder.encode_reverse(symbols, fam, means, stds)
    ref = O.AnsCoder()
    ref.encode_gaussian_reverse(symbols, -100, 100, means, stds)
    assert coder.get_compressed().tolist() == ref.get_compressed().tolist()
    assert coder.decode(fam, means, stds).tolist() == symbols.tolist()
    assert coder.is_empty()
    # iid model, appended to a non-empty coder in two calls
    iid = constriction.stream.model.QuantizedGaussian(-50, 50, 3.2, 9.6)
    cdf = O.GaussianModel(-50, 50, 3.2, 9.6, 24, 32).cdf_table()
    msg = O.synth_symbols(7, 0, 1, 30000, -50, cdf, 24)[0]
    coder.encode_reverse(msg[15000:], iid)
    coder.encode_reverse(msg[:15000], iid)
    ref = O.AnsCoder()
    ref.encode_iid_table_reverse(msg, cdf, -50, 24)
    assert coder.get_compressed().tolist() == ref.get_compressed().tolist()
    assert coder.decode(iid, 30000).tolist() == msg.tolist()
    # range coder
    enc = constriction.stream.queue.RangeEncoder()
    enc.encode(symbols[:5000], fam, means[:5000], stds[:5000])
    enc.encode(msg[:5000], iid)
    oenc = O.RangeEncoder()
    oenc.encode(symbols[:5000], [O.GaussianModel(-100, 100, m, s) for m, s in zip(means[:5000], stds[:5000])])
    oenc.encode(msg[:5000], O.TableModel(cdf, -50, 24))
    assert enc.get_compressed().tolist() == oenc.get_compressed().tolist()
    dec = enc.get_decoder()
    assert dec.decode(fam, means[:5000], stds[:5000]).tolist() == symbols[:5000].tolist()
    assert dec.decode(iid, 5000).tolist() == msg[:5000].tolist()
    assert dec.maybe_exhausted()
