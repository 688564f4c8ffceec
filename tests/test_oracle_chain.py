"""The CPU restatement of stream::chain::ChainCoder (oracle.ChainCoder, src/stream/chain.rs) against the reference's own
vectors: tests/python/test_constriction.py:58-126 (`test_chain_gaussian`, `test_chain_independence`) and the
(DefaultChainCoder / SmallChainCoder) round trips of chain.rs:1225-1388 (`generic_restore_many`)."""
import numpy as np
import pytest

from oracle import oracle as O


def test_chain_independence_golden():
    data = np.array([0x80d14131, 0xdda97c6c, 0x5017a640, 0x01170a3e], np.uint32)
    probabilities = np.array([[0.1, 0.7, 0.1, 0.1], [0.2, 0.2, 0.1, 0.5], [0.2, 0.1, 0.4, 0.3]])
    models = [O.TableModel(O.categorical_fast_cdf(p, 24), 0, 24) for p in probabilities]
    assert O.AnsCoder(data, seal=True).decode(models).tolist() == [0, 0, 2]
    assert O.ChainCoder(data, seal=True).decode(models).tolist() == [0, 3, 3]
    probabilities[0, :] = [0.09, 0.71, 0.1, 0.1]
    models = [O.TableModel(O.categorical_fast_cdf(p, 24), 0, 24) for p in probabilities]
    assert O.AnsCoder(data, seal=True).decode(models).tolist() == [1, 0, 0]
    assert O.ChainCoder(data, seal=True).decode(models).tolist() == [1, 3, 3]     # only the symbol whose model changed


def test_chain_gaussian_restores_the_data():
    original = np.random.RandomState(123).randint(2**32, size=100, dtype=np.uint32)
    means, stds = np.arange(50, dtype=np.float64), np.full(50, 10.0)
    models = [O.GaussianModel(-100, 100, m, s, 24, 32) for m, s in zip(means, stds)]
    decoder = O.ChainCoder(original, seal=True)
    symbols = decoder.decode(models)
    prefix, suffix = decoder.get_remainders()
    assert len(prefix) + len(suffix) < len(original)
    e1 = O.ChainCoder(suffix, is_remainders=True)
    e1.encode_reverse(symbols, models)
    p1, s1 = e1.get_data(unseal=True)
    assert len(p1) == 0 and np.array_equal(np.concatenate((prefix, s1)), original)
    e2 = O.ChainCoder(np.concatenate((prefix, suffix)), is_remainders=True)
    e2.encode_reverse(symbols, models)
    assert np.array_equal(np.concatenate(e2.get_data(unseal=True)), original)
    decoder.encode_reverse(symbols, models)
    p3, s3 = decoder.get_data(unseal=True)
    assert len(p3) == 0 and np.array_equal(s3, original)


@pytest.mark.parametrize("W,S,P,words,n", [(32, 64, 24, 4, 0), (32, 64, 24, 5, 2), (32, 64, 24, 20, 10), (32, 64, 24, 19, 20),
                                           (32, 64, 24, 300, 250), (32, 64, 16, 300, 250), (16, 32, 16, 300, 250),
                                           (16, 32, 8, 300, 250), (16, 32, 12, 300, 250)])
def test_restore_many(W, S, P, words, n):
    """generic_restore_many (chain.rs:1302-1387): decode n symbols from random words whose last word has a random number of
    leading zero bits, then put them back in three ways; every way restores the words."""
    rng = np.random.default_rng(words * 1000 + n + P)
    compressed = rng.integers(0, 1 << W, words, dtype=np.uint64).astype(np.uint32)
    lz = int(rng.integers(0, W - 1))
    compressed[-1] = (int(compressed[-1]) | (1 << (W - lz - 1))) & ((1 << W) - 1 >> lz)
    means = rng.uniform(-100, 100, n); stds = rng.uniform(0.001, 10.001, n)
    models = [O.GaussianModel(-100, 100, m, s, P, 32 if W == 32 else 16) for m, s in zip(means, stds)]
    coder = O.ChainCoder(compressed, W=W, S=S, P=P)
    symbols = coder.decode(models)
    prefix, suffix = coder.get_remainders()
    c2 = O.ChainCoder(np.concatenate((prefix, suffix)), is_remainders=True, W=W, S=S, P=P)
    c3 = O.ChainCoder(suffix, is_remainders=True, W=W, S=S, P=P)
    for c, pre in ((coder, np.zeros(0, np.uint32)), (c2, np.zeros(0, np.uint32)), (c3, prefix)):
        c.encode_reverse(symbols, models)
        a, b = c.get_data()
        assert np.array_equal(np.concatenate((pre, a, b)), compressed)


def _vector_models(spec, P=24):
    """oracle models for a model spec of tests/golden/chain_vectors.json (the scipy ones tabulated through the LeakyQuantizer)"""
    import golden_util
    kind = spec["kind"]
    if kind in ("gaussian", "categorical_fast_rows"):
        return golden_util.models_for({"model": spec}, P, O)[0]
    stats = pytest.importorskip("scipy.stats")
    lo, hi = spec["lo"], spec["hi"]
    table = lambda cdf: O.TableModel(golden_util.leaky_table(cdf, lo, hi, P), lo, P)
    if kind == "scipy_cauchy":
        return table(stats.cauchy(loc=spec["loc"], scale=spec["scale"]).cdf)
    if kind == "scipy_cauchy_family":
        return [table(lambda x, a=a, b=b: stats.cauchy.cdf(x, a, b)) for a, b in zip(spec["locs"], spec["scales"])]
    if kind == "scipy_binom_family":
        return [table(lambda x, p=p: stats.binom.cdf(x, n=spec["n"], p=p)) for p in spec["ps"]]
    raise ValueError(kind)


def _load_chain_vectors():
    import json
    from pathlib import Path
    with open(Path(__file__).parent / "golden" / "chain_vectors.json") as f:
        return json.load(f)["vectors"]


@pytest.mark.parametrize("vec", _load_chain_vectors(), ids=lambda v: v["name"])
def test_chain_vectors_through_the_oracle(vec):
    """tests/golden/chain_vectors.json pins oracle.ChainCoder: expected symbols of the reference's tests, words restored."""
    words = np.array(vec["words"], np.uint32)
    models = _vector_models(vec["model"])
    coder = O.ChainCoder(words, seal=vec["seal"])
    got = coder.decode(models) if isinstance(models, list) else coder.decode(models, vec["n"])
    if vec["symbols"] is not None:
        assert got.tolist() == vec["symbols"]
    if "ans_symbols" in vec:       # the same words through the stack coder: EVERY symbol depends on the first model there
        assert O.AnsCoder(words, seal=True).decode(models).tolist() == vec["ans_symbols"]
    if vec.get("restore"):
        coder.encode_reverse(got, models)
        assert np.concatenate(coder.get_data(unseal=vec["seal"])).tolist() == words.tolist()
