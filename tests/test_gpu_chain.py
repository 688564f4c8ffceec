"""GPU parity tests of the chain coder: the drop-in `constriction_amd.stream.chain.ChainCoder` against the reference's
vectors, kept as data in tests/golden/chain_vectors.json (each entry cites its source), and against the CPU oracle
(oracle.ChainCoder = src/stream/chain.rs restated), and the batched C entry points for many chains at once."""
import json
from pathlib import Path

import numpy as np
import pytest

import golden_util

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


def _chain_vectors():
    with open(Path(__file__).parent / "golden" / "chain_vectors.json") as f:
        return json.load(f)["vectors"]


def _dropin_model(M, spec):
    """(model, per-symbol parameter arrays) of the drop-in for a model spec of tests/golden/chain_vectors.json"""
    kind = spec["kind"]
    if kind == "categorical_fast_rows":
        return M.Categorical(perfect=False), (np.array(spec["probs"], np.float64),)
    if kind == "gaussian":
        return M.QuantizedGaussian(spec["lo"], spec["hi"]), (np.array(spec["means"]), np.array(spec["stds"]))
    stats = pytest.importorskip("scipy.stats")
    lo, hi = spec["lo"], spec["hi"]
    if kind == "scipy_cauchy":
        frozen = stats.cauchy(loc=spec["loc"], scale=spec["scale"])
        return M.CustomModel(frozen.cdf, frozen.ppf, lo, hi), ()
    if kind == "scipy_cauchy_family":
        return (M.CustomModel(lambda x, a, b: stats.cauchy.cdf(x, a, b), lambda x, a, b: stats.cauchy.ppf(x, a, b), lo, hi),
                (np.array(spec["locs"]), np.array(spec["scales"])))
    if kind == "scipy_binom_family":
        n = spec["n"]
        return (M.CustomModel(lambda x, p: stats.binom.cdf(x, n=n, p=p), lambda x, p: stats.binom.ppf(x, n=n, p=p), lo, hi),
                (np.array(spec["ps"]),))
    raise ValueError(kind)


@pytest.mark.parametrize("vec", _chain_vectors(), ids=lambda v: v["name"])
def test_chain_reference_vectors(constriction, O, vec):
    """Every vector of tests/golden/chain_vectors.json through the drop-in: the symbols the reference expects (and the
    oracle's, where the oracle has the model), and -- `restore` -- every way of putting them back restores the words."""
    Chain, M = constriction.stream.chain.ChainCoder, constriction.stream.model
    words = np.array(vec["words"], np.uint32)
    model, params = _dropin_model(M, vec["model"])
    args = params if params else (vec["n"],)
    coder = Chain(words, False, vec["seal"])
    got = coder.decode(model, *args)
    if vec["symbols"] is not None:
        assert got.tolist() == vec["symbols"]
    if vec["model"]["kind"] in ("gaussian", "categorical_fast_rows"):
        omodels, _ = golden_util.models_for({"model": vec["model"]}, 24, O)
        want = O.ChainCoder(words, seal=vec["seal"])
        assert got.tolist() == want.decode(omodels).tolist()
        for mine, theirs in zip(coder.get_remainders(), want.get_remainders()):
            assert mine.tolist() == theirs.tolist()
    if not vec.get("restore"):
        return
    prefix, suffix = coder.get_remainders()
    if vec.get("shrinks"):
        assert len(prefix) + len(suffix) < len(words)
    unseal = vec["seal"]
    # (a) the coder that decoded them takes them back
    coder.encode_reverse(got, model, *params)
    back = coder.get_data(unseal=unseal)
    assert np.concatenate(back).tolist() == words.tolist()
    if vec["seal"]:
        assert len(back[0]) == 0
        # (b) a coder made of all remainders, (c) one made of the suffix alone (the prefix was never touched)
        whole = Chain(np.concatenate((prefix, suffix)), is_remainders=True)
        whole.encode_reverse(got, model, *params)
        assert np.concatenate(whole.get_data(unseal=True)).tolist() == words.tolist()
        tail = Chain(suffix, is_remainders=True)
        tail.encode_reverse(got, model, *params)
        head, rest = tail.get_data(unseal=True)
        assert len(head) == 0 and np.concatenate((prefix, rest)).tolist() == words.tolist()


@pytest.mark.parametrize("kind", ["gaussian", "table", "rows"])
@pytest.mark.parametrize("n_words,n", [(4, 1), (40, 30), (700, 600), (3000, 2500)])
def test_chain_dropin_vs_oracle(constriction, O, kind, n_words, n):
    rng = np.random.default_rng(n_words + n)
    data = rng.integers(1, 1 << 32, n_words, dtype=np.uint64).astype(np.uint32)
    M = constriction.stream.model
    if kind == "gaussian":
        means, stds = rng.uniform(-80, 80, n), np.exp(rng.uniform(-2, 4, n))
        model, params = M.QuantizedGaussian(-100, 100), (means, stds)
        omodels = [O.GaussianModel(-100, 100, m, s, 24, 32) for m, s in zip(means, stds)]
    elif kind == "table":
        probs = rng.random(37) + 0.01
        model, params = M.Categorical(probs / probs.sum(), perfect=False), ()
        omodels = O.TableModel(O.categorical_fast_cdf(probs / probs.sum(), 24), 0, 24)
    else:
        probs = rng.random((n, 9)) + 0.01
        probs /= probs.sum(axis=1, keepdims=True)
        model, params = M.Categorical(perfect=False), (probs,)
        omodels = [O.TableModel(O.categorical_fast_cdf(p, 24), 0, 24) for p in probs]
    coder = constriction.stream.chain.ChainCoder(data, seal=True)
    want = O.ChainCoder(data, seal=True)
    symbols = coder.decode(model, *params) if params else coder.decode(model, n)
    assert symbols.tolist() == (want.decode(omodels) if params else want.decode(omodels, n)).tolist()
    for got, exp in zip(coder.get_remainders(), want.get_remainders()):
        assert got.tolist() == exp.tolist()
    # a second batch of symbols off the same coder, then everything back in reverse order
    more = min(n, (n_words * 32) // 24 - n - 3)
    if more > 0:
        s2 = coder.decode(model, *[p[:more] for p in params]) if params else coder.decode(model, more)
        w2 = want.decode(omodels[:more] if params else omodels, more)
        assert s2.tolist() == w2.tolist()
        coder.encode_reverse(s2, model, *[p[:more] for p in params])
        want.encode_reverse(w2, omodels[:more] if params else omodels)
    coder.encode_reverse(symbols, model, *params)
    want.encode_reverse(symbols, omodels)
    got, exp = coder.get_data(unseal=True), want.get_data(unseal=True)
    assert got[0].tolist() == exp[0].tolist() and got[1].tolist() == exp[1].tolist()
    assert np.array_equal(np.concatenate(got), data)


def test_chain_errors(constriction):
    M = constriction.stream.model
    with pytest.raises(ValueError):
        constriction.stream.chain.ChainCoder(np.array([1, 0], np.uint32))               # ends in a zero word, not sealed
    with pytest.raises(AssertionError):
        constriction.stream.chain.ChainCoder(np.array([1, 2, 3], np.uint32), True, True)  # cannot seal remainders
    coder = constriction.stream.chain.ChainCoder(np.array([5, 6, 7], np.uint32), seal=True)
    model = M.QuantizedGaussian(-10, 10, 0.0, 3.0)
    with pytest.raises(AssertionError):
        coder.decode(model, 50)                                                        # out of compressed data
    coder = constriction.stream.chain.ChainCoder(np.array([5, 6, 7, 8], np.uint32), seal=True)
    sym = coder.decode(model, 2)
    with pytest.raises(KeyError):
        coder.encode_reverse(np.array([99], np.int32), model)                           # impossible symbol
    coder.encode_reverse(sym, model)
    assert np.concatenate(coder.get_data(unseal=True)).tolist() == [5, 6, 7, 8]
    with pytest.raises(AssertionError):
        coder.encode_reverse(np.zeros(40, np.int32), model)                             # out of remainders
    # the same on the three-kernel path (64 symbols and more): the coder is left as it was
    data = np.arange(1, 21, dtype=np.uint32)
    coder = constriction.stream.chain.ChainCoder(data, seal=True)
    fam = M.QuantizedGaussian(-10, 10)
    with pytest.raises(AssertionError):
        coder.decode(fam, np.zeros(200), np.full(200, 3.0))
    with pytest.raises(KeyError):
        coder.decode(fam, np.zeros(20), np.concatenate([np.full(19, 3.0), [-1.0]]))       # (short path: invalid model)
    big = constriction.stream.chain.ChainCoder(np.arange(1, 101, dtype=np.uint32), seal=True)
    with pytest.raises(KeyError):
        big.decode(fam, np.zeros(75), np.concatenate([np.full(74, 3.0), [-1.0]]))         # (three-kernel path: invalid model)
    sym = coder.decode(fam, np.zeros(20), np.full(20, 3.0))
    coder.encode_reverse(sym, fam, np.zeros(20), np.full(20, 3.0))
    assert np.concatenate(coder.get_data(unseal=True)).tolist() == data.tolist()


@pytest.mark.parametrize("cfg", [(32, 64, 24), (16, 32, 12), (16, 32, 16)], ids=lambda c: "W%dS%dP%d" % c)
@pytest.mark.parametrize("n_streams", [3, 70])
@pytest.mark.parametrize("layout", [0, 1])
def test_chain_batch_entry_points(constriction, O, cfg, n_streams, layout):
    """Many chains at once through the C ABI (one wave per chain below 64 chains, one lane per chain from 64 on): decode
    per-symbol Gaussians, then re-encode them; every chain against the oracle."""
    W, S, P = cfg
    rng = np.random.default_rng(n_streams * 7 + P)
    n_per, stride = 90, 100
    lo, hi = (-100, 100) if P > 8 else (-20, 20)
    words = rng.integers(1, 1 << W, (n_streams, stride), dtype=np.uint64).astype(np.uint32)
    mu = rng.uniform(lo, hi, (n_streams, n_per)); sd = np.exp(rng.uniform(-1, 3, (n_streams, n_per)))
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    t = (lambda a: a.T) if layout == 1 else (lambda a: a)
    oracles, heads = [], np.zeros((n_streams, 2), dtype=np.uint64)
    n_pop = np.zeros(n_streams, np.uint32)
    for s in range(n_streams):
        c = O.ChainCoder(words[s], W=W, S=S, P=P)
        oracles.append(c)
        heads[s, 0] = c.rem_head; heads[s, 1] = c.comp_head
        n_pop[s] = len(c.compressed)
    from constriction_amd import batched as B
    chains = B.ChainBatch(dev(words.view(np.int32)), dev(n_pop.view(np.int32)), dev(heads.view(np.int64)), cfg)
    d_mu, d_sd = dev(t(mu)), dev(t(sd))
    lay = "symbol_major" if layout == 1 else "stream_major"
    d_sym, d_push, d_n_push, d_status = B.chain_decode_gaussian(chains, lo, hi, d_mu, d_sd, lay)
    d_heads, d_n_pop = chains.heads, chains.n_words
    torch.cuda.synchronize()
    sym = t(d_sym.cpu().numpy())
    assert (d_status.cpu().numpy() == 0).all()
    got_heads = d_heads.cpu().numpy().view(np.uint64)
    pushed, n_pushed, left = d_push.cpu().numpy().view(np.uint32), d_n_push.cpu().numpy(), d_n_pop.cpu().numpy()
    for s in range(n_streams):
        c = oracles[s]
        models = [O.GaussianModel(lo, hi, m, d, P, 32 if W == 32 else 16) for m, d in zip(mu[s], sd[s])]
        assert sym[s].tolist() == c.decode(models).tolist(), f"chain {s}"
        assert int(got_heads[s, 0]) == c.rem_head and int(got_heads[s, 1] & 0xffffffff) == c.comp_head
        assert left[s] == len(c.compressed) and pushed[s, : n_pushed[s]].tolist() == c.remainders
    # back again: pop the remainders just pushed, push onto (what is left of) compressed
    rem = B.ChainBatch(d_push, d_n_push.clone(), chains.heads, cfg)
    d_back, d_n_back, d_status = B.chain_encode_gaussian(rem, d_sym, lo, hi, d_mu, d_sd, lay)
    torch.cuda.synchronize()
    assert (d_status.cpu().numpy() == 0).all()
    back, n_back = d_back.cpu().numpy().view(np.uint32), d_n_back.cpu().numpy()
    got_heads = d_heads.cpu().numpy().view(np.uint64)
    for s in range(n_streams):
        c = oracles[s]
        models = [O.GaussianModel(lo, hi, m, d, P, 32 if W == 32 else 16) for m, d in zip(mu[s], sd[s])]
        before = len(c.compressed)
        c.encode_reverse(sym[s], models)
        assert back[s, : n_back[s]].tolist() == c.compressed[before:], f"chain {s}"
        assert int(got_heads[s, 0]) == c.rem_head and int(got_heads[s, 1] & 0xffffffff) == c.comp_head
        assert np.array_equal(np.concatenate(c.get_data()), words[s])
