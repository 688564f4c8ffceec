"""GPU parity tests of the chain coder: the drop-in `constriction_amd.stream.chain.ChainCoder` against the reference's own
vectors (tests/python/test_constriction.py:58-126, test_docexamples.py:932-995) and against the CPU oracle
(oracle.ChainCoder = src/stream/chain.rs restated), and the batched C entry points for many chains at once."""
import ctypes as C

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


def test_chain_independence(constriction):
    data = np.array([0x80d14131, 0xdda97c6c, 0x5017a640, 0x01170a3e], np.uint32)
    probabilities = np.array([[0.1, 0.7, 0.1, 0.1], [0.2, 0.2, 0.1, 0.5], [0.2, 0.1, 0.4, 0.3]])
    model = constriction.stream.model.Categorical(perfect=False)
    chain = constriction.stream.chain.ChainCoder(data, False, True)
    assert np.all(chain.decode(model, probabilities) == [0, 3, 3])
    probabilities[0, :] = np.array([0.09, 0.71, 0.1, 0.1])
    chain = constriction.stream.chain.ChainCoder(data, False, True)
    assert np.all(chain.decode(model, probabilities) == [1, 3, 3])


def test_chain_gaussian(constriction, O):
    rng = np.random.RandomState(123)
    original_data = rng.randint(2**32, size=100, dtype=np.uint32)
    decoder = constriction.stream.chain.ChainCoder(original_data, seal=True)
    model = constriction.stream.model.QuantizedGaussian(-100, 100)
    means = np.arange(50, dtype=np.float64)
    stds = np.array([10.0] * 50, dtype=np.float64)
    symbols = decoder.decode(model, means, stds)

    want = O.ChainCoder(original_data, seal=True)
    models = [O.GaussianModel(-100, 100, m, s, 24, 32) for m, s in zip(means, stds)]
    assert symbols.tolist() == want.decode(models).tolist()

    remainders_prefix, remainders_suffix = decoder.get_remainders()
    wp, ws = want.get_remainders()
    assert remainders_prefix.tolist() == wp.tolist() and remainders_suffix.tolist() == ws.tolist()
    assert len(remainders_prefix) + len(remainders_suffix) < len(original_data)

    encoder1 = constriction.stream.chain.ChainCoder(remainders_suffix, is_remainders=True)
    encoder1.encode_reverse(symbols, model, means, stds)
    recovered_prefix1, recovered_suffix1 = encoder1.get_data(unseal=True)
    assert len(recovered_prefix1) == 0
    assert np.all(np.concatenate((remainders_prefix, recovered_suffix1)) == original_data)

    remainders = np.concatenate((remainders_prefix, remainders_suffix))
    encoder2 = constriction.stream.chain.ChainCoder(remainders, is_remainders=True)
    encoder2.encode_reverse(symbols, model, means, stds)
    assert np.all(np.concatenate(encoder2.get_data(unseal=True)) == original_data)

    encoder3 = decoder
    encoder3.encode_reverse(symbols, model, means, stds)
    recovered_prefix3, recovered_suffix3 = encoder3.get_data(unseal=True)
    assert len(recovered_prefix3) == 0
    assert np.all(recovered_suffix3 == original_data)


def test_custom_model_chain(constriction):
    scipy_stats = pytest.importorskip("scipy.stats")
    compressed = np.array([0xa5dd25f7, 0xfaef49b5, 0xd5b12228, 0x156ceb98, 0x71a0a92b,
                           0x99e6d365, 0x2eebfadb, 0x404a567b, 0xf6cbdc09, 0xe63f3848], dtype=np.uint32)
    model_scipy = scipy_stats.cauchy(loc=10.3, scale=5.8)
    model = constriction.stream.model.CustomModel(model_scipy.cdf, model_scipy.ppf, -100, 100)
    coder = constriction.stream.chain.ChainCoder(compressed, False, False)
    symbols = coder.decode(model, 4)
    assert np.all(symbols == np.array([18, 6, 33, 59]))
    coder.encode_reverse(symbols, model)
    assert np.all(np.hstack(coder.get_data()) == compressed)

    model = constriction.stream.model.CustomModel(lambda x, loc, scale: scipy_stats.cauchy.cdf(x, loc, scale),
                                                  lambda x, loc, scale: scipy_stats.cauchy.ppf(x, loc, scale), -100, 100)
    params = np.array([(7.3, 3.9), (11.5, 5.2), (-3.2, 4.9), (25.9, 7.1)])
    coder = constriction.stream.chain.ChainCoder(compressed, False, False)
    symbols = coder.decode(model, params[:, 0].copy(), params[:, 1].copy())
    assert np.all(symbols == np.array([13, 7, 16, 85]))
    coder.encode_reverse(symbols, model, params[:, 0].copy(), params[:, 1].copy())
    assert np.all(np.hstack(coder.get_data()) == compressed)

    model = constriction.stream.model.CustomModel(lambda x, params: scipy_stats.binom.cdf(x, n=10, p=params),
                                                  lambda x, params: scipy_stats.binom.ppf(x, n=10, p=params), 0, 10)
    success_probabilities = np.array([0.3, 0.7, 0.2, 0.6])
    coder = constriction.stream.chain.ChainCoder(compressed, False, False)
    symbols = coder.decode(model, success_probabilities)
    assert np.all(symbols == np.array([4, 6, 4, 9]))
    coder.encode_reverse(symbols, model, success_probabilities)
    assert np.all(np.hstack(coder.get_data()) == compressed)


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
