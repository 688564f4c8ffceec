"""GPU parity tests for BASELINE config C3: one quantized-Gaussian table per stream, tables resident in LDS."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def B():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    from constriction_amd import batched
    return batched


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def c3_params(n_streams, seed=0xC0FFEE):
    """mu in [-10, 10], sigma log-uniform in [0.5, 16]."""
    rng = np.random.default_rng(seed)
    mu = -10 + 20 * rng.random(n_streams)
    sigma = np.exp(np.log(0.5) + rng.random(n_streams) * np.log(32))
    return mu, sigma


def oracle_tables(O, lo, hi, mu, sigma, P):
    return np.stack([O.GaussianModel(lo, hi, m, s, P, 32).cdf_table() for m, s in zip(mu, sigma)])


@pytest.mark.parametrize("cfg", [(32, 64, 12), (16, 32, 12), (32, 64, 16), (16, 32, 16)], ids=lambda c: "W%dS%dP%d" % c)
@pytest.mark.parametrize("lo,hi", [(-127, 127), (-50, 50), (0, 1)])
@pytest.mark.parametrize("n_streams,n_per", [(1, 50), (64, 128), (300, 101), (1000, 64)])
@pytest.mark.parametrize("layout", ["stream_major", "symbol_major"])
def test_per_stream_tables_roundtrip_parity(B, O, cfg, lo, hi, n_streams, n_per, layout):
    W, S, P = cfg
    mu, sigma = c3_params(n_streams, seed=n_streams * 31 + n_per)
    model = B.Model.quantized_gaussian_per_stream(lo, hi, dev(mu), dev(sigma), P)
    cdfs = oracle_tables(O, lo, hi, mu, sigma, P)
    for s in (0, n_streams // 2, n_streams - 1):
        assert model.cdf(s).tolist() == cdfs[s].tolist()
    sym = O.synth_symbols(0xC0FFEE, 0, n_streams, n_per, lo, cdfs, P, per_stream_tables=True)
    want_words, want_n, want_status = O.ans_encode_batch(sym, lo, cdfs, P, W, S)
    enc = B.ans_encode(dev(sym if layout == "stream_major" else sym.T), model, cfg, layout)
    torch.cuda.synchronize()
    words, n_words, status = enc.to_numpy()
    assert status.tolist() == want_status.tolist() and (status == 0).all()
    assert n_words.tolist() == want_n.tolist()
    for s in range(n_streams):
        assert words[s, : n_words[s]].tolist() == want_words[s, : want_n[s]].tolist(), f"stream {s}"
    dec, dstatus = B.ans_decode(enc, model, n_per, layout)
    torch.cuda.synchronize()
    got = dec.cpu().numpy()
    assert (dstatus.cpu().numpy() == 0).all()
    assert np.array_equal(got.T if layout == "symbol_major" else got, sym)


@pytest.mark.parametrize("case", ["narrow", "wide", "flat", "mixed", "two_symbols"])
@pytest.mark.parametrize("n_streams,n_per", [(64, 96), (130, 100), (256, 32), (257, 4096 + 3)])
def test_compact_rows_extremes(B, O, case, n_streams, n_per):
    """The compact per-stream tables (cst_ans_pt.hip) at their limits: rows of a single bin, rows without any run of
    unit probabilities, every probability equal to 1 (n = 2^P), both in one workgroup; full waves take the generated
    main loops, the ragged last wave and the 3 extra symbols the compiler-scheduled paths."""
    P, cfg = 12, (32, 64, 12)
    rng = np.random.default_rng(n_streams * 7 + n_per)
    lo, hi = -127, 127
    if case == "narrow":
        mu, sigma = rng.uniform(-100, 100, n_streams), np.full(n_streams, 1e-3)
    elif case == "wide":
        mu, sigma = rng.uniform(-10, 10, n_streams), np.full(n_streams, 1e4)      # nearly uniform: no unit runs at all
    elif case == "flat":
        lo, hi, P, cfg = 0, 255, 8, (32, 64, 8)                                    # n = 2^P: every probability is 1
        mu, sigma = rng.uniform(0, 255, n_streams), rng.uniform(0.5, 300, n_streams)
    elif case == "two_symbols":
        lo, hi = 0, 1
        mu, sigma = rng.uniform(-1, 2, n_streams), rng.uniform(0.1, 3, n_streams)
    else:
        mu = rng.uniform(-200, 200, n_streams)
        sigma = np.exp(rng.uniform(np.log(1e-3), np.log(1e3), n_streams))
    model = B.Model.quantized_gaussian_per_stream(lo, hi, dev(mu), dev(sigma), P)
    cdfs = oracle_tables(O, lo, hi, mu, sigma, P)
    sym = O.synth_symbols(0xC0FFEE, 0, n_streams, n_per, lo, cdfs, P, per_stream_tables=True)
    # force the tails too: first / last symbols of the support and the ones next to them
    sym[:, :4] = np.array([lo, hi, lo + 1, hi - 1])[: sym[:, :4].shape[1]] if hi - lo > 2 else sym[:, :4]
    want_words, want_n, want_status = O.ans_encode_batch(sym, lo, cdfs, P, 32, 64)
    enc = B.ans_encode(dev(sym), model, cfg)
    torch.cuda.synchronize()
    words, n_words, status = enc.to_numpy()
    assert (status == 0).all() and (want_status == 0).all()
    assert n_words.tolist() == want_n.tolist()
    for s in range(n_streams):
        assert words[s, : n_words[s]].tolist() == want_words[s, : want_n[s]].tolist(), f"stream {s}"
    dec, dstatus = B.ans_decode(enc, model, n_per)
    # and from the packed layout (unaligned streams)
    packed, offsets = B.compact(enc)
    dec2, _ = B.ans_decode((packed, enc.n_words), model, n_per, offsets=offsets, config=cfg)
    torch.cuda.synchronize()
    assert (dstatus.cpu().numpy() == 0).all()
    assert np.array_equal(dec.cpu().numpy(), sym) and np.array_equal(dec2.cpu().numpy(), sym)


def test_per_stream_errors(B, O):
    mu, sigma = c3_params(70)
    model = B.Model.quantized_gaussian_per_stream(-127, 127, dev(mu), dev(sigma), 12)
    cdfs = oracle_tables(O, -127, 127, mu, sigma, 12)
    sym = O.synth_symbols(1, 0, 70, 40, -127, cdfs, 12, per_stream_tables=True)
    sym[9, 3] = 128
    enc = B.ans_encode(dev(sym), model, (32, 64, 12))
    torch.cuda.synchronize()
    st = enc.status.cpu().numpy()
    assert st[9] == 1 and (np.delete(st, 9) == 0).all()
    sigma_bad = sigma.copy()
    sigma_bad[5] = 0.0
    with pytest.raises(ValueError):
        B.Model.quantized_gaussian_per_stream(-127, 127, dev(mu), dev(sigma_bad), 12)
    with pytest.raises(Exception):
        B.ans_encode(dev(sym[:10]), model, (32, 64, 12))   # model has 70 tables, batch has 10 streams


def test_config_c3_full_size(B, O):
    """65 536 streams x 4096 symbols, per-stream (mu, sigma), support -127..127, P = 12."""
    n_streams, n_per, P, lo, hi = 65536, 4096, 12, -127, 127
    mu, sigma = c3_params(n_streams)
    model = B.Model.quantized_gaussian_per_stream(lo, hi, dev(mu), dev(sigma), P)
    sample = [0, 1, 4095, 4096, 33333, 65535]
    cdfs = oracle_tables(O, lo, hi, mu[sample], sigma[sample], P)
    for k, s in enumerate(sample):
        assert model.cdf(s).tolist() == cdfs[k].tolist()
    # symbols: clipped rounded Gaussians generated on the device (any in-support symbols are valid input)
    g = torch.Generator(device="cuda").manual_seed(1234)
    z = torch.randn((n_streams, n_per), generator=g, device="cuda", dtype=torch.float32)
    sym = torch.clamp(torch.round(z * dev(sigma.astype(np.float32))[:, None] + dev(mu.astype(np.float32))[:, None]), lo, hi).to(torch.int32)
    enc = B.ans_encode(sym, model, (32, 64, P))
    dec, status = B.ans_decode(enc, model, n_per)
    torch.cuda.synchronize()
    assert int(enc.status.abs().sum().item()) == 0 and int(status.abs().sum().item()) == 0
    assert torch.equal(dec, sym)
    host = sym[sample].cpu().numpy()
    want_words, want_n, _ = O.ans_encode_batch(host, lo, cdfs, P)
    for k, s in enumerate(sample):
        assert enc.stream(s).tolist() == want_words[k, : want_n[k]].tolist()
