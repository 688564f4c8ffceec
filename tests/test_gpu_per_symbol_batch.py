"""GPU parity tests of the BATCHED per-symbol-parameter API (constriction_amd.batched.*_gaussian): the reference's
flagship call `encode_reverse(symbols, QuantizedGaussian(lo, hi), means, stds)` / `decode(family, means, stds)`
(src/pybindings/stream/stack.rs:567-588, 733-751; queue.rs:343-410, 598-661) for many independent coders at once,
against one CPU oracle coder per stream."""
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


def workload(n_streams, n_per, lo, hi, seed):
    rng = np.random.default_rng(seed)
    mu = rng.uniform(lo * 0.6, hi * 0.6, (n_streams, n_per))
    sd = np.exp(rng.uniform(np.log(0.3), np.log(40.0), (n_streams, n_per)))
    sym = np.clip(np.rint(mu + sd * rng.standard_normal((n_streams, n_per))), lo, hi).astype(np.int32)
    sym[:, :2] = np.array([lo, hi])[: min(2, n_per)] if n_per >= 2 else sym[:, :2]     # the ends of the support too
    return sym, mu, sd


@pytest.mark.parametrize("coder", ["ans", "range"])
@pytest.mark.parametrize("cfg", [(32, 64, 24), (32, 64, 12), (16, 32, 12)], ids=lambda c: "W%dS%dP%d" % c)
@pytest.mark.parametrize("n_streams,n_per", [(1, 300), (65, 40), (1000, 21)])
@pytest.mark.parametrize("layout", ["stream_major", "symbol_major"])
@pytest.mark.parametrize("encoder", ["two_pass", "fused"])
def test_gaussian_per_symbol_batch_parity(B, O, coder, cfg, n_streams, n_per, layout, encoder, knob):
    # batches of >= 16 384 streams take the fused encoder kernel (entries computed and coded in one kernel, nothing but inputs
    # and words in HBM); CST_FUSED_MIN_STREAMS moves that threshold so that the small parity shapes run it too
    knob(CST_FUSED_MIN_STREAMS="1" if encoder == "fused" else "1000000000")
    W, S, P = cfg
    lo, hi = (-100, 100) if P == 24 else (-60, 60)
    sym, mu, sd = workload(n_streams, n_per, lo, hi, n_streams * 13 + n_per + P)
    # one oracle coder per stream
    want = []
    for s in range(n_streams):
        if coder == "ans":
            c = O.AnsCoder(W=W, S=S)
            c.encode_gaussian_reverse(sym[s], lo, hi, mu[s], sd[s], P, 32 if W == 32 else 16)
        else:
            c = O.RangeEncoder(W=W, S=S)
            c.encode(sym[s], [O.GaussianModel(lo, hi, m, d, P, 32 if W == 32 else 16) for m, d in zip(mu[s], sd[s])], P)
        want.append(c.get_compressed())
    t = (lambda a: a.T) if layout == "symbol_major" else (lambda a: a)
    enc_fn = B.ans_encode_gaussian if coder == "ans" else B.range_encode_gaussian
    dec_fn = B.ans_decode_gaussian if coder == "ans" else B.range_decode_gaussian
    enc = enc_fn(dev(t(sym)), lo, hi, dev(t(mu)), dev(t(sd)), cfg, layout)
    torch.cuda.synchronize()
    words, n_words, status = enc.to_numpy()
    assert (status == 0).all()
    for s in range(n_streams):
        assert words[s, : n_words[s]].tolist() == want[s].tolist(), f"stream {s}"
    dec, dstatus = dec_fn(enc, lo, hi, dev(t(mu)), dev(t(sd)), layout)
    torch.cuda.synchronize()
    assert (dstatus.cpu().numpy() == 0).all()
    assert np.array_equal(t(dec.cpu().numpy()), sym)
    if coder == "ans":
        packed, offsets = B.compact(enc)
        dec2, _ = B.ans_decode_gaussian((packed, enc.n_words), lo, hi, dev(t(mu)), dev(t(sd)), layout, offsets=offsets, config=cfg)
        torch.cuda.synchronize()
        assert np.array_equal(t(dec2.cpu().numpy()), sym)


def test_gaussian_per_symbol_batch_errors(B, O):
    lo, hi, cfg = -30, 30, (32, 64, 24)
    sym, mu, sd = workload(70, 25, lo, hi, 5)
    sym[9, 3] = hi + 1                                  # impossible symbol -> that stream only
    sd[11, 7] = 0.0                                     # invalid model -> that stream only (the reference panics)
    enc = B.ans_encode_gaussian(dev(sym), lo, hi, dev(mu), dev(sd), cfg)
    torch.cuda.synchronize()
    st = enc.status.cpu().numpy()
    assert st[9] == 1 and st[11] == 1 and (np.delete(st, [9, 11]) == 0).all()
    with pytest.raises(ValueError):
        B.ans_encode_gaussian(dev(sym), lo, hi, dev(mu[:, :5]), dev(sd), cfg)
    # supports wider than 65536 symbols work in both directions (P = 24)
    lo2, hi2 = -40000, 40000
    sym2, mu2, sd2 = workload(3, 50, lo2, hi2, 11)
    sd2 *= 300
    sym2 = np.clip(np.rint(mu2 + sd2 * np.random.default_rng(1).standard_normal(sym2.shape)), lo2, hi2).astype(np.int32)
    enc2 = B.ans_encode_gaussian(dev(sym2), lo2, hi2, dev(mu2), dev(sd2), cfg)
    dec2, st2 = B.ans_decode_gaussian(enc2, lo2, hi2, dev(mu2), dev(sd2))
    torch.cuda.synchronize()
    assert (enc2.status.cpu().numpy() == 0).all() and (st2.cpu().numpy() == 0).all()
    assert np.array_equal(dec2.cpu().numpy(), sym2)
    c = O.AnsCoder()
    c.encode_gaussian_reverse(sym2[1], lo2, hi2, mu2[1], sd2[1], 24, 32)
    assert enc2.stream(1).tolist() == c.get_compressed().tolist()


@pytest.mark.parametrize("n_streams", [130, 3])          # a lane per stream / cdf rows + a wave per stream
@pytest.mark.parametrize("cfg", [(32, 64, 24), (16, 32, 12)], ids=lambda c: "W%dS%dP%d" % c)
def test_gaussian_per_symbol_decode_of_random_words_extreme_models(B, O, cfg, n_streams):
    """Decoding RANDOM words draws every quantile, the far tails included, and the models here are the hard ones for a
    search that starts from an inverse-CDF guess: needle-thin and very wide Gaussians, means far outside the support
    (all the mass in the leak), supports of two symbols.  The lane-per-stream decoder must find the reference's symbol."""
    W, S, P = cfg
    rng = np.random.default_rng(P)
    n_per = 70
    for lo, hi in ((-100, 100), (0, 1), (-5, 2000 if P == 24 else 900), (-127, 127)):
        mu = rng.uniform(lo - 50.0, hi + 50.0, (n_streams, n_per))
        sd = np.exp(rng.uniform(np.log(1e-7), np.log(1e6), (n_streams, n_per)))
        mu[:, 0] = lo - 1e9; mu[:, 1] = hi + 1e9; sd[:, 2] = 1e-300; sd[:, 3] = 1e300
        stride = 160
        words = rng.integers(1, 1 << W, (n_streams, stride), dtype=np.uint64).astype(np.uint32)
        enc = B.EncodedBatch(dev(words.view(np.int32)), dev(np.full(n_streams, stride, np.int32)), dev(np.zeros(n_streams, np.int32)), cfg)
        dec, st = B.ans_decode_gaussian(enc, lo, hi, dev(mu), dev(sd))
        torch.cuda.synchronize()
        dec = dec.cpu().numpy()
        assert (st.cpu().numpy() == 0).all()
        for s in range(0, n_streams, 3 if n_streams > 3 else 1):
            comp = words[s] if W == 32 else words[s].astype(np.uint16)
            c = O.AnsCoder(comp, W=W, S=S)
            want = c.decode_gaussian(n_per, lo, hi, mu[s], sd[s], P, 32 if W == 32 else 16)
            assert dec[s].tolist() == list(want), f"stream {s} support [{lo}, {hi}]"


@pytest.mark.parametrize("coder", ["ans", "range"])
@pytest.mark.parametrize("layout", ["stream_major", "symbol_major"])
def test_gaussian_per_symbol_few_long_streams_in_pieces(B, O, coder, layout):
    """Fewer streams than lanes: the decoder tabulates cdf rows piece by piece (64 MiB of rows at a time) and parks the
    coders between pieces.  40 streams x 14 000 symbols = nine pieces of 1600 symbols; one stream has an invalid model
    in a later piece (all other streams must be unaffected, and so must that stream's symbols before it)."""
    lo, hi, cfg = -100, 100, (32, 64, 24)
    n_streams, n_per = 40, 14000
    sym, mu, sd = workload(n_streams, n_per, lo, hi, 99)
    t = (lambda a: a.T) if layout == "symbol_major" else (lambda a: a)
    enc_fn = B.ans_encode_gaussian if coder == "ans" else B.range_encode_gaussian
    dec_fn = B.ans_decode_gaussian if coder == "ans" else B.range_decode_gaussian
    enc = enc_fn(dev(t(sym)), lo, hi, dev(t(mu)), dev(t(sd)), cfg, layout)
    torch.cuda.synchronize()
    for s in (0, 17, 39):
        if coder == "ans":
            c = O.AnsCoder()
            c.encode_gaussian_reverse(sym[s], lo, hi, mu[s], sd[s], 24, 32)
        else:
            c = O.RangeEncoder()
            c.encode(sym[s], [O.GaussianModel(lo, hi, m, d, 24, 32) for m, d in zip(mu[s], sd[s])], 24)
        assert enc.stream(s).tolist() == c.get_compressed().tolist()
    dec, st = dec_fn(enc, lo, hi, dev(t(mu)), dev(t(sd)), layout)
    torch.cuda.synchronize()
    assert (st.cpu().numpy() == 0).all()
    assert np.array_equal(t(dec.cpu().numpy()), sym)
    sd_bad = sd.copy()
    sd_bad[5, 9000] = -1.0
    dec, st = dec_fn(enc, lo, hi, dev(t(mu)), dev(t(sd_bad)), layout)
    torch.cuda.synchronize()
    st = st.cpu().numpy()
    assert st[5] == 1 and (np.delete(st, 5) == 0).all()
    got = t(dec.cpu().numpy())
    assert np.array_equal(np.delete(got, 5, axis=0), np.delete(sym, 5, axis=0))
    assert np.array_equal(got[5, :8960], sym[5, :8960])            # (whole 64-symbol groups before the failure are delivered)


def test_release_scratch(B):
    lo, hi = -20, 20
    sym, mu, sd = workload(100, 64, lo, hi, 3)
    enc = B.ans_encode_gaussian(dev(sym), lo, hi, dev(mu), dev(sd))
    B.release_scratch()
    dec, st = B.ans_decode_gaussian(enc, lo, hi, dev(mu), dev(sd))       # (and everything still works afterwards)
    torch.cuda.synchronize()
    assert (st.cpu().numpy() == 0).all() and np.array_equal(dec.cpu().numpy(), sym)


@pytest.mark.parametrize("P", [8, 13, 17, 18, 19, 21, 23, 24])
def test_fused_encoder_step_variants(B, O, P, knob):
    """The fused Gaussian encoder codes with entries that carry 1 / p as an f64 from P = 18 on (encode_step_inv: the quotient
    from one f64 product, within 0.27 of the true one at P = 18) and with floor(2^64 / p) below: both sides of that switch,
    the precisions next to it, symbols of probability 1 / 2^P (far off the means) next to near-certain ones; words of every
    stream against the oracle."""
    knob(CST_FUSED_MIN_STREAMS="1")
    lo, hi = -100, 100
    n_streams, n_per = 70, 16 * 12 + 5
    rng = np.random.default_rng(P)
    mu = rng.uniform(-60, 60, (n_streams, n_per))
    sd = np.exp(rng.uniform(np.log(0.05), np.log(40.0), (n_streams, n_per)))
    sym = np.clip(np.rint(mu + sd * rng.standard_normal((n_streams, n_per))), lo, hi).astype(np.int32)
    far = rng.random((n_streams, n_per)) < 0.1
    sym[far] = rng.integers(lo, hi + 1, size=int(far.sum()))                # symbols the model gives the minimal probability
    enc = B.ans_encode_gaussian(dev(sym), lo, hi, dev(mu), dev(sd), (32, 64, P))
    torch.cuda.synchronize()
    words, n_words, status = enc.to_numpy()
    assert (status == 0).all()
    for s in range(n_streams):
        c = O.AnsCoder(W=32, S=64)
        c.encode_gaussian_reverse(sym[s], lo, hi, mu[s], sd[s], P, 32)
        assert words[s, : n_words[s]].tolist() == c.get_compressed().tolist(), f"stream {s}"


@pytest.mark.parametrize("coder", ["ans", "range"])
@pytest.mark.parametrize("layout", ["stream_major", "symbol_major"])
@pytest.mark.parametrize("n_per", [16, 32, 80, 96])
def test_fused_encoder_walks_whole_tiles(B, O, coder, layout, n_per, knob):
    """Full waves (a multiple of 32 streams) over rows of whole 16-symbol tiles: the fused Gaussian encoder then walks its three
    input matrices by adding strides to one running index per lane instead of computing every item's index (the shape of the
    bench's f1 entry), and in stream-major matrices asks for the symbols of both tiles of a 128-byte line at once (one tile, one
    pair, an odd and an even number of tiles); 96 streams in both layouts and both coders, words against the oracle."""
    knob(CST_FUSED_MIN_STREAMS="1")
    lo, hi, P = -100, 100, 24
    n_streams = 96
    sym, mu, sd = workload(n_streams, n_per, lo, hi, 4242 + n_per)
    t = (lambda a: a.T) if layout == "symbol_major" else (lambda a: a)
    enc_fn = B.ans_encode_gaussian if coder == "ans" else B.range_encode_gaussian
    enc = enc_fn(dev(t(sym)), lo, hi, dev(t(mu)), dev(t(sd)), (32, 64, P), layout)
    torch.cuda.synchronize()
    words, n_words, status = enc.to_numpy()
    assert (status == 0).all()
    for s in range(n_streams):
        if coder == "ans":
            c = O.AnsCoder(W=32, S=64)
            c.encode_gaussian_reverse(sym[s], lo, hi, mu[s], sd[s], P, 32)
        else:
            c = O.RangeEncoder(W=32, S=64)
            c.encode(sym[s], [O.GaussianModel(lo, hi, m, d, P, 32) for m, d in zip(mu[s], sd[s])], P)
        assert words[s, : n_words[s]].tolist() == c.get_compressed().tolist(), f"stream {s}"


def test_float32_parameters_are_widened_as_the_reference_does(B, O):
    """the reference's Python API takes float32 parameter arrays and casts them to f64 (src/pybindings/mod.rs:187-214; its doc
    example passes float32): the batched calls do the same -- the words of the widened values, for both coders"""
    lo, hi = -100, 100
    sym, mu, sd = workload(130, 64, lo, hi, 21)
    mu32, sd32 = mu.astype(np.float32), sd.astype(np.float32)
    enc = B.ans_encode_gaussian(dev(sym), lo, hi, dev(mu32), dev(sd32))
    dec, st = B.ans_decode_gaussian(enc, lo, hi, dev(mu32), dev(sd32))
    torch.cuda.synchronize()
    assert (enc.status.cpu().numpy() == 0).all() and (st.cpu().numpy() == 0).all() and np.array_equal(dec.cpu().numpy(), sym)
    for s in (0, 64, 129):
        c = O.AnsCoder()
        c.encode_gaussian_reverse(sym[s], lo, hi, mu32[s].astype(np.float64), sd32[s].astype(np.float64), 24, 32)
        assert enc.stream(s).tolist() == c.get_compressed().tolist(), f"stream {s}"
    enc_r = B.range_encode_gaussian(dev(sym), lo, hi, dev(mu32), dev(sd32))
    dec_r, st_r = B.range_decode_gaussian(enc_r, lo, hi, dev(mu32), dev(sd32))
    assert (st_r.cpu().numpy() == 0).all() and np.array_equal(dec_r.cpu().numpy(), sym)
