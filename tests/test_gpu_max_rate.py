"""Adversarial RATES (round 5): data made of symbols of probability 2^-P -- P bits each, the most a symbol can cost -- with a likely
symbol here and there, so that the words are consumed at the maximum the decoders' word windows are sized for (12 words per 32-symbol
tile at (32,64,12), 24 at P = 24 and for 16-bit words) and the refills fall on every phase of a tile.  All-tail data alone emits on the
same steps tile after tile and missed a stale read of the first refill candidate of a tile in the per-stream-table decoder (found in
round 5: wrong symbols at position 1 of a tile, only above ~11.5 bits per symbol).  Every decoder family, against the input and the
CPU oracle's words."""
import os

import numpy as np
import pytest

from kernel_names import with_jump  # noqa: E402

pytestmark = pytest.mark.gpu
# (runs of the suite through the alternate kernel paths -- profiles/r05_alt_paths.txt -- do not take the kernels the tests name)
ALT = any(os.environ.get(k) for k in ("CST_NO_N8", "CST_NO_PC_ENCODER", "CST_SMALL_KERNELS", "CST_PC_COMBINED", "CST_NO_PC_WIDE"))
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


def spiky_cdf(n, P):
    """symbol 0 takes everything the n - 1 others (probability 2^-P each) leave"""
    cdf = np.zeros(n + 1, np.uint32)
    cdf[1] = (1 << P) - (n - 1)
    cdf[2:] = cdf[1] + np.arange(1, n, dtype=np.uint32)
    return cdf


def high_rate_symbols(rng, n_streams, n_per, n, frac):
    tails = rng.integers(1, n, (n_streams, n_per), dtype=np.int32)
    return np.where(rng.random((n_streams, n_per)) < frac, 0, tails).astype(np.int32)


CASES = [(coder, cfg) for coder in ("ans", "range") for cfg in ((32, 64, 12), (32, 64, 24), (32, 64, 16), (16, 32, 12), (32, 64, 8))]


@pytest.mark.parametrize("frac", [0.0, 0.01, 0.05, 0.2])
@pytest.mark.parametrize("coder,cfg", CASES, ids=lambda v: v if isinstance(v, str) else "W%dS%dP%d" % v)
def test_shared_table_decoders_at_the_maximum_rate(B, O, coder, cfg, frac):
    W, S, P = cfg
    n = 101 if P >= 8 else 16
    cdf = spiky_cdf(n, P)
    model = B.Model.from_cdf(cdf, 0, P)
    rng = np.random.default_rng(int(frac * 1000) + P)
    sym = high_rate_symbols(rng, 192, 2048, n, frac)
    if coder == "ans":
        want_words, want_n, _ = O.ans_encode_batch(sym, 0, cdf, P, W, S)
        enc = B.ans_encode(dev(sym), model, cfg)
    else:
        want_words, want_n, _ = O.rc_encode_batch(sym, 0, cdf, P, W, S)
        enc = B.range_encode(dev(sym), model, cfg)
    words, n_words, status = enc.to_numpy()
    assert (status == 0).all() and n_words.tolist() == want_n.tolist()
    for s in range(192):
        assert np.array_equal(words[s, : n_words[s]], want_words[s, : want_n[s]]), s
    checks = []
    if coder == "ans":
        checks.append(("plain", *B.ans_decode(enc, model, 2048, cold=False)))
        checks.append(("cold", *B.ans_decode(enc, model, 2048, cold=True)))
        symT = dev(np.ascontiguousarray(sym.T))
        encT = B.ans_encode(symT, model, cfg, layout="symbol_major")
        dT, sT = B.ans_decode(encT, model, 2048, layout="symbol_major")
        checks.append(("symbol-major", dT.t().contiguous(), sT))
        if W == 16:
            pk = B.ans_encode(dev(sym), model, cfg, packed16=True)
            checks.append(("packed16", *B.ans_decode(pk, model, 2048)))
        e2, ck = B.ans_encode_checkpointed(dev(sym), model, 512, cfg)
        assert torch.equal(e2.n_words, enc.n_words) and all(e2.stream(s).tolist() == enc.stream(s).tolist() for s in (0, 77, 191))
        d2, s2 = B.ans_decode_checkpointed(e2, ck, model, 2048)
        checks.append(("jump points", d2, s2))
    else:
        checks.append(("plain", *B.range_decode(enc, model, 2048)))
        e2, ck = B.range_encode_checkpointed(dev(sym), model, 512, cfg)
        d2, s2 = B.range_decode_checkpointed(e2, ck, model, 2048)
        checks.append(("jump points", d2, s2))
    torch.cuda.synchronize()
    for name, dec, st in checks:
        assert int(st.abs().sum()) == 0, name
        assert np.array_equal(dec.cpu().numpy(), sym), name


@pytest.mark.parametrize("frac", [0.0, 0.01, 0.03, 0.1, 0.3])
def test_per_stream_table_decoders_at_the_maximum_rate(B, O, frac):
    n, k, P = 320, 2048, 12
    rng = np.random.default_rng(int(frac * 100))
    mu_h, sd_h = rng.uniform(-5, 5, n), rng.uniform(0.4, 0.8, n)
    model = B.Model.quantized_gaussian_per_stream(-127, 127, dev(mu_h), dev(sd_h), P)
    tails = rng.choice(np.concatenate([np.arange(-127, -40), np.arange(40, 128)]), (n, k)).astype(np.int32)
    likely = np.rint(mu_h)[:, None].astype(np.int32) + np.zeros((n, k), np.int32)
    sym = np.where(rng.random((n, k)) < frac, likely, tails).astype(np.int32)
    cdfs = np.stack([O.GaussianModel(-127, 127, a, b, P, 32).cdf_table() for a, b in zip(mu_h, sd_h)])
    want_words, want_n, _ = O.ans_encode_batch(sym, -127, cdfs, P)
    enc = B.ans_encode(dev(sym), model, (32, 64, P))
    words, n_words, status = enc.to_numpy()
    assert (status == 0).all() and n_words.tolist() == want_n.tolist()
    for s in (0, 100, 319):
        assert words[s, : n_words[s]].tolist() == want_words[s, : want_n[s]].tolist()
    dec, st = B.ans_decode(enc, model, k)
    assert int(st.abs().sum()) == 0 and np.array_equal(dec.cpu().numpy(), sym), "plain"
    for chunks in (4, 8):
        e2, ck = B.ans_encode_checkpointed(dev(sym), model, k // chunks, (32, 64, P))
        d2, s2 = B.ans_decode_checkpointed(e2, ck, model, k)
        assert int(s2.abs().sum()) == 0 and np.array_equal(d2.cpu().numpy(), sym), f"k = {chunks}"


@pytest.mark.parametrize("jp", [0, "auto"], ids=["plain", "auto_jump"])
@pytest.mark.parametrize("frac", [0.0, 0.01, 0.05, 0.2])
@pytest.mark.parametrize("P", [12, 8])
def test_int8_native_kernels_at_the_maximum_rate(B, O, P, frac, jp):
    """the loops that read / write int8 matrices themselves (cst_ans_n8.hip, ans_encode_pc_n8_kernel): 12 words per 32-symbol tile,
    refills on every phase, four tiles per pass of the decoder's statement"""
    n = 101
    cdf = spiky_cdf(n, P)
    model = B.Model.from_cdf(cdf, 0, P)
    rng = np.random.default_rng(int(frac * 1000) + P + 7)
    sym = high_rate_symbols(rng, 256, 2048, n, frac)
    want_words, want_n, _ = O.ans_encode_batch(sym, 0, cdf, P, 32, 64)
    d = dev(sym.astype(np.int8))
    enc = B.ans_encode(d, model, (32, 64, P), jump_points=jp)        # (auto: max-rate words THROUGH jump points noted on the way)
    assert ALT or B.last_kernel() == with_jump("ans_encode_pc_n8_kernel", enc)
    assert ALT or os.environ.get("CST_AUTO_JUMP") or (enc.jump is not None) == (jp == "auto")
    words, n_words, status = enc.to_numpy()
    assert (status == 0).all() and n_words.tolist() == want_n.tolist()
    for s in range(256):
        assert np.array_equal(words[s, : n_words[s]], want_words[s, : want_n[s]]), f"stream {s}"
    dec, st = B.ans_decode(enc, model, 2048, dtype=torch.int8)
    assert ALT or B.last_kernel() == "ans_decode_n8_kernel"
    assert (st.cpu().numpy() == 0).all() and torch.equal(dec, d)


@pytest.mark.parametrize("frac", [0.0, 0.01, 0.05])
@pytest.mark.parametrize("dtype", ["int32", "int8"])
def test_small_footprint_kernels_at_the_maximum_rate(B, O, dtype, frac):
    """the kernels for more than one wave per SIMD (cst_ans_small.hip; int8: ans_decode_small_n8_kernel) are only taken by batches of
    more streams than 256 per CU -- the other tests of this file never reach them.  Their decoder shared the stale first-candidate
    read of the per-stream-table decoder (fixed in round 5: scripts/gen_decode_loop_small.py, tail(last=True))."""
    from constriction_amd import _native as N
    P, n = 12, 101
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    n_streams, n_per = cus * 256 + 512, 256
    cdf = spiky_cdf(n, P)
    model = B.Model.from_cdf(cdf, 0, P)
    rng = np.random.default_rng(int(frac * 1000) + 99)
    sym = high_rate_symbols(rng, n_streams, n_per, n, frac)
    d = dev(sym if dtype == "int32" else sym.astype(np.int8))
    enc = B.ans_encode(d, model, (32, 64, P))
    dec, st = B.ans_decode(enc, model, n_per, dtype=d.dtype)
    assert ALT or B.last_kernel() == ("ans_decode_small_kernel" if dtype == "int32" else "ans_decode_small_n8_kernel")
    assert (st.cpu().numpy() == 0).all() and (enc.status.cpu().numpy() == 0).all()
    wrong = (dec != d).any(dim=1).nonzero().flatten()
    assert wrong.numel() == 0, f"{wrong.numel()} streams decode wrongly, first {wrong[:4].tolist()}"
    words, n_words, _ = enc.to_numpy()
    for s in (0, 4097, n_streams - 1):
        w, nw, _ = O.ans_encode_batch(sym[s: s + 1], 0, cdf, P, 32, 64)
        assert np.array_equal(words[s, : n_words[s]], w[0, : nw[0]]), f"stream {s}"


@pytest.mark.parametrize("frac", [0.0, 0.02, 0.2])
@pytest.mark.parametrize("coder", ["ans", "range"])
@pytest.mark.parametrize("geo", ["big", "small"])
def test_per_symbol_gaussian_decoders_at_the_maximum_rate(B, O, knob, coder, geo, frac):
    """f1 (every symbol its own mean and std) at 24 bits per symbol: needle-thin models far from their symbol -- the leaky
    quantizer's floor of 2^-24 -- with a likely symbol here and there; both geometries of the lane decoder (CST_LANE_GEO), both
    coders, and the jump-point form of the ANS coder"""
    knob(CST_LANE_GEO=geo)
    rng = np.random.default_rng(int(frac * 1000) + 5)
    n_streams, n_per, lo, hi = 128, 512, -100, 100
    sym = rng.integers(60, 101, (n_streams, n_per)).astype(np.int32)
    mu = np.full((n_streams, n_per), -90.0)
    sd = np.full((n_streams, n_per), 0.05)
    likely = rng.random((n_streams, n_per)) < frac
    mu = np.where(likely, sym.astype(np.float64), mu)
    sd = np.where(likely, 0.3, sd)
    enc_f, dec_f = (B.ans_encode_gaussian, B.ans_decode_gaussian) if coder == "ans" else (B.range_encode_gaussian, B.range_decode_gaussian)
    enc = enc_f(dev(sym), lo, hi, dev(mu), dev(sd))
    dec, st = dec_f(enc, lo, hi, dev(mu), dev(sd))
    assert (st.cpu().numpy() == 0).all() and (enc.status.cpu().numpy() == 0).all()
    assert np.array_equal(dec.cpu().numpy(), sym)
    if frac == 0.0:
        assert float(enc.n_words.float().mean()) > 0.74 * n_per          # 24 bits per symbol: three words per four symbols
    if coder == "ans":
        for s in (0, 77, 127):
            c = O.AnsCoder()
            c.encode_gaussian_reverse(sym[s], lo, hi, mu[s], sd[s], 24, 32)
            assert enc.stream(s).tolist() == c.get_compressed().tolist(), f"stream {s}"
        e2, ck = B.ans_encode_gaussian_checkpointed(dev(sym), lo, hi, dev(mu), dev(sd), 128)
        d2, s2 = B.ans_decode_gaussian_checkpointed(e2, ck, lo, hi, dev(mu), dev(sd))
        assert (s2.cpu().numpy() == 0).all() and np.array_equal(d2.cpu().numpy(), sym)


@pytest.mark.parametrize("frac", [0.0, 0.05])
def test_ragged_decoder_at_the_maximum_rate(B, O, frac):
    """many small coders in one launch (cst_ans_ragged.hip) on all-tail data"""
    P, n = 24, 64
    cdf = spiky_cdf(n, P)
    model = B.Model.from_cdf(cdf, 0, P)
    rng = np.random.default_rng(int(frac * 1000) + 3)
    seqs = [high_rate_symbols(rng, 1, int(k), n, frac)[0] for k in rng.integers(1, 700, 300)]
    symbols, offsets = B.ragged(seqs)
    enc = B.ans_encode_ragged(symbols, offsets, model, (32, 64, P))
    dec, st = B.ans_decode_ragged(enc, model, offsets)
    assert (st.cpu().numpy() == 0).all() and torch.equal(dec, symbols)


# ---- the same rates on PACKED words: every stream starts at an arbitrary word of one buffer, so the word windows run at every
# ---- alignment phase of their 16-byte chunks (slabs are 64-byte aligned: the tests above only see phase 0) ----

def gapped(words, n_words, rng, max_gap=19):
    """(packed int32 buffer, offsets int64[n + 1], n_words) with 0 .. max_gap - 1 words of junk in front of every stream"""
    n_streams = len(n_words)
    gaps = rng.integers(0, max_gap, n_streams)
    offsets = np.zeros(n_streams + 1, np.int64)
    offsets[1:] = np.cumsum(gaps + n_words)
    offsets[:-1] += gaps                                   # stream s occupies [offsets[s], offsets[s] + n_words[s])
    total = int((gaps + n_words).sum()) + 64
    buf = rng.integers(1, 2 ** 32, total, dtype=np.uint64).astype(np.uint32)
    for s in range(n_streams):
        buf[offsets[s]: offsets[s] + n_words[s]] = words[s, : n_words[s]]
    return dev(buf.view(np.int32)), dev(offsets), dev(n_words.astype(np.int32))


@pytest.mark.parametrize("frac", [0.0, 0.01, 0.05, 0.2])
@pytest.mark.parametrize("coder,cfg", CASES, ids=lambda v: v if isinstance(v, str) else "W%dS%dP%d" % v)
def test_shared_table_decoders_at_the_maximum_rate_on_packed_words(B, O, coder, cfg, frac):
    W, S, P = cfg
    n = 101 if P >= 8 else 16
    cdf = spiky_cdf(n, P)
    model = B.Model.from_cdf(cdf, 0, P)
    rng = np.random.default_rng(int(frac * 1000) + P + 31)
    sym = high_rate_symbols(rng, 192, 2048, n, frac)
    enc = (B.ans_encode if coder == "ans" else B.range_encode)(dev(sym), model, cfg)
    words, n_words, status = enc.to_numpy()
    assert (status == 0).all()
    if W == 16:
        words = words.astype(np.uint32)                    # (one 16-bit word per 32-bit slot in the packed form too)
    buf, offsets, nw = gapped(words.view(np.uint32), n_words, rng)
    if coder == "ans":
        for cold in (False, True):
            dec, st = B.ans_decode((buf, nw), model, 2048, offsets=offsets, config=cfg, cold=cold)
            assert int(st.abs().sum()) == 0 and np.array_equal(dec.cpu().numpy(), sym), f"cold={cold} [{B.last_kernel()}]"
        if W == 32 and P <= 12:
            d8, st = B.ans_decode((buf, nw), model, 2048, offsets=offsets, config=cfg, dtype=torch.int8)
            assert int(st.abs().sum()) == 0 and np.array_equal(d8.cpu().numpy(), sym.astype(np.int8)), B.last_kernel()
    else:
        dec, st = B.range_decode((buf, nw), model, 2048, offsets=offsets, config=cfg)
        assert int(st.abs().sum()) == 0 and np.array_equal(dec.cpu().numpy(), sym), B.last_kernel()


@pytest.mark.parametrize("frac", [0.0, 0.01, 0.05])
@pytest.mark.parametrize("dtype", ["int32", "int8"])
def test_small_footprint_decoders_at_the_maximum_rate_on_packed_words(B, O, dtype, frac):
    P, n = 12, 101
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    n_streams, n_per = cus * 256 + 512, 256
    cdf = spiky_cdf(n, P)
    model = B.Model.from_cdf(cdf, 0, P)
    rng = np.random.default_rng(int(frac * 1000) + 77)
    sym = high_rate_symbols(rng, n_streams, n_per, n, frac)
    enc = B.ans_encode(dev(sym), model, (32, 64, P))
    words, n_words, status = enc.to_numpy()
    assert (status == 0).all()
    buf, offsets, nw = gapped(words.view(np.uint32), n_words, rng, max_gap=5)
    dt = torch.int32 if dtype == "int32" else torch.int8
    dec, st = B.ans_decode((buf, nw), model, n_per, offsets=offsets, config=(32, 64, P), dtype=dt)
    assert ALT or B.last_kernel() == ("ans_decode_small_kernel" if dtype == "int32" else "ans_decode_small_n8_kernel")
    wrong = (dec != dev(sym).to(dt)).any(dim=1).nonzero().flatten()
    assert int(st.abs().sum()) == 0 and wrong.numel() == 0, f"{wrong.numel()} streams decode wrongly, first {wrong[:4].tolist()}"


@pytest.mark.parametrize("frac", [0.0, 0.03, 0.3])
def test_per_stream_table_decoder_at_the_maximum_rate_on_packed_words(B, O, frac):
    n, k, P = 320, 2048, 12
    rng = np.random.default_rng(int(frac * 100) + 9)
    mu_h, sd_h = rng.uniform(-5, 5, n), rng.uniform(0.4, 0.8, n)
    model = B.Model.quantized_gaussian_per_stream(-127, 127, dev(mu_h), dev(sd_h), P)
    tails = rng.choice(np.concatenate([np.arange(-127, -40), np.arange(40, 128)]), (n, k)).astype(np.int32)
    likely = np.rint(mu_h)[:, None].astype(np.int32) + np.zeros((n, k), np.int32)
    sym = np.where(rng.random((n, k)) < frac, likely, tails).astype(np.int32)
    enc = B.ans_encode(dev(sym), model, (32, 64, P))
    words, n_words, status = enc.to_numpy()
    assert (status == 0).all()
    buf, offsets, nw = gapped(words.view(np.uint32), n_words, rng)
    dec, st = B.ans_decode((buf, nw), model, k, offsets=offsets, config=(32, 64, P))
    assert int(st.abs().sum()) == 0 and np.array_equal(dec.cpu().numpy(), sym), B.last_kernel()


@pytest.mark.parametrize("frac", [0.0, 0.02])
@pytest.mark.parametrize("coder", ["ans", "range"])
@pytest.mark.parametrize("geo", ["big", "small"])
def test_per_symbol_gaussian_decoders_at_the_maximum_rate_on_packed_words(B, O, knob, coder, geo, frac):
    knob(CST_LANE_GEO=geo)
    rng = np.random.default_rng(int(frac * 1000) + 6)
    n_streams, n_per, lo, hi = 128, 512, -100, 100
    sym = rng.integers(60, 101, (n_streams, n_per)).astype(np.int32)
    likely = rng.random((n_streams, n_per)) < frac
    mu = np.where(likely, sym.astype(np.float64), -90.0)
    sd = np.where(likely, 0.3, 0.05)
    enc_f, dec_f = (B.ans_encode_gaussian, B.ans_decode_gaussian) if coder == "ans" else (B.range_encode_gaussian, B.range_decode_gaussian)
    enc = enc_f(dev(sym), lo, hi, dev(mu), dev(sd))
    words, n_words, status = enc.to_numpy()
    assert (status == 0).all()
    buf, offsets, nw = gapped(words.view(np.uint32), n_words, rng)
    dec, st = dec_f((buf, nw), lo, hi, dev(mu), dev(sd), offsets=offsets, config=(32, 64, 24))
    assert int(st.abs().sum()) == 0 and np.array_equal(dec.cpu().numpy(), sym), B.last_kernel()


@pytest.mark.parametrize("jp", [0, "auto"], ids=["plain", "auto_jump"])
@pytest.mark.parametrize("frac", [0.0, 0.01, 0.2])
@pytest.mark.parametrize("n_streams", [256, "small"])
def test_int16_native_decoders_at_the_maximum_rate(B, O, n_streams, frac, jp):
    """the decoders that write int16 matrices themselves (two tiles per pass), one and two waves per SIMD, slabs and packed words"""
    P, n = 12, 101
    if n_streams == "small":
        n_streams, n_per = torch.cuda.get_device_properties(0).multi_processor_count * 256 + 320, 256
    else:
        n_per = 2048
    cdf = spiky_cdf(n, P)
    model = B.Model.from_cdf(cdf, 1000, P)                 # symbols 1000 .. 1100: int16, not int8
    rng = np.random.default_rng(int(frac * 1000) + 17)
    sym = high_rate_symbols(rng, n_streams, n_per, n, frac) + 1000
    d = dev(sym.astype(np.int16))
    enc = B.ans_encode(d, model, (32, 64, P), jump_points=jp)
    assert ALT or B.last_kernel() == with_jump("ans_encode_pc_n16_kernel", enc)
    plain = B.ans_encode(dev(sym), model, (32, 64, P), jump_points=jp)
    assert torch.equal(enc.n_words, plain.n_words) and int(enc.status.abs().sum()) == 0
    used = torch.arange(plain.words.shape[1], device="cuda")[None, :] < plain.n_words[:, None]
    assert bool(((enc.words == plain.words) | ~used).all()), "the int16 encoder's words differ from the int32 encoder's"
    dec, st = B.ans_decode(enc, model, n_per, dtype=torch.int16)
    assert ALT or B.last_kernel() == ("ans_decode_small_n16_kernel" if n_streams > 65536 else "ans_decode_n16_kernel")
    assert int(st.abs().sum()) == 0 and torch.equal(dec, d)
    words, n_words, _ = enc.to_numpy()
    buf, offsets, nw = gapped(words.view(np.uint32), n_words, rng, max_gap=5)
    dec, st = B.ans_decode((buf, nw), model, n_per, offsets=offsets, config=(32, 64, P), dtype=torch.int16)
    assert int(st.abs().sum()) == 0 and torch.equal(dec, d)


@pytest.mark.parametrize("jp", [0, "auto"], ids=["plain", "auto_jump"])
@pytest.mark.parametrize("frac", [0.0, 0.01, 0.05, 0.2])
@pytest.mark.parametrize("P", [24, 16])
@pytest.mark.parametrize("dtype", ["int8", "int16"])
def test_narrow_high_precision_kernels_at_the_maximum_rate(B, O, dtype, P, frac, jp):
    """ans_decode_b16_narrow_kernel at 24 words per 32-symbol tile (P = 24: every symbol of probability 2^-24), slabs and packed words;
    the tails also take the statement's out-of-line walk over the cdf table on nearly every step"""
    n = 101
    cdf = spiky_cdf(n, P)
    model = B.Model.from_cdf(cdf, 0, P)
    rng = np.random.default_rng(int(frac * 1000) + P + 3)
    sym = high_rate_symbols(rng, 200, 2048, n, frac)
    dt = torch.int8 if dtype == "int8" else torch.int16
    d = dev(sym).to(dt)
    enc = B.ans_encode(dev(sym), model, (32, 64, P), jump_points=jp)
    narrow = B.ans_encode(d, model, (32, 64, P), jump_points=jp)            # ... and the narrow encoder (its storer moves two word groups per tile)
    assert ALT or B.last_kernel() == with_jump("ans_encode_pc_n%d_kernel<wide>" % (8 * d.element_size()), narrow)
    assert torch.equal(narrow.n_words, enc.n_words) and int(narrow.status.abs().sum()) == 0
    wa, na, _ = enc.to_numpy()
    wb, _, _ = narrow.to_numpy()
    assert all(np.array_equal(wa[s, : na[s]], wb[s, : na[s]]) for s in range(wa.shape[0]))
    dec, st = B.ans_decode(enc, model, 2048, dtype=dt)
    assert ALT or B.last_kernel() == ("ans_decode_b16_n8_kernel" if dtype == "int8" else "ans_decode_b16_n16_kernel")
    assert int(st.abs().sum()) == 0 and torch.equal(dec, d)
    words, n_words, _ = enc.to_numpy()
    buf, offsets, nw = gapped(words.view(np.uint32), n_words, rng)
    dec, st = B.ans_decode((buf, nw), model, 2048, offsets=offsets, config=(32, 64, P), dtype=dt)
    assert int(st.abs().sum()) == 0 and torch.equal(dec, d)


_B16_SMALL = {"int32": "ans_decode_b16_small_kernel", "int16": "ans_decode_b16_small_n16_kernel", "int8": "ans_decode_b16_small_n8_kernel"}


@pytest.mark.parametrize("frac", [0.0, 0.01, 0.05, 0.2])
@pytest.mark.parametrize("P", [24, 16])
@pytest.mark.parametrize("dtype", ["int32", "int16", "int8"])
def test_small_footprint_high_precision_decoders_at_the_maximum_rate(B, O, dtype, P, frac):
    """ans_decode_b16_narrow_kernel<BYTES, SMALL> (12 < P <= 24 at more than 256 streams per CU: two waves per SIMD, 16-slot word rings
    with a window every quarter tile -- 8 symbols of probability 2^-24 take the 6 words a window supplies), slabs and packed words"""
    n = 101
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    n_streams, n_per = cus * 256 + 300, 256
    cdf = spiky_cdf(n, P)
    model = B.Model.from_cdf(cdf, 0, P)
    rng = np.random.default_rng(int(frac * 1000) + P + 5)
    sym = high_rate_symbols(rng, n_streams, n_per, n, frac)
    dt = {"int32": torch.int32, "int16": torch.int16, "int8": torch.int8}[dtype]
    d = dev(sym).to(dt)
    enc = B.ans_encode(dev(sym), model, (32, 64, P))
    assert int(enc.status.abs().sum()) == 0
    words, n_words, _ = enc.to_numpy()
    for s in (0, 4097, n_streams - 1):
        w, nw, _ = O.ans_encode_batch(sym[s: s + 1], 0, cdf, P, 32, 64)
        assert np.array_equal(words[s, : n_words[s]], w[0, : nw[0]]), f"stream {s}"
    guard = torch.full((n_streams * n_per + 4096,), 77, dtype=dt, device="cuda")
    out = guard[: n_streams * n_per].view(n_streams, n_per)
    dec, st = B.ans_decode(enc, model, n_per, out=out)
    assert ALT or B.last_kernel() == _B16_SMALL[dtype]
    wrong = (dec != d).any(dim=1).nonzero().flatten()
    assert int(st.abs().sum()) == 0 and wrong.numel() == 0, f"{wrong.numel()} streams decode wrongly, first {wrong[:4].tolist()}"
    assert (guard[n_streams * n_per:] == 77).all(), "symbols were written behind the matrix"
    buf, offsets, nw = gapped(words.view(np.uint32), n_words, rng, max_gap=5)
    out.fill_(55)
    dec, st = B.ans_decode((buf, nw), model, n_per, offsets=offsets, config=(32, 64, P), out=out)
    assert ALT or B.last_kernel() == _B16_SMALL[dtype]
    wrong = (dec != d).any(dim=1).nonzero().flatten()
    assert int(st.abs().sum()) == 0 and wrong.numel() == 0, f"packed words: {wrong.numel()} streams decode wrongly, first {wrong[:4].tolist()}"


@pytest.mark.parametrize("dtype", ["int32", "int8"])
@pytest.mark.parametrize("P", [24, 13])
def test_small_footprint_high_precision_decoders_on_model_data_and_jump_points(B, O, dtype, P):
    """... on Gaussian data (the tails walk the cdf table out of line), and as the decoder of a batch with jump points (the virtual streams
    of 65 536 x k chunks are what brings a batch of one wave per SIMD above 256 streams per CU)"""
    lo, hi = -100, 100
    cdf = O.GaussianModel(lo, hi, 7.3, 11.0, P, 32).cdf_table()
    model = B.Model.from_cdf(cdf, lo, P)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    n_streams, n_per, k = cus * 128 + 64, 512, 4
    dt = torch.int32 if dtype == "int32" else torch.int8
    cdf_dev = torch.from_numpy(cdf.astype(np.int64)).cuda()
    import bench
    sym = bench.synth_symbols_device(77 + P, 0, n_streams, n_per, lo, cdf_dev, P)
    d = sym.to(dt)
    enc, ck = B.ans_encode_checkpointed(d, model, n_per // k, (32, 64, P))
    plain = B.ans_encode(sym, model, (32, 64, P))
    assert torch.equal(enc.n_words, plain.n_words) and int(enc.status.abs().sum()) == 0
    host = sym[:3].cpu().numpy()
    want_words, want_n, _ = O.ans_encode_batch(host, lo, cdf, P)
    for s in range(3):
        assert enc.stream(s).tolist() == want_words[s, : want_n[s]].tolist()
    dec, st = B.ans_decode_checkpointed(enc, ck, model, n_per, dtype=dt)
    assert ALT or B.last_kernel() == _B16_SMALL[dtype]
    assert int(st.abs().sum()) == 0 and torch.equal(dec, d)
    big = torch.cat([d, d, d])[: cus * 256 + 70]                 # ... and whole streams, a partial last wave
    enc2 = B.ans_encode(big, model, (32, 64, P))
    dec, st = B.ans_decode(enc2, model, n_per, dtype=dt)
    assert ALT or B.last_kernel() == _B16_SMALL[dtype]
    assert int(st.abs().sum()) == 0 and torch.equal(dec, big)
