"""GPU parity tests for the batched range coder (BASELINE config C4) against the CPU oracle."""
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


@pytest.mark.parametrize("cfg", [(32, 64, 12), (32, 64, 24), (16, 32, 12), (16, 32, 16)], ids=lambda c: "W%dS%dP%d" % c)
@pytest.mark.parametrize("n_streams,n_per", [(1, 1), (1, 777), (64, 32), (65, 100), (300, 257), (129, 4096), (7, 0), (128, 4096), (192, 75),
                                             (256, 2050)])
@pytest.mark.parametrize("layout", ["stream_major", "symbol_major"])
def test_range_roundtrip_parity(B, O, cfg, n_streams, n_per, layout):
    W, S, P = cfg
    cdf = O.GaussianModel(-50, 50, 3.2, 9.6, P, 32).cdf_table()
    model = B.Model.quantized_gaussian(-50, 50, 3.2, 9.6, P)
    sym = O.synth_symbols(0xC0FFEE, 0, n_streams, n_per, -50, cdf, P)
    want_words, want_n, want_status = O.rc_encode_batch(sym, -50, cdf, P, W, S)
    enc = B.range_encode(dev(sym if layout == "stream_major" else sym.T), model, cfg, layout)
    torch.cuda.synchronize()
    words, n_words, status = enc.to_numpy()
    assert status.tolist() == want_status.tolist()
    assert n_words.tolist() == want_n.tolist()
    for s in range(n_streams):
        assert words[s, : n_words[s]].tolist() == want_words[s, : want_n[s]].tolist(), f"stream {s}"
    dec, dstatus = B.range_decode(enc, model, n_per, layout)
    torch.cuda.synchronize()
    got = dec.cpu().numpy()
    assert (dstatus.cpu().numpy() == 0).all()
    assert np.array_equal(got.T if layout == "symbol_major" else got, sym)


@pytest.mark.parametrize("P", [8, 12, 16, 24])
def test_range_symbol_major_main_loops(B, O, P):
    """symbols[t][stream] through the hand-scheduled range coder statements (cst_range_{encode,decode}_loop*_sm.inc: whole
    waves, any row count): rows of 4096 + 7 symbols (ragged tail after the last full tile), a skewed model (carries,
    inverted runs -> the slow-path repeat), decoding n + 40 symbols past the end, words against the oracle."""
    n_streams, n_per = 320, 4096 + 7
    rng = np.random.default_rng(P)
    n_sym = 40
    probs = rng.dirichlet(np.ones(n_sym) * 0.3) if P > 8 else rng.dirichlet(np.ones(n_sym) * 2.0)
    cdf = O.categorical_fast_cdf(probs, P)
    model = B.Model.from_cdf(cdf, -7, P)
    sym = O.synth_symbols(21, 0, n_streams, n_per, -7, cdf, P)
    want_words, want_n, want_status = O.rc_encode_batch(sym, -7, cdf, P)
    d_sym = dev(sym.T)
    enc = B.range_encode(d_sym, model, (32, 64, P), "symbol_major")
    torch.cuda.synchronize()
    words, n_words, status = enc.to_numpy()
    assert status.tolist() == want_status.tolist() and n_words.tolist() == want_n.tolist()
    width = min(words.shape[1], want_words.shape[1])
    assert int(n_words.max()) <= width
    mask = np.arange(width, dtype=np.uint32)[None, :] < n_words[:, None]
    assert np.array_equal(np.where(mask, words[:, :width], 0), np.where(mask, want_words[:, :width], 0))
    dec, dstatus = B.range_decode(enc, model, n_per, "symbol_major")
    torch.cuda.synchronize()
    assert (dstatus.cpu().numpy() == 0).all() and np.array_equal(dec.cpu().numpy().T, sym)
    # past the end: the reference's decoder keeps going on zeros (queue.rs:1020-1024)
    # (until a quantile of 2^P or more turns up: then the reference returns InvalidData, queue.rs:989-993, and the stream's
    # remaining symbols are undefined)
    more, more_st = B.range_decode(enc, model, n_per + 40, "symbol_major")
    want_more, want_more_st = O.rc_decode_batch(want_words, want_n, n_per + 40, -7, cdf, P)
    torch.cuda.synchronize()
    assert more_st.cpu().numpy().tolist() == want_more_st.tolist()
    fine = want_more_st == 0
    assert fine.sum() > 0 and np.array_equal(more.cpu().numpy().T[fine], want_more[fine])
    assert np.array_equal(more.cpu().numpy().T[:, :n_per], sym)
    # an impossible symbol in one stream flags that stream only
    bad = sym.copy()
    bad[77, 1000] = -8
    st = B.range_encode(dev(bad.T), model, (32, 64, P), "symbol_major").status.cpu().numpy()
    assert st[77] == 1 and (np.delete(st, 77) == 0).all()


def test_range_skewed_model_exercises_carry_paths(B, O):
    """Very skewed tables make long runs of 0xFFFFFFFF / carries far more likely (lazy carry, queue.rs:126-142)."""
    P = 12
    cdf = np.array([0, 4093, 4094, 4095, 4096], dtype=np.uint32)
    model = B.Model.from_cdf(cdf, 0, P)
    rng = np.random.default_rng(3)
    sym = rng.choice(4, size=(512, 3000), p=[0.97, 0.01, 0.01, 0.01]).astype(np.int32)
    sym[:, :50] = 3  # push `lower` close to the top of the range
    want_words, want_n, _ = O.rc_encode_batch(sym, 0, cdf, P)
    enc = B.range_encode(dev(sym), model, (32, 64, P))
    dec, st = B.range_decode(enc, model, 3000)
    torch.cuda.synchronize()
    words, n_words, status = enc.to_numpy()
    assert (status == 0).all() and n_words.tolist() == want_n.tolist()
    for s in range(512):
        assert words[s, : n_words[s]].tolist() == want_words[s, : want_n[s]].tolist()
    assert np.array_equal(dec.cpu().numpy(), sym)


@pytest.mark.parametrize("P", [12, 24])
def test_range_invalid_data(B, O, P):
    """Random words decoded with a model whose last quantiles are unreachable give InvalidData
    exactly where the oracle says (queue.rs:989-993); full waves go through the hand-scheduled loops, whose symbol
    check must send them to the exact step."""
    cdf = O.GaussianModel(-50, 50, 3.2, 9.6, P, 32).cdf_table()
    model = B.Model.quantized_gaussian(-50, 50, 3.2, 9.6, P)
    rng = np.random.default_rng(8)
    words = rng.integers(0, 2 ** 32, (200, 40), dtype=np.uint64).astype(np.uint32)
    n_words = rng.integers(0, 41, 200).astype(np.uint32)
    want, want_status = O.rc_decode_batch(words, n_words, 64, -50, cdf, P)
    enc = B.EncodedBatch(dev(words.view(np.int32)), dev(n_words.view(np.int32)), torch.zeros(200, dtype=torch.int32, device="cuda"), (32, 64, P))
    got, status = B.range_decode(enc, model, 64)
    torch.cuda.synchronize()
    assert status.cpu().numpy().tolist() == want_status.tolist()
    ok = want_status == 0
    assert np.array_equal(got.cpu().numpy()[ok], want[ok])


@pytest.mark.parametrize("P", [12, 24])
def test_range_full_size_c4(B, O, P):
    """Config C4 at full size, both precisions of the bench: 65 536 x 4096 round trip + sampled bit-exactness."""
    n_streams, n_per = 65536, 4096
    cdf = O.GaussianModel(-50, 50, 3.2, 9.6, P, 32).cdf_table()
    model = B.Model.quantized_gaussian(-50, 50, 3.2, 9.6, P)
    base = O.synth_symbols(0xC0FFEE, 0, 256, n_per, -50, cdf, P)
    dsym = dev(base).repeat(n_streams // 256, 1)
    shift = torch.arange(n_streams, device="cuda") // 256
    idx = (torch.arange(n_per, device="cuda")[None, :] + 7 * shift[:, None]) % n_per
    dsym = torch.gather(dsym, 1, idx).contiguous()
    enc = B.range_encode(dsym, model, (32, 64, P))
    dec, status = B.range_decode(enc, model, n_per)
    torch.cuda.synchronize()
    assert int(enc.status.abs().sum().item()) == 0 and int(status.abs().sum().item()) == 0
    assert torch.equal(dec, dsym)
    sample = [0, 255, 256, 40000, 65535]
    want_words, want_n, _ = O.rc_encode_batch(dsym[sample].cpu().numpy(), -50, cdf, P)
    for k, s in enumerate(sample):
        assert enc.stream(s).tolist() == want_words[k, : want_n[k]].tolist()


def test_range_alphabet_too_large_for_lds(B, O):
    """6000 symbols at P = 24: the encoder entries are read from HBM / L2, the decoder uses its bucket index."""
    P = 24
    rng = np.random.default_rng(77)
    cdf = O.categorical_fast_cdf(rng.dirichlet(np.ones(6000) * 0.2), P)
    model = B.Model.from_cdf(cdf, -3000, P)
    sym = O.synth_symbols(9, 0, 150, 200, -3000, cdf, P)
    want_words, want_n, want_status = O.rc_encode_batch(sym, -3000, cdf, P)
    enc = B.range_encode(dev(sym), model, (32, 64, P))
    torch.cuda.synchronize()
    words, n_words, status = enc.to_numpy()
    assert status.tolist() == want_status.tolist() and n_words.tolist() == want_n.tolist()
    for s in range(150):
        assert words[s, : n_words[s]].tolist() == want_words[s, : want_n[s]].tolist()
    dec, dstatus = B.range_decode(enc, model, 200)
    torch.cuda.synchronize()
    assert (dstatus.cpu().numpy() == 0).all() and np.array_equal(dec.cpu().numpy(), sym)


def _straddling_streams(cdf, P, n_streams, n_per, seed, stay):
    """Symbols chosen by following the encoder's interval (queue.rs:612-705 in Python integers): while the interval
    straddles a word boundary, the symbol whose bin contains the boundary is taken with probability `stay` -- Inverted
    situations of many held-back words (queue.rs:126-142), far beyond what model-distributed data ever produces --
    and another one otherwise (resolution with or without a carry)."""
    rng = np.random.default_rng(seed)
    n = len(cdf) - 1
    top = 1 << 64
    out = np.zeros((n_streams, n_per), dtype=np.int32)
    longest = 0
    for s in range(n_streams):
        lower, rng_ = 0, top - 1
        held = 0
        for t in range(n_per):
            scale = rng_ >> P
            pick = None
            if lower + rng_ >= top and rng.random() < stay:
                for i in range(n):
                    if lower + scale * int(cdf[i]) < top <= lower + scale * int(cdf[i + 1]):
                        pick = i
            if pick is None:
                pick = int(rng.integers(0, n))
            out[s, t] = pick
            lower = lower + scale * int(cdf[pick])
            rng_ = scale * int(cdf[pick + 1] - cdf[pick])
            if lower >= top:
                lower -= top
            if lower + rng_ < top:
                held = 0
            if rng_ < (1 << 32):
                lower = (lower << 32) % top
                rng_ <<= 32
                held = held + 1 if lower + rng_ >= top else 0
                longest = max(longest, held)
    return out, longest


@pytest.mark.parametrize("P", [12, 24])
def test_range_long_inverted_runs(B, O, P):
    """Carries that travel through many held-back words, some of which have already left the LDS ring for HBM."""
    probs = np.array([1, 3, 1 << (P - 2), (1 << P) - 8 - (1 << (P - 2)), 2, 2], dtype=np.int64)
    cdf = np.concatenate([[0], np.cumsum(probs)]).astype(np.uint32)
    model = B.Model.from_cdf(cdf, 0, P)
    sym, longest = _straddling_streams(cdf, P, 192, 640, 5 + P, 0.995)
    assert longest >= 24          # (the fixture does what it is for: runs longer than a 64-byte group)
    sym[100:] = np.where(np.random.default_rng(1).random(sym[100:].shape) < 0.5, sym[100:], 3)
    want_words, want_n, _ = O.rc_encode_batch(sym, 0, cdf, P)
    enc = B.range_encode(dev(sym), model, (32, 64, P))
    dec, st = B.range_decode(enc, model, sym.shape[1])
    torch.cuda.synchronize()
    words, n_words, status = enc.to_numpy()
    assert (status == 0).all() and n_words.tolist() == want_n.tolist()
    for s in range(len(sym)):
        assert words[s, : n_words[s]].tolist() == want_words[s, : want_n[s]].tolist(), f"stream {s}"
    assert np.array_equal(dec.cpu().numpy(), sym)


@pytest.mark.parametrize("P", [13, 16, 20, 24])
def test_range_bucket_entry_decoder_walks_the_tails(B, O, P):
    """12 < P <= 24: the decoder's lookup is one 16-byte bucket entry + a walk over the cdf table beyond its third symbol.
    256 symbols, 100 + 155 of them with probability 1 / 2^P crowded into the first and last bucket, data drawn uniformly
    over the alphabet; full and partial waves, and decoding past the end of the data."""
    n = 256
    probs = np.ones(n, dtype=np.int64)
    probs[100] = (1 << P) - (n - 1) - 500
    probs[37] += 300; probs[200] += 200
    cdf = np.concatenate([[0], np.cumsum(probs)]).astype(np.uint32)
    model = B.Model.from_cdf(cdf, -7, P)
    rng = np.random.default_rng(100 + P)
    for n_streams, n_per in ((192, 640), (70, 96), (64, 64), (3, 100)):
        sym = (rng.integers(0, n, (n_streams, n_per)) - 7).astype(np.int32)
        sym[:, ::3] = 93
        want_words, want_n, _ = O.rc_encode_batch(sym, -7, cdf, P)
        enc = B.range_encode(dev(sym), model, (32, 64, P))
        torch.cuda.synchronize()
        words, n_words, status = enc.to_numpy()
        assert (status == 0).all() and n_words.tolist() == want_n.tolist()
        for s in range(n_streams):
            assert words[s, : n_words[s]].tolist() == want_words[s, : want_n[s]].tolist()
        dec, st = B.range_decode(enc, model, n_per)
        torch.cuda.synchronize()
        assert (st.cpu().numpy() == 0).all() and np.array_equal(dec.cpu().numpy(), sym)
        want_more, want_st = O.rc_decode_batch(want_words, want_n, n_per + 36, -7, cdf, P)
        more, st3 = B.range_decode(enc, model, n_per + 36)
        torch.cuda.synchronize()
        assert st3.cpu().numpy().tolist() == want_st.tolist()
        ok = want_st == 0
        assert np.array_equal(more.cpu().numpy()[ok], want_more[ok])


@pytest.mark.parametrize("n_per", [128, 131, 4095, 4100])
@pytest.mark.parametrize("base_shift", [0, 1, 3])
@pytest.mark.parametrize("P", [12, 24])
def test_range_rows_of_any_length(B, O, n_per, base_shift, P):
    """The range coder's main-loop statements on rows that do not start on cache-line boundaries (row_skew, cst_ans_kernels.hpp:
    the symbols in front of a row's next 128-byte boundary and behind its last whole tile are coded outside the statements, which
    continue from the state they left); symbol buffers 4 and 12 bytes off a 16-byte boundary, full waves plus a partial one,
    a skewed model whose carries travel (the encoder's slow-path repeat from the first symbol); words against the oracle,
    decoded symbols against the input, nothing written outside the rows."""
    n_streams = 200
    rng = np.random.default_rng(P + n_per)
    cdf = O.categorical_fast_cdf(rng.dirichlet(np.ones(40) * 0.3), P)
    model = B.Model.from_cdf(cdf, -7, P)
    sym = O.synth_symbols(n_per, 0, n_streams, n_per, -7, cdf, P)
    want_words, want_n, want_status = O.rc_encode_batch(sym, -7, cdf, P)
    src = torch.zeros(n_streams * n_per + 8, dtype=torch.int32, device="cuda")
    src[base_shift: base_shift + sym.size] = dev(sym).reshape(-1)
    enc = B.range_encode(src[base_shift: base_shift + sym.size].view(n_streams, n_per), model, (32, 64, P))
    torch.cuda.synchronize()
    words, n_words, status = enc.to_numpy()
    assert status.tolist() == want_status.tolist() and n_words.tolist() == want_n.tolist()
    for i in range(n_streams):
        assert words[i, : n_words[i]].tolist() == want_words[i, : want_n[i]].tolist(), i
    buf = torch.full((n_streams * n_per + 8,), -99, dtype=torch.int32, device="cuda")
    out = buf[base_shift: base_shift + n_streams * n_per].view(n_streams, n_per)
    dec, st = B.range_decode(enc, model, n_per, out=out)
    torch.cuda.synchronize()
    assert (st.cpu().numpy() == 0).all() and np.array_equal(out.cpu().numpy(), sym)
    assert (buf[:base_shift].cpu().numpy() == -99).all() and (buf[base_shift + n_streams * n_per:].cpu().numpy() == -99).all()
