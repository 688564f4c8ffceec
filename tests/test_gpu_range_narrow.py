"""int8 / int16 symbol matrices through the range coder (ABI 5: cst_range_{en,de}code_batch[_ckpt]_sym).  The reference's RangeEncoder /
RangeDecoder are generic over the symbol type (src/stream/queue.rs:612, 968; src/stream/model/quantize.rs:229-255).  int8 rows of whole
32-symbol tiles are read by the hand-scheduled encoder and written by the sub-lane decoder THEMSELVES (round 6); everything else
converts next to the int32 kernels.  Either way: the words, counts, status and jump points of the CPU oracle on the widened values, and
the input back."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ALT = any(os.environ.get(k) for k in ("CST_NO_N8", "CST_AUTO_JUMP"))
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
    cdf = np.zeros(n + 1, np.uint32)
    cdf[1] = (1 << P) - (n - 1)
    cdf[2:] = cdf[1] + np.arange(1, n, dtype=np.uint32)
    return cdf


def _check_words(O, enc, sym, lo, cdf, P):
    want_words, want_n, want_st = O.rc_encode_batch(sym, lo, cdf, P)
    words, n_words, status = enc.to_numpy()
    assert status.tolist() == want_st.tolist() and n_words.tolist() == want_n.tolist()
    for s in range(len(sym)):
        assert np.array_equal(words[s, : n_words[s]], want_words[s, : want_n[s]]), f"stream {s}"


@pytest.mark.parametrize("jump", [0, 2, 8, "auto"])
@pytest.mark.parametrize("P", [8, 12, 16, 24])
@pytest.mark.parametrize("support", [(-50, 50), (-128, 127), (5, 60), (0, 1)], ids=lambda s: "%d..%d" % s)
@pytest.mark.parametrize("n_streams,n_per", [(64, 32), (256, 256), (300, 1024), (1024, 2048), (65, 4096)])
def test_int8_matrices_inside_the_range_coder_loops(B, O, n_streams, n_per, support, P, jump):
    lo, hi = support
    if hi - lo + 1 > (1 << P) // 2:
        pytest.skip("alphabet too large for the precision")
    if jump not in (0, "auto") and (n_per % jump != 0 or (n_per // jump) % 32 != 0):
        pytest.skip("chunks of whole tiles")
    cdf = O.GaussianModel(lo, hi, 0.3 * lo + 0.7 * hi - 20, 9.6, P, 32).cdf_table()
    model = B.Model.from_cdf(cdf, lo, P)
    sym = O.synth_symbols(2000 + P, 0, n_streams, n_per, lo, cdf, P)
    d = dev(sym.astype(np.int8))
    enc = B.range_encode(d, model, (32, 64, P), jump_points=jump)
    k = enc.jump.pos.shape[1] if enc.jump is not None else 0
    assert ALT or B.last_kernel() == ("range_encode_ckpt_n8_kernel" if k else "range_encode_n8_kernel"), B.last_kernel()
    _check_words(O, enc, sym, lo, cdf, P)
    if k:
        pos, lower, rng = O.range_jump_table(sym, lo, cdf, P, n_per // k)
        assert np.array_equal(enc.jump.pos.cpu().numpy().view(np.uint32), pos)
        assert np.array_equal(enc.jump.lower.cpu().numpy().view(np.uint64), lower) and np.array_equal(enc.jump.range.cpu().numpy().view(np.uint64), rng)
    guard = torch.full((n_streams * n_per + 4096,), 77, dtype=torch.int8, device="cuda")
    out = guard[: n_streams * n_per].view(n_streams, n_per)
    dec, st = B.range_decode(enc, model, n_per, out=out)
    assert ALT or B.last_kernel() == ("range_decode_sub_n8_kernel" if k else "range_decode_n8_kernel"), B.last_kernel()
    assert dec.dtype == torch.int8 and int(st.abs().sum()) == 0 and torch.equal(dec, d)
    assert bool((guard[n_streams * n_per:] == 77).all()), "symbols were written behind the matrix"
    # ... and the int32 decoder agrees on the same words
    wide, st = B.range_decode(enc, model, n_per)
    assert int(st.abs().sum()) == 0 and torch.equal(wide.to(torch.int8), d)


@pytest.mark.parametrize("frac", [0.0, 0.03, 0.2])
@pytest.mark.parametrize("P", [12, 24])
@pytest.mark.parametrize("jump", [0, 4, 32])
def test_int8_range_loops_at_the_maximum_rate(B, O, jump, P, frac):
    """symbols of probability 2^-P: a word per symbol at P = 24 ... and with jump = 32 a jump point on every tile"""
    n, n_streams, n_per = 101, 256 + 19, 1024
    cdf = spiky_cdf(n, P)
    model = B.Model.from_cdf(cdf, 0, P)
    rng = np.random.default_rng(P + jump + int(100 * frac))
    tails = rng.integers(1, n, (n_streams, n_per), dtype=np.int32)
    sym = np.where(rng.random((n_streams, n_per)) < frac, 0, tails).astype(np.int32)
    d = dev(sym.astype(np.int8))
    enc = B.range_encode(d, model, (32, 64, P), jump_points=jump)
    _check_words(O, enc, sym, 0, cdf, P)
    dec, st = B.range_decode(enc, model, n_per, dtype=torch.int8)
    assert int(st.abs().sum()) == 0 and torch.equal(dec, d)


def test_int8_range_coder_reports_what_int32_reports(B, O):
    """impossible symbols (below / above the support, at the type's ends), invalid data and a corrupt jump point: the int32 kernels' status"""
    P, n_streams, n_per, lo = 12, 256, 256, -50
    cdf = O.GaussianModel(lo, 50, 3.2, 9.6, P, 32).cdf_table()
    model = B.Model.from_cdf(cdf, lo, P)
    sym = O.synth_symbols(66, 0, n_streams, n_per, lo, cdf, P).astype(np.int8)
    bad = sym.copy()
    bad[3, 10] = 51; bad[69, 0] = -128; bad[70, 255] = 127; bad[255, 128] = -51
    for jump in (0, 2):
        enc = B.range_encode(dev(bad), model, (32, 64, P), jump_points=jump)
        want_words, want_n, want_st = O.rc_encode_batch(bad.astype(np.int32), lo, cdf, P)
        words, n_words, status = enc.to_numpy()
        assert status.tolist() == want_st.tolist() and sorted(np.flatnonzero(status).tolist()) == [3, 69, 70, 255]
        ok = np.flatnonzero(status == 0)
        assert n_words[ok].tolist() == want_n[ok].tolist()
        for s in ok[::7]:
            assert np.array_equal(words[s, : n_words[s]], want_words[s, : want_n[s]])
    good = B.range_encode(dev(sym), model, (32, 64, P), jump_points=2)
    good.jump.pos[17, 1] = 1 << 20                                   # a jump point beyond its stream
    dec, st = B.range_decode_checkpointed(good, good.jump, model, n_per, dtype=torch.int8)
    st = st.cpu().numpy()
    assert st[17, 1] == 3 and st.sum() == 3
    keep = np.ones(n_streams, bool); keep[17] = False
    assert torch.equal(dec[torch.from_numpy(keep).cuda()], dev(sym)[torch.from_numpy(keep).cuda()])
    # a model whose support does not fit the type cannot be decoded into it
    wide_model = B.Model.from_cdf(O.GaussianModel(-300, 300, 0.0, 70.0, 12, 32).cdf_table(), -300, 12)
    with pytest.raises(ValueError):
        B.range_decode(good, wide_model, n_per, dtype=torch.int8)


@pytest.mark.parametrize("dtype", [torch.int8, torch.int16], ids=["int8", "int16"])
@pytest.mark.parametrize("n_streams,n_per", [(3, 17), (70, 100), (256, 1000)])
@pytest.mark.parametrize("layout", ["stream_major", "symbol_major"])
def test_shapes_that_convert(B, O, dtype, n_streams, n_per, layout):
    """rows that are not whole tiles, symbol-major batches, int16: widened / narrowed next to the int32 kernels -- same results"""
    P = 12
    lo, hi = (-50, 50) if dtype == torch.int8 else (-300, 300)
    cdf = O.GaussianModel(lo, hi, 3.2, 9.6 if dtype == torch.int8 else 70.0, P, 32).cdf_table()
    model = B.Model.from_cdf(cdf, lo, P)
    sym = O.synth_symbols(5, 0, n_streams, n_per, lo, cdf, P)
    host = sym if layout == "stream_major" else np.ascontiguousarray(sym.T)
    d = dev(host).to(dtype)
    enc = B.range_encode(d, model, (32, 64, P), layout=layout)
    _check_words(O, enc, sym, lo, cdf, P)
    dec, st = B.range_decode(enc, model, n_per, layout=layout, dtype=dtype)
    assert dec.dtype == dtype and int(st.abs().sum()) == 0 and torch.equal(dec, d)


def test_int8_range_round_trip_at_full_size_without_conversion_kernels(B, O):
    """C4's batch as an int8 matrix: both directions inside the loops, two lanes per stream by default"""
    import bench
    P = 12
    model = B.Model.quantized_gaussian(bench.LO, bench.HI, bench.MEAN, bench.STD, P)
    cdf = model.cdf()
    sym = bench.synth_symbols_device(bench.SEED, 0, 65536, 4096, bench.LO, torch.from_numpy(cdf.astype(np.int64)).cuda(), P)
    d = sym.to(torch.int8)
    enc = B.range_encode(d, model, (32, 64, P))
    assert ALT or B.last_kernel() == "range_encode_ckpt_n8_kernel"
    assert ALT or (enc.jump is not None and enc.jump.pos.shape == (65536, 2))
    plain = B.range_encode(sym, model, (32, 64, P), jump_points=0)
    used = torch.arange(plain.words.shape[1], device="cuda")[None, :] < plain.n_words[:, None]
    assert torch.equal(enc.n_words, plain.n_words) and bool(((enc.words == plain.words) | ~used).all())
    dec, st = B.range_decode(enc, model, 4096, dtype=torch.int8)
    assert ALT or B.last_kernel() == "range_decode_sub_n8_kernel"
    assert int(st.abs().sum()) == 0 and torch.equal(dec, d)
    rows = [0, 1, 4095, 65535]
    _check_words(O, B.EncodedBatch(enc.words[rows].contiguous(), enc.n_words[rows].contiguous(), enc.status[rows].contiguous(), enc.config),
                 sym[rows].cpu().numpy(), bench.LO, cdf, P)
