"""GPU tests of the range coder's jump points (cst_range_encode_batch_ckpt / cst_range_decode_batch_ckpt): RangeEncoder::pos() /
RangeDecoder::seek of the reference (src/stream/queue.rs:172-196, 900-926, test :1333-1396) for the batched coder -- words,
jump tables and decoded chunks against the CPU oracle; the sub-lane decoder (k lanes per stream, two waves per SIMD) and the
fall-back onto the ordinary batched decode for the shapes it does not take."""
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


SHAPES = [
    (256, 256, 64),      # k = 4, chunks of whole tiles, full waves
    (128, 512, 256),     # k = 2
    (300, 256, 64),      # partial last wave
    (70, 128, 32),       # one-tile chunks
    (1000, 96, 32),      # k = 3: any k
    (64, 1024, 64),      # k = 16
    (130, 96, 24),       # chunks that are not whole tiles: per-symbol paths on both sides
    (3, 96, 96),         # k = 1
    (67, 200, 64),       # interval does not divide n: encode only
    (2, 9000, 1000),     # long streams: the end-of-data statement
]


# (symbol-major: the encoder side only, a few shapes)
LAYOUT_SHAPES = [("stream_major", *sh) for sh in SHAPES] + [("symbol_major", *sh) for sh in SHAPES if sh[:2] in ((256, 256), (300, 256), (3, 96))]


@pytest.mark.parametrize("cfg", [(32, 64, 12), (32, 64, 24), (32, 64, 16), (16, 32, 12)], ids=lambda c: "W%dS%dP%d" % c)
@pytest.mark.parametrize("layout,n_streams,n_per,interval", LAYOUT_SHAPES, ids=lambda v: str(v))
def test_range_jump_points(B, O, cfg, layout, n_streams, n_per, interval):
    W, S, P = cfg
    lo = -30
    cdf = O.GaussianModel(lo, 30, 1.5, 6.0, P, 32 if W == 32 else 16).cdf_table()
    model = B.Model.from_cdf(cdf, lo, P)
    sym = O.synth_symbols(77, 0, n_streams, n_per, lo, cdf, P)
    want_words, want_n, _ = O.rc_encode_batch(sym, lo, cdf, P, W, S)
    want_pos, want_lower, want_range = O.range_jump_table(sym, lo, cdf, P, interval, W, S)
    enc, ck = B.range_encode_checkpointed(dev(sym if layout == "stream_major" else sym.T), model, interval, cfg, layout)
    torch.cuda.synchronize()
    words, n_words, status = enc.to_numpy()
    assert (status == 0).all() and n_words.tolist() == want_n.tolist()
    for s in range(n_streams):
        assert words[s, : n_words[s]].tolist() == want_words[s, : want_n[s]].tolist(), f"stream {s}"
    assert np.array_equal(ck.pos.cpu().numpy().view(np.uint32), want_pos)
    assert np.array_equal(ck.lower.cpu().numpy().view(np.uint64), want_lower)
    assert np.array_equal(ck.range.cpu().numpy().view(np.uint64), want_range)
    if n_per % interval or layout != "stream_major":
        return
    dec, dstatus = B.range_decode_checkpointed(enc, ck, model, n_per)
    torch.cuda.synchronize()
    assert (dstatus.cpu().numpy() == 0).all()
    assert np.array_equal(dec.cpu().numpy(), sym)


def test_range_jump_points_on_a_skewed_model(B, O):
    """carry-heavy data (a near-deterministic model): held words, carries into them, the exact repeat of the encoder"""
    P, lo, n_streams, n_per, interval = 12, 0, 192, 2048, 512
    probs = np.array([0.97, 0.01, 0.01, 0.01])
    cdf = O.categorical_fast_cdf(probs, P)
    model = B.Model.from_cdf(cdf, lo, P)
    rng = np.random.default_rng(4)
    sym = rng.choice(4, size=(n_streams, n_per), p=probs).astype(np.int32)
    want_words, want_n, _ = O.rc_encode_batch(sym, lo, cdf, P)
    want_pos, want_lower, want_range = O.range_jump_table(sym, lo, cdf, P, interval)
    enc, ck = B.range_encode_checkpointed(dev(sym), model, interval, (32, 64, P))
    torch.cuda.synchronize()
    words, n_words, status = enc.to_numpy()
    assert (status == 0).all() and n_words.tolist() == want_n.tolist()
    for s in range(n_streams):
        assert words[s, : n_words[s]].tolist() == want_words[s, : want_n[s]].tolist()
    assert np.array_equal(ck.pos.cpu().numpy().view(np.uint32), want_pos)
    assert np.array_equal(ck.lower.cpu().numpy().view(np.uint64), want_lower)
    assert np.array_equal(ck.range.cpu().numpy().view(np.uint64), want_range)
    dec, dstatus = B.range_decode_checkpointed(enc, ck, model, n_per)
    assert (dstatus.cpu().numpy() == 0).all() and np.array_equal(dec.cpu().numpy(), sym)


@pytest.mark.parametrize("P", [12, 24])
def test_range_jump_points_corrupt_side_information(B, O, P):
    """jump points are caller data: a position beyond its stream flags THAT chunk; a wrong state decodes garbage or reports
    invalid data, and never reads outside the stream"""
    lo, n_streams, n_per, interval = -30, 256, 256, 64
    cdf = O.GaussianModel(lo, 30, 1.5, 6.0, P, 32).cdf_table()
    model = B.Model.from_cdf(cdf, lo, P)
    sym = O.synth_symbols(3, 0, n_streams, n_per, lo, cdf, P)
    enc, ck = B.range_encode_checkpointed(dev(sym), model, interval, (32, 64, P))
    ck.pos[5, 2] = 0x7fffffff
    ck.range[9, 1] = 0                   # (RangeCoderState never holds range = 0)
    ck.lower[200, 3] += 12345678901
    dec, dstatus = B.range_decode_checkpointed(enc, ck, model, n_per)
    torch.cuda.synchronize()
    st, got = dstatus.cpu().numpy(), dec.cpu().numpy()
    assert st[5, 2] == 3
    touched = {(5, 2), (9, 1), (200, 3)}
    for s in range(n_streams):
        for j in range(n_per // interval):
            if (s, j) not in touched:
                assert st[s, j] == 0 and np.array_equal(got[s, j * interval:(j + 1) * interval], sym[s, j * interval:(j + 1) * interval]), (s, j)


def test_range_jump_points_full_size_k2(B, O):
    """config C4 at full size with two jump points per stream: sampled streams against the oracle, every stream round trips"""
    import bench
    P, n, k = 12, 65536, 4096
    m = B.Model.quantized_gaussian(-50, 50, 3.2, 9.6, P)
    cdf_h = m.cdf()
    sym = bench.synth_symbols_device(0xC0FFEE, 0, n, k, -50, torch.from_numpy(cdf_h.astype(np.int64)).cuda(), P)
    enc, ck = B.range_encode_checkpointed(sym, m, k // 2, (32, 64, P))
    plain = B.range_encode(sym, m, (32, 64, P))
    assert torch.equal(plain.n_words, enc.n_words)
    dec, st = B.range_decode_checkpointed(enc, ck, m, k)
    assert int(st.abs().sum()) == 0 and torch.equal(dec, sym)
    rows = [0, 1, 63, 64, 4097, 65535]
    host = sym[rows].cpu().numpy()
    want_pos, want_lower, want_range = O.range_jump_table(host, -50, cdf_h, P, k // 2)
    assert np.array_equal(ck.pos[rows].cpu().numpy().view(np.uint32), want_pos)
    assert np.array_equal(ck.lower[rows].cpu().numpy().view(np.uint64), want_lower)
    assert np.array_equal(ck.range[rows].cpu().numpy().view(np.uint64), want_range)


@pytest.mark.parametrize("P", [12, 24])
def test_range_jump_points_on_packed_words(B, O, P):
    """the words of compact() decoded through the batch's jump table (cst_range_decode_batch_ckpt with d_offsets)"""
    lo, n_streams, n_per, k = -50, 300, 512, 4
    cdf = O.GaussianModel(lo, 50, 3.2, 9.6, P, 32).cdf_table()
    model = B.Model.from_cdf(cdf, lo, P)
    sym = O.synth_symbols(70 + P, 0, n_streams, n_per, lo, cdf, P)
    d = torch.from_numpy(sym).cuda()
    enc, ck = B.range_encode_checkpointed(d, model, n_per // k, (32, 64, P))
    packed, offsets = B.compact(enc)
    total = int(offsets[-1].item())
    packed = packed[:total].clone()
    dec, st = B.range_decode_checkpointed((packed, enc.n_words), ck, model, n_per, offsets=offsets, config=(32, 64, P))
    assert int(st.abs().sum()) == 0 and torch.equal(dec, d)
    with pytest.raises(ValueError):
        B.range_decode_checkpointed((packed, enc.n_words), ck, model, n_per)
