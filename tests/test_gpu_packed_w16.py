"""GPU tests of CST_FLAG_PACKED_W16 (ABI 4): the (16,32) preset with its compressed words two per 32-bit slot -- the bytes of the
reference's Vec<u16> (SmallAnsCoder, src/stream/stack.rs:153) -- against the CPU oracle's words, through the hand-scheduled loops
(whole tiles, aligned slabs) and the symbol-by-symbol paths; compaction at 16-bit granularity; decoding from slabs and from the
packed buffer."""
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


@pytest.mark.parametrize("P", [8, 10, 12])
@pytest.mark.parametrize("n_streams,n_per", [(1, 1), (1, 40), (64, 64), (64, 128), (70, 96), (300, 101), (256, 4096), (130, 2048), (5, 0)])
def test_packed_words_are_the_reference_coders_u16_words(B, O, P, n_streams, n_per):
    lo, hi = (-50, 50) if P >= 10 else (-20, 20)
    cdf = O.GaussianModel(lo, hi, 3.2, 9.6 if P >= 10 else 4.0, P, 16).cdf_table()
    model = B.Model.from_cdf(cdf, lo, P)
    cfg = (16, 32, P)
    sym = O.synth_symbols(11, 0, n_streams, n_per, lo, cdf, P) if n_per else np.zeros((n_streams, 0), np.int32)
    want_words, want_n, _ = O.ans_encode_batch(sym, lo, cdf, P, 16, 32)
    enc = B.ans_encode(dev(sym), model, cfg, packed16=True)
    assert enc.packed16 and B.last_kernel() == "ans_encode_w16pk_kernel"
    torch.cuda.synchronize()
    words, n_words, status = enc.to_numpy()
    assert words.dtype == np.uint16 and (status == 0).all() and n_words.tolist() == want_n.tolist()
    for s in range(n_streams):
        assert words[s, : n_words[s]].tolist() == want_words[s, : want_n[s]].tolist(), f"stream {s}"
    # the unpacked call gives the same words, one per slot
    plain = B.ans_encode(dev(sym), model, cfg)
    pw, pn, _ = plain.to_numpy()
    assert pn.tolist() == n_words.tolist()
    for s in (0, n_streams // 2, n_streams - 1):
        assert pw[s, : pn[s]].tolist() == words[s, : n_words[s]].tolist()
    dec, dstatus = B.ans_decode(enc, model, n_per)
    assert B.last_kernel() == "ans_decode_w16pk_kernel"
    assert (dstatus.cpu().numpy() == 0).all() and np.array_equal(dec.cpu().numpy(), sym)
    # compaction in halfwords, decoding from the packed buffer (odd offsets: streams start on any 2-byte boundary)
    packed, offsets = B.compact(enc)
    torch.cuda.synchronize()
    off = offsets.cpu().numpy()
    assert off.tolist() == np.concatenate([[0], np.cumsum(want_n.astype(np.int64))]).tolist()
    flat = packed.cpu().numpy().view(np.uint16)
    for s in range(n_streams):
        assert flat[off[s]: off[s + 1]].tolist() == want_words[s, : want_n[s]].tolist(), f"stream {s}"
    dec2, st2 = B.ans_decode((packed, enc.n_words), model, n_per, offsets=offsets, config=cfg)
    assert (st2.cpu().numpy() == 0).all() and np.array_equal(dec2.cpu().numpy(), sym)


def test_packed_words_errors_and_bounds(B, O):
    P, lo = 12, -50
    cdf = O.GaussianModel(lo, 50, 3.2, 9.6, P, 16).cdf_table()
    model = B.Model.from_cdf(cdf, lo, P)
    sym = O.synth_symbols(12, 0, 128, 256, lo, cdf, P)
    bad = sym.copy()
    bad[7, 100] = 51                                    # impossible symbol
    enc = B.ans_encode(dev(bad), model, (16, 32, P), packed16=True)
    st = enc.status.cpu().numpy()
    assert st[7] == 1 and (np.delete(st, 7) == 0).all()
    small = B.EncodedBatch(torch.empty((128, 64), dtype=torch.int16, device="cuda"), torch.empty(128, dtype=torch.int32, device="cuda"),
                           torch.empty(128, dtype=torch.int32, device="cuda"), (16, 32, P))
    B.ans_encode(dev(sym), model, (16, 32, P), out=small)          # slabs of 64 halfwords: too small
    assert (small.status.cpu().numpy() == 2).all() and (small.n_words.cpu().numpy() == 0).all()
    enc = B.ans_encode(dev(sym), model, (16, 32, P), packed16=True)
    enc.n_words[3] = 100000                             # a count that leaves its slab: flagged, nothing read
    enc.words[9, int(enc.n_words[9]) - 1] = 0           # a stream that ends in a zero word
    dec, dstatus = B.ans_decode(enc, model, 256)
    d = dstatus.cpu().numpy()
    assert d[3] == 3 and d[9] == 3 and (np.delete(d, [3, 9]) == 0).all()
    got = dec.cpu().numpy()
    keep = np.ones(128, bool); keep[[3, 9]] = False
    assert np.array_equal(got[keep], sym[keep])
    from constriction_amd import _native as N
    with pytest.raises(N.BackendError):
        B.ans_encode(dev(sym), model if False else B.Model.from_cdf(O.GaussianModel(lo, 50, 3.2, 9.6, P, 32).cdf_table(), lo, P), (32, 64, P), packed16=True)
