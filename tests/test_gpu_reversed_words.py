"""Reversed word order (`AnsCoder::from_reversed_compressed`, /root/reference/src/stream/stack.rs:734-748; `Cursor::into_reversed`,
src/backends.rs:1424-1448): `cst_words_reverse` converts a batch between the reference's default order and the order a decoder
consumes the words in.  The oracle's words reversed on the CPU are the expected values."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def B():
    from constriction_amd import batched
    return batched


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def _batch(B, n_streams, n_per, seed):
    rng = np.random.default_rng(seed)
    m = B.Model.quantized_gaussian(-40, 40, 1.5, 7.0, 12)
    sym = np.clip(np.rint(rng.normal(1.5, 7.0, (n_streams, n_per))), -40, 40).astype(np.int32)
    return m, sym


@pytest.mark.parametrize("n_streams,n_per", [(1, 1), (3, 70), (130, 333), (64, 4096)])
def test_reversed_slabs_match_the_oracle_and_reverse_back(B, O, n_streams, n_per):
    m, sym = _batch(B, n_streams, n_per, n_streams + n_per)
    enc = B.ans_encode(torch.from_numpy(sym).cuda(), m, (32, 64, 12))
    rev = B.reverse_words(enc)
    torch.cuda.synchronize()
    words, n_words, _ = rev.to_numpy()
    cdf = m.cdf()
    for s in range(0, n_streams, max(1, n_streams // 7)):
        c = O.AnsCoder(W=32, S=64)
        c.encode_iid_table_reverse(sym[s], cdf, -40, 12)
        want = c.get_compressed()[::-1]                      # (what Cursor::into_reversed leaves in the buffer)
        assert words[s, : n_words[s]].tolist() == want.tolist(), f"stream {s}"
    # its own inverse, out of place and in place
    back = B.reverse_words(rev)
    B.reverse_words(rev, out=rev)
    torch.cuda.synchronize()
    w0, n0, _ = enc.to_numpy()
    w1, _, _ = rev.to_numpy()
    w2, _, _ = back.to_numpy()
    for s in range(n_streams):
        assert np.array_equal(w0[s, : n0[s]], w1[s, : n0[s]]) and np.array_equal(w0[s, : n0[s]], w2[s, : n0[s]])
    dec = B.ans_decode(rev, m, n_per)
    dec = dec[0] if isinstance(dec, tuple) else dec
    assert np.array_equal(dec.cpu().numpy(), sym)


def test_reversed_packed_words(B):
    m, sym = _batch(B, 77, 200, 5)
    enc = B.ans_encode(torch.from_numpy(sym).cuda(), m, (32, 64, 12))
    packed, offsets = B.compact(enc)
    rev = B.reverse_words((packed, enc.n_words), offsets=offsets)
    torch.cuda.synchronize()
    off = offsets.cpu().numpy()
    p, r = packed.cpu().numpy(), rev.cpu().numpy()
    for s in range(77):
        assert np.array_equal(r[off[s]: off[s + 1]], p[off[s]: off[s + 1]][::-1])
    with pytest.raises(ValueError):
        B.reverse_words((packed, enc.n_words))


def test_reversed_words_of_two_messages_and_an_empty_stream(B, O):
    """the oracle's coder over a message, its words reversed on the host = what Cursor::into_reversed holds; a stream without
    words stays empty"""
    cdf = np.array([0, 1000, 3000, 4096], dtype=np.uint32)
    msg = np.array([0, 2, 1, 1, 2, 0, 2, 2, 1] * 11, dtype=np.int32)
    m = B.Model.from_cdf(cdf, 0, 12)
    enc = B.ans_encode(torch.from_numpy(np.stack([msg, msg[::-1].copy()])).cuda(), m, (32, 64, 12))
    enc.n_words[1] = 0
    rev = B.reverse_words(enc)
    torch.cuda.synchronize()
    c = O.AnsCoder(W=32, S=64)
    c.encode_iid_table_reverse(msg, cdf, 0, 12)
    assert rev.stream(0).tolist() == c.get_compressed()[::-1].tolist()
    assert rev.stream(1).size == 0
