"""GPU parity tests over the reference's whole (Word, State, PRECISION) grid -- the thirteen combinations of
src/stream/stack.rs:1293-1356 (`compress_many_*`; tests/random_data.rs:161-192 has the same ones): three presets run on the
hand-scheduled kernels, the others on cst_ans_generic.hip.  Words, counts and status of every stream against the CPU oracle
(whose coder state is 64 bits wide whatever S is), decoded symbols against the input, both layouts, decoding past the end."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

GRID = [(32, 64, 32), (32, 64, 24), (32, 64, 16), (32, 64, 8),
        (16, 64, 16), (16, 64, 12), (16, 64, 8), (8, 64, 8),
        (16, 32, 16), (16, 32, 12), (16, 32, 8), (8, 32, 8), (8, 16, 8)]


@pytest.fixture(scope="module")
def B():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: torch.cuda.is_available() is False")
    from constriction_amd import batched
    return batched


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def random_cdf(rng, n, P):
    """n symbols, every probability >= 1, sum 2^P (a wrapped u32 at P = 32, as the reference's Probability = u32 holds it)"""
    total = 1 << P
    w = rng.dirichlet(np.full(n, 0.3))
    counts = np.maximum(1, np.floor(w * (total - n)).astype(np.int64) + 1)
    counts[np.argmax(counts)] += total - int(counts.sum())
    assert counts.min() >= 1 and int(counts.sum()) == total
    cdf = np.concatenate([[0], np.cumsum(counts)])
    return (cdf & 0xffffffff).astype(np.uint32), counts / total


@pytest.mark.parametrize("W,S,P", GRID)
@pytest.mark.parametrize("layout", ["stream_major", "symbol_major"])
def test_every_combination_of_the_reference_grid(B, O, W, S, P, layout):
    rng = np.random.default_rng(W * 1000 + S * 10 + P)
    n_sym = int(min(200, (1 << P) // 2))
    lo = -37
    cdf, probs = random_cdf(rng, n_sym, P)
    model = B.Model.from_cdf(cdf, lo, P)
    for n_streams, n_per in ((1, 1), (3, 0), (70, 257), (300, 64)):
        sym = (rng.choice(n_sym, size=(n_streams, n_per), p=probs) + lo).astype(np.int32)
        if n_streams > 5 and n_per > 5:
            sym[2, 3] = lo + n_sym                       # impossible symbols
            sym[5, n_per - 1] = lo - 1
        want_words, want_n, want_status = O.ans_encode_batch(sym, lo, cdf, P, W, S)
        d = dev(sym if layout == "stream_major" else sym.T)
        enc = B.ans_encode(d, model, (W, S, P), layout)
        torch.cuda.synchronize()
        words, n_words, status = enc.to_numpy()
        assert status.tolist() == want_status.tolist(), (n_streams, n_per)
        assert n_words.tolist() == want_n.tolist()
        width = min(words.shape[1], want_words.shape[1])
        assert int(n_words.max(initial=0)) <= width
        mask = np.arange(width, dtype=np.uint32)[None, :] < n_words[:, None]
        assert np.array_equal(np.where(mask, words[:, :width], 0), np.where(mask, want_words[:, :width], 0))
        assert (words[mask.nonzero()[0], mask.nonzero()[1]] >> W == 0).all() if W < 32 else True
        dec, dstatus = B.ans_decode(enc, model, n_per + 3, layout)       # three symbols past the end: legal and deterministic
        torch.cuda.synchronize()
        good = status == 0
        got = dec.cpu().numpy() if layout == "stream_major" else dec.cpu().numpy().T
        assert (dstatus.cpu().numpy()[good] == 0).all()
        assert np.array_equal(got[good][:, :n_per], sym[good])
        want_dec, _ = O.ans_decode_batch(np.where(mask, want_words[:, :width], 0)[good], want_n[good], n_per + 3, lo, cdf, P, W, S)
        assert np.array_equal(got[good], want_dec)


def test_unsupported_combinations_are_rejected(B):
    from constriction_amd import _native as N
    cdf = np.array([0, 100, 256], dtype=np.uint32)
    model = B.Model.from_cdf(cdf, 0, 8)
    d = torch.zeros((4, 8), dtype=torch.int32, device="cuda")
    for cfg in ((8, 8, 8), (32, 32, 8), (64, 128, 8), (16, 16, 8), (24, 64, 8)):
        with pytest.raises(N.BackendError):
            B.ans_encode(d, model, cfg)
