"""GPU parity tests of the producer / consumer form of the (32,64), P <= 12 ANS encoder (constriction_amd/csrc/cst_ans_pc.hip:
coder waves + helper waves; the path BASELINE config C2 takes): words, counts and status of every stream against the CPU
oracle, and against the one-wave kernel (CST_NO_PC_ENCODER=1) on shapes the oracle does not finish in seconds."""
import os

import numpy as np
import pytest

from kernel_names import with_jump  # noqa: E402

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
ALT = any(os.environ.get(k) for k in ("CST_NO_PC_ENCODER", "CST_SMALL_KERNELS", "CST_PC_COMBINED", "CST_NO_PC_WIDE"))    # (A/B runs: scripts/alt_paths.sh)


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


def _one_wave(fn):
    from constriction_amd import _native
    os.environ["CST_NO_PC_ENCODER"] = "1"
    _native.reload_knobs()            # (the library reads its switches once, at load)
    try:
        return fn()
    finally:
        del os.environ["CST_NO_PC_ENCODER"]
        _native.reload_knobs()


def _aligned_symbols(sym):
    """the kernel takes rows that are whole 128-byte aligned tiles: a fresh device allocation is"""
    d = dev(sym)
    assert d.data_ptr() % 128 == 0
    return d


@pytest.mark.parametrize("P", [8, 11, 12, 13, 16, 24])          # (12 < P <= 24, round 5: the wide step in the coder waves, two word groups per tile)
@pytest.mark.parametrize("jp", [0, "auto"], ids=["plain", "auto_jump"])
@pytest.mark.parametrize("n_streams,n_per", [(256, 64), (256, 96), (512, 160), (1024, 992), (768, 4096), (300, 128)])
def test_pc_encoder_matches_the_oracle(B, O, P, n_streams, n_per, jp):
    lo, hi = -60, 60
    cdf = O.GaussianModel(lo, hi, 2.5, 7.0 if P > 8 else 9.0, P, 32).cdf_table()
    model = B.Model.from_cdf(cdf, lo, P)
    sym = O.synth_symbols(0xABC + P, 0, n_streams, n_per, lo, cdf, P)
    sym[3, 5] = hi + 1                                   # impossible symbols: above, below, far away, in the last and first tile
    sym[70, n_per - 1] = lo - 1
    sym[200, 0] = 2 ** 30
    sym[255, n_per // 2] = -2 ** 31
    want_words, want_n, want_status = O.ans_encode_batch(sym, lo, cdf, P)
    enc = B.ans_encode(_aligned_symbols(sym), model, (32, 64, P), jump_points=jp)
    assert ALT or B.last_kernel() == with_jump("ans_encode_pc_kernel<wide>" if P > 12 else "ans_encode_pc_kernel", enc)
    torch.cuda.synchronize()
    words, n_words, status = enc.to_numpy()
    assert status.tolist() == want_status.tolist() and sorted(np.flatnonzero(status).tolist()) == [3, 70, 200, 255]
    assert n_words.tolist() == want_n.tolist()
    width = min(words.shape[1], want_words.shape[1])
    assert int(n_words.max()) <= width
    words = words[:, :width]
    mask = np.arange(width, dtype=np.uint32)[None, :] < n_words[:, None]
    assert np.array_equal(np.where(mask, words, 0), np.where(mask, want_words[:, :width], 0))
    # ... and it really was the producer / consumer kernel: the one-wave kernel gives the same, from the same call
    enc1 = _one_wave(lambda: B.ans_encode(_aligned_symbols(sym), model, (32, 64, P)))
    torch.cuda.synchronize()
    w1, n1, s1 = enc1.to_numpy()
    assert n1.tolist() == n_words.tolist() and s1.tolist() == status.tolist()
    assert np.array_equal(np.where(mask, w1[:, :width], 0), np.where(mask, words, 0))
    dec, dstatus = B.ans_decode(enc, model, n_per)
    torch.cuda.synchronize()
    good = status == 0
    assert (dstatus.cpu().numpy()[good] == 0).all() and np.array_equal(dec.cpu().numpy()[good], sym[good])


@pytest.mark.parametrize("P", [12, 24])
def test_pc_encoder_capacity_and_raw_state(B, O, P):
    """slabs that are too small (CST_STREAM_CAPACITY, nothing written behind the slab) and coders that continue from a given
    state (CST_FLAG_RAW_STATE: AnsCoder::encode_symbols_reverse on a non-empty coder, stack.rs:784-849)"""
    n_streams, n_per, lo = 512, 256, -50
    cdf = O.GaussianModel(lo, 50, 3.2, 9.6, P, 32).cdf_table()
    model = B.Model.from_cdf(cdf, lo, P)
    sym = O.synth_symbols(77, 0, n_streams, n_per, lo, cdf, P)
    want_words, want_n, _ = O.ans_encode_batch(sym, lo, cdf, P)
    stride = 32                                           # < the ~45 words a stream needs: every stream overflows its slab
    guard = torch.full((n_streams * stride + 4096,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
    from constriction_amd import _native as N
    import ctypes as C
    n_words = torch.zeros(n_streams, dtype=torch.int32, device="cuda")
    status = torch.zeros(n_streams, dtype=torch.int32, device="cuda")
    d = _aligned_symbols(sym)
    N.check(N.lib().cst_ans_encode_batch(model._h, N.CoderConfig(32, 64, P), C.c_void_p(d.data_ptr()), n_streams, n_per, 0,
                                         C.c_void_p(guard.data_ptr()), stride, C.c_void_p(n_words.data_ptr()), None,
                                         C.c_void_p(status.data_ptr()), 0, None), "cst_ans_encode_batch")
    torch.cuda.synchronize()
    over = want_n > stride
    assert over.all() and (status.cpu().numpy() == 2).all() and (n_words.cpu().numpy() == 0).all()
    assert (guard[n_streams * stride:].cpu().numpy() == 0x5A5A5A5A).all(), "words were written behind the last slab"
    # raw state: two halves of every row coded by two calls = the whole row coded by one
    half = n_per // 2
    full = B.ans_encode(d, model, (32, 64, P))
    torch.cuda.synchronize()
    fw, fn, fs = full.to_numpy()
    assert fn.tolist() == want_n.tolist()
    st = torch.zeros(n_streams, dtype=torch.int64, device="cuda")
    stride2 = B.max_words(n_per, (32, 64, P))
    w2 = torch.zeros((n_streams, stride2), dtype=torch.int32, device="cuda")
    n2a = torch.zeros(n_streams, dtype=torch.int32, device="cuda")
    n2b = torch.zeros_like(n2a)
    second = d[:, half:].contiguous()                      # (encoding runs backwards: the SECOND half of a row is coded first)
    first = d[:, :half].contiguous()
    assert second.data_ptr() % 128 == 0 and first.data_ptr() % 128 == 0
    lib = N.lib()
    N.check(lib.cst_ans_encode_batch(model._h, N.CoderConfig(32, 64, P), C.c_void_p(second.data_ptr()), n_streams, half, 0,
                                     C.c_void_p(w2.data_ptr()), stride2, C.c_void_p(n2a.data_ptr()), C.c_void_p(st.data_ptr()),
                                     C.c_void_p(status.data_ptr()), 1, None), "raw 1")
    torch.cuda.synchronize()
    na = n2a.cpu().numpy()
    assert len(set(na.tolist())) > 1                        # (different streams have emitted different numbers of words so far)
    # continue every coder behind the words it has emitted: a second buffer, rows shifted by the words of the first call
    w3 = torch.zeros((n_streams, stride2), dtype=torch.int32, device="cuda")
    N.check(lib.cst_ans_encode_batch(model._h, N.CoderConfig(32, 64, P), C.c_void_p(first.data_ptr()), n_streams, half, 0,
                                     C.c_void_p(w3.data_ptr()), stride2, C.c_void_p(n2b.data_ptr()), C.c_void_p(st.data_ptr()),
                                     C.c_void_p(status.data_ptr()), 1, None), "raw 2")
    torch.cuda.synchronize()
    nb, a, b, state = n2b.cpu().numpy(), w2.cpu().numpy().view(np.uint32), w3.cpu().numpy().view(np.uint32), st.cpu().numpy().view(np.uint64)
    for s in (0, 1, 63, 64, 255, 256, 511):
        tail = [int(state[s]) & 0xffffffff, int(state[s]) >> 32]
        while tail and tail[-1] == 0:
            tail.pop()
        got = a[s, : na[s]].tolist() + b[s, : nb[s]].tolist() + tail
        assert got == want_words[s, : want_n[s]].tolist(), s
