"""GPU parity tests of the lane-quad form of the (32,64), P <= 12 ANS decoder (constriction_amd/csrc/cst_ans_dq.hip: 64-byte
word groups moved by four lanes; taken with CST_FLAG_COLD_WORDS -- `cold=True` -- or CST_DQ_DECODER=1): symbols and status of every stream against the CPU oracle and
against ans_decode_kernel, slabs and the packed layout (streams that start anywhere inside a 64-byte segment), decoding past the
end of the data, empty and invalid streams."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


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


def _dq(fn):
    from constriction_amd import _native
    os.environ["CST_DQ_DECODER"] = "1"
    _native.reload_knobs()            # (the library reads its switches once, at load)
    try:
        return fn()
    finally:
        del os.environ["CST_DQ_DECODER"]
        _native.reload_knobs()


@pytest.mark.parametrize("P", [8, 11, 12])
@pytest.mark.parametrize("n_streams,n_per", [(64, 64), (192, 96), (512, 160), (1024, 992), (320, 4096)])
def test_dq_decoder_matches_the_oracle_and_the_lane_decoder(B, O, P, n_streams, n_per):
    lo, hi = -60, 60
    cdf = O.GaussianModel(lo, hi, 2.5, 7.0 if P > 8 else 9.0, P, 32).cdf_table()
    model = B.Model.from_cdf(cdf, lo, P)
    sym = O.synth_symbols(0xD0 + P, 0, n_streams, n_per, lo, cdf, P)
    if n_per >= 992:
        sym[5, :] = lo + 60                      # (a stream of the most probable symbol only: next to no words)
    enc = B.ans_encode(dev(sym), model, (32, 64, P))
    torch.cuda.synchronize()
    words, n_words, status = enc.to_numpy()
    assert (status == 0).all()
    extra = 0 if n_per % 64 else 32               # (decoding past the end of the data is legal: stack.rs:1062-1065)
    want, want_st = O.ans_decode_batch(words, n_words, n_per + extra, lo, cdf, P)
    packed, offsets = B.compact(enc)
    for source, kw in ((enc, {}), ((packed, enc.n_words), {"offsets": offsets, "config": (32, 64, P)})):
        got, st = _dq(lambda: B.ans_decode(source, model, n_per + extra, **kw))
        ref, ref_st = B.ans_decode(source, model, n_per + extra, **kw)
        torch.cuda.synchronize()
        assert st.cpu().numpy().tolist() == want_st.tolist() == ref_st.cpu().numpy().tolist()
        assert np.array_equal(got.cpu().numpy(), want) and np.array_equal(ref.cpu().numpy(), want)
        assert np.array_equal(got.cpu().numpy()[:, :n_per], sym)


def test_dq_decoder_empty_and_invalid_streams(B, O):
    """a stream without words decodes from state 0, a stream whose last word is 0 is invalid data (stack.rs:299-318)"""
    P, n_streams, n_per, lo = 12, 128, 128, -50
    cdf = O.GaussianModel(lo, 50, 3.2, 9.6, P, 32).cdf_table()
    model = B.Model.from_cdf(cdf, lo, P)
    sym = O.synth_symbols(91, 0, n_streams, n_per, lo, cdf, P)
    enc = B.ans_encode(dev(sym), model, (32, 64, P))
    torch.cuda.synchronize()
    words, n_words, _ = enc.to_numpy()
    words = words.copy(); n_words = n_words.copy()
    n_words[3] = 0
    n_words[64] = 1
    words[77, n_words[77] - 1] = 0
    want, want_st = O.ans_decode_batch(words, n_words, n_per, lo, cdf, P)
    src = (dev(words.view(np.int32)), dev(n_words.view(np.int32)))
    got, st = _dq(lambda: B.ans_decode(src, model, n_per, config=(32, 64, P)))
    torch.cuda.synchronize()
    assert st.cpu().numpy().tolist() == want_st.tolist() and want_st[77] != 0
    ok = want_st == 0
    assert np.array_equal(got.cpu().numpy()[ok], want[ok])


def test_cold_words_hint_decodes_the_same(B, O):
    """CST_FLAG_COLD_WORDS is a hint: the same symbols and status with and without it (and it is the lane-quad kernel that ran:
    its LDS footprint of 136 KiB shows up as one workgroup per CU -- not observable from here, so the check is the results)"""
    P, n_streams, n_per, lo = 12, 1024, 1024, -50
    cdf = O.GaussianModel(lo, 50, 3.2, 9.6, P, 32).cdf_table()
    model = B.Model.from_cdf(cdf, lo, P)
    sym = O.synth_symbols(5, 0, n_streams, n_per, lo, cdf, P)
    enc = B.ans_encode(dev(sym), model, (32, 64, P))
    hot, st_hot = B.ans_decode(enc, model, n_per)
    cold, st_cold = B.ans_decode(enc, model, n_per, cold=True)
    torch.cuda.synchronize()
    assert torch.equal(hot, cold) and torch.equal(st_hot, st_cold) and np.array_equal(cold.cpu().numpy(), sym)
    # a shape the lane-quad kernel does not take (a partial wave): the hint is ignored
    enc2 = B.ans_encode(dev(sym[:100]), model, (32, 64, P))
    d2, s2 = B.ans_decode(enc2, model, n_per, cold=True)
    torch.cuda.synchronize()
    assert np.array_equal(d2.cpu().numpy(), sym[:100]) and (s2.cpu().numpy() == 0).all()


@pytest.mark.parametrize("flags", [1, 1 | 2])           # CST_FLAG_RAW_STATE, without and with CST_FLAG_COLD_WORDS
def test_batched_decoders_continue_from_a_raw_state(B, O, flags):
    """CST_FLAG_RAW_STATE at batch shapes (AnsCoder::decode_symbols on a coder that has already decoded: stack.rs:1070-1100):
    the two halves of every row decoded by two calls -- state and word count handed from one to the next -- are the row"""
    import ctypes as C
    from constriction_amd import _native as N
    P, n_streams, n_per, lo = 12, 256, 512, -50
    cdf = O.GaussianModel(lo, 50, 3.2, 9.6, P, 32).cdf_table()
    model = B.Model.from_cdf(cdf, lo, P)
    sym = O.synth_symbols(321, 0, n_streams, n_per, lo, cdf, P)
    enc = B.ans_encode(dev(sym), model, (32, 64, P))
    torch.cuda.synchronize()
    words, n_words, status = enc.to_numpy()
    assert (status == 0).all() and (n_words >= 2).all()
    # from_compressed by hand (stack.rs:440-462): the last word, then the one before (a state of two words here)
    state = np.array([(int(words[s, n_words[s] - 1]) << 32) | int(words[s, n_words[s] - 2]) for s in range(n_streams)], dtype=np.uint64)
    assert (state >= (1 << 32)).all()
    d_state, d_n = dev(state.view(np.int64)), dev((n_words - 2).astype(np.int32))
    d_n_out = torch.zeros(n_streams, dtype=torch.int32, device="cuda")
    d_status = torch.zeros(n_streams, dtype=torch.int32, device="cuda")
    half = n_per // 2
    outs = []
    for _ in range(2):
        out = torch.empty((n_streams, half), dtype=torch.int32, device="cuda")
        N.check(N.lib().cst_ans_decode_batch(model._h, N.CoderConfig(32, 64, P), C.c_void_p(enc.words.data_ptr()), None, enc.words.shape[1],
                                             enc.words.numel(), C.c_void_p(d_n.data_ptr()), C.c_void_p(out.data_ptr()), n_streams, half, 0,
                                             C.c_void_p(d_state.data_ptr()), C.c_void_p(d_n_out.data_ptr()), C.c_void_p(d_status.data_ptr()),
                                             flags, None), "cst_ans_decode_batch")
        torch.cuda.synchronize()
        assert (d_status.cpu().numpy() == 0).all()
        d_n.copy_(d_n_out)
        outs.append(out.cpu().numpy())
    assert np.array_equal(np.concatenate(outs, axis=1), sym)
    assert (d_n.cpu().numpy() == 0).all()                # every word consumed


def test_cold_words_are_recognised_by_provenance(B, O):
    """batched.ans_decode decides by itself how to read the words (round 5): chunk loads for the EncodedBatch the last ans_encode
    on this stream filled, whole 64-byte segments by lane quads for everything else -- and the choice never changes a result."""
    P, n_streams, n_per = 12, 256, 256
    cdf = O.GaussianModel(-50, 50, 3.2, 9.6, P, 32).cdf_table()
    model = B.Model.from_cdf(cdf, -50, P)
    sym = torch.from_numpy(O.synth_symbols(9, 0, n_streams, n_per, -50, cdf, P)).cuda()
    other = torch.from_numpy(O.synth_symbols(10, 0, n_streams, n_per, -50, cdf, P)).cuda()
    enc = B.ans_encode(sym, model, (32, 64, P))
    d0, s0 = B.ans_decode(enc, model, n_per)
    assert B.last_kernel() == "ans_decode_kernel"                      # fresh: the encoder has just left them in the caches
    d1, _ = B.ans_decode(enc, model, n_per)
    assert B.last_kernel() == "ans_decode_kernel"                      # (reading them does not make them colder)
    enc_other = B.ans_encode(other, model, (32, 64, P))
    d2, s2 = B.ans_decode(enc, model, n_per)
    assert B.last_kernel() == "ans_decode_dq_kernel"                   # another batch went through the caches since
    d3, _ = B.ans_decode(enc_other, model, n_per)
    assert B.last_kernel() == "ans_decode_kernel"
    foreign = B.EncodedBatch(enc.words.clone(), enc.n_words.clone(), enc.status, enc.config)     # words that came from elsewhere
    d4, s4 = B.ans_decode(foreign, model, n_per)
    assert B.last_kernel() == "ans_decode_dq_kernel"
    packed, offsets = B.compact(enc)
    d5, s5 = B.ans_decode((packed, enc.n_words), model, n_per, offsets=offsets)
    assert B.last_kernel() == "ans_decode_dq_kernel"
    d6, _ = B.ans_decode(foreign, model, n_per, cold=False)
    assert B.last_kernel() == "ans_decode_kernel"
    d7, _ = B.ans_decode(enc_other, model, n_per, cold=True)
    assert B.last_kernel() == "ans_decode_dq_kernel"
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        side.wait_stream(torch.cuda.default_stream())
        d8, _ = B.ans_decode(enc_other, model, n_per)                   # another HIP stream: no claim about its caches' history
        assert B.last_kernel() == "ans_decode_dq_kernel"
    torch.cuda.synchronize()
    for d in (d0, d1, d2, d4, d5, d6):
        assert torch.equal(d, sym)
    for d in (d3, d7, d8):
        assert torch.equal(d, other)
    for s in (s0, s2, s4, s5):
        assert int(s.abs().sum()) == 0
