"""One table per stream with THIRTY-TWO lanes per stream: ans_decode_pt_lut_kernel (cst_ans_ptlut.hip, round 6) -- every stream's whole
quantile -> symbol table in LDS (the reference's lookup decoder model, src/stream/model/categorical/lookup_contiguous.rs:564-605: ONE
table read per symbol), built by the workgroup from the stream's cdf row; the lanes are the jump points (stack.rs:1107-1139) the
default encode call notes every N / 32 symbols.

Parity: the jump tables are the oracle's AnsCoder.pos() at those symbols, the words those of the plain call and of the oracle, the
decoded symbols the input -- for int32 and int8 matrices, every precision the kernel takes, partial workgroups (24 streams each),
data at the maximum word rate, packed words behind offsets, and jump points that point outside their stream."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ALT = any(os.environ.get(k) for k in ("CST_AUTO_JUMP", "CST_NO_N8", "CST_PT_LUT", "CST_PT_SUB_WAVES"))


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


def _case(B, O, n_streams, n_per, P, lo, hi, seed, tails=False):
    rng = np.random.default_rng(seed)
    mu = rng.uniform(lo / 4, hi / 4, n_streams)
    sd = np.exp(rng.uniform(np.log(0.3), np.log(max(1.0, (hi - lo) / 8)), n_streams))
    model = B.Model.quantized_gaussian_per_stream(lo, hi, dev(mu), dev(sd), P)
    cdfs = model.cdfs_device().cpu().numpy().astype(np.uint32)
    assert cdfs.shape == (n_streams, hi - lo + 2)
    u = rng.integers(0, 1 << P, (n_streams, n_per))
    if tails:                                # the two ends of every table: probability 1 / 2^P each, P bits per symbol
        u = rng.integers(0, 2, (n_streams, n_per)) * ((1 << P) - 1)
    sym = np.stack([np.searchsorted(cdfs[s], u[s], side="right") - 1 + lo for s in range(n_streams)]).astype(np.int32)
    return model, cdfs, sym


def _oracle_check(O, sym, lo, cdfs, P, enc, streams):
    words, n_words, status = enc.to_numpy()
    assert (status == 0).all()
    for s in streams:
        c = O.AnsCoder()
        c.encode_iid_table_reverse(sym[s], cdfs[s], lo, P)
        assert words[s, : n_words[s]].tolist() == c.get_compressed().tolist(), f"stream {s}"
        wp, ws = O.ans_jump_table(sym[s: s + 1], lo, cdfs[s], P, enc.jump.interval)
        assert np.array_equal(enc.jump.pos[s].cpu().numpy().view(np.uint32), wp[0]), f"stream {s}"
        assert np.array_equal(enc.jump.state[s].cpu().numpy().view(np.uint64), ws[0]), f"stream {s}"


@pytest.mark.skipif(ALT, reason="an alternate kernel path is forced")
@pytest.mark.parametrize("dtype", [torch.int32, torch.int8], ids=["int32", "int8"])
@pytest.mark.parametrize("n_streams,n_per,P,lo,hi", [(24, 2048, 12, -127, 127), (25, 2048, 12, -127, 127), (1, 4096, 12, -127, 127),
                                                       (100, 2048, 11, -100, 90), (47, 2048, 10, -20, 20), (333, 2048, 8, -3, 4),
                                                       (48, 4096, 12, -128, 127), (600, 2048, 9, 0, 60)])
def test_default_call_takes_32_lanes_per_stream(B, O, dtype, n_streams, n_per, P, lo, hi):
    model, cdfs, sym = _case(B, O, n_streams, n_per, P, lo, hi, 11 * n_streams + P)
    d = dev(sym).to(dtype)
    plain = B.ans_encode(d, model, (32, 64, P), jump_points=0)
    auto = B.ans_encode(d, model, (32, 64, P))
    assert auto.jump is not None and auto.jump.pos.shape == (n_streams, 32) and auto.jump.interval == n_per // 32
    assert torch.equal(plain.n_words, auto.n_words)
    used = torch.arange(auto.words.shape[1], device="cuda")[None, :] < auto.n_words[:, None]
    assert bool(((plain.words == auto.words) | ~used).all())
    _oracle_check(O, sym, lo, cdfs, P, auto, sorted({0, n_streams // 2, n_streams - 1}))
    dec, st = B.ans_decode(auto, model, n_per, dtype=dtype)
    assert B.last_kernel() == ("ans_decode_pt_lut_kernel" if dtype == torch.int32 else "ans_decode_pt_lut_n8_kernel")
    assert st.shape == (n_streams,) and int(st.abs().sum()) == 0 and torch.equal(dec, d)
    dec_p, st_p = B.ans_decode(plain, model, n_per, dtype=dtype)
    assert torch.equal(dec_p, d) and int(st_p.abs().sum()) == 0


@pytest.mark.skipif(ALT, reason="an alternate kernel path is forced")
@pytest.mark.parametrize("dtype", [torch.int32, torch.int8], ids=["int32", "int8"])
@pytest.mark.parametrize("P", [12, 9])
def test_maximum_word_rate_through_every_jump_point(B, O, dtype, P):
    """every symbol costs P bits: a lane reads a word on three steps out of eight (P = 12) -- the ring's windows at their limit"""
    n_streams, n_per, lo, hi = 50, 2048, -127, 127
    model, cdfs, sym = _case(B, O, n_streams, n_per, P, lo, hi, 5, tails=True)
    d = dev(sym).to(dtype)
    enc = B.ans_encode(d, model, (32, 64, P))
    assert enc.jump is not None and enc.jump.pos.shape[1] == 32
    _oracle_check(O, sym, lo, cdfs, P, enc, (0, 17, n_streams - 1))
    assert int(enc.n_words.min()) >= n_per * P // 32 * 9 // 10
    dec, st = B.ans_decode(enc, model, n_per, dtype=dtype)
    assert B.last_kernel().startswith("ans_decode_pt_lut") and int(st.abs().sum()) == 0 and torch.equal(dec, d)


@pytest.mark.skipif(ALT, reason="an alternate kernel path is forced")
def test_packed_words_and_bad_jump_points(B, O):
    n_streams, n_per, P, lo, hi = 70, 2048, 12, -127, 127
    model, cdfs, sym = _case(B, O, n_streams, n_per, P, lo, hi, 77)
    d = dev(sym)
    enc = B.ans_encode(d, model, (32, 64, P))
    ck = enc.jump
    packed, offsets = B.compact(enc)
    dec, st = B.ans_decode_checkpointed(packed, ck, model, n_per, offsets=offsets, config=(32, 64, P))
    assert B.last_kernel() == "ans_decode_pt_lut_kernel" and int(st.abs().sum()) == 0 and torch.equal(dec, d)
    # a jump point beyond its slab: that chunk is flagged, the others decode, nothing is read out of bounds
    bad = B.Checkpoints(ck.interval, ck.pos.clone(), ck.state.clone())
    bad.pos[3, 5] = 1 << 30
    dec, st = B.ans_decode_checkpointed(enc, bad, model, n_per)
    assert st.shape == (n_streams, 32) and int(st[3, 5]) != 0 and int(st.abs().sum()) == abs(int(st[3, 5]))
    ok = torch.ones((n_streams, 32), dtype=torch.bool, device="cuda"); ok[3, 5] = False
    assert torch.equal(dec.view(n_streams, 32, -1)[ok], d.view(n_streams, 32, -1)[ok])


def test_the_knob_switches_it_off(B, O, knob):
    n_streams, n_per, P, lo, hi = 30, 2048, 12, -127, 127
    model, cdfs, sym = _case(B, O, n_streams, n_per, P, lo, hi, 3)
    d = dev(sym)
    knob(CST_PT_LUT="0")
    enc = B.ans_encode(d, model, (32, 64, P))
    assert enc.jump is None or enc.jump.pos.shape[1] <= 16
    dec, st = B.ans_decode(enc, model, n_per)
    assert not B.last_kernel().startswith("ans_decode_pt_lut") and torch.equal(dec, d)
