"""Pins the CPU oracle (oracle/oracle.c) against the reference's own golden vectors.

CPU-only.  Every vector in tests/golden/reference_vectors.json was transcribed from the
reference's tests / doc-tests (see its 'source' field)."""
import numpy as np
import pytest

from conftest import golden_vectors
from golden_util import models_for
from oracle import oracle as O


def _run(vec):
    W, S, P = vec["W"], vec["S"], vec["P"]
    prob_bits = 32 if W == 32 else 16
    is_ans = vec["coder"] == "ans"
    init = vec.get("init")
    enc_steps = [s for s in vec["steps"] if s["op"] == "encode"]
    dec_steps = [s for s in vec["steps"] if s["op"] == "decode"]

    compressed = None
    if enc_steps:
        coder = O.AnsCoder(W=W, S=S) if is_ans else O.RangeEncoder(W=W, S=S)
        for st in enc_steps:
            models, _ = models_for(st, P, O, prob_bits)
            if is_ans:
                coder.encode_reverse(st["symbols"], models, P)
            else:
                coder.encode(st["symbols"], models, P)
        compressed = coder.get_compressed()
        assert compressed.tolist() == vec["expect_compressed"], vec["id"]
        if "expect_compressed_hex" in vec:
            assert [int(h, 16) for h in vec["expect_compressed_hex"]] == vec["expect_compressed"]
        if "expect_num_bits" in vec:
            assert coder.num_bits() == vec["expect_num_bits"]
        if "expect_num_valid_bits" in vec:
            assert coder.num_valid_bits() == vec["expect_num_valid_bits"]
        if vec.get("roundtrip"):
            dec = O.AnsCoder(compressed, W=W, S=S) if is_ans else O.RangeDecoder(compressed, W=W, S=S)
            # ANS pops in reverse call order, the range coder replays calls in order
            for st in (reversed(enc_steps) if is_ans else enc_steps):
                models, _ = models_for(st, P, O, prob_bits)
                got = dec.decode(models, len(st["symbols"]), P)
                assert got.tolist() == st["symbols"], vec["id"]
            if is_ans:
                assert dec.is_empty()

    if init is not None:
        words = np.asarray(init["compressed"], dtype=np.uint32)
        if "compressed_hex" in init:
            assert [int(h, 16) for h in init["compressed_hex"]] == init["compressed"]
        dec = (O.AnsCoder(words, seal=init.get("seal", False), W=W, S=S) if is_ans else O.RangeDecoder(words, W=W, S=S))
        for st in dec_steps:
            models, n = models_for(st, P, O, prob_bits)
            got = dec.decode(models, n if n is not None else len(st["expect"]), P)
            assert got.tolist() == st["expect"], vec["id"]
        if vec.get("expect_empty_after"):
            assert dec.is_empty()


@pytest.mark.parametrize("vec", golden_vectors(), ids=lambda v: v["id"])
def test_golden_vector(vec):
    _run(vec)


def test_sizes(golden):
    """G10: src/stream/stack.rs:1249-1291."""
    sz = golden["sizes"]
    model = O.GaussianModel(sz["lo"], sz["hi"], sz["mean"], sz["std"], 24, 32)
    for case in sz["cases"]:
        symbols = case.get("symbols")
        if symbols is None:
            a, b = case["symbols_range"]
            symbols = list(range(a, b))
        coder = O.AnsCoder()
        # the reference test encodes in forward order (encode_iid_symbols)
        coder.encode_reverse(symbols[::-1], model, 24)
        words = coder.get_compressed()
        assert len(words) == case["expect_num_words"]
        dec = O.AnsCoder(words)
        got = dec.decode(model, len(symbols), 24)
        assert got.tolist() == symbols[::-1]
        assert dec.is_empty()


def test_trailing_zero_word_rejected():
    """stack.rs:299-318 / pybindings/stream/stack.rs:220-234."""
    with pytest.raises(ValueError):
        O.AnsCoder(np.array([5, 0], dtype=np.uint32))
    assert O.AnsCoder(np.array([], dtype=np.uint32)).is_empty()
    # from_binary (seal) accepts it and is never empty (stack.rs:341-360 doc example)
    assert not O.AnsCoder(np.array([], dtype=np.uint32), seal=True).is_empty()
    c = O.AnsCoder(np.array([0x89ABCDEF, 0x01234567], dtype=np.uint32), seal=True)
    assert c.get_compressed().tolist() == [0x89ABCDEF, 0x01234567, 1]  # stack.rs:916-939


def test_fast_c_loops_match_generic():
    """The whole-array C loops must agree with the per-symbol generic path."""
    rng = np.random.default_rng(7)
    lo, hi = -100, 100
    means = rng.uniform(-20, 20, 300)
    stds = rng.uniform(0.3, 30, 300)
    symbols = np.clip(np.round(rng.normal(means, stds)), lo, hi).astype(np.int32)
    a, b = O.AnsCoder(), O.AnsCoder()
    a.encode_reverse(symbols, [O.GaussianModel(lo, hi, m, s) for m, s in zip(means, stds)])
    b.encode_gaussian_reverse(symbols, lo, hi, means, stds)
    assert a.get_compressed().tolist() == b.get_compressed().tolist()
    assert b.decode_gaussian(len(symbols), lo, hi, means, stds).tolist() == symbols.tolist()
    assert b.is_empty()
    # iid table path
    model = O.GaussianModel(-50, 50, 3.2, 9.6, 12, 16)
    cdf = model.cdf_table()
    sym = O.synth_symbols(0xC0FFEE, 0, 1, 2000, -50, cdf, 12)[0]
    a, b = O.AnsCoder(), O.AnsCoder()
    a.encode_reverse(sym, model, 12)
    b.encode_iid_table_reverse(sym, cdf, -50, 12)
    assert a.get_compressed().tolist() == b.get_compressed().tolist()
    assert b.decode_iid_table(2000, cdf, -50, 12).tolist() == sym.tolist()
