"""GPU tests of checkpointed streams (cst_ans_encode_batch_ckpt / cst_ans_decode_batch_ckpt): the reference's Pos / Seek
jump tables (src/stream/stack.rs:1107-1139, its test :1456-1548) for the batched coder, and BASELINE config C1 (ONE stream
of 10^6 symbols, QuantizedGaussian(-50, 50, 3.2, 9.6), P = 24) decoded on a thousand lanes."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
# (runs of the suite through the alternate kernel paths -- profiles/r05_alt_paths.txt -- do not take the kernels the tests name)
ALT = any(os.environ.get(k) for k in ("CST_NO_N8", "CST_NO_PC_ENCODER", "CST_SMALL_KERNELS", "CST_PC_COMBINED", "CST_NO_PC_WIDE"))
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


@pytest.mark.parametrize("cfg", [(32, 64, 24), (32, 64, 12), (16, 32, 12)], ids=lambda c: "W%dS%dP%d" % c)
@pytest.mark.parametrize("n_streams,n_per,interval", [(1, 96, 32), (3, 120, 40), (70, 64, 64), (5, 100, 1), (2, 90, 100), (67, 1200, 400)])
def test_checkpoints_are_pos_and_state_of_the_reference_coder(B, O, cfg, n_streams, n_per, interval):
    W, S, P = cfg
    cdf = O.GaussianModel(-30, 30, 1.5, 6.0, P, 32 if W == 32 else 16).cdf_table()
    model = B.Model.from_cdf(cdf, -30, P)
    sym = O.synth_symbols(21, 0, n_streams, n_per, -30, cdf, P)
    want_words, want_n, _ = O.ans_encode_batch(sym, -30, cdf, P, W, S)
    enc, ck = B.ans_encode_checkpointed(dev(sym), model, interval, cfg)
    torch.cuda.synchronize()
    words, n_words, status = enc.to_numpy()
    assert (status == 0).all() and n_words.tolist() == want_n.tolist()
    pos, state = ck.pos.cpu().numpy(), ck.state.cpu().numpy().view(np.uint64)
    n_chunks = (n_per + interval - 1) // interval
    assert pos.shape == (n_streams, n_chunks)
    for s in range(n_streams):
        assert words[s, : n_words[s]].tolist() == want_words[s, : want_n[s]].tolist()
        for j in range(n_chunks):
            c = O.AnsCoder(W=W, S=S)                       # AnsCoder::pos() after encoding symbols [j * K, n) in reverse
            c.encode_iid_table_reverse(sym[s, j * interval:], cdf, -30, P)
            p, st = c.pos()
            assert (int(pos[s, j]), int(state[s, j])) == (p, st), (s, j)
    if n_per % interval == 0:
        dec, dstatus = B.ans_decode_checkpointed(enc, ck, model, n_per)
        torch.cuda.synchronize()
        assert (dstatus.cpu().numpy() == 0).all() and np.array_equal(dec.cpu().numpy(), sym)


def test_config_c1_one_stream_of_a_million_symbols(B, O):
    """BASELINE config C1 on the GPU: words identical to the CPU coder's, decoded on 1000 lanes through the checkpoints
    and (slowly, on one lane) without them."""
    lo, hi, mean, std, P, n = -50, 50, 3.2, 9.6, 24, 1_000_000
    model = B.Model.quantized_gaussian(lo, hi, mean, std, P)
    cdf = O.GaussianModel(lo, hi, mean, std, P, 32).cdf_table()
    assert model.cdf().tolist() == cdf.tolist()
    sym = O.synth_symbols(0xC0FFEE, 0, 1, n, lo, cdf, P)
    want_words, want_n, _ = O.ans_encode_batch(sym, lo, cdf, P)
    enc, ck = B.ans_encode_checkpointed(dev(sym), model, 1000, (32, 64, P))
    torch.cuda.synchronize()
    assert int(enc.status[0]) == 0 and int(enc.n_words[0]) == int(want_n[0])
    assert np.array_equal(enc.stream(0), want_words[0, : want_n[0]])
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    dec, dstatus = B.ans_decode_checkpointed(enc, ck, model, n)
    e1.record()
    dec1, st1 = B.ans_decode(enc, model, n)
    e2.record()
    torch.cuda.synchronize()
    assert (dstatus.cpu().numpy() == 0).all() and np.array_equal(dec.cpu().numpy(), sym)
    assert int(st1[0]) == 0 and np.array_equal(dec1.cpu().numpy(), sym)
    print(f"C1 decode: {e0.elapsed_time(e1):.2f} ms on 1000 lanes (checkpoints), {e1.elapsed_time(e2):.2f} ms on one lane")
    assert e0.elapsed_time(e1) * 20 < e1.elapsed_time(e2)


# ---------------------------------------------------------------------------------------------------------------------
# round 5: jump points for ONE TABLE PER STREAM (config C3) -- the compact-row encoder notes them on its way, the sub-lane
# decoder runs k lanes per stream that share the stream's table in LDS (cst_ans_pt.hip)
# ---------------------------------------------------------------------------------------------------------------------

def _c3_model(B, O, n_streams, lo, hi, P, seed):
    rng = np.random.default_rng(seed)
    mu = -10 + 20 * rng.random(n_streams)
    sigma = np.exp(np.log(0.5) + rng.random(n_streams) * np.log(32))
    model = B.Model.quantized_gaussian_per_stream(lo, hi, dev(mu), dev(sigma), P)
    cdfs = np.stack([O.GaussianModel(lo, hi, m, s, P, 32).cdf_table() for m, s in zip(mu, sigma)])
    return model, cdfs


@pytest.mark.parametrize("n_streams,n_per,interval", [
    (256, 256, 64),      # k = 4: one full workgroup of the sub-lane decoder, chunks of whole tiles
    (128, 512, 256),     # k = 2
    (1024, 1024, 128),   # k = 8, several workgroups
    (300, 256, 64),      # partial last wave of streams (and of virtual streams)
    (70, 128, 32),       # k = 4, one-tile chunks, fewer streams than a workgroup holds
    (64, 1024, 64),      # k = 16
    (130, 96, 24),       # k = 4, chunks that are NOT whole tiles: per-symbol paths on both sides
    (5, 96, 32),         # k = 3: not a power of two -> decoded whole (the jump points are side information)
    (3, 96, 96),         # k = 1
    (67, 200, 64),       # interval does not divide n: encode only
], ids=lambda v: str(v))
def test_per_stream_jump_points(B, O, n_streams, n_per, interval):
    lo, hi, P = -127, 127, 12
    cfg = (32, 64, P)
    model, cdfs = _c3_model(B, O, n_streams, lo, hi, P, seed=n_streams * 7 + n_per + interval)
    sym = O.synth_symbols(0xC0FFEE, 0, n_streams, n_per, lo, cdfs, P, per_stream_tables=True)
    want_words, want_n, _ = O.ans_encode_batch(sym, lo, cdfs, P)
    want_pos, want_state = O.ans_jump_table(sym, lo, cdfs, P, interval)
    enc, ck = B.ans_encode_checkpointed(dev(sym), model, interval, cfg)
    torch.cuda.synchronize()
    words, n_words, status = enc.to_numpy()
    assert (status == 0).all() and n_words.tolist() == want_n.tolist()
    for s in range(n_streams):
        assert words[s, : n_words[s]].tolist() == want_words[s, : want_n[s]].tolist(), f"stream {s}"
    assert np.array_equal(ck.pos.cpu().numpy().view(np.uint32), want_pos)
    assert np.array_equal(ck.state.cpu().numpy().view(np.uint64), want_state)
    if n_per % interval:
        return
    dec, dstatus = B.ans_decode_checkpointed(enc, ck, model, n_per)
    torch.cuda.synchronize()
    assert (dstatus.cpu().numpy() == 0).all()
    assert np.array_equal(dec.cpu().numpy(), sym)
    # ... and the words decode the same without the jump points
    plain, pstatus = B.ans_decode(enc, model, n_per)
    assert (pstatus.cpu().numpy() == 0).all() and torch.equal(plain, dec)


def test_per_stream_jump_points_corrupt_positions(B, O):
    """positions are caller data: one that leaves its slab flags THAT chunk and reads nothing (as every count does)"""
    lo, hi, P, n_streams, n_per, interval = -127, 127, 12, 256, 256, 64
    model, cdfs = _c3_model(B, O, n_streams, lo, hi, P, seed=99)
    sym = O.synth_symbols(1, 0, n_streams, n_per, lo, cdfs, P, per_stream_tables=True)
    enc, ck = B.ans_encode_checkpointed(dev(sym), model, interval, (32, 64, P))
    bad = [(0, 1), (17, 3), (255, 0)]
    for s, j in bad:
        ck.pos[s, j] = 0x7fffffff
    dec, dstatus = B.ans_decode_checkpointed(enc, ck, model, n_per)
    torch.cuda.synchronize()
    st = dstatus.cpu().numpy()
    got = dec.cpu().numpy()
    for s in range(n_streams):
        for j in range(n_per // interval):
            if (s, j) in bad:
                assert st[s, j] == 3
            else:
                assert st[s, j] == 0 and np.array_equal(got[s, j * interval:(j + 1) * interval], sym[s, j * interval:(j + 1) * interval])


# ---------------------------------------------------------------------------------------------------------------------
# jump points for per-symbol Gaussians (the reference's flagship call): cst_ans_{encode,decode}_gaussian_batch_ckpt
# ---------------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("n_streams,n_per,interval", [(64, 64, 16), (130, 96, 32), (70, 128, 64), (256, 48, 48)], ids=lambda v: str(v))
def test_per_symbol_gaussian_jump_points(B, O, n_streams, n_per, interval, knob):
    knob(CST_FUSED_MIN_STREAMS="1")          # the fused encoder on small batches (it is the one that notes jump points)
    rng = np.random.default_rng(n_streams + n_per)
    mu = rng.uniform(-30, 30, (n_streams, n_per)); sd = np.exp(rng.uniform(-1, 3, (n_streams, n_per)))
    sym = np.clip(np.rint(mu + sd * rng.standard_normal((n_streams, n_per))), -100, 100).astype(np.int32)
    enc, ck = B.ans_encode_gaussian_checkpointed(dev(sym), -100, 100, dev(mu), dev(sd), interval)
    plain = B.ans_encode_gaussian(dev(sym), -100, 100, dev(mu), dev(sd))
    torch.cuda.synchronize()
    assert (enc.status.cpu().numpy() == 0).all() and torch.equal(enc.n_words, plain.n_words)
    pos, state = ck.pos.cpu().numpy().view(np.uint32), ck.state.cpu().numpy().view(np.uint64)
    for s in (0, n_streams // 2, n_streams - 1):
        assert enc.stream(s).tolist() == plain.stream(s).tolist()
        for j in range(n_per // interval):
            c = O.AnsCoder()
            c.encode_gaussian_reverse(sym[s, j * interval:], -100, 100, mu[s, j * interval:], sd[s, j * interval:], 24, 32)
            assert (int(pos[s, j]), int(state[s, j])) == c.pos(), (s, j)
            if j == 0:
                assert enc.stream(s).tolist() == c.get_compressed().tolist()
    dec, dstatus = B.ans_decode_gaussian_checkpointed(enc, ck, -100, 100, dev(mu), dev(sd))
    torch.cuda.synchronize()
    assert (dstatus.cpu().numpy() == 0).all() and np.array_equal(dec.cpu().numpy(), sym)


@pytest.mark.parametrize("n_streams,n_per,interval", [(64, 64, 16), (130, 96, 32), (70, 128, 64), (256, 48, 48)], ids=lambda v: str(v))
def test_per_symbol_gaussian_jump_points_range_coder(B, O, n_streams, n_per, interval, knob):
    """round 6: the same for the range coder (cst_range_{encode,decode}_gaussian_batch_ckpt): RangeEncoder.pos() in front of every chunk
    (queue.rs:182-196) from the CPU oracle coding the stream's prefix alone, the words of the plain call, every chunk decoded through its
    jump point"""
    knob(CST_FUSED_MIN_STREAMS="1")
    rng = np.random.default_rng(n_streams + n_per + 1)
    mu = rng.uniform(-30, 30, (n_streams, n_per)); sd = np.exp(rng.uniform(-1, 3, (n_streams, n_per)))
    sym = np.clip(np.rint(mu + sd * rng.standard_normal((n_streams, n_per))), -100, 100).astype(np.int32)
    enc = B.range_encode_gaussian(dev(sym), -100, 100, dev(mu), dev(sd), jump_points=n_per // interval)
    assert B.last_kernel() == "range_encode_gaussian_fused_kernel<ckpt>" and enc.jump.pos.shape == (n_streams, n_per // interval)
    plain = B.range_encode_gaussian(dev(sym), -100, 100, dev(mu), dev(sd), jump_points=0)
    torch.cuda.synchronize()
    assert plain.jump is None and (enc.status.cpu().numpy() == 0).all() and torch.equal(enc.n_words, plain.n_words)
    pos = enc.jump.pos.cpu().numpy().view(np.uint32)
    lower, rng_ = enc.jump.lower.cpu().numpy().view(np.uint64), enc.jump.range.cpu().numpy().view(np.uint64)
    for s in (0, n_streams // 2, n_streams - 1):
        assert enc.stream(s).tolist() == plain.stream(s).tolist()
        models = [O.GaussianModel(-100, 100, float(m), float(v), 24, 32) for m, v in zip(mu[s], sd[s])]
        for j in range(n_per // interval):
            c = O.RangeEncoder()
            c.encode(sym[s, : j * interval], models[: j * interval], 24)
            want = c.pos()
            assert (int(pos[s, j]), (int(lower[s, j]), int(rng_[s, j]))) == want, (s, j)
        c = O.RangeEncoder()
        c.encode(sym[s], models, 24)
        assert enc.stream(s).tolist() == c.get_compressed().tolist()
    dec, dstatus = B.range_decode_gaussian(enc, -100, 100, dev(mu), dev(sd))
    torch.cuda.synchronize()
    assert dstatus.shape == (n_streams,) and (dstatus.cpu().numpy() == 0).all() and np.array_equal(dec.cpu().numpy(), sym)
    dec2, st2 = B.range_decode_gaussian(plain, -100, 100, dev(mu), dev(sd))
    assert np.array_equal(dec2.cpu().numpy(), sym)
    enc.jump.pos[3, -1] = 1 << 28                                  # a jump point beyond its stream: flagged, nothing read out of bounds
    _, st3 = B.range_decode_gaussian(enc, -100, 100, dev(mu), dev(sd))
    st3 = st3.cpu().numpy()
    assert st3[3] == 3 and st3.sum() == 3


# ---- round 5: jump points at the speed of the plain encoder (producer / consumer encoder), and int8 matrices through them ----

@pytest.mark.parametrize("dtype", ["int32", "int8"])
@pytest.mark.parametrize("P", [12, 9, 16, 24])
@pytest.mark.parametrize("n_streams,n_per,interval", [(256, 256, 128), (256, 1024, 256), (512, 768, 384), (256, 4096, 512), (256, 512, 512),
                                                      (256, 256, 32), (256, 384, 96), (70, 256, 128), (1, 128, 64), (321, 512, 128)])
def test_producer_consumer_encoders_note_jump_points(B, O, dtype, P, n_streams, n_per, interval):
    """cst_ans_encode_batch_ckpt / _ckpt_sym on shapes the producer / consumer encoders take: the words of the plain encoder, the
    jump table of the CPU oracle (AnsCoder::pos() in front of every chunk: stack.rs:1107-1139) for EVERY stream, and the chunks
    decoded on their own lanes (int8: by the decoder loops themselves where the chunks are whole 128-symbol lines)"""
    lo = -50
    cdf = O.GaussianModel(lo, 50, 3.2, 9.6, P, 32).cdf_table()
    model = B.Model.from_cdf(cdf, lo, P)
    sym = O.synth_symbols(500 + interval, 0, n_streams, n_per, lo, cdf, P)
    want_words, want_n, _ = O.ans_encode_batch(sym, lo, cdf, P)
    want_pos, want_state = O.ans_jump_table(sym, lo, cdf, P, interval)
    if dtype == "int8" and n_per % 128 != 0:
        pytest.skip("int8 rows are whole 128-symbol lines")
    d = dev(sym if dtype == "int32" else sym.astype(np.int8))
    enc, ck = B.ans_encode_checkpointed(d, model, interval, (32, 64, P))
    tag = "<wide, ckpt>" if P > 12 else "<ckpt>"              # (12 < P <= 24: the wide step, two word groups per tile)
    assert ALT or B.last_kernel() == ("ans_encode_pc_kernel" if dtype == "int32" else "ans_encode_pc_n8_kernel") + tag
    torch.cuda.synchronize()
    words, n_words, status = enc.to_numpy()
    assert (status == 0).all() and n_words.tolist() == want_n.tolist()
    for s in range(n_streams):
        assert np.array_equal(words[s, : n_words[s]], want_words[s, : want_n[s]]), f"stream {s}"
    assert np.array_equal(ck.pos.cpu().numpy().astype(np.uint32), want_pos)
    assert np.array_equal(ck.state.cpu().numpy().view(np.uint64), want_state)
    dec, dstatus = B.ans_decode_checkpointed(enc, ck, model, n_per, dtype=d.dtype)
    if dtype == "int8" and interval % 128 == 0 and P <= 12:
        assert ALT or B.last_kernel() in ("ans_decode_n8_kernel", "ans_decode_small_n8_kernel")
    assert dec.dtype == d.dtype and (dstatus.cpu().numpy() == 0).all() and torch.equal(dec, d)


def test_int8_jump_points_on_shapes_that_convert(B, O):
    """int8 / int16 matrices whose shape the native kernels do not take: converted next to the int32 calls, same jump table"""
    P, lo = 12, -50
    cdf = O.GaussianModel(lo, 50, 3.2, 9.6, P, 32).cdf_table()
    model = B.Model.from_cdf(cdf, lo, P)
    for dtype, n_streams, n_per, interval in ((torch.int8, 70, 120, 40), (torch.int16, 256, 256, 128), (torch.int8, 256, 192, 64)):
        sym = O.synth_symbols(9, 0, n_streams, n_per, lo, cdf, P)
        want_pos, want_state = O.ans_jump_table(sym, lo, cdf, P, interval)
        d = dev(sym).to(dtype)
        enc, ck = B.ans_encode_checkpointed(d, model, interval, (32, 64, P))
        assert np.array_equal(ck.pos.cpu().numpy().astype(np.uint32), want_pos)
        assert np.array_equal(ck.state.cpu().numpy().view(np.uint64), want_state)
        dec, dstatus = B.ans_decode_checkpointed(enc, ck, model, n_per, dtype=dtype)
        assert (dstatus.cpu().numpy() == 0).all() and torch.equal(dec, d)


@pytest.mark.parametrize("dtype", ["int32", "int8"])
@pytest.mark.parametrize("P", [12, 24])
def test_jump_points_on_packed_words(B, O, dtype, P):
    """the words of compact() (what container.load or a gather hands over) decoded through the jump table of the batch: a jump point counts
    words from the start of ITS stream, wherever the stream lies (cst_ans_decode_batch_ckpt[_sym] with d_offsets)"""
    lo, n_streams, n_per, k = -50, 300, 512, 4
    cdf = O.GaussianModel(lo, 50, 3.2, 9.6, P, 32).cdf_table()
    model = B.Model.from_cdf(cdf, lo, P)
    sym = O.synth_symbols(900 + P, 0, n_streams, n_per, lo, cdf, P)
    d = dev(sym if dtype == "int32" else sym.astype(np.int8))
    enc, ck = B.ans_encode_checkpointed(d, model, n_per // k, (32, 64, P))
    packed, offsets = B.compact(enc)
    total = int(offsets[-1].item())
    packed = packed[:total].clone()
    dec, st = B.ans_decode_checkpointed(packed, ck, model, n_per, dtype=d.dtype, offsets=offsets, config=(32, 64, P))
    assert dec.dtype == d.dtype and int(st.abs().sum()) == 0 and torch.equal(dec, d)
    with pytest.raises(ValueError):
        B.ans_decode_checkpointed(packed, ck, model, n_per, dtype=d.dtype)
    # a jump point beyond its stream's words: that chunk alone is flagged
    bad = B.Checkpoints(ck.interval, ck.pos.clone(), ck.state)
    bad.pos[7, 2] = 0x7FFFFFF0
    dec, st = B.ans_decode_checkpointed(packed, bad, model, n_per, dtype=d.dtype, offsets=offsets, config=(32, 64, P))
    st = st.cpu().numpy()
    assert st[7, 2] == 3 and (np.delete(st.reshape(-1), 7 * k + 2) == 0).all()


def test_jump_points_travel_in_the_container(B, O, tmp_path):
    """encode with jump points -> compact -> container file -> load -> decode k lanes per stream from the packed words"""
    from constriction_amd import container
    P, lo, n_streams, n_per, k = 24, -50, 200, 256, 2
    cdf = O.GaussianModel(lo, 50, 3.2, 9.6, P, 32).cdf_table()
    model = B.Model.from_cdf(cdf, lo, P)
    sym = O.synth_symbols(31, 0, n_streams, n_per, lo, cdf, P)
    d = dev(sym.astype(np.int8))
    enc, ck = B.ans_encode_checkpointed(d, model, n_per // k, (32, 64, P))
    packed, offsets = B.compact(enc)
    path = tmp_path / "batch.cst"
    container.save(path, packed, offsets, (32, 64, P), jump_points=ck)
    words, off, cfg, (interval, pos, state) = container.load_with_jump_points(path)
    want_words, want_n, _ = O.ans_encode_batch(sym, lo, cdf, P)
    for s in (0, 77, n_streams - 1):
        assert words[int(off[s]): int(off[s + 1])].tolist() == want_words[s, : want_n[s]].tolist()
    jump = B.Checkpoints(interval, dev(pos.view(np.int32)), dev(state.view(np.int64)))
    dec, st = B.ans_decode_checkpointed(dev(words.view(np.int32)), jump, model, n_per, dtype=torch.int8, offsets=dev(off.astype(np.int64)), config=cfg)
    assert int(st.abs().sum()) == 0 and torch.equal(dec, d)
