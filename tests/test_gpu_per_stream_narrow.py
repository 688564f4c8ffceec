"""int8 symbol matrices on the per-stream-table coder (config C3: one quantized Gaussian per stream, support -127 .. 127 -- it fits int8;
the reference's coders and models are generic over the symbol type, src/stream/model/quantize.rs:229-255).  Round 6: the compact-row
encoder reads int8 tiles itself ("ans_encode_pt_n8_kernel[<ckpt>]") and the sub-lane decoder's byte tiles hold the symbols and leave as
they are ("ans_decode_pt_sub_n8_kernel"); other shapes convert next to the int32 kernels.  Either way the CPU oracle's words, counts,
status and jump points on the widened values, and the input back."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ALT = any(os.environ.get(k) for k in ("CST_NO_N8", "CST_PT_SUB_WAVES", "CST_AUTO_JUMP"))
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


def _batch(B, O, n_streams, n_per, seed, P=12, lo=-127, hi=127, narrow_models=False):
    rng = np.random.default_rng(seed)
    mu = rng.uniform(-10, 10, n_streams)
    sd = rng.uniform(0.4, 0.8, n_streams) if narrow_models else np.exp(rng.uniform(np.log(0.5), np.log(16.0), n_streams))
    model = B.Model.quantized_gaussian_per_stream(lo, hi, dev(mu), dev(sd), P)
    cdfs = np.stack([O.GaussianModel(lo, hi, float(m), float(s), P, 32).cdf_table() for m, s in zip(mu, sd)])
    return model, cdfs, mu, sd


@pytest.mark.parametrize("jump", [0, 2, 4, 8, 16, "auto"])
@pytest.mark.parametrize("n_streams,n_per", [(256, 512), (300, 1024), (1024, 2048), (70, 4096)])
def test_int8_matrices_inside_the_per_stream_table_loops(B, O, n_streams, n_per, jump):
    model, cdfs, _, _ = _batch(B, O, n_streams, n_per, n_streams + n_per)
    sym = np.stack([O.synth_symbols(9, s, 1, n_per, -127, cdfs[s], 12)[0] for s in range(n_streams)])
    d = dev(sym.astype(np.int8))
    enc = B.ans_encode(d, model, (32, 64, 12), jump_points=jump)
    k = enc.jump.pos.shape[1] if enc.jump is not None else 0
    assert ALT or B.last_kernel() == ("ans_encode_pt_n8_kernel<ckpt>" if k else "ans_encode_pt_n8_kernel"), B.last_kernel()
    want_words, want_n, _ = O.ans_encode_batch(sym, -127, cdfs, 12)
    words, n_words, status = enc.to_numpy()
    assert (status == 0).all() and n_words.tolist() == want_n.tolist()
    for s in range(n_streams):
        assert np.array_equal(words[s, : n_words[s]], want_words[s, : want_n[s]]), f"stream {s}"
    if k:
        pos, state = O.ans_jump_table(sym, -127, cdfs, 12, n_per // k)
        assert np.array_equal(enc.jump.pos.cpu().numpy().view(np.uint32), pos) and np.array_equal(enc.jump.state.cpu().numpy().view(np.uint64), state)
    guard = torch.full((n_streams * n_per + 4096,), 77, dtype=torch.int8, device="cuda")
    out = guard[: n_streams * n_per].view(n_streams, n_per)
    dec, st = B.ans_decode(enc, model, n_per, out=out)
    if k >= 2:
        assert ALT or B.last_kernel() == "ans_decode_pt_sub_n8_kernel", B.last_kernel()
    assert dec.dtype == torch.int8 and int(st.abs().sum()) == 0 and torch.equal(dec, d)
    assert bool((guard[n_streams * n_per:] == 77).all()), "symbols were written behind the matrix"
    wide, st = B.ans_decode(enc, model, n_per)                       # the int32 decoders agree on the same words
    assert int(st.abs().sum()) == 0 and torch.equal(wide.to(torch.int8), d)


@pytest.mark.parametrize("frac", [0.0, 0.05])
@pytest.mark.parametrize("jump", [0, 8, 32])
def test_int8_per_stream_loops_at_the_maximum_rate(B, O, jump, frac):
    """needle-thin models far from their symbols (~12 bits per symbol); jump = 32: a jump point on every tile"""
    n_streams, n_per = 320, 1024
    model, cdfs, mu, _ = _batch(B, O, n_streams, n_per, jump + int(100 * frac), narrow_models=True)
    rng = np.random.default_rng(5 + jump)
    tails = rng.choice(np.concatenate([np.arange(-127, -40), np.arange(40, 128)]), (n_streams, n_per)).astype(np.int32)
    sym = np.where(rng.random((n_streams, n_per)) < frac, np.rint(mu)[:, None].astype(np.int32), tails).astype(np.int32)
    d = dev(sym.astype(np.int8))
    enc = B.ans_encode(d, model, (32, 64, 12), jump_points=jump)
    want_words, want_n, _ = O.ans_encode_batch(sym, -127, cdfs, 12)
    words, n_words, status = enc.to_numpy()
    assert (status == 0).all() and n_words.tolist() == want_n.tolist()
    for s in range(n_streams):
        assert np.array_equal(words[s, : n_words[s]], want_words[s, : want_n[s]]), f"stream {s}"
    dec, st = B.ans_decode(enc, model, n_per, dtype=torch.int8)
    assert int(st.abs().sum()) == 0 and torch.equal(dec, d)


def test_int8_per_stream_coder_reports_what_int32_reports(B, O):
    """a support that is narrower than the type: impossible symbols at both ends; a jump point beyond its slab"""
    n_streams, n_per, lo, hi = 256, 512, -100, 100
    model, cdfs, _, _ = _batch(B, O, n_streams, n_per, 3, lo=lo, hi=hi)
    sym = np.stack([O.synth_symbols(4, s, 1, n_per, lo, cdfs[s], 12)[0] for s in range(n_streams)]).astype(np.int8)
    bad = sym.copy()
    bad[3, 10] = 101; bad[69, 0] = -128; bad[70, 511] = 127; bad[255, 128] = -101
    for jump in (0, 4):
        enc = B.ans_encode(dev(bad), model, (32, 64, 12), jump_points=jump)
        _, want_n, want_st = O.ans_encode_batch(bad.astype(np.int32), lo, cdfs, 12)
        _, n_words, status = enc.to_numpy()
        assert status.tolist() == want_st.tolist() and sorted(np.flatnonzero(status).tolist()) == [3, 69, 70, 255]
        ok = status == 0
        assert n_words[ok].tolist() == want_n[ok].tolist()
    good = B.ans_encode(dev(sym), model, (32, 64, 12), jump_points=4)
    good.jump.pos[17, 2] = good.words.shape[1] + 5
    dec, st = B.ans_decode_checkpointed(good, good.jump, model, n_per, dtype=torch.int8)
    st = st.cpu().numpy()
    assert st[17, 2] == 3 and st.sum() == 3
    keep = np.ones(n_streams, bool); keep[17] = False
    assert torch.equal(dec[torch.from_numpy(keep).cuda()], dev(sym)[torch.from_numpy(keep).cuda()])


@pytest.mark.parametrize("dtype", [torch.int8, torch.int16], ids=["int8", "int16"])
@pytest.mark.parametrize("n_streams,n_per", [(3, 17), (70, 100), (256, 1000)])
def test_shapes_that_convert(B, O, dtype, n_streams, n_per):
    model, cdfs, _, _ = _batch(B, O, n_streams, n_per, 11)
    sym = np.stack([O.synth_symbols(2, s, 1, n_per, -127, cdfs[s], 12)[0] for s in range(n_streams)])
    d = dev(sym).to(dtype)
    enc = B.ans_encode(d, model, (32, 64, 12))
    want_words, want_n, _ = O.ans_encode_batch(sym, -127, cdfs, 12)
    words, n_words, status = enc.to_numpy()
    assert (status == 0).all() and n_words.tolist() == want_n.tolist()
    for s in range(n_streams):
        assert np.array_equal(words[s, : n_words[s]], want_words[s, : want_n[s]])
    dec, st = B.ans_decode(enc, model, n_per, dtype=dtype)
    assert dec.dtype == dtype and int(st.abs().sum()) == 0 and torch.equal(dec, d)


def test_c3_int8_round_trip_at_full_size_without_conversion_kernels(B, O):
    import bench
    n, k = 65536, 4096
    mu, sigma = bench.c3_parameters(bench.SEED, 0, n, k, torch.device("cuda"))
    model = B.Model.quantized_gaussian_per_stream(-127, 127, mu, sigma, 12)
    sym = bench.synth_symbols_per_stream(bench.SEED, 0, k, -127, model.cdfs_device(), 12)
    d = sym.to(torch.int8)
    enc = B.ans_encode(d, model, (32, 64, 12))
    assert ALT or B.last_kernel() == "ans_encode_pt_n8_kernel<ckpt>"
    assert ALT or (enc.jump is not None and enc.jump.pos.shape == (n, 8))
    plain = B.ans_encode(sym, model, (32, 64, 12), jump_points=0)
    used = torch.arange(plain.words.shape[1], device="cuda")[None, :] < plain.n_words[:, None]
    assert torch.equal(enc.n_words, plain.n_words) and bool(((enc.words == plain.words) | ~used).all())
    dec, st = B.ans_decode(enc, model, k, dtype=torch.int8)
    assert ALT or B.last_kernel() == "ans_decode_pt_sub_n8_kernel"
    assert int(st.abs().sum()) == 0 and torch.equal(dec, d)
    rows = [0, 255, 256, 65535]
    cdfs = np.stack([O.GaussianModel(-127, 127, float(mu[s]), float(sigma[s]), 12, 32).cdf_table() for s in rows])
    want_words, want_n, _ = O.ans_encode_batch(sym[rows].cpu().numpy(), -127, cdfs, 12)
    for i, s in enumerate(rows):
        assert enc.stream(s).tolist() == want_words[i, : want_n[i]].tolist()
