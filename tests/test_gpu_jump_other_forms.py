"""Jump points (the reference's Pos / Seek, src/stream/stack.rs:1107-1139) on the two batch forms the checkpointing kernels do not
take themselves: the (16,32) preset with PACKED words (the reference's Vec<u16>) and SYMBOL-MAJOR symbol matrices.  AnsCoder::pos()
is (words in the bulk, state) whatever the word type or the matrix layout: the table is the oracle's, the words are those of the
plain call, and decoding from the jump points (batched.ans_decode / ans_decode_checkpointed: the plain batched decoder on the
chunks as streams of their own, raw states) gives the input back -- chunk by chunk, too, which is what Seek is for."""
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


def _model(B, O, P, W):
    lo, hi = -50, 50
    cdf = O.GaussianModel(lo, hi, 3.2, 9.6, P, 32 if W == 32 else 16).cdf_table()
    return B.Model.from_cdf(cdf, lo, P), cdf, lo


@pytest.mark.parametrize("k", [2, 4, 16])
@pytest.mark.parametrize("n_streams,n_per,P", [(256, 1024, 12), (300, 2048, 12), (77, 512, 8), (1024, 4096, 12)])
def test_packed16_jump_points(B, O, n_streams, n_per, P, k):
    model, cdf, lo = _model(B, O, P, 16)
    sym = O.synth_symbols(31 + k, 0, n_streams, n_per, lo, cdf, P)
    d = dev(sym)
    cfg = (16, 32, P)
    plain = B.ans_encode(d, model, cfg, packed16=True)
    assert plain.jump is None                                   # "auto" notes none on the packed preset
    enc = B.ans_encode(d, model, cfg, packed16=True, jump_points=k)
    assert B.last_kernel() == "ans_encode_w16pk_kernel<ckpt>"           # chunks of whole tiles: the packed encoder notes them on its way
    assert enc.packed16 and enc.jump is not None and enc.jump.pos.shape == (n_streams, k) and enc.jump.interval == n_per // k
    assert torch.equal(enc.n_words, plain.n_words) and torch.equal(enc.status, plain.status)
    used = torch.arange(enc.words.shape[1], device="cuda")[None, :] < enc.n_words[:, None]
    assert bool(((enc.words == plain.words) | ~used).all())
    wp, ws = O.ans_jump_table(sym, lo, cdf, P, n_per // k, W=16, S=32)
    assert np.array_equal(enc.jump.pos.cpu().numpy().view(np.uint32), wp)
    assert np.array_equal(enc.jump.state.cpu().numpy().view(np.uint64), ws)
    for s in (0, n_streams - 1):                                # the words are the oracle's
        want_words, want_n, _ = O.ans_encode_batch(sym[s: s + 1], lo, cdf, P, W=16, S=32)
        assert enc.stream(s).tolist() == want_words[0, : want_n[0]].tolist()
    dec, st = B.ans_decode(enc, model, n_per)                   # finds the table on the batch
    assert st.shape == (n_streams,) and int(st.abs().sum()) == 0 and torch.equal(dec, d)
    dec, st = B.ans_decode_checkpointed(enc, enc.jump, model, n_per)
    assert st.shape == (n_streams, k) and int(st.abs().sum()) == 0 and torch.equal(dec, d)
    dec, st = B.ans_decode(plain, model, n_per)
    assert torch.equal(dec, d)


@pytest.mark.parametrize("n_streams,n_per,interval", [(256, 1024, 32), (100, 640, 64), (65, 96, 96), (3, 2048, 16), (40, 1000, 250), (256, 1056, 48)])
def test_packed16_jump_points_every_tile_at_the_maximum_word_rate(B, O, n_streams, n_per, interval):
    """every symbol costs P bits (the rarest symbols only) and a jump point sits on every tile (or on odd chunks: the scratch-slab
    path): words and tables against the oracle, every chunk decodes from its point"""
    P, lo = 12, 0
    p = np.full(64, 1, dtype=np.int64); p[0] = (1 << P) - 63
    cdf = np.concatenate([[0], np.cumsum(p)]).astype(np.uint32)
    model = B.Model.from_cdf(cdf, lo, P)
    rng = np.random.default_rng(n_streams + n_per)
    sym = rng.integers(1, 64, (n_streams, n_per)).astype(np.int32)
    d = dev(sym)
    k = n_per // interval
    enc = B.ans_encode(d, model, (16, 32, P), packed16=True, jump_points=k)
    fast = interval % 32 == 0
    assert (B.last_kernel() == "ans_encode_w16pk_kernel<ckpt>") == fast
    want_words, want_n, want_st = O.ans_encode_batch(sym, lo, cdf, P, W=16, S=32)
    words, n_words, status = enc.to_numpy()
    assert (status == 0).all() and n_words.tolist() == want_n.tolist()
    for s in range(n_streams):
        assert np.array_equal(words[s, : n_words[s]], want_words[s, : want_n[s]].astype(np.uint16)), f"stream {s}"
    wp, ws = O.ans_jump_table(sym, lo, cdf, P, interval, W=16, S=32)
    assert np.array_equal(enc.jump.pos.cpu().numpy().view(np.uint32), wp) and np.array_equal(enc.jump.state.cpu().numpy().view(np.uint64), ws)
    dec, st = B.ans_decode_checkpointed(enc, enc.jump, model, n_per)
    assert int(st.abs().sum()) == 0 and torch.equal(dec, d)


def test_packed16_bad_jump_point_is_flagged(B, O):
    model, cdf, lo = _model(B, O, 12, 16)
    n_streams, n_per, k = 200, 1024, 4
    d = dev(O.synth_symbols(5, 0, n_streams, n_per, lo, cdf, 12))
    enc = B.ans_encode(d, model, (16, 32, 12), packed16=True, jump_points=k)
    bad = B.Checkpoints(enc.jump.interval, enc.jump.pos.clone(), enc.jump.state.clone())
    bad.pos[7, 2] = 1 << 29
    bad.pos[9, 0] = int(enc.n_words[9]) + 1
    dec, st = B.ans_decode_checkpointed(enc, bad, model, n_per)
    assert int(st[7, 2]) != 0 and int(st[9, 0]) != 0 and int((st != 0).sum()) == 2
    ok = st == 0
    assert torch.equal(dec.view(n_streams, k, -1)[ok], d.view(n_streams, k, -1)[ok])


@pytest.mark.parametrize("k", [2, 8])
@pytest.mark.parametrize("n_streams,n_per,W,P", [(256, 1024, 32, 12), (320, 2048, 32, 24), (256, 512, 16, 12), (1000, 1024, 32, 12)])
def test_symbol_major_jump_points(B, O, n_streams, n_per, W, P, k):
    model, cdf, lo = _model(B, O, P, W)
    cfg = (W, 2 * W, P)
    sym = O.synth_symbols(77 + k, 0, n_streams, n_per, lo, cdf, P)
    d = dev(np.ascontiguousarray(sym.T))                        # [n_per][n_streams]
    plain = B.ans_encode(d, model, cfg, layout="symbol_major")
    assert plain.jump is None
    enc = B.ans_encode(d, model, cfg, layout="symbol_major", jump_points=k)
    assert enc.jump is not None and enc.jump.pos.shape == (n_streams, k)
    assert torch.equal(enc.n_words, plain.n_words)
    used = torch.arange(enc.words.shape[1], device="cuda")[None, :] < enc.n_words[:, None]
    assert bool(((enc.words == plain.words) | ~used).all())
    wp, ws = O.ans_jump_table(sym, lo, cdf, P, n_per // k, W=W, S=2 * W)
    assert np.array_equal(enc.jump.pos.cpu().numpy().view(np.uint32), wp)
    assert np.array_equal(enc.jump.state.cpu().numpy().view(np.uint64), ws)
    dec, st = B.ans_decode(enc, model, n_per, layout="symbol_major")
    assert dec.shape == (n_per, n_streams) and int(st.abs().sum()) == 0 and torch.equal(dec, d)
    dec, st = B.ans_decode_checkpointed(enc, enc.jump, model, n_per, layout="symbol_major")
    assert st.shape == (n_streams, k) and int(st.abs().sum()) == 0 and torch.equal(dec, d)
    # Seek: ONE chunk of all streams, from its jump point alone (the plain decoder with raw states on the chunk's rows)
    j, K = k - 1, n_per // k
    one = B.Checkpoints(K, enc.jump.pos[:, j: j + 1].contiguous(), enc.jump.state[:, j: j + 1].contiguous())
    part, st = B.ans_decode_checkpointed(enc, one, model, K, layout="symbol_major")
    assert int(st.abs().sum()) == 0 and torch.equal(part, d[j * K:(j + 1) * K])


def test_decoding_never_changes_a_jump_table(B, O):
    """CST_FLAG_RAW_STATE makes the decoders' state array in AND out: every path that decodes from jump points must hand them a copy.
    Decode every family twice from the same batch; the table is what it was."""
    lo, P, n, N = -50, 12, 512, 2048
    cdf = O.GaussianModel(lo, 50, 3.2, 9.6, P, 32).cdf_table()
    model = B.Model.from_cdf(cdf, lo, P)
    cdf16 = O.GaussianModel(lo, 50, 3.2, 9.6, P, 16).cdf_table()
    model16 = B.Model.from_cdf(cdf16, lo, P)
    d = dev(O.synth_symbols(3, 0, n, N, lo, cdf, P))
    d16 = dev(O.synth_symbols(3, 0, n, N, lo, cdf16, P))

    def twice(enc, decode, want):
        if enc.jump is None:            # (an alternate kernel path is forced -- scripts/alt_paths.sh -- and the library took no points)
            return
        before = {k: v.clone() for k, v in vars(enc.jump).items() if isinstance(v, torch.Tensor)}
        a, _ = decode(enc)
        b, _ = decode(enc)
        assert torch.equal(a, want) and torch.equal(b, want)
        assert all(torch.equal(v, getattr(enc.jump, k)) for k, v in before.items())

    for dt in (torch.int32, torch.int8):
        x = d.to(dt)
        twice(B.ans_encode(x, model, (32, 64, P)), lambda e: B.ans_decode(e, model, N, dtype=dt), x)
        twice(B.range_encode(x, model, (32, 64, P)), lambda e: B.range_decode(e, model, N, dtype=dt), x)
    twice(B.ans_encode(d16, model16, (16, 32, P), jump_points=2), lambda e: B.ans_decode(e, model16, N), d16)
    twice(B.ans_encode(d16, model16, (16, 32, P), packed16=True, jump_points=4), lambda e: B.ans_decode(e, model16, N), d16)
    dt_ = d.t().contiguous()
    twice(B.ans_encode(dt_, model, (32, 64, P), layout="symbol_major", jump_points=2), lambda e: B.ans_decode(e, model, N, layout="symbol_major"), dt_)
    rng = np.random.default_rng(1)
    mu, sd = rng.uniform(-10, 10, n), np.exp(rng.uniform(np.log(0.5), np.log(16.0), n))
    pt = B.Model.quantized_gaussian_per_stream(-127, 127, dev(mu), dev(sd), 12)
    u = torch.randint(0, 4096, (n, N), device="cuda")
    sp = (torch.searchsorted(pt.cdfs_device().to(torch.int64), u, right=True) - 1 - 127).to(torch.int32)
    twice(B.ans_encode(sp, pt, (32, 64, 12)), lambda e: B.ans_decode(e, pt, N), sp)
    ns, ng = 16384, 512
    g = torch.Generator(device="cuda").manual_seed(5)
    mu = (torch.rand((ns, ng), generator=g, device="cuda", dtype=torch.float64) - 0.5) * 60
    sg = torch.exp(torch.rand((ns, ng), generator=g, device="cuda", dtype=torch.float64) * 4 - 1)
    sy = torch.clamp(torch.round(mu + sg * torch.randn((ns, ng), generator=g, device="cuda", dtype=torch.float64)), -100, 100).to(torch.int32)
    twice(B.ans_encode_gaussian(sy, -100, 100, mu, sg), lambda e: B.ans_decode_gaussian(e, -100, 100, mu, sg), sy)
    twice(B.range_encode_gaussian(sy, -100, 100, mu, sg), lambda e: B.range_decode_gaussian(e, -100, 100, mu, sg), sy)
