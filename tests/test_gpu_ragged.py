"""GPU parity tests of the ragged entry points (streams of different lengths, one launch): every stream's words against the
CPU oracle coding that stream alone, through the C ABI (`cst_ans_{encode,decode}_ragged`)."""
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


def oracle_words(O, doc, lo, cdf, cfg):
    W, S, P = cfg
    words, n, st = O.ans_encode_batch(np.asarray(doc, dtype=np.int32)[None, :], lo, cdf, P, W, S)
    return words[0, : n[0]], int(st[0])


@pytest.mark.parametrize("cfg", [(32, 64, 24), (32, 64, 12), (16, 32, 12), (32, 64, 16)], ids=lambda c: "W%dS%dP%d" % c)
def test_ragged_streams_vs_oracle(B, O, cfg):
    """Documents of 0 .. 3000 symbols, 1000 of them (partial last wave): words, counts, status of every stream equal the
    oracle's for that stream alone; decoding returns the documents."""
    W, S, P = cfg
    rng = np.random.default_rng(P * 7 + W)
    n_sym, lo = 90, -17
    cdf = O.categorical_fast_cdf(rng.dirichlet(np.ones(n_sym) * 0.4), P)
    model = B.Model.from_cdf(cdf, lo, P)
    lengths = np.concatenate([rng.integers(0, 200, 900), rng.integers(200, 3000, 95), [0, 1, 2, 3, 4]])
    rng.shuffle(lengths)
    docs = [O.synth_symbols(int(k), 0, 1, int(n), lo, cdf, P)[0] if n else np.zeros(0, np.int32) for k, n in enumerate(lengths)]
    flat, offsets = B.ragged(docs)
    enc = B.ans_encode_ragged(flat, offsets, model, cfg)
    torch.cuda.synchronize()
    assert (enc.status.cpu().numpy() == 0).all()
    n_words = enc.n_words.cpu().numpy()
    for s in list(range(0, len(docs), 7)) + [len(docs) - 1]:
        want, st = oracle_words(O, docs[s], lo, cdf, cfg)
        assert st == 0 and n_words[s] == len(want) and enc.stream(s).tolist() == want.tolist(), f"stream {s} of {len(docs[s])} symbols"
    dec, status = B.ans_decode_ragged(enc, model, offsets)
    torch.cuda.synchronize()
    assert (status.cpu().numpy() == 0).all() and torch.equal(dec, flat)


def test_ragged_like_the_reference_index(B, O):
    """tests/issue52.rs: documents over a small alphabet, one coder each, an EOF symbol first (decoded last); the words of
    a document are those of the drop-in AnsCoder for it (constriction_amd.stream.stack), i.e. the reference's."""
    from constriction_amd.stream import model as M, stack
    text = ["the quick brown fox", "", "jumps", "over the lazy dog " * 40, "a"] * 30
    alphabet = sorted(set("".join(text)))
    eof = len(alphabet)
    probs = np.ones(eof + 1) / (eof + 1)
    docs = [np.array([alphabet.index(c) for c in doc] + [eof], dtype=np.int32) for doc in text]     # reversed coding: EOF is encoded first
    cdf = O.categorical_fast_cdf(probs, 24)
    model = B.Model.from_cdf(cdf, 0, 24)
    flat, offsets = B.ragged(docs)
    enc = B.ans_encode_ragged(flat, offsets, model)
    torch.cuda.synchronize()
    single = M.Categorical(probs, perfect=False)
    for s in (0, 1, 2, 3, 4, len(docs) - 1):
        coder = stack.AnsCoder()
        coder.encode_reverse(docs[s], single)
        assert enc.stream(s).tolist() == coder.get_compressed().tolist()
    dec, status = B.ans_decode_ragged(enc, model, offsets)
    out, off = dec.cpu().numpy(), offsets.cpu().numpy()
    assert ["".join(alphabet[i] for i in out[off[s]: off[s + 1] - 1]) for s in range(len(docs))] == text
    assert all(out[off[s + 1] - 1] == eof for s in range(len(docs)))
    # ... and as the reference's decompress does it: no lengths stored, every document ends at its EOF symbol
    dec2, off2, status2 = B.ans_decode_until(enc, model, eof)
    torch.cuda.synchronize()
    assert (status2.cpu().numpy() == 0).all() and torch.equal(off2, offsets) and torch.equal(dec2, dec)
    # a limit below the longest document: those streams report CAPACITY and decode to NOTHING (a batch of documents without a
    # terminator -- corrupt data, a wrong eof symbol -- must not ask for n_streams x max_symbols symbols), the others are unaffected
    dec3, off3, status3 = B.ans_decode_until(enc, model, eof, max_symbols=100)
    lens, lens3 = np.diff(off), np.diff(off3.cpu().numpy())
    assert lens3.tolist() == np.where(lens > 100, 0, lens).tolist()
    assert status3.cpu().tolist() == [2 if n > 100 else 0 for n in lens]
    o3 = dec3.cpu().numpy()
    assert all(o3[off3[s]: off3[s + 1]].tolist() == out[off[s]: off[s] + lens3[s]].tolist() for s in range(len(docs)))
    # no document has a terminator at all (a wrong eof symbol): nothing is decoded, every stream says so
    dec4, off4, status4 = B.ans_decode_until(enc, model, eof + 5, max_symbols=1 << 16)
    assert dec4.numel() == 0 and int(off4[-1]) == 0 and (status4.cpu().numpy() == 2).all()


def test_ragged_status_and_bounds(B, O):
    """an impossible symbol flags its stream only; corrupt counts / offsets are decoded as empty streams (INVALID_DATA) and
    nothing outside the buffer is read; no streams at all is a no-op"""
    P, lo = 12, 0
    cdf = O.categorical_fast_cdf(np.ones(20) / 20, P)
    model = B.Model.from_cdf(cdf, lo, P)
    docs = [np.arange(n) % 20 for n in (5, 64, 0, 300, 17)]
    docs[3] = docs[3].copy(); docs[3][100] = 20
    flat, offsets = B.ragged(docs)
    enc = B.ans_encode_ragged(flat, offsets, model, (32, 64, P))
    torch.cuda.synchronize()
    assert enc.status.cpu().tolist() == [0, 0, 0, 1, 0] and enc.n_words.cpu().tolist()[3] == 0
    good = B.ragged([d for k, d in enumerate(docs) if k != 3])
    enc = B.ans_encode_ragged(*good, model, (32, 64, P))
    ref, _ = B.ans_decode_ragged(enc, model, good[1])
    enc.n_words[1] = 1 << 30                      # leaves the buffer
    enc.word_offsets[2] = 1 << 40
    dec, status = B.ans_decode_ragged(enc, model, good[1])
    torch.cuda.synchronize()
    assert status.cpu().tolist() == [0, 3, 3, 0]
    off = good[1].cpu().numpy()
    assert torch.equal(dec[off[3]: off[4]], ref[off[3]: off[4]]) and torch.equal(dec[: off[1]], ref[: off[1]])
    empty = B.ans_encode_ragged(torch.zeros(0, dtype=torch.int32, device="cuda"), torch.zeros(1, dtype=torch.int64, device="cuda"), model, (32, 64, P))
    assert empty.n_words.numel() == 0
    # word offsets that run BACKWARDS (corrupt metadata on the encoder side): that stream gets a slab of no words -- CAPACITY,
    # nothing written -- instead of an "unbounded" one over its neighbours
    import ctypes as C
    from constriction_amd import _native as N
    flat, offsets = good
    n = offsets.numel() - 1
    woff = torch.tensor([0, 64, 32, 512, 1024], dtype=torch.int64, device="cuda")      # stream 1: [64, 32)
    words = torch.full((2048,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
    n_words = torch.zeros(n, dtype=torch.int32, device="cuda")
    status = torch.zeros(n, dtype=torch.int32, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    N.check(N.lib().cst_ans_encode_ragged(model._h, N.CoderConfig(32, 64, P), p(flat), p(offsets), n, p(words), p(woff), 0, p(n_words), p(status), None),
            "cst_ans_encode_ragged")
    torch.cuda.synchronize()
    assert status.cpu().tolist()[1] == 2 and n_words.cpu().tolist()[1] == 0
    assert status.cpu().tolist()[0] == 0 and status.cpu().tolist()[3] == 0
    w = words.cpu().numpy()
    assert (w[1024:] == 0x5A5A5A5A).all() and (w[32:64][n_words.cpu().numpy()[0]:] == 0x5A5A5A5A).all()


@pytest.mark.parametrize("cfg,n_sym", [((32, 64, 6), 40), ((16, 32, 5), 20), ((32, 64, 16), 5000), ((32, 64, 20), 20000), ((16, 32, 16), 5000),
                                       ((32, 64, 22), 1000), ((32, 64, 24), 256)],
                         ids=lambda v: "W%dS%dP%d" % v if isinstance(v, tuple) else "n%d" % v)
def test_ragged_kernel_variants(B, O, cfg, n_sym):
    """Every variant of the ragged kernels (cst_ans_ragged.hip): the generic coder steps (P < 8, 16-bit words) and the
    hand-scheduled ones, encoder tables in LDS (<= 4096 symbols) and in HBM, decoder tables as 16-byte bucket entries in LDS
    (<= 1024 symbols), as cdf + 16-bit bucket index in LDS, and in HBM (more than 64 KiB of cdf) -- documents whose lengths
    cover every residue of the group of eight, a partial last wave; words of every 5th document and the terminator-delimited
    decode against the oracle's coder for that document alone."""
    W, S, P = cfg
    rng = np.random.default_rng(n_sym + P)
    lo = -3
    w = rng.gamma(0.3, 1.0, n_sym) + 1e-9
    p = np.maximum(1, np.floor(w / w.sum() * ((1 << P) - n_sym)).astype(np.int64))
    p[int(np.argmax(p))] += (1 << P) - int(p.sum())
    cdf = np.concatenate([[0], np.cumsum(p)]).astype(np.uint32)
    model = B.Model.from_cdf(cdf, lo, P)
    lengths = np.concatenate([np.arange(0, 41), rng.integers(0, 700, 150)])
    eof = lo + int(np.argmin(p))
    docs = []
    for n in lengths:
        d = lo + rng.choice(n_sym, size=int(n), p=p / p.sum()).astype(np.int32)
        d[d == eof] = lo + int(np.argmax(p))
        docs.append(np.concatenate([[eof], d]).astype(np.int32))          # coded in reverse: the terminator is decoded last
    flat, offsets = B.ragged(docs)
    enc = B.ans_encode_ragged(flat, offsets, model, cfg)
    torch.cuda.synchronize()
    assert (enc.status.cpu().numpy() == 0).all()
    n_words = enc.n_words.cpu().numpy()
    for s in range(0, len(docs), 5):
        want, st = oracle_words(O, docs[s], lo, cdf, cfg)
        assert st == 0 and n_words[s] == len(want) and enc.stream(s).tolist() == want.tolist(), f"stream {s} of {len(docs[s])} symbols"
    dec, status = B.ans_decode_ragged(enc, model, offsets)
    torch.cuda.synchronize()
    assert (status.cpu().numpy() == 0).all() and torch.equal(dec, flat)
    # decoding yields the document from its first symbol on: a stream that STARTS with the terminator is one symbol long for
    # ans_decode_until -- so code the documents reversed (terminator last in decoding order)
    rdocs = [d[::-1].copy() for d in docs]
    rflat, roff = B.ragged(rdocs)
    renc = B.ans_encode_ragged(rflat, roff, model, cfg)
    dec2, off2, status2 = B.ans_decode_until(renc, model, eof)
    torch.cuda.synchronize()
    assert (status2.cpu().numpy() == 0).all() and torch.equal(off2, roff) and torch.equal(dec2, rflat)


def test_ragged_schedule_does_not_change_results(B, O):
    """`*_ragged_ordered`: lane slot i codes stream order[i].  Sorted by length, shuffled, or the identity: the same words,
    counts, symbols and status per STREAM; an entry that is not a stream index leaves its slot idle (its stream is not
    coded, nothing outside the arrays is touched)."""
    P, lo, cfg = 24, 0, (32, 64, 24)
    rng = np.random.default_rng(5)
    cdf = O.categorical_fast_cdf(rng.dirichlet(np.ones(50) * 0.5), P)
    model = B.Model.from_cdf(cdf, lo, P)
    lengths = np.exp(rng.uniform(np.log(1), np.log(1500), 700)).astype(np.int64)
    docs = [rng.integers(0, 50, int(n)).astype(np.int32) for n in lengths]
    flat, offsets = B.ragged(docs)
    ref = B.ans_encode_ragged(flat, offsets, model, cfg, order=None)
    ref_dec, ref_st = B.ans_decode_ragged(ref, model, offsets, order=None)
    assert ref.order is None and torch.equal(ref_dec, flat)
    perm = torch.from_numpy(rng.permutation(len(docs)).astype(np.int32)).cuda()
    for order in ("sorted", perm):
        enc = B.ans_encode_ragged(flat, offsets, model, cfg, order=order)
        torch.cuda.synchronize()
        assert enc.order is not None and sorted(enc.order.cpu().tolist()) == list(range(len(docs)))
        assert torch.equal(enc.n_words, ref.n_words) and torch.equal(enc.status, ref.status)
        for s in range(0, len(docs), 9):
            assert enc.stream(s).tolist() == ref.stream(s).tolist()
        for dec_order in ("auto", None, "sorted"):
            dec, st = B.ans_decode_ragged(enc, model, offsets, order=dec_order)
            assert torch.equal(dec, flat) and torch.equal(st, ref_st)
    srt = B.ragged_order(torch.from_numpy(lengths).cuda())
    assert np.all(np.diff(lengths[srt.cpu().numpy()]) <= 0)
    # entries that are no stream indices: their slots idle, the streams they displaced keep what the buffers held
    bad = torch.arange(len(docs), dtype=torch.int32, device="cuda")
    bad[3] = -1
    bad[10] = len(docs)
    enc = B.ans_encode_ragged(flat, offsets, model, cfg, order=bad)
    torch.cuda.synchronize()
    keep = np.ones(len(docs), bool); keep[[3, 10]] = False
    assert torch.equal(enc.n_words[torch.from_numpy(keep).cuda()], ref.n_words[torch.from_numpy(keep).cuda()])
    out = torch.full_like(flat, -7)
    dec, st = B.ans_decode_ragged(ref, model, offsets, out=out, order=bad)
    off = offsets.cpu().numpy()
    assert (dec[off[3]: off[4]] == -7).all() and (dec[off[10]: off[11]] == -7).all()
    assert torch.equal(dec[off[11]:], flat[off[11]:]) and torch.equal(dec[: off[3]], flat[: off[3]])


# ---- jump points for ragged batches (round 6: cst_ans_{encode,decode}_ragged_jump) ----

@pytest.mark.parametrize("cfg", [(32, 64, 24), (32, 64, 12), (16, 32, 12)], ids=lambda c: "W%dS%dP%d" % c)
@pytest.mark.parametrize("every", [8, 64, 256, 1024])
def test_ragged_jump_points(B, O, cfg, every):
    """AnsCoder.pos() in front of every `every` symbols of every document, noted by the encoder on its way: the words are the plain
    call's, the table is the CPU oracle's for each document alone, and the decoder that runs the chunks side by side returns the
    documents (empty documents, documents shorter than a chunk, lengths that are not multiples of anything)."""
    W, S, P = cfg
    rng = np.random.default_rng(P + every)
    n_sym, lo = 90, -17
    cdf = O.categorical_fast_cdf(rng.dirichlet(np.ones(n_sym) * 0.4), P)
    model = B.Model.from_cdf(cdf, lo, P)
    lengths = np.concatenate([rng.integers(0, 200, 400), rng.integers(200, 3000, 60), [0, 1, 7, 8, 9, every - 1, every, every + 1, 2 * every, 2047]])
    rng.shuffle(lengths)
    docs = [O.synth_symbols(int(k), 0, 1, int(n), lo, cdf, P)[0] if n else np.zeros(0, np.int32) for k, n in enumerate(lengths)]
    flat, offsets = B.ragged(docs)
    plain = B.ans_encode_ragged(flat, offsets, model, cfg, jump_every=0)
    enc = B.ans_encode_ragged(flat, offsets, model, cfg, jump_every=every)
    assert B.last_kernel() == "ans_encode_ragged_kernel<jump>"
    assert plain.jump is None and enc.jump is not None and enc.jump.interval == every
    assert torch.equal(enc.n_words, plain.n_words) and torch.equal(enc.status, plain.status) and int(enc.status.abs().sum()) == 0
    for s in range(0, len(docs), 5):
        assert enc.stream(s).tolist() == plain.stream(s).tolist(), f"stream {s}"
    co = enc.jump.chunk_offsets.cpu().numpy()
    assert co.tolist() == np.concatenate([[0], np.cumsum((lengths + every - 1) // every)]).tolist()
    pos, state = enc.jump.pos.cpu().numpy().view(np.uint32), enc.jump.state.cpu().numpy().view(np.uint64)
    for s in list(range(0, len(docs), 9)) + [int(np.argmax(lengths))]:
        if len(docs[s]) == 0:
            continue
        wp, ws = O.ans_jump_table(docs[s][None, :], lo, cdf, P, every, W, S)
        assert pos[co[s]: co[s + 1]].tolist() == wp[0].tolist() and state[co[s]: co[s + 1]].tolist() == ws[0].tolist(), f"stream {s} ({len(docs[s])} symbols)"
    dec, status = B.ans_decode_ragged(enc, model, offsets)
    assert B.last_kernel() == "ans_decode_ragged_kernel<jump>"
    assert int(status.abs().sum()) == 0 and torch.equal(dec, flat)
    dec2, status2 = B.ans_decode_ragged(plain, model, offsets)
    assert torch.equal(dec2, flat) and torch.equal(status2, status)


def test_ragged_jump_points_are_checked(B, O):
    """a table that does not describe its streams, a jump point beyond its stream's words: INVALID_DATA for that stream, the others
    decode; no access outside the buffers (guarded output)"""
    P, lo = 12, 0
    cdf = O.categorical_fast_cdf(np.ones(40) / 40, P)
    model = B.Model.from_cdf(cdf, lo, P)
    rng = np.random.default_rng(2)
    lengths = rng.integers(300, 900, 200)
    docs = [O.synth_symbols(int(k), 0, 1, int(n), lo, cdf, P)[0] for k, n in enumerate(lengths)]
    flat, offsets = B.ragged(docs)
    enc = B.ans_encode_ragged(flat, offsets, model, (32, 64, P), jump_every=64)
    co = enc.jump.chunk_offsets.cpu().numpy()
    enc.jump.pos[int(co[17]) + 2] = 1 << 30                       # a jump point beyond everything
    guard = torch.full((flat.numel() + 4096,), 77, dtype=torch.int32, device="cuda")
    dec, status = B.ans_decode_ragged(enc, model, offsets, out=guard[: flat.numel()])
    st = status.cpu().numpy()
    assert st[17] == 3 and st.sum() == 3
    assert bool((guard[flat.numel():] == 77).all())
    off = offsets.cpu().numpy()
    ok = np.ones(flat.numel(), bool); ok[off[17]: off[18]] = False
    assert torch.equal(dec[torch.from_numpy(ok).cuda()], flat[torch.from_numpy(ok).cuda()])
    # a table made for other lengths: one chunk too few for stream 5
    bad = enc.jump.chunk_offsets.clone(); bad[6:] -= 1
    enc2 = B.ans_encode_ragged(flat, offsets, model, (32, 64, P), jump_every=64)
    enc2.jump.chunk_offsets = bad
    _, status = B.ans_decode_ragged(enc2, model, offsets)
    assert int(status[5]) == 3
