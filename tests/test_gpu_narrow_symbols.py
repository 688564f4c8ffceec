"""GPU tests of narrow symbol matrices (ABI 4: cst_ans_encode_batch_sym / cst_ans_decode_batch_sym, cst_symbols_widen / _narrow):
int8 / int16 matrices give the words of the int32 call on the widened values (the CPU oracle's), and decode back into the narrow
type.  The reference's coders are generic over the symbol type (src/stream/model/quantize.rs:229-255)."""
import os

import numpy as np
import pytest

from kernel_names import with_jump  # noqa: E402

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


@pytest.mark.parametrize("dtype", [torch.int8, torch.int16], ids=["int8", "int16"])
@pytest.mark.parametrize("cfg", [(32, 64, 12), (32, 64, 24), (16, 32, 12)], ids=lambda c: "W%dS%dP%d" % c)
@pytest.mark.parametrize("n_streams,n_per", [(1, 1), (3, 17), (64, 128), (300, 101), (256, 4096)])
@pytest.mark.parametrize("layout", ["stream_major", "symbol_major"])
def test_narrow_matrices_code_like_int32(B, O, dtype, cfg, n_streams, n_per, layout):
    W, S, P = cfg
    lo, hi = (-50, 50) if dtype == torch.int8 else (-300, 300)
    if P < 10 and hi > 100:
        pytest.skip("alphabet larger than 2^P")
    cdf = O.GaussianModel(lo, hi, 3.2, 9.6 if dtype == torch.int8 else 70.0, P, 32 if W == 32 else 16).cdf_table()
    model = B.Model.from_cdf(cdf, lo, P)
    sym = O.synth_symbols(5, 0, n_streams, n_per, lo, cdf, P)
    want_words, want_n, _ = O.ans_encode_batch(sym, lo, cdf, P, W, S)
    host = sym if layout == "stream_major" else np.ascontiguousarray(sym.T)
    narrow = torch.from_numpy(host).to(dtype).cuda()
    enc = B.ans_encode(narrow, model, cfg, layout)
    torch.cuda.synchronize()
    words, n_words, status = enc.to_numpy()
    assert (status == 0).all() and n_words.tolist() == want_n.tolist()
    for s in range(n_streams):
        assert words[s, : n_words[s]].tolist() == want_words[s, : want_n[s]].tolist(), f"stream {s}"
    dec, dstatus = B.ans_decode(enc, model, n_per, layout, dtype=dtype)
    assert dec.dtype == dtype and (dstatus.cpu().numpy() == 0).all()
    assert torch.equal(dec, narrow)


def test_narrow_matrices_report_what_int32_reports(B, O):
    """an impossible symbol in an int8 matrix is an impossible symbol; a support that does not fit the type cannot be decoded into it"""
    P = 12
    cdf = O.GaussianModel(-50, 50, 3.2, 9.6, P, 32).cdf_table()
    model = B.Model.from_cdf(cdf, -50, P)
    sym = O.synth_symbols(6, 0, 70, 64, -50, cdf, P).astype(np.int8)
    sym[3, 10] = 51
    sym[69, 0] = -128
    enc = B.ans_encode(torch.from_numpy(sym).cuda(), model, (32, 64, P))
    st = enc.status.cpu().numpy()
    assert st[3] == 1 and st[69] == 1 and (np.delete(st, [3, 69]) == 0).all()
    wide = O.GaussianModel(-200, 200, 0.0, 50.0, P, 32).cdf_table()
    big = B.Model.from_cdf(wide, -200, P)
    ok = B.ans_encode(torch.from_numpy(O.synth_symbols(7, 0, 64, 64, -200, wide, P)).cuda(), big, (32, 64, P))
    from constriction_amd import _native as N
    with pytest.raises(N.BackendError):
        B.ans_decode(ok, big, 64, dtype=torch.int8)
    dec, _ = B.ans_decode(ok, big, 64, dtype=torch.int16)          # ... but into int16
    assert dec.dtype == torch.int16


# ---- int8 matrices INSIDE the hand-scheduled loops (round 5: cst_ans_n8.hip, ans_encode_pc_n8_kernel) ----

def _aligned_i8(host):
    """an int8 matrix on a 128-byte boundary (what the native kernels ask for; torch's allocator gives 512)"""
    t = torch.from_numpy(np.ascontiguousarray(host)).to(torch.int8).cuda()
    assert t.data_ptr() % 128 == 0
    return t


@pytest.mark.parametrize("jp", [0, "auto"], ids=["plain", "auto_jump"])
@pytest.mark.parametrize("P", [8, 10, 12])
@pytest.mark.parametrize("support", [(-50, 50), (-128, 127), (5, 60), (-128, -100), (0, 0 + 1)], ids=lambda s: "%d..%d" % s)
@pytest.mark.parametrize("n_streams,n_per", [(256, 128), (256, 256), (512, 384), (256, 4096), (768, 1152)])
def test_int8_native_kernels_code_like_the_oracle(B, O, P, support, n_streams, n_per, jp):
    """the kernels that read / write the int8 matrix themselves: words, counts and status of the CPU oracle on the widened values
    (every stream), decoded symbols = the input.  Shapes: one line per row (no previous group to store, no second line to stage),
    two lines, an odd number of lines, the headline row length, several workgroups."""
    lo, hi = support
    if hi - lo + 1 > (1 << P) // 2:
        pytest.skip("alphabet too large for the precision")
    cdf = O.GaussianModel(lo, hi, 0.3 * lo + 0.7 * hi - 20, 9.6, P, 32).cdf_table()
    model = B.Model.from_cdf(cdf, lo, P)
    sym = O.synth_symbols(1000 + P, 0, n_streams, n_per, lo, cdf, P)
    want_words, want_n, _ = O.ans_encode_batch(sym, lo, cdf, P)
    d = _aligned_i8(sym)
    enc = B.ans_encode(d, model, (32, 64, P), jump_points=jp)
    assert ALT or B.last_kernel() == with_jump("ans_encode_pc_n8_kernel", enc)
    torch.cuda.synchronize()
    words, n_words, status = enc.to_numpy()
    assert (status == 0).all() and n_words.tolist() == want_n.tolist()
    for s in range(n_streams):
        assert np.array_equal(words[s, : n_words[s]], want_words[s, : want_n[s]]), f"stream {s}"
    out = torch.full((n_streams, n_per), 99, dtype=torch.int8, device="cuda")
    dec, dstatus = B.ans_decode(enc, model, n_per, out=out)
    assert ALT or B.last_kernel() == "ans_decode_n8_kernel"
    assert dec.dtype == torch.int8 and (dstatus.cpu().numpy() == 0).all()
    assert torch.equal(dec, d)
    # ... and the int32 kernels agree on the same words
    wide, _ = B.ans_decode(enc, model, n_per)
    assert torch.equal(wide.to(torch.int8), d)


def test_int8_native_kernels_report_what_int32_reports(B, O):
    """impossible symbols in the native encoder (below and above the support, at the type's ends), invalid and empty streams in the
    native decoder: the status of the int32 kernels, the other streams untouched"""
    P, n_streams, n_per, lo = 12, 256, 256, -50
    cdf = O.GaussianModel(lo, 50, 3.2, 9.6, P, 32).cdf_table()
    model = B.Model.from_cdf(cdf, lo, P)
    sym = O.synth_symbols(66, 0, n_streams, n_per, lo, cdf, P).astype(np.int8)
    want_words, want_n, _ = O.ans_encode_batch(sym.astype(np.int32), lo, cdf, P)
    bad = sym.copy()
    bad[3, 10] = 51; bad[69, 0] = -128; bad[70, 255] = 127; bad[255, 128] = -51
    enc = B.ans_encode(_aligned_i8(bad), model, (32, 64, P))
    assert ALT or B.last_kernel() == with_jump("ans_encode_pc_n8_kernel", enc)
    torch.cuda.synchronize()
    words, n_words, st = enc.to_numpy()
    flagged = [3, 69, 70, 255]
    assert (st[flagged] == 1).all() and (np.delete(st, flagged) == 0).all() and (n_words[flagged] == 0).all()
    for s in range(n_streams):
        if s not in flagged:
            assert np.array_equal(words[s, : n_words[s]], want_words[s, : want_n[s]]), f"stream {s}"
    # decoder: a stream without words decodes from state 0, a stream whose last word is 0 is invalid data (stack.rs:299-318)
    good = B.ans_encode(_aligned_i8(sym), model, (32, 64, P))
    torch.cuda.synchronize()
    w, n, _ = good.to_numpy()
    w = w.copy(); n = n.copy()
    n[5] = 0; n[64] = 1; w[77, n[77] - 1] = 0
    want, want_st = O.ans_decode_batch(w, n, n_per, lo, cdf, P)
    src = (torch.from_numpy(w.view(np.int32)).cuda(), torch.from_numpy(n.view(np.int32)).cuda())
    got, gst = B.ans_decode(src, model, n_per, config=(32, 64, P), dtype=torch.int8)
    assert ALT or B.last_kernel() == "ans_decode_n8_kernel"
    assert gst.cpu().numpy().tolist() == want_st.tolist() and want_st[77] != 0
    ok = want_st == 0
    assert np.array_equal(got.cpu().numpy()[ok], want[ok].astype(np.int8))


def test_int8_native_decoder_packed_words_and_raw_state(B, O):
    """the native decoder on the packed + offsets form (into_compressed's contiguous words) and continuing from a raw state
    (CST_FLAG_RAW_STATE: the two halves of every row by two calls)"""
    import ctypes as C
    from constriction_amd import _native as N
    P, n_streams, n_per, lo = 12, 256, 512, -50
    cdf = O.GaussianModel(lo, 50, 3.2, 9.6, P, 32).cdf_table()
    model = B.Model.from_cdf(cdf, lo, P)
    sym = O.synth_symbols(67, 0, n_streams, n_per, lo, cdf, P)
    d = _aligned_i8(sym)
    enc = B.ans_encode(d, model, (32, 64, P))
    packed, offsets = B.compact(enc)
    dec, st = B.ans_decode((packed, enc.n_words), model, n_per, offsets=offsets, config=(32, 64, P), dtype=torch.int8)
    assert ALT or B.last_kernel() == "ans_decode_n8_kernel"
    assert torch.equal(dec, d) and (st.cpu().numpy() == 0).all()
    words, n_words, _ = enc.to_numpy()
    state = np.array([(int(words[s, n_words[s] - 1]) << 32) | int(words[s, n_words[s] - 2]) for s in range(n_streams)], dtype=np.uint64)
    d_state = torch.from_numpy(state.view(np.int64)).cuda()
    d_n = torch.from_numpy((n_words - 2).astype(np.int32)).cuda()
    d_n_out = torch.zeros(n_streams, dtype=torch.int32, device="cuda")
    d_status = torch.zeros(n_streams, dtype=torch.int32, device="cuda")
    half = n_per // 2
    outs = []
    scratch = torch.empty(N.lib().cst_symbols_scratch_bytes(n_streams, half, 1), dtype=torch.uint8, device="cuda")   # (only the conversion path touches it)
    for _ in range(2):
        out = torch.empty((n_streams, half), dtype=torch.int8, device="cuda")
        N.check(N.lib().cst_ans_decode_batch_sym(model._h, N.CoderConfig(32, 64, P), C.c_void_p(enc.words.data_ptr()), None, enc.words.shape[1],
                                                 enc.words.numel(), C.c_void_p(d_n.data_ptr()), C.c_void_p(out.data_ptr()), 1, n_streams, half, 0,
                                                 C.c_void_p(d_state.data_ptr()), C.c_void_p(d_n_out.data_ptr()), C.c_void_p(d_status.data_ptr()),
                                                 1, C.c_void_p(scratch.data_ptr()), None), "cst_ans_decode_batch_sym")
        assert ALT or B.last_kernel() == "ans_decode_n8_kernel"
        torch.cuda.synchronize()
        assert (d_status.cpu().numpy() == 0).all()
        d_n.copy_(d_n_out)
        outs.append(out.cpu().numpy())
    assert np.array_equal(np.concatenate(outs, axis=1), sym.astype(np.int8))
    assert (d_n.cpu().numpy() == 0).all()


def test_int8_shapes_the_native_kernels_do_not_take(B, O):
    """rows that are not whole 128-symbol lines, unaligned matrices: the conversion path, same results"""
    P, lo = 12, -50
    cdf = O.GaussianModel(lo, 50, 3.2, 9.6, P, 32).cdf_table()
    model = B.Model.from_cdf(cdf, lo, P)
    for n_streams, n_per, skew in ((256, 100, 0), (100, 100, 0), (256, 128, 1)):
        sym = O.synth_symbols(70 + n_per, 0, n_streams, n_per, lo, cdf, P)
        want_words, want_n, _ = O.ans_encode_batch(sym, lo, cdf, P)
        flat = torch.zeros(n_streams * n_per + 256, dtype=torch.int8, device="cuda")
        d = flat[skew: skew + n_streams * n_per].view(n_streams, n_per)
        d.copy_(torch.from_numpy(sym).to(torch.int8))
        enc = B.ans_encode(d, model, (32, 64, P))
        assert ALT or B.last_kernel() != "ans_encode_pc_n8_kernel"
        torch.cuda.synchronize()
        words, n_words, status = enc.to_numpy()
        assert (status == 0).all() and n_words.tolist() == want_n.tolist()
        out = flat.new_zeros(n_streams * n_per + 256)[skew: skew + n_streams * n_per].view(n_streams, n_per)
        dec, _ = B.ans_decode(enc, model, n_per, out=out)
        assert torch.equal(dec, d)


def _conv_scratch(n_streams, n_per, symbol_bytes=1, interval=0):
    """the scratch of the conversion path (the A/B runs of scripts/alt_paths.sh switch the native kernels off: the same calls then convert)"""
    from constriction_amd import _native as N
    lib = N.lib()
    nbytes = lib.cst_ckpt_sym_scratch_bytes(n_streams, n_per, interval, symbol_bytes) if interval else lib.cst_symbols_scratch_bytes(n_streams, n_per, symbol_bytes)
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device="cuda")


def _pc_name(bytes_, P, jump=False):
    """the encoder kernel's name as cst_last_kernel reports it (12 < P <= 24: the wide step, two word groups per tile)"""
    tags = (["wide"] if P > 12 else []) + (["ckpt"] if jump else [])
    return "ans_encode_pc_n%d_kernel" % (8 * bytes_) + ("<%s>" % ", ".join(tags) if tags else "")


@pytest.mark.parametrize("P", [12, 16, 24])
def test_int8_native_encoder_capacity_and_raw_state(B, O, P):
    """the int8 encoder on slabs that are too small (CST_STREAM_CAPACITY, nothing written behind the slabs) and continuing from a given
    state (CST_FLAG_RAW_STATE: AnsCoder::encode_symbols_reverse on a non-empty coder, stack.rs:784-849): the two halves of every row
    coded by two calls are the row coded by one"""
    import ctypes as C
    from constriction_amd import _native as N
    n_streams, n_per, lo = 512, 512, -50
    cdf = O.GaussianModel(lo, 50, 3.2, 9.6, P, 32).cdf_table()
    model = B.Model.from_cdf(cdf, lo, P)
    sym = O.synth_symbols(78, 0, n_streams, n_per, lo, cdf, P)
    want_words, want_n, _ = O.ans_encode_batch(sym, lo, cdf, P)
    d = _aligned_i8(sym)
    lib, cfg = N.lib(), N.CoderConfig(32, 64, P)
    stride = 32                                           # < the ~90 words a stream needs: every stream overflows its slab
    guard = torch.full((n_streams * stride + 4096,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
    n_words = torch.zeros(n_streams, dtype=torch.int32, device="cuda")
    status = torch.zeros(n_streams, dtype=torch.int32, device="cuda")
    N.check(lib.cst_ans_encode_batch_sym(model._h, cfg, C.c_void_p(d.data_ptr()), 1, n_streams, n_per, 0, C.c_void_p(guard.data_ptr()), stride,
                                         C.c_void_p(n_words.data_ptr()), None, C.c_void_p(status.data_ptr()), 0,
                                         C.c_void_p(_conv_scratch(n_streams, n_per).data_ptr()), None), "capacity")
    assert ALT or B.last_kernel() == _pc_name(1, P)
    torch.cuda.synchronize()
    assert (want_n > stride).all() and (status.cpu().numpy() == 2).all() and (n_words.cpu().numpy() == 0).all()
    assert (guard[n_streams * stride:].cpu().numpy() == 0x5A5A5A5A).all(), "words were written behind the last slab"
    # raw state: the SECOND half of a row is coded first (encoding runs backwards), the first half continues behind its words
    half = n_per // 2
    second, first = d[:, half:].contiguous(), d[:, :half].contiguous()
    assert second.data_ptr() % 128 == 0 and first.data_ptr() % 128 == 0
    full = B.max_words(n_per, (32, 64, P))
    st = torch.zeros(n_streams, dtype=torch.int64, device="cuda")
    wa = torch.zeros((n_streams, full), dtype=torch.int32, device="cuda")
    wb = torch.zeros((n_streams, full), dtype=torch.int32, device="cuda")
    na, nb = torch.zeros(n_streams, dtype=torch.int32, device="cuda"), torch.zeros(n_streams, dtype=torch.int32, device="cuda")
    for part, w, n in ((second, wa, na), (first, wb, nb)):
        N.check(lib.cst_ans_encode_batch_sym(model._h, cfg, C.c_void_p(part.data_ptr()), 1, n_streams, half, 0, C.c_void_p(w.data_ptr()), full,
                                             C.c_void_p(n.data_ptr()), C.c_void_p(st.data_ptr()), C.c_void_p(status.data_ptr()), 1,
                                             C.c_void_p(_conv_scratch(n_streams, half).data_ptr()), None), "raw")
        assert ALT or B.last_kernel() == _pc_name(1, P)
    torch.cuda.synchronize()
    a, b = wa.cpu().numpy().view(np.uint32), wb.cpu().numpy().view(np.uint32)
    ka, kb, state = na.cpu().numpy(), nb.cpu().numpy(), st.cpu().numpy().view(np.uint64)
    for s in range(n_streams):
        tail = [int(state[s] & 0xffffffff), int(state[s] >> 32)] if state[s] >> 32 else [int(state[s])]      # into_compressed: the state's words
        got = a[s, : ka[s]].tolist() + b[s, : kb[s]].tolist() + tail
        assert got == want_words[s, : want_n[s]].tolist(), f"stream {s}"


@pytest.mark.parametrize("n_streams", [1, 70, 257, 330, 582])
@pytest.mark.parametrize("jump", [0, 2])
@pytest.mark.parametrize("P", [12, 24])
def test_int8_native_encoder_takes_partial_workgroups(B, O, n_streams, jump, P):
    """any number of streams: the coder lanes behind the last stream code the last stream again and store nothing (their slabs have
    capacity 0; nothing is written behind the last slab); the jump points they note are the last stream's own.  (A split into a
    native head and a converted tail was measured first and dropped: two launches run one after the other, and a launch lasts as
    long as its longest STREAM -- 65 636 x 4096 int8: 0.63 ms split against 0.54 converted.)"""
    import ctypes as C
    from constriction_amd import _native as N
    n_per, lo = 256, -50
    cdf = O.GaussianModel(lo, 50, 3.2, 9.6, P, 32).cdf_table()
    model = B.Model.from_cdf(cdf, lo, P)
    sym = O.synth_symbols(4000 + n_streams, 0, n_streams, n_per, lo, cdf, P)
    sym[n_streams // 2, 7] = 51 if n_streams > 1 else sym[0, 7]          # an impossible symbol somewhere in the middle
    want_words, want_n, want_st = O.ans_encode_batch(sym, lo, cdf, P)
    d = _aligned_i8(sym)
    stride = B.max_words(n_per, (32, 64, P))
    guard = torch.full((n_streams * stride + 8192,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
    n_words = torch.zeros(n_streams + 64, dtype=torch.int32, device="cuda")
    status = torch.full((n_streams + 64,), -7, dtype=torch.int32, device="cuda")
    k = max(jump, 1)
    pos = torch.full((n_streams + 64, k), -3, dtype=torch.int32, device="cuda")
    state = torch.full((n_streams + 64, k), -3, dtype=torch.int64, device="cuda")
    lib, cfg = N.lib(), N.CoderConfig(32, 64, P)
    if jump:
        N.check(lib.cst_ans_encode_batch_ckpt_sym(model._h, cfg, C.c_void_p(d.data_ptr()), 1, n_streams, n_per, 0, C.c_void_p(guard.data_ptr()), stride,
                                                  C.c_void_p(n_words.data_ptr()), n_per // jump, C.c_void_p(pos.data_ptr()), C.c_void_p(state.data_ptr()),
                                                  C.c_void_p(status.data_ptr()), C.c_void_p(_conv_scratch(n_streams, n_per, 1, n_per // jump).data_ptr()), None),
                "ckpt_sym")
        assert ALT or B.last_kernel() == _pc_name(1, P, True)
    else:
        N.check(lib.cst_ans_encode_batch_sym(model._h, cfg, C.c_void_p(d.data_ptr()), 1, n_streams, n_per, 0, C.c_void_p(guard.data_ptr()), stride,
                                             C.c_void_p(n_words.data_ptr()), None, C.c_void_p(status.data_ptr()), 0,
                                             C.c_void_p(_conv_scratch(n_streams, n_per).data_ptr()), None), "sym")
        assert ALT or B.last_kernel() == _pc_name(1, P)
    torch.cuda.synchronize()
    st, nw = status.cpu().numpy(), n_words.cpu().numpy()
    assert st[:n_streams].tolist() == want_st.tolist() and (st[n_streams:] == -7).all() and (nw[n_streams:] == 0).all()
    words = guard.cpu().numpy().view(np.uint32)
    assert (words[n_streams * stride:] == 0x5A5A5A5A).all(), "words were written behind the last slab"
    rows = words[: n_streams * stride].reshape(n_streams, stride)
    for s in range(n_streams):
        if want_st[s] == 0:
            assert nw[s] == want_n[s] and np.array_equal(rows[s, : want_n[s]], want_words[s, : want_n[s]]), f"stream {s}"
    if jump:
        ok = want_st == 0
        wp, ws = O.ans_jump_table(sym, lo, cdf, P, n_per // jump)
        assert np.array_equal(pos.cpu().numpy().view(np.uint32)[:n_streams][ok], wp[ok])
        assert np.array_equal(state.cpu().numpy().view(np.uint64)[:n_streams][ok], ws[ok])
        assert (pos.cpu().numpy()[n_streams:] == -3).all() and (state.cpu().numpy()[n_streams:] == -3).all(), "jump points were written behind the arrays"


@pytest.mark.parametrize("n_streams", [1, 63, 65, 300, 1000])
@pytest.mark.parametrize("n_per", [128, 512])
def test_int8_native_decoder_takes_partial_waves(B, O, n_streams, n_per):
    """any number of streams: the spare lanes of the last wave decode the last stream again (same reads, same bytes to the same places)"""
    P, lo = 12, -50
    cdf = O.GaussianModel(lo, 50, 3.2, 9.6, P, 32).cdf_table()
    model = B.Model.from_cdf(cdf, lo, P)
    sym = O.synth_symbols(6000 + n_streams, 0, n_streams, n_per, lo, cdf, P)
    enc = B.ans_encode(torch.from_numpy(sym).cuda(), model, (32, 64, P))
    guard = torch.full((n_streams * n_per + 4096,), 77, dtype=torch.int8, device="cuda")
    out = guard[: n_streams * n_per].view(n_streams, n_per)
    dec, st = B.ans_decode(enc, model, n_per, out=out)
    assert ALT or B.last_kernel() == "ans_decode_n8_kernel"
    assert (st.cpu().numpy() == 0).all() and np.array_equal(dec.cpu().numpy(), sym.astype(np.int8))
    assert (guard[n_streams * n_per:].cpu().numpy() == 77).all(), "symbols were written behind the matrix"
    packed, offsets = B.compact(enc)
    out.fill_(55)
    dec, st = B.ans_decode((packed, enc.n_words), model, n_per, offsets=offsets, config=(32, 64, P), out=out)
    assert (st.cpu().numpy() == 0).all() and np.array_equal(dec.cpu().numpy(), sym.astype(np.int8))


# ---- int16 matrices inside the decoder loops (a 128-byte line is 64 symbols: two tiles per pass) ----

@pytest.mark.parametrize("P", [10, 12])
@pytest.mark.parametrize("support", [(-300, 300), (-100, 100), (1000, 1200), (-32768, -32700), (32000, 32767)], ids=lambda s: "%d..%d" % s)
@pytest.mark.parametrize("n_streams,n_per", [(1, 64), (70, 128), (256, 192), (300, 4096), (768, 1152)])
def test_int16_native_decoders_decode_like_the_oracle(B, O, P, support, n_streams, n_per):
    lo, hi = support
    if hi - lo + 1 > (1 << P) // 2:
        pytest.skip("alphabet too large for the precision")
    cdf = O.GaussianModel(lo, hi, 0.4 * lo + 0.6 * hi, 30.0, P, 32).cdf_table()
    model = B.Model.from_cdf(cdf, lo, P)
    sym = O.synth_symbols(2000 + P, 0, n_streams, n_per, lo, cdf, P)
    want_words, want_n, _ = O.ans_encode_batch(sym, lo, cdf, P)
    d = torch.from_numpy(sym).to(torch.int16).cuda()
    enc = B.ans_encode(d, model, (32, 64, P))
    torch.cuda.synchronize()
    words, n_words, status = enc.to_numpy()
    assert (status == 0).all() and n_words.tolist() == want_n.tolist()
    for s in range(0, n_streams, max(1, n_streams // 40)):
        assert np.array_equal(words[s, : n_words[s]], want_words[s, : want_n[s]]), f"stream {s}"
    guard = torch.full((n_streams * n_per + 4096,), 77, dtype=torch.int16, device="cuda")
    out = guard[: n_streams * n_per].view(n_streams, n_per)
    dec, st = B.ans_decode(enc, model, n_per, out=out)
    assert ALT or B.last_kernel() == "ans_decode_n16_kernel"
    assert (st.cpu().numpy() == 0).all() and torch.equal(dec, d)
    assert (guard[n_streams * n_per:].cpu().numpy() == 77).all(), "symbols were written behind the matrix"
    packed, offsets = B.compact(enc)
    out.fill_(55)
    dec, st = B.ans_decode((packed, enc.n_words), model, n_per, offsets=offsets, config=(32, 64, P), out=out)
    assert (st.cpu().numpy() == 0).all() and torch.equal(dec, d)


@pytest.mark.parametrize("P", [10, 12, 16, 24])
@pytest.mark.parametrize("support", [(-300, 300), (-100, 100), (1000, 1200), (-32768, -32700), (32000, 32767), (-2000, -1000)], ids=lambda s: "%d..%d" % s)
@pytest.mark.parametrize("n_streams,n_per,jump", [(1, 64, 0), (70, 128, 2), (256, 192, 3), (300, 4096, 4), (768, 1152, 0), (512, 64, 1)])
def test_int16_native_encoder_codes_like_the_oracle(B, O, P, support, n_streams, n_per, jump):
    """ans_encode_pc_n16_kernel (lines of 64 symbols, table addresses by v_mad_i32_i16): every stream's words, counts and status against
    the CPU oracle on the widened values; impossible symbols below and above the support; partial workgroups; jump points"""
    lo, hi = support
    if hi - lo + 1 > (1 << P) // 2:
        pytest.skip("alphabet too large for the precision")
    cdf = O.GaussianModel(lo, hi, 0.4 * lo + 0.6 * hi, 30.0, P, 32).cdf_table()
    model = B.Model.from_cdf(cdf, lo, P)
    sym = O.synth_symbols(3000 + P, 0, n_streams, n_per, lo, cdf, P)
    bad = []
    if n_streams >= 70:
        for r, v in ((3, lo - 1), (n_streams - 2, hi + 1), (n_streams // 2, -32768), (n_streams // 3, 32767)):
            if -32768 <= v <= 32767 and not lo <= v <= hi:
                sym[r, (7 * r) % n_per] = v
                bad.append(r)
    want_words, want_n, want_st = O.ans_encode_batch(sym, lo, cdf, P)
    assert all(want_st[r] == 1 for r in bad)
    d = torch.from_numpy(sym).to(torch.int16).cuda()
    assert d.data_ptr() % 128 == 0
    if jump:
        enc, ck = B.ans_encode_checkpointed(d, model, n_per // jump, (32, 64, P))
        assert ALT or B.last_kernel() == _pc_name(2, P, True)
    else:
        enc = B.ans_encode(d, model, (32, 64, P))
        assert ALT or B.last_kernel() == with_jump(_pc_name(2, P), enc)
    torch.cuda.synchronize()
    words, n_words, status = enc.to_numpy()
    assert status.tolist() == want_st.tolist()
    for s in range(n_streams):
        if want_st[s] == 0:
            assert n_words[s] == want_n[s] and np.array_equal(words[s, : n_words[s]], want_words[s, : want_n[s]]), f"stream {s}"
    ok = want_st == 0
    if jump:
        wp, ws = O.ans_jump_table(sym, lo, cdf, P, n_per // jump)
        assert np.array_equal(ck.pos.cpu().numpy().view(np.uint32)[ok], wp[ok]) and np.array_equal(ck.state.cpu().numpy().view(np.uint64)[ok], ws[ok])
    if ok.all():
        dec, st = B.ans_decode(enc, model, n_per, dtype=torch.int16)
        assert (st.cpu().numpy() == 0).all() and torch.equal(dec, d)
        if jump and (n_per // jump) % 64 == 0:
            dec, st = B.ans_decode_checkpointed(enc, ck, model, n_per, dtype=torch.int16)
            assert ALT or P > 12 or B.last_kernel() in ("ans_decode_n16_kernel", "ans_decode_small_n16_kernel")
            assert (st.cpu().numpy() == 0).all() and torch.equal(dec, d)


@pytest.mark.parametrize("dtype", [torch.int8, torch.int16, torch.int32], ids=["int8", "int16", "int32"])
def test_jump_points_travel_with_the_batch(B, O, dtype):
    """ans_encode(..., jump_points=k) notes AnsCoder.pos() on its way and hangs the table on the batch; ans_decode finds it there and
    decodes every part on a lane of its own -- the words are the plain encoder's, the symbols the input"""
    P, n_streams, n_per, lo = 12, 512, 1024, -50
    cdf = O.GaussianModel(lo, 50, 3.2, 9.6, P, 32).cdf_table()
    model = B.Model.from_cdf(cdf, lo, P)
    sym = O.synth_symbols(99, 0, n_streams, n_per, lo, cdf, P)
    d = torch.from_numpy(sym).to(dtype).cuda()
    plain = B.ans_encode(d, model, (32, 64, P))
    enc = B.ans_encode(d, model, (32, 64, P), jump_points=4)
    assert enc.jump.pos.shape == (n_streams, 4) and torch.equal(enc.n_words, plain.n_words)
    used = torch.arange(plain.words.shape[1], device="cuda")[None, :] < plain.n_words[:, None]
    assert bool(((enc.words == plain.words) | ~used).all())
    wp, ws = O.ans_jump_table(sym, lo, cdf, P, n_per // 4)
    assert np.array_equal(enc.jump.pos.cpu().numpy().view(np.uint32), wp) and np.array_equal(enc.jump.state.cpu().numpy().view(np.uint64), ws)
    dec, st = B.ans_decode(enc, model, n_per, dtype=dtype)
    assert st.shape == (n_streams,) and int(st.abs().sum()) == 0 and torch.equal(dec, d)
    again = B.ans_encode(d, model, (32, 64, P), jump_points=4, out=enc)          # into the same buffers
    assert again is enc and torch.equal(B.ans_decode(enc, model, n_per, dtype=dtype)[0], d)
    with pytest.raises(ValueError):
        B.ans_encode(d, model, (32, 64, P), jump_points=3)


# ---- narrow matrices at 12 < P <= 24 (the reference's default precision): the bucket-entry decoder writes them itself ----

@pytest.mark.parametrize("dtype", [torch.int8, torch.int16], ids=["int8", "int16"])
@pytest.mark.parametrize("P", [13, 16, 24])
@pytest.mark.parametrize("n_streams,n_per", [(1, 128), (70, 256), (256, 384), (300, 4096), (512, 128)])
def test_narrow_kernels_at_high_precision(B, O, dtype, P, n_streams, n_per):
    lo, hi = (-100, 100) if dtype == torch.int8 else (900, 1150)
    cdf = O.GaussianModel(lo, hi, 0.5 * (lo + hi) + 7.3, 11.0, P, 32).cdf_table()
    model = B.Model.from_cdf(cdf, lo, P)
    sym = O.synth_symbols(7000 + P, 0, n_streams, n_per, lo, cdf, P)
    d = torch.from_numpy(sym).to(dtype).cuda()
    enc = B.ans_encode(d, model, (32, 64, P))
    if d.data_ptr() % 128 == 0 and n_per % (128 // d.element_size()) == 0:
        assert ALT or B.last_kernel() == with_jump(_pc_name(d.element_size(), P), enc)
    torch.cuda.synchronize()
    want_words, want_n, _ = O.ans_encode_batch(sym, lo, cdf, P)
    words, n_words, status = enc.to_numpy()
    assert (status == 0).all() and n_words.tolist() == want_n.tolist()
    for s in range(0, n_streams, max(1, n_streams // 30)):
        assert np.array_equal(words[s, : n_words[s]], want_words[s, : want_n[s]]), f"stream {s}"
    guard = torch.full((n_streams * n_per + 4096,), 77, dtype=dtype, device="cuda")
    out = guard[: n_streams * n_per].view(n_streams, n_per)
    dec, st = B.ans_decode(enc, model, n_per, out=out)
    assert ALT or B.last_kernel() == ("ans_decode_b16_n8_kernel" if dtype == torch.int8 else "ans_decode_b16_n16_kernel")
    assert (st.cpu().numpy() == 0).all() and torch.equal(dec, d)
    assert (guard[n_streams * n_per:].cpu().numpy() == 77).all(), "symbols were written behind the matrix"
    packed, offsets = B.compact(enc)
    out.fill_(55)
    dec, st = B.ans_decode((packed, enc.n_words), model, n_per, offsets=offsets, config=(32, 64, P), out=out)
    assert (st.cpu().numpy() == 0).all() and torch.equal(dec, d)


# ---- the range coder takes narrow tensors through the exported conversions (its kernels code int32) ----

@pytest.mark.parametrize("dtype", [torch.int8, torch.int16], ids=["int8", "int16"])
@pytest.mark.parametrize("cfg", [(32, 64, 12), (32, 64, 24), (16, 32, 12)], ids=lambda c: "W%dS%dP%d" % c)
@pytest.mark.parametrize("layout", ["stream_major", "symbol_major"])
def test_range_coder_takes_narrow_tensors(B, O, dtype, cfg, layout):
    W, S, P = cfg
    lo, hi = (-100, 100) if dtype == torch.int8 else (-300, 300)
    cdf = O.GaussianModel(lo, hi, 2.5, 30.0, P, W).cdf_table()
    model = B.Model.from_cdf(cdf, lo, P)
    n_streams, n_per = 256, 101
    sym = O.synth_symbols(11 + P, 0, n_streams, n_per, lo, cdf, P)
    want_words, want_n, _ = O.rc_encode_batch(sym, lo, cdf, P, W, S)
    d = torch.from_numpy(sym if layout == "stream_major" else np.ascontiguousarray(sym.T)).to(dtype).cuda()
    enc = B.range_encode(d, model, cfg, layout=layout)
    torch.cuda.synchronize()
    words, n_words, status = enc.to_numpy()
    assert (status == 0).all() and n_words.tolist() == want_n.tolist()
    for s in range(n_streams):
        assert np.array_equal(words[s, : n_words[s]], want_words[s, : want_n[s]]), f"stream {s}"
    dec, st = B.range_decode(enc, model, n_per, layout=layout, dtype=dtype)
    assert dec.dtype == dtype and (st.cpu().numpy() == 0).all() and torch.equal(dec, d)
    out = torch.empty_like(d)
    dec, st = B.range_decode(enc, model, n_per, layout=layout, out=out)
    assert dec.data_ptr() == out.data_ptr() and torch.equal(out, d)
    # a support that does not fit the type is refused, as by cst_ans_decode_batch_sym
    wide_model = B.Model.from_cdf(np.array([0, 1 << (P - 1), 1 << P], dtype=np.uint32), 40000, P)
    with pytest.raises(ValueError):
        B.range_decode(enc, wide_model, n_per, layout=layout, dtype=dtype)
