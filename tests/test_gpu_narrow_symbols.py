"""GPU tests of narrow symbol matrices (ABI 4: cst_ans_encode_batch_sym / cst_ans_decode_batch_sym, cst_symbols_widen / _narrow):
int8 / int16 matrices give the words of the int32 call on the widened values (the CPU oracle's), and decode back into the narrow
type.  The reference's coders are generic over the symbol type (src/stream/model/quantize.rs:229-255)."""
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
