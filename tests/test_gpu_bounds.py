"""GPU tests of memory safety on corrupt metadata (`words_capacity` of the decode entry points, ABI 3).

The reference decoder pops from a Vec and cannot read outside it (src/backends.rs:495-507); the batched decoders take
counts and offsets from the caller, so a slice that leaves the caller's buffer must become CST_STREAM_INVALID_DATA, not an
out-of-bounds read.  Every test decodes a batch in which SOME streams carry corrupt counts / offsets -- far outside the
allocation: an unchecked kernel faults on them -- and checks that exactly those streams are flagged while every other
stream still decodes to its symbols."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

INVALID = 3


@pytest.fixture(scope="module")
def B():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    from constriction_amd import batched
    return batched


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _symbols(rng, n_streams, n_per, lo=-50, hi=50):
    return np.clip(np.rint(rng.normal(3.2, 9.6, (n_streams, n_per))), lo, hi).astype(np.int32)


CASES = [("ans", (32, 64, 12)), ("ans", (32, 64, 24)), ("ans", (16, 32, 12)), ("range", (32, 64, 12)), ("range", (32, 64, 24))]


@pytest.mark.parametrize("coder,config", CASES, ids=lambda c: str(c))
@pytest.mark.parametrize("n_streams,n_per", [(256, 4096), (70, 300), (512, 96)])
def test_corrupt_counts_in_slabs(B, coder, config, n_streams, n_per):
    """slab form: a count larger than the slab (checked always) and one that would run off the end of the buffer"""
    rng = np.random.default_rng(n_streams + n_per)
    sym = _symbols(rng, n_streams, n_per)
    model = B.Model.quantized_gaussian(-50, 50, 3.2, 9.6, config[2])
    enc_f, dec_f = (B.ans_encode, B.ans_decode) if coder == "ans" else (B.range_encode, B.range_decode)
    enc = enc_f(dev(sym), model, config)
    good = enc.n_words.clone()
    bad_streams = [0, 5, 63, 64, n_streams - 1]
    n = good.cpu().numpy().astype(np.int64)
    n[bad_streams] = [enc.stride + 1, 0x7FFFFFF0, 0xFFFFFFFF, 1 << 20, 0xFFFFFFF0]
    enc.n_words = dev(n.astype(np.uint32).view(np.int32))
    dec, status = dec_f(enc, model, n_per)
    torch.cuda.synchronize()
    st = status.cpu().numpy()
    ok = np.ones(n_streams, bool)
    ok[bad_streams] = False
    assert (st[~ok] == INVALID).all() and (st[ok] == 0).all()
    assert np.array_equal(dec.cpu().numpy()[ok], sym[ok])


@pytest.mark.parametrize("coder,config", CASES, ids=lambda c: str(c))
def test_corrupt_offsets_in_a_packed_buffer(B, coder, config):
    """packed form (what container.load hands over): offsets beyond the buffer, counts that run past its end"""
    n_streams, n_per = 300, 1000
    rng = np.random.default_rng(7)
    sym = _symbols(rng, n_streams, n_per)
    model = B.Model.quantized_gaussian(-50, 50, 3.2, 9.6, config[2])
    enc_f, dec_f = (B.ans_encode, B.ans_decode) if coder == "ans" else (B.range_encode, B.range_decode)
    enc = enc_f(dev(sym), model, config)
    packed, offsets = B.compact(enc)
    torch.cuda.synchronize()
    total = int(offsets[-1].item())
    packed = packed[:total].clone()                                   # the buffer IS its valid words: capacity = total
    off = offsets.cpu().numpy().astype(np.int64)
    n = enc.n_words.cpu().numpy().astype(np.int64)
    off[3] = total + 1
    off[64] = 1 << 40
    off[65] = -8                                                      # (2^64 - 8 as unsigned)
    n[100] = total                                                    # starts inside, ends far outside
    n[299] = n[299] + 1                                               # one word past the end of the buffer
    bad = [3, 64, 65, 100, 299]
    dec, status = dec_f((packed, dev(n.astype(np.uint32).view(np.int32))), model, n_per, offsets=dev(off), config=config)
    torch.cuda.synchronize()
    st = status.cpu().numpy()
    ok = np.ones(n_streams, bool)
    ok[bad] = False
    assert (st[~ok] == INVALID).all() and (st[ok] == 0).all()
    assert np.array_equal(dec.cpu().numpy()[ok], sym[ok])


def test_corrupt_counts_per_stream_tables(B):
    """C3's kernels (one table per stream): compact rows (P = 12, 255 symbols) and the full-row fallback (P = 16)"""
    n_streams, n_per = 512, 512
    rng = np.random.default_rng(11)
    mu = rng.uniform(-10, 10, n_streams)
    sd = np.exp(rng.uniform(np.log(0.5), np.log(16), n_streams))
    sym = np.clip(np.rint(mu[:, None] + sd[:, None] * rng.standard_normal((n_streams, n_per))), -127, 127).astype(np.int32)
    for P in (12, 16):
        model = B.Model.quantized_gaussian_per_stream(-127, 127, dev(mu), dev(sd), P)
        enc = B.ans_encode(dev(sym), model, (32, 64, P))
        n = enc.n_words.cpu().numpy().astype(np.int64)
        bad = [1, 255, 256, 511]
        n[bad] = [enc.stride + 7, 0xFFFFFFFF, 1 << 24, 0x80000000]
        enc.n_words = dev(n.astype(np.uint32).view(np.int32))
        dec, status = B.ans_decode(enc, model, n_per)
        torch.cuda.synchronize()
        st = status.cpu().numpy()
        ok = np.ones(n_streams, bool)
        ok[bad] = False
        assert (st[~ok] == INVALID).all() and (st[ok] == 0).all(), P
        assert np.array_equal(dec.cpu().numpy()[ok], sym[ok])


@pytest.mark.parametrize("coder", ["ans", "range"])
@pytest.mark.parametrize("n_streams", [3, 130])
def test_corrupt_counts_per_symbol_models(B, coder, n_streams):
    """the per-symbol Gaussian decoders: one wave per stream (few streams) and one lane per stream"""
    n_per = 200
    rng = np.random.default_rng(13)
    mu = rng.uniform(-30, 30, (n_streams, n_per))
    sd = np.exp(rng.uniform(-1, 3, (n_streams, n_per)))
    sym = np.clip(np.rint(mu + sd * rng.standard_normal((n_streams, n_per))), -100, 100).astype(np.int32)
    enc_f, dec_f = (B.ans_encode_gaussian, B.ans_decode_gaussian) if coder == "ans" else (B.range_encode_gaussian, B.range_decode_gaussian)
    enc = enc_f(dev(sym), -100, 100, dev(mu), dev(sd))
    n = enc.n_words.cpu().numpy().astype(np.int64)
    bad = [0, n_streams - 1]
    n[bad] = [0xFFFFFFFF, enc.stride + 1]
    enc.n_words = dev(n.astype(np.uint32).view(np.int32))
    dec, status = dec_f(enc, -100, 100, dev(mu), dev(sd))
    torch.cuda.synchronize()
    st = status.cpu().numpy()
    ok = np.ones(n_streams, bool)
    ok[bad] = False
    assert (st[~ok] == INVALID).all() and (st[ok] == 0).all()
    assert np.array_equal(dec.cpu().numpy()[ok], sym[ok])


def test_corrupt_checkpoints(B):
    """jump tables: a checkpoint position beyond the stream's slab (stack.rs:1117-1139: `seek` fails there)"""
    n_streams, n_per, interval = 4, 4096, 256
    rng = np.random.default_rng(17)
    sym = _symbols(rng, n_streams, n_per)
    model = B.Model.quantized_gaussian(-50, 50, 3.2, 9.6, 24)
    enc, ck = B.ans_encode_checkpointed(dev(sym), model, interval)
    pos = ck.pos.cpu().numpy().astype(np.int64)
    pos[1, 3] = 0x7FFFFFFF
    pos[3, 15] = n_streams * enc.stride + 1
    ck.pos = dev(pos.astype(np.uint32).view(np.int32))
    dec, status = B.ans_decode_checkpointed(enc, ck, model, n_per)
    torch.cuda.synchronize()
    st = status.cpu().numpy()
    want = np.zeros_like(st)
    want[1, 3] = want[3, 15] = INVALID
    assert np.array_equal(st, want)
    d = dec.cpu().numpy().reshape(n_streams, n_per // interval, interval)
    s = sym.reshape(n_streams, n_per // interval, interval)
    assert np.array_equal(d[want == 0], s[want == 0])


@pytest.mark.parametrize("dtype", ["int8", "int16", "int32"])
@pytest.mark.parametrize("P", [12, 24])
@pytest.mark.parametrize("shape", ["one wave per SIMD", "small footprint"])
def test_corrupt_metadata_narrow_and_small_footprint_decoders(B, dtype, P, shape):
    """the decoders that write int8 / int16 matrices themselves (cst_ans_n8.hip, ans_decode_b16_narrow_kernel) and the small-footprint
    forms of every type (more than 256 streams per CU): corrupt counts in slabs and corrupt offsets / counts in a packed buffer flag
    exactly their streams, everybody else decodes, nothing is written behind the matrix"""
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    n_streams, n_per = (300, 512) if shape == "one wave per SIMD" else (cus * 256 + 100, 128)
    rng = np.random.default_rng(P + n_streams)
    sym = _symbols(rng, n_streams, n_per)
    dt = {"int8": torch.int8, "int16": torch.int16, "int32": torch.int32}[dtype]
    d = dev(sym).to(dt)
    model = B.Model.quantized_gaussian(-50, 50, 3.2, 9.6, P)
    config = (32, 64, P)
    enc = B.ans_encode(d, model, config)
    n = enc.n_words.cpu().numpy().astype(np.int64)
    bad = [0, 5, 63, 64, n_streams - 1]
    n_bad = n.copy()
    n_bad[bad] = [enc.stride + 1, 0x7FFFFFF0, 0xFFFFFFFF, 1 << 20, 0xFFFFFFF0]
    ok = np.ones(n_streams, bool)
    ok[bad] = False
    guard = torch.full((n_streams * n_per + 4096,), 77, dtype=dt, device="cuda")
    out = guard[: n_streams * n_per].view(n_streams, n_per)
    good_counts = enc.n_words
    enc.n_words = dev(n_bad.astype(np.uint32).view(np.int32))
    dec, status = B.ans_decode(enc, model, n_per, out=out)
    kernel = B.last_kernel()
    st = status.cpu().numpy()
    assert (st[~ok] == INVALID).all() and (st[ok] == 0).all(), kernel
    assert torch.equal(dec[torch.from_numpy(ok).cuda()], d[torch.from_numpy(ok).cuda()]), kernel
    assert (guard[n_streams * n_per:] == 77).all(), kernel
    enc.n_words = good_counts
    packed, offsets = B.compact(enc)
    total = int(offsets[-1].item())
    packed = packed[:total].clone()
    off = offsets.cpu().numpy().astype(np.int64)
    m = n.copy()
    off[3] = total + 1
    off[64] = 1 << 40
    off[65] = -8
    m[100] = total
    m[n_streams - 1] += 1
    bad = [3, 64, 65, 100, n_streams - 1]
    ok[:] = True
    ok[bad] = False
    out.fill_(55)
    dec, status = B.ans_decode((packed, dev(m.astype(np.uint32).view(np.int32))), model, n_per, offsets=dev(off), config=config, out=out)
    kernel = B.last_kernel()
    st = status.cpu().numpy()
    assert (st[~ok] == INVALID).all() and (st[ok] == 0).all(), kernel
    assert torch.equal(dec[torch.from_numpy(ok).cuda()], d[torch.from_numpy(ok).cuda()]), kernel
    assert (guard[n_streams * n_per:] == 77).all(), kernel
