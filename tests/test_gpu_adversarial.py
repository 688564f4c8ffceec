"""Adversarial shapes that nothing had driven to an extreme before ("what has only ever run near the typical case"):

 * every checkpointing encoder at the MAXIMUM word rate with a jump point on EVERY tile (every lane emits on every step and the
   coder waves note a jump point while the storers' ring wraps) -- words, counts and the jump tables themselves against the CPU
   oracle, every chunk decoded through its jump point;
 * the decoders on words that end at the LAST BYTE of their allocation (an exact-size hipMalloc; a read behind it faults, a write
   behind the symbol matrix shows in a guard pattern) -- in a subprocess, so that a fault fails the test and not the session;
 * compaction with more than 4 GiB of packed words in front of the last streams (64-bit offsets end to end);
 * batches just above one wave of streams per SIMD (65 537 .. 65 600 streams: a partial last wave on the small-footprint kernels,
   a last workgroup of one stream) for every coder family.
The reference's own edge cases for this path: stack.rs:1456-1548 (seek), queue.rs:1333-1396, backends.rs:470-555."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = Path(__file__).resolve().parents[1]
# (runs of the suite through the alternate kernel paths do not take the kernels the tests name)
ALT = any(os.environ.get(k) for k in ("CST_NO_N8", "CST_NO_PC_ENCODER", "CST_SMALL_KERNELS", "CST_PC_COMBINED", "CST_NO_PC_WIDE"))


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


def spiky_cdf(n, P):
    """symbol 0 takes everything the n - 1 others (probability 2^-P each) leave"""
    cdf = np.zeros(n + 1, np.uint32)
    cdf[1] = (1 << P) - (n - 1)
    cdf[2:] = cdf[1] + np.arange(1, n, dtype=np.uint32)
    return cdf


def high_rate_symbols(rng, n_streams, n_per, n, frac):
    tails = rng.integers(1, n, (n_streams, n_per), dtype=np.int32)
    return np.where(rng.random((n_streams, n_per)) < frac, 0, tails).astype(np.int32)


# ---- 1. jump points on every tile at the maximum rate ----

@pytest.mark.parametrize("frac", [0.0, 0.03, 0.2])
@pytest.mark.parametrize("interval", [32, 64, 128])
@pytest.mark.parametrize("dtype,P", [("int32", 12), ("int32", 8), ("int32", 24), ("int32", 16), ("int8", 12), ("int8", 24), ("int16", 12), ("int16", 24)])
def test_ans_jump_points_on_every_tile_at_the_maximum_rate(B, O, dtype, P, interval, frac):
    n, n_streams, n_per = 101, 512 + 37, 1024                  # (a partial last workgroup too)
    if dtype != "int32" and interval % 128 != 0:
        pytest.skip("narrow matrices: chunks of whole 128-symbol lines (other intervals convert: tests/test_gpu_checkpoints.py)")
    cdf = spiky_cdf(n, P)
    model = B.Model.from_cdf(cdf, 0, P)
    rng = np.random.default_rng(1000 * P + interval + int(100 * frac))
    sym = high_rate_symbols(rng, n_streams, n_per, n, frac)
    dt = {"int32": torch.int32, "int8": torch.int8, "int16": torch.int16}[dtype]
    d = dev(sym).to(dt)
    enc = B.ans_encode(d, model, (32, 64, P), jump_points=n_per // interval)
    assert ALT or "ckpt" in B.last_kernel(), B.last_kernel()    # noted on the way by the producer / consumer coder waves
    want_words, want_n, _ = O.ans_encode_batch(sym, 0, cdf, P)
    words, n_words, status = enc.to_numpy()
    assert (status == 0).all() and n_words.tolist() == want_n.tolist()
    for s in range(n_streams):
        assert np.array_equal(words[s, : n_words[s]], want_words[s, : want_n[s]]), f"stream {s}"
    pos, state = O.ans_jump_table(sym, 0, cdf, P, interval)
    assert np.array_equal(enc.jump.pos.cpu().numpy().view(np.uint32), pos), "jump positions"
    assert np.array_equal(enc.jump.state.cpu().numpy().view(np.uint64), state), "jump states"
    dec, st = B.ans_decode_checkpointed(enc, enc.jump, model, n_per, dtype=dt)
    assert int(st.abs().sum()) == 0 and torch.equal(dec, d)


@pytest.mark.parametrize("frac", [0.0, 0.03, 0.2])
@pytest.mark.parametrize("interval", [32, 64, 256])
@pytest.mark.parametrize("P", [12, 24, 16])
def test_range_jump_points_on_every_tile_at_the_maximum_rate(B, O, P, interval, frac):
    n, n_streams, n_per = 101, 512 + 37, 1024
    cdf = spiky_cdf(n, P)
    model = B.Model.from_cdf(cdf, 0, P)
    rng = np.random.default_rng(77 * P + interval + int(100 * frac))
    sym = high_rate_symbols(rng, n_streams, n_per, n, frac)
    d = dev(sym)
    enc = B.range_encode(d, model, (32, 64, P), jump_points=n_per // interval)
    assert B.last_kernel() == "range_encode_ckpt_kernel"
    want_words, want_n, _ = O.rc_encode_batch(sym, 0, cdf, P)
    words, n_words, status = enc.to_numpy()
    assert (status == 0).all() and n_words.tolist() == want_n.tolist()
    for s in range(n_streams):
        assert np.array_equal(words[s, : n_words[s]], want_words[s, : want_n[s]]), f"stream {s}"
    pos, lower, rng_ = O.range_jump_table(sym, 0, cdf, P, interval)
    assert np.array_equal(enc.jump.pos.cpu().numpy().view(np.uint32), pos)
    assert np.array_equal(enc.jump.lower.cpu().numpy().view(np.uint64), lower) and np.array_equal(enc.jump.range.cpu().numpy().view(np.uint64), rng_)
    dec, st = B.range_decode_checkpointed(enc, enc.jump, model, n_per)
    assert int(st.abs().sum()) == 0 and torch.equal(dec, d)


@pytest.mark.parametrize("frac", [0.0, 0.05])
@pytest.mark.parametrize("interval", [32, 64])
def test_per_stream_table_jump_points_on_every_tile_at_the_maximum_rate(B, O, interval, frac):
    """needle-thin models far from their symbols: one table per stream, ~12 bits per symbol"""
    P, n_streams, n_per = 12, 320, 1024
    rng = np.random.default_rng(interval + int(100 * frac))
    mu, sd = rng.uniform(-5, 5, n_streams), rng.uniform(0.4, 0.8, n_streams)
    model = B.Model.quantized_gaussian_per_stream(-127, 127, dev(mu), dev(sd), P)
    tails = rng.choice(np.concatenate([np.arange(-127, -40), np.arange(40, 128)]), (n_streams, n_per)).astype(np.int32)
    sym = np.where(rng.random((n_streams, n_per)) < frac, np.rint(mu)[:, None].astype(np.int32), tails).astype(np.int32)
    cdfs = np.stack([O.GaussianModel(-127, 127, a, b, P, 32).cdf_table() for a, b in zip(mu, sd)])
    enc = B.ans_encode(dev(sym), model, (32, 64, P), jump_points=n_per // interval)
    assert B.last_kernel() == "ans_encode_pt_kernel<ckpt>"
    want_words, want_n, _ = O.ans_encode_batch(sym, -127, cdfs, P)
    words, n_words, status = enc.to_numpy()
    assert (status == 0).all() and n_words.tolist() == want_n.tolist()
    for s in range(n_streams):
        assert np.array_equal(words[s, : n_words[s]], want_words[s, : want_n[s]]), f"stream {s}"
    pos, state = O.ans_jump_table(sym, -127, cdfs, P, interval)
    assert np.array_equal(enc.jump.pos.cpu().numpy().view(np.uint32), pos) and np.array_equal(enc.jump.state.cpu().numpy().view(np.uint64), state)
    dec, st = B.ans_decode_checkpointed(enc, enc.jump, model, n_per)
    assert int(st.abs().sum()) == 0 and np.array_equal(dec.cpu().numpy(), sym)


# ---- 2. words that end at the last byte of their allocation ----

_EXACT = r'''
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, %(root)r)
from constriction_amd import batched as B, _native as N
from oracle import oracle as O
hip = ctypes.CDLL("libamdhip64.so")
P, lo, n_streams, n_per = 12, -50, %(n_streams)d, %(n_per)d
cdf = O.GaussianModel(lo, 50, 3.2, 9.6, P, 32).cdf_table()
model = B.Model.from_cdf(cdf, lo, P)
sym = O.synth_symbols(%(seed)d, 0, n_streams, n_per, lo, cdf, P)
d = torch.from_numpy(sym).cuda()
enc = B.ans_encode(d, model, (32, 64, P), jump_points=0)
packed, offsets = B.compact(enc)
total = int(offsets[-1])
torch.cuda.synchronize()
# the packed words in an allocation of EXACTLY their size, rounded up to the allocator's 2-MiB granule at the FRONT: the last
# word is the last word of the mapping
granule = 2 << 20
nbytes = (4 * total + granule - 1) // granule * granule
ptr = ctypes.c_void_p()
assert hip.hipMalloc(ctypes.byref(ptr), ctypes.c_size_t(nbytes)) == 0
base = ptr.value + nbytes - 4 * total
assert hip.hipMemcpy(ctypes.c_void_p(base), ctypes.c_void_p(packed.data_ptr()), ctypes.c_size_t(4 * total), 3) == 0
off = offsets.clone()
n_words = enc.n_words
lib = N.lib()
cfg = N.CoderConfig(32, 64, P)
for flags, name in ((N.FLAG_COLD_WORDS, "ans_decode_dq_kernel"), (N.FLAG_NONE, None)):
    guard = torch.full((n_streams * n_per + 4096,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
    status = torch.full((n_streams,), -1, dtype=torch.int32, device="cuda")
    rc = lib.cst_ans_decode_batch(model._h, cfg, ctypes.c_void_p(base), ctypes.c_void_p(off.data_ptr()), 0, total, ctypes.c_void_p(n_words.data_ptr()),
                                  ctypes.c_void_p(guard.data_ptr()), n_streams, n_per, 0, None, None, ctypes.c_void_p(status.data_ptr()), flags, None)
    assert rc == 0, rc
    torch.cuda.synchronize()
    if name and %(expect_dq)d:
        assert B.last_kernel() == name, B.last_kernel()
    assert int(status.abs().sum()) == 0
    assert torch.equal(guard[: n_streams * n_per].view(n_streams, n_per), d)
    assert bool((guard[n_streams * n_per:] == 0x5A5A5A5A).all()), "symbols were written behind the matrix"
# the range decoder on its own words, placed the same way
renc = B.range_encode(d, model, (32, 64, P), jump_points=0)
rp, ro = B.compact(renc)
rt = int(ro[-1]); torch.cuda.synchronize()
rbytes = (4 * rt + granule - 1) // granule * granule
rptr = ctypes.c_void_p()
assert hip.hipMalloc(ctypes.byref(rptr), ctypes.c_size_t(rbytes)) == 0
rbase = rptr.value + rbytes - 4 * rt
assert hip.hipMemcpy(ctypes.c_void_p(rbase), ctypes.c_void_p(rp.data_ptr()), ctypes.c_size_t(4 * rt), 3) == 0
out = torch.empty((n_streams, n_per), dtype=torch.int32, device="cuda")
status = torch.full((n_streams,), -1, dtype=torch.int32, device="cuda")
rc = lib.cst_range_decode_batch(model._h, cfg, ctypes.c_void_p(rbase), ctypes.c_void_p(ro.data_ptr()), 0, rt, ctypes.c_void_p(renc.n_words.data_ptr()),
                                ctypes.c_void_p(out.data_ptr()), n_streams, n_per, 0, None, ctypes.c_void_p(status.data_ptr()), 0, None)
assert rc == 0
torch.cuda.synchronize()
assert int(status.abs().sum()) == 0 and torch.equal(out, d)
print("EXACT_OK")
'''


@pytest.mark.parametrize("n_streams,n_per,expect_dq", [(1024, 2048, 1), (320, 96, 1), (65, 33, 0)])
def test_decoders_on_words_that_end_with_their_allocation(n_streams, n_per, expect_dq):
    """packed words whose last word is the last word of a hipMalloc'ed mapping: the lane-quad decoder (whole 64-byte segments), the
    chunk-load decoder (16-byte chunks) and the range decoder must not read behind it (a fault ends the subprocess)"""
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    code = _EXACT % {"root": str(ROOT), "n_streams": n_streams, "n_per": n_per, "seed": n_streams + n_per, "expect_dq": expect_dq}
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=str(ROOT),
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert res.returncode == 0 and "EXACT_OK" in res.stdout, (res.stdout[-1500:] + res.stderr[-3000:])


# ---- 3. more than 4 GiB of packed words ----

def test_compaction_beyond_4_gib(B):
    """64-bit offsets end to end: 1 179 648 slabs of 1024 words, every one full (4.5 GiB packed), a few of them short"""
    free, _ = torch.cuda.mem_get_info()
    if free < 12 << 30:
        pytest.skip("needs 10 GiB of device memory")
    n_streams, stride = 1_179_648, 1024
    words = torch.empty((n_streams, stride), dtype=torch.int32, device="cuda")
    for a in range(0, n_streams, 65536):                       # word w of stream s = s * 1024 + w (mod 2^32): position-coded
        b = min(a + 65536, n_streams)
        words[a:b] = (torch.arange(a, b, device="cuda", dtype=torch.int64)[:, None] * stride +
                      torch.arange(stride, device="cuda", dtype=torch.int64)[None, :]).to(torch.int32)
    n_words = torch.full((n_streams,), stride, dtype=torch.int32, device="cuda")
    short = {5: 0, 77_777: 1, 1_048_576: 3, n_streams - 2: 1000, n_streams - 1: 17}
    for s, k in short.items():
        n_words[s] = k
    enc = B.EncodedBatch(words, n_words, torch.zeros(n_streams, dtype=torch.int32, device="cuda"), (32, 64, 12))
    packed, offsets = B.compact(enc)
    torch.cuda.synchronize()
    want = torch.zeros(n_streams + 1, dtype=torch.int64)
    want[1:] = torch.cumsum(n_words.cpu().to(torch.int64), 0)
    beyond = int((want >= (1 << 30)).nonzero()[0])             # the first stream that starts behind 4 GiB of packed words
    assert int(want[-1]) > (1 << 30) + (1 << 26) and beyond < n_streams - 100_000, "the prefix must pass 4 GiB (2^30 words)"
    assert torch.equal(offsets.cpu(), want)
    for s in (0, 4, 5, 6, 77_777, 77_778, 1_048_575, 1_048_576, 1_048_577, beyond - 1, beyond, beyond + 1, 1_100_000, n_streams - 3, n_streams - 2,
              n_streams - 1):
        k, o = int(n_words[s]), int(want[s])
        got = packed[o: o + k].cpu().numpy().view(np.uint32)
        exp = ((s * stride + np.arange(k, dtype=np.int64)) & 0xFFFFFFFF).astype(np.uint32)
        assert np.array_equal(got, exp), f"stream {s} at offset {o}"
    # a decoder reads its stream from beyond the 4-GiB mark through the same offsets (an empty model-free check: the word slice)
    del packed, words


# ---- 4. just above one wave of streams per SIMD ----

def _slots():
    return torch.cuda.get_device_properties(0).multi_processor_count * 256


@pytest.mark.parametrize("extra", [1, 63, 64])
@pytest.mark.parametrize("family", ["c2_int32", "c2_int8", "c2_int16", "p24_int32", "p24_int8", "w16", "w16_packed", "range12", "range24", "symbol_major"])
def test_table_coders_just_above_one_wave_per_simd(B, O, family, extra):
    n_streams, n_per, lo = _slots() + extra, 512, -50
    P = 24 if family.startswith("p24") or family == "range24" else 12
    cfg = (16, 32, 12) if family.startswith("w16") else (32, 64, P)
    cdf = O.GaussianModel(lo, 50, 3.2, 9.6, P, cfg[0]).cdf_table()
    model = B.Model.from_cdf(cdf, lo, P)
    import bench
    sym = bench.synth_symbols_device(0xADD + extra, 0, n_streams, n_per, lo, torch.from_numpy(cdf.astype(np.int64)).cuda(), P)
    dt = torch.int8 if family.endswith("int8") else torch.int16 if family.endswith("int16") else torch.int32
    d = sym.to(dt)
    rows = [0, 1, 255, 256, _slots() - 1, _slots(), n_streams - 1]
    host = sym[rows].cpu().numpy()
    if family.startswith("range"):
        enc = B.range_encode(d, model, cfg)
        dec, st = B.range_decode(enc, model, n_per)
        want_words, want_n, _ = O.rc_encode_batch(host, lo, cdf, P)
    elif family == "symbol_major":
        enc = B.ans_encode(d.t().contiguous(), model, cfg, layout="symbol_major")
        dec, st = B.ans_decode(enc, model, n_per, layout="symbol_major")
        dec = dec.t().contiguous()
        want_words, want_n, _ = O.ans_encode_batch(host, lo, cdf, P)
    else:
        enc = B.ans_encode(d, model, cfg, packed16=(family == "w16_packed"))
        dec, st = B.ans_decode(enc, model, n_per, dtype=dt)
        want_words, want_n, _ = O.ans_encode_batch(host, lo, cdf, P, cfg[0], cfg[1])
    assert int(enc.status.abs().sum()) == 0 and int(st.abs().sum()) == 0
    wrong = (dec != d).any(dim=1).nonzero().flatten()
    assert wrong.numel() == 0, f"{wrong.numel()} streams decode wrongly, first {wrong[:4].tolist()} [{B.last_kernel()}]"
    for i, s in enumerate(rows):
        assert enc.stream(s).tolist() == want_words[i, : want_n[i]].tolist(), f"stream {s}"


@pytest.mark.parametrize("extra", [1, 64])
def test_per_stream_tables_just_above_one_wave_per_simd(B, O, extra):
    n_streams, n_per, P = _slots() + extra, 512, 12
    import bench
    mu, sigma = bench.c3_parameters(3, 0, n_streams, n_per, torch.device("cuda"))
    model = B.Model.quantized_gaussian_per_stream(-127, 127, mu, sigma, P)
    sym = bench.synth_symbols_per_stream(3, 0, n_per, -127, model.cdfs_device(), P)
    rows = [0, 255, 256, _slots() - 1, _slots(), n_streams - 1]
    cdfs = np.stack([O.GaussianModel(-127, 127, float(mu[s]), float(sigma[s]), P, 32).cdf_table() for s in rows])
    want_words, want_n, _ = O.ans_encode_batch(sym[rows].cpu().numpy(), -127, cdfs, P)
    for jp in (0, "auto"):
        enc = B.ans_encode(sym, model, (32, 64, P), jump_points=jp)
        dec, st = B.ans_decode(enc, model, n_per)
        assert int(enc.status.abs().sum()) == 0 and int(st.abs().sum()) == 0 and torch.equal(dec, sym), jp
        for i, s in enumerate(rows):
            assert enc.stream(s).tolist() == want_words[i, : want_n[i]].tolist(), f"stream {s} ({jp})"


@pytest.mark.parametrize("extra", [1, 64])
@pytest.mark.parametrize("coder", ["ans", "range"])
def test_per_symbol_gaussians_just_above_one_wave_per_simd(B, O, coder, extra):
    n_streams, n_per, lo, hi = _slots() + extra, 64, -100, 100
    g = torch.Generator(device="cuda").manual_seed(extra)
    mu = (torch.rand((n_streams, n_per), generator=g, device="cuda", dtype=torch.float64) - 0.5) * 60
    sd = torch.exp(torch.rand((n_streams, n_per), generator=g, device="cuda", dtype=torch.float64) * 4 - 1)
    sym = torch.clamp(torch.round(mu + sd * torch.randn((n_streams, n_per), generator=g, device="cuda", dtype=torch.float64)), lo, hi).to(torch.int32)
    enc_f, dec_f = (B.ans_encode_gaussian, B.ans_decode_gaussian) if coder == "ans" else (B.range_encode_gaussian, B.range_decode_gaussian)
    enc = enc_f(sym, lo, hi, mu, sd)
    dec, st = dec_f(enc, lo, hi, mu, sd)
    assert int(enc.status.abs().sum()) == 0 and int(st.abs().sum()) == 0 and torch.equal(dec, sym)
    h_sym, h_mu, h_sd = sym.cpu().numpy(), mu.cpu().numpy(), sd.cpu().numpy()
    for s in (0, _slots() - 1, _slots(), n_streams - 1):
        c = O.AnsCoder() if coder == "ans" else O.RangeEncoder()
        if coder == "ans":
            c.encode_gaussian_reverse(h_sym[s], lo, hi, h_mu[s], h_sd[s], 24, 32)
        else:
            c.encode(h_sym[s], [O.GaussianModel(lo, hi, float(m), float(v), 24, 32) for m, v in zip(h_mu[s], h_sd[s])], 24)
        assert enc.stream(s).tolist() == c.get_compressed().tolist(), f"stream {s}"
