"""GPU parity tests (run on the MI355X box with `-m gpu`): the HIP path, called through the C ABI,
against the CPU oracle on the same seeded inputs -- bit-exact words, symbols, counts and status."""
import ctypes as C

import os

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


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


# ------------------------------------------------------------------ f64 special functions and models

def test_device_erf_bit_exact(B, O):
    from constriction_amd import _native as N
    rng = np.random.default_rng(1)
    x = np.concatenate([
        rng.uniform(-7, 7, 400_000), rng.normal(0, 1, 300_000), rng.uniform(-0.9, 0.9, 100_000),
        rng.uniform(0.8, 1.3, 100_000) * rng.choice([-1, 1], 100_000), 10.0 ** rng.uniform(-320, 3, 50_000),
        -(10.0 ** rng.uniform(-320, 3, 50_000)),
        np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 0.84375, 1.25, 2.857142857142857, 6.0, 1e-300, 5e-324, 2.0 ** -28]),
    ])
    x = np.concatenate([x, np.nextafter(x[-12:], np.inf), np.nextafter(x[-12:], -np.inf), -x[-12:]])
    dx = dev(x)
    lib = O.load()
    want = np.array([lib.cst_oracle_erf(float(v)) for v in x])
    for fn in ("cst_debug_erf", "cst_debug_erf_tab"):       # the branchy erf and the table-driven one of the per-symbol kernels
        dout = torch.empty_like(dx)
        N.check(getattr(N.lib(), fn)(dx.data_ptr(), dout.data_ptr(), dx.numel(), None), fn)
        torch.cuda.synchronize()
        got = dout.cpu().numpy()
        same = (got.view(np.uint64) == want.view(np.uint64)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), f"{fn}: {(~same).sum()} of {len(x)} erf values differ, first at x={x[~same][0]!r}"


def test_fast_erf_error_bound(B, O):
    """The per-symbol kernels evaluate erf the cheap way first (cst_math.hpp: erf_fast_poly, 96 polynomials of degree 7) and
    rely on |fast - exact| <= 2^-46 to decide where the exact evaluation is needed.  Measured here on 6 M arguments over every
    interval, the interval boundaries, the exact evaluation's branch boundaries and the special values: the real deviation is
    2 ulp of 1 (2.2e-16: the polynomial's 2^-52 plus the exact evaluation's own last bit)."""
    from constriction_amd import _native as N
    rng = np.random.default_rng(2)
    edges = np.concatenate([[0.0, 2.0 ** -28, 0.84375, 1.25, 2.857142857142857, 6.0, 5.999999, 27.0, 1e-300, 5e-324, 1e300],
                            np.arange(1, 97) / 16.0])
    x = np.concatenate([rng.uniform(-6.5, 6.5, 3_000_000), rng.normal(0, 1.5, 2_000_000), rng.uniform(0.8, 1.3, 500_000) * rng.choice([-1, 1], 500_000),
                        10.0 ** rng.uniform(-320, 2, 250_000), -(10.0 ** rng.uniform(-320, 2, 250_000)),
                        edges, -edges, np.nextafter(edges, np.inf), np.nextafter(edges, -np.inf), [np.inf, -np.inf]])
    dx = dev(x)
    out = torch.empty_like(dx)
    N.check(N.lib().cst_debug_erf_fast(1, dx.data_ptr(), out.data_ptr(), dx.numel(), None), "cst_debug_erf_fast")
    torch.cuda.synchronize()
    err = out.cpu().numpy()
    assert np.isfinite(err).all()
    worst = float(err.max())
    assert worst <= 2.0 ** -46, f"fast erf deviates by {worst:.3e} at x = {x[int(err.argmax())]!r}"
    assert worst < 1e-15                                     # (in practice 2.2e-16: four orders of magnitude of room)
    # a NaN argument comes back as +-1 (no NaN test in the fast path): the caller's guard sends it to the exact evaluation,
    # because free_weight * (1 +- 1) / 2 is an integer (test_quick_left_cumulatives_equal_the_exact_ones: overflowing scales)
    nan = dev(np.array([np.nan]))
    o1 = torch.empty_like(nan)
    N.check(N.lib().cst_debug_erf_fast(0, nan.data_ptr(), o1.data_ptr(), 1, None), "cst_debug_erf_fast")
    torch.cuda.synchronize()
    assert abs(abs(o1.cpu().numpy()[0]) - 1.0) < 1e-15


def test_quick_left_cumulatives_equal_the_exact_ones(B, O):
    """leaky_gaussian_left_quick (fast erf + exact fallback next to integers) against leaky_gaussian_left on the device:
    40 M random evaluations over four supports and scales from needle-thin to huge -- no difference; the fallback is taken
    about twice in a million -- and 20 000 CRAFTED cases whose free_weight * cdf lies within 1e-9 ... 1e-6 of an integer
    (found by bisection on the oracle's cdf), where the fast value alone would truncate wrongly about every other time."""
    from constriction_amd import _native as N
    lib = N.lib()
    rng = np.random.default_rng(3)
    total = fallbacks = 0
    for lo, hi, P in ((-100, 100, 24), (-127, 127, 12), (-3000, 3000, 24), (0, 1, 24), (-50, 50, 16)):
        n = 4_000_000
        mu = rng.uniform(lo - 20, hi + 20, n)
        sd = np.exp(rng.uniform(-3, 6, n))
        d_mu, d_sd = dev(mu), dev(sd)
        # (a) indices anywhere in the support -- mostly where the cdf saturates or nearly does (|arg| in [5.3, 6): the exact
        # evaluation is taken there because free_weight * cdf lies within the guard of free_weight or 0) -- and (b) indices
        # next to the mean, what a coder asks for: there the fallback must be rare
        near = np.clip(np.rint(mu + sd * rng.standard_normal(n)) - lo + rng.integers(0, 2, n), 0, hi - lo + 1)
        wide = sd >= 0.5                  # (a bin edge of a narrower model sits 5 sigma and more from its mean: the band again)
        for which, idx in (("anywhere", rng.integers(0, hi - lo + 2, n)), ("near", near)):
            counts = torch.zeros(2, dtype=torch.int64, device="cuda")
            d_idx = dev(idx.astype(np.int32))
            N.check(lib.cst_debug_gaussian_left_quick(P, lo, hi, d_idx.data_ptr(), d_mu.data_ptr(), d_sd.data_ptr(), n, counts.data_ptr(), None), "quick")
            torch.cuda.synchronize()
            c = counts.cpu().numpy()
            assert c[0] == 0, (lo, hi, P, which, int(c[0]))
        counts = torch.zeros(2, dtype=torch.int64, device="cuda")
        d_idx, d_mu, d_sd = dev(near[wide].astype(np.int32)), dev(mu[wide]), dev(sd[wide])
        N.check(lib.cst_debug_gaussian_left_quick(P, lo, hi, d_idx.data_ptr(), d_mu.data_ptr(), d_sd.data_ptr(), int(wide.sum()), counts.data_ptr(), None), "quick")
        torch.cuda.synchronize()
        c = counts.cpu().numpy()
        assert c[0] == 0
        total += int(wide.sum())
        fallbacks += int(c[1])
    assert 0 < fallbacks < total * 1e-4, (fallbacks, total)
    # crafted: mu such that free_weight * cdf(x; mu, sd) sits next to an integer k
    ol = O.load()
    lo, hi, P = -100, 100, 24
    fw = float(((1 << P) - 1) - (hi - lo))
    m = 20000
    idx = rng.integers(1, hi - lo + 1, m).astype(np.int32)
    sd = np.exp(rng.uniform(-1, 4, m))
    mus = np.empty(m)
    for j in range(m):
        x = float(lo + int(idx[j])) - 0.5
        target = (float(rng.integers(1000, (1 << P) - 1000)) + rng.choice([-1.0, 1.0]) * 10.0 ** rng.uniform(-9, -6)) / fw
        a, b = x - 12 * sd[j], x + 12 * sd[j]            # cdf(x; mu) decreases in mu
        for _ in range(70):
            mid = 0.5 * (a + b)
            if ol.cst_oracle_gaussian_cdf(x, mid, float(sd[j])) > target:
                a = mid
            else:
                b = mid
        mus[j] = a
    counts = torch.zeros(2, dtype=torch.int64, device="cuda")
    d_idx, d_mu, d_sd = dev(idx), dev(mus), dev(sd)
    N.check(lib.cst_debug_gaussian_left_quick(P, lo, hi, d_idx.data_ptr(), d_mu.data_ptr(), d_sd.data_ptr(), m, counts.data_ptr(), None), "quick")
    torch.cuda.synchronize()
    c = counts.cpu().numpy()
    assert c[0] == 0 and c[1] > m // 2, c.tolist()      # (most crafted cases must have taken the exact path)
    want = np.array([O.GaussianModel(lo, hi, float(mus[j]), float(sd[j]), P, 32).lcp(lo + int(idx[j]))[0] for j in range(0, m, 50)])
    l = torch.empty(len(want), dtype=torch.int32, device="cuda")
    p = torch.empty_like(l)
    sym, mu50, sd50 = dev((lo + idx[::50]).astype(np.int32)), dev(mus[::50]), dev(sd[::50])
    N.check(lib.cst_debug_gaussian_lcp(P, 32, lo, hi, sym.data_ptr(), mu50.data_ptr(), sd50.data_ptr(), l.data_ptr(), p.data_ptr(),
                                       len(want), None), "lcp")
    torch.cuda.synchronize()
    assert np.array_equal(l.cpu().numpy().view(np.uint32), want.astype(np.uint32))


@pytest.mark.parametrize("lo,hi,P,prob_bits", [(-100, 100, 24, 32), (-50, 50, 12, 16), (-127, 127, 12, 16), (0, 1, 1, 16),
                                               (-2000, 2000, 16, 16), (-30000, 30000, 24, 32)])
def test_device_gaussian_lcp_bit_exact(B, O, lo, hi, P, prob_bits):
    from constriction_amd import _native as N
    rng = np.random.default_rng(abs(lo) * 7 + hi + P)
    n = 200_000
    sym = rng.integers(lo - 2, hi + 3, n).astype(np.int32)
    mean = rng.uniform(lo - 20, hi + 20, n)
    std = np.exp(rng.uniform(np.log(1e-3), np.log(5.0 * (hi - lo)), n))
    std[:50] = 1e-40  # the reference's own test uses sigma = 1e-40 (quantize.rs:906-935)
    dl = torch.empty(n, dtype=torch.int32, device="cuda")
    dp = torch.empty(n, dtype=torch.int32, device="cuda")
    ds, dm, dsd = dev(sym), dev(mean), dev(std)
    N.check(N.lib().cst_debug_gaussian_lcp(P, prob_bits, lo, hi, ds.data_ptr(), dm.data_ptr(), dsd.data_ptr(),
                                           dl.data_ptr(), dp.data_ptr(), n, None), "lcp")
    torch.cuda.synchronize()
    gl, gp = dl.cpu().numpy().view(np.uint32), dp.cpu().numpy().view(np.uint32)
    lib = O.load()
    l, p = C.c_uint32(), C.c_uint32()
    bad = 0
    for i in range(n):
        rc = lib.cst_oracle_leaky_gaussian_lcp(int(sym[i]), lo, hi, P, prob_bits, float(mean[i]), float(std[i]), C.byref(l), C.byref(p))
        if rc == 1:
            ok = gl[i] == 0xFFFFFFFF and gp[i] == 0
        else:
            ok = gl[i] == l.value and gp[i] == p.value
        bad += not ok
    assert bad == 0


@pytest.mark.parametrize("lo,hi,mean,std,P", [(-50, 50, 3.2, 9.6, 12), (-50, 50, 3.2, 9.6, 24), (-100, 100, 12.6, 7.3, 24),
                                              (-127, 127, 3.2, 5.1, 24), (-127, 127, -300.6, 1e-4, 12), (0, 255, 100.0, 123.45, 16)])
def test_gaussian_model_table(B, O, lo, hi, mean, std, P):
    m = B.Model.quantized_gaussian(lo, hi, mean, std, P)
    want = O.GaussianModel(lo, hi, mean, std, P, 32).cdf_table()
    assert m.cdf().tolist() == want.tolist()
    assert (m.precision, m.min_symbol, m.n_symbols, m.n_tables) == (P, lo, hi - lo + 1, 1)


@pytest.mark.parametrize("std", [1e-40, 0.0001, 0.1, 3.5, 123.45, 1234.56])
def test_leakily_quantized_normal_invariants(B, O, std):
    """The reference's `leakily_quantized_normal` grid (src/stream/model/quantize.rs:906-935 with test_entropy_model,
    src/stream/model.rs:960-989): LeakyQuantizer<_,_,u32,24>(-127..=127) x 6 sigmas x 9 means.  For every model the
    device table must equal the oracle's bit for bit, be strictly increasing from 0 to 2^24 (every probability >= 1,
    cumulative sums to 2^24), and decoding a batch of quantiles at the left end, the last quantile and the middle of
    every bin must return that bin's symbol (quantile_function inverts left_cumulative_and_probability)."""
    lo, hi, P = -127, 127, 24
    n = hi - lo + 1
    for mean in [-300.6, -127.5, -100.2, -4.5, 0.0, 50.3, 127.5, 180.2, 2000.0]:
        m = B.Model.quantized_gaussian(lo, hi, mean, std, P)
        cdf = m.cdf().astype(np.int64)
        assert cdf.tolist() == O.GaussianModel(lo, hi, mean, std, P, 32).cdf_table().tolist()
        assert cdf[0] == 0 and cdf[-1] == 1 << P and (np.diff(cdf) >= 1).all()
        # quantile -> symbol through the decoder: a one-symbol stream whose state IS the quantile (state < 2^24 decodes
        # with q = state and no refill); three probes per bin, one stream each
        left, prob = cdf[:-1], np.diff(cdf)
        q = np.concatenate([left, left + prob - 1, left + prob // 2]).astype(np.uint32)
        words = q.reshape(-1, 1).copy()
        keep = q != 0                                   # compressed data never ends in a zero word; q = 0 <=> empty stream
        n_words = keep.astype(np.uint32)
        enc = B.EncodedBatch(dev(words.view(np.int32)), dev(n_words.view(np.int32)),
                             torch.zeros(len(q), dtype=torch.int32, device="cuda"), (32, 64, P))
        got, status = B.ans_decode(enc, m, 1)
        torch.cuda.synchronize()
        assert (status.cpu().numpy() == 0).all()
        want = np.concatenate([np.arange(n), np.arange(n), np.arange(n)]) + lo
        assert np.array_equal(got.cpu().numpy()[:, 0], want)


def test_model_errors(B):
    with pytest.raises(ValueError):
        B.Model.quantized_gaussian(-50, 50, 0.0, 0.0, 12)      # std <= 0 (reference: assert!)
    with pytest.raises(ValueError):
        B.Model.quantized_gaussian(0, 5000, 0.0, 1.0, 12)      # support larger than 2^P
    with pytest.raises(ValueError):
        B.Model.quantized_gaussian(3, 3, 0.0, 1.0, 12)         # degenerate support
    with pytest.raises(ValueError):
        B.Model.from_cdf([0, 5, 5, 4096], 0, 12)               # zero-probability symbol


# ------------------------------------------------------------------ batched ANS, shared table

CONFIGS = [(32, 64, 12), (16, 32, 12), (32, 64, 24), (16, 32, 16), (32, 64, 8)]


def make_model(B, O, P, lo=-50, hi=50, mean=3.2, std=9.6):
    gm = O.GaussianModel(lo, hi, mean, std, P, 32)
    cdf = gm.cdf_table()
    return B.Model.quantized_gaussian(lo, hi, mean, std, P), cdf


@pytest.mark.parametrize("cfg", CONFIGS, ids=lambda c: "W%dS%dP%d" % c)
@pytest.mark.parametrize("n_streams,n_per", [(1, 1), (1, 1000), (63, 37), (64, 32), (65, 128), (300, 100), (257, 4096), (1000, 65),
                                             (5, 0)])
@pytest.mark.parametrize("layout", ["stream_major", "symbol_major"])
def test_roundtrip_parity_vs_oracle(B, O, cfg, n_streams, n_per, layout):
    W, S, P = cfg
    model, cdf = make_model(B, O, P)
    sym = O.synth_symbols(0xC0FFEE, 0, n_streams, n_per, -50, cdf, P)
    want_words, want_n, want_status = O.ans_encode_batch(sym, -50, cdf, P, W, S)
    dsym = dev(sym if layout == "stream_major" else sym.T)
    enc = B.ans_encode(dsym, model, cfg, layout)
    torch.cuda.synchronize()
    words, n_words, status = enc.to_numpy()
    assert status.tolist() == want_status.tolist()
    assert n_words.tolist() == want_n.tolist()
    for s in range(n_streams):
        assert words[s, : n_words[s]].tolist() == want_words[s, : want_n[s]].tolist(), f"stream {s}"
    dec, dstatus = B.ans_decode(enc, model, n_per, layout)
    torch.cuda.synchronize()
    got = dec.cpu().numpy()
    if layout == "symbol_major":
        got = got.T
    assert (dstatus.cpu().numpy() == 0).all()
    assert np.array_equal(got, sym)
    # packed layout + offsets decodes identically
    packed, offsets = B.compact(enc)
    torch.cuda.synchronize()
    off = offsets.cpu().numpy()
    assert off[0] == 0 and np.array_equal(np.diff(off), want_n.astype(np.int64))
    pk = packed.cpu().numpy().view(np.uint32)
    for s in (0, n_streams // 2, n_streams - 1):
        assert pk[off[s]: off[s + 1]].tolist() == want_words[s, : want_n[s]].tolist()
    dec2, _ = B.ans_decode((packed, enc.n_words), model, n_per, layout, offsets=offsets, config=cfg)
    torch.cuda.synchronize()
    got2 = dec2.cpu().numpy()
    assert np.array_equal(got2.T if layout == "symbol_major" else got2, sym)


def test_unaligned_base_pointer(B, O):
    """stream-major rows that are not 16-byte aligned take the scalar tile path."""
    P = 12
    model, cdf = make_model(B, O, P)
    sym = O.synth_symbols(1, 0, 130, 64, -50, cdf, P)
    want_words, want_n, _ = O.ans_encode_batch(sym, -50, cdf, P)
    buf = torch.empty(130 * 64 + 1, dtype=torch.int32, device="cuda")
    view = buf[1:].view(130, 64)
    view.copy_(dev(sym))
    enc = B.ans_encode(view, model, (32, 64, 12))
    torch.cuda.synchronize()
    words, n_words, _ = enc.to_numpy()
    assert n_words.tolist() == want_n.tolist()
    for s in range(130):
        assert words[s, : n_words[s]].tolist() == want_words[s, : want_n[s]].tolist()
    out = torch.empty(130 * 64 + 1, dtype=torch.int32, device="cuda")
    dec, _ = B.ans_decode(enc, model, 64, out=out[1:].view(130, 64))
    torch.cuda.synchronize()
    assert np.array_equal(dec.cpu().numpy(), sym)


def test_impossible_symbol_and_capacity(B, O):
    P = 12
    model, cdf = make_model(B, O, P)
    sym = O.synth_symbols(2, 0, 200, 256, -50, cdf, P)
    sym[3, 17] = 51          # outside the support -> ImpossibleSymbol (src/lib.rs:376-385)
    sym[77, 255] = -51
    sym[199, 0] = 2 ** 31 - 1
    want_words, want_n, want_status = O.ans_encode_batch(sym, -50, cdf, P)
    enc = B.ans_encode(dev(sym), model, (32, 64, 12))
    torch.cuda.synchronize()
    words, n_words, status = enc.to_numpy()
    assert status.tolist() == want_status.tolist()
    assert [int(status[i]) for i in (3, 77, 199)] == [1, 1, 1]
    assert n_words.tolist() == want_n.tolist()
    # too small a slab -> CAPACITY for the streams that overflow, others unaffected
    stride = int(np.median(want_n))
    w2, n2, st2 = O.ans_encode_batch(sym, -50, cdf, P, stride=stride)
    enc2 = B.ans_encode(dev(sym), model, (32, 64, 12), stride=stride)
    torch.cuda.synchronize()
    words2, n_words2, status2 = enc2.to_numpy()
    assert status2.tolist() == st2.tolist() and 2 in status2.tolist() and 0 in status2.tolist()
    assert n_words2.tolist() == n2.tolist()
    for s in np.nonzero(status2 == 0)[0]:
        assert words2[s, : n_words2[s]].tolist() == w2[s, : n2[s]].tolist()


def test_capacity_overflow_in_main_loop(B, O):
    """Slabs that are 64-byte aligned but too small: the encoder's main-loop statement must drop the words past the
    capacity, report CAPACITY for exactly the streams the oracle reports it for, and leave the others bit-exact."""
    P = 12
    model, cdf = make_model(B, O, P)
    sym = O.synth_symbols(21, 0, 192, 1024, -50, cdf, P)
    sym[::2, ::3] = -50                                  # every other stream needs many more words
    _, full_n, _ = O.ans_encode_batch(sym, -50, cdf, P)
    stride = (int(full_n[1::2].max()) + 15) // 16 * 16   # multiple of 16 words: the aligned path; fits the odd streams
    w2, n2, st2 = O.ans_encode_batch(sym, -50, cdf, P, stride=stride)
    assert 2 in st2.tolist() and 0 in st2.tolist()
    enc = B.ans_encode(dev(sym), model, (32, 64, 12), stride=stride)
    torch.cuda.synchronize()
    words, n_words, status = enc.to_numpy()
    assert status.tolist() == st2.tolist() and n_words.tolist() == n2.tolist()
    for s in np.nonzero(status == 0)[0]:
        assert words[s, : n_words[s]].tolist() == w2[s, : n2[s]].tolist()


def test_decode_invalid_and_past_end(B, O):
    """Trailing zero word -> INVALID_DATA (stack.rs:299-318); decoding past the end is legal and
    deterministic (stack.rs:1062-1065)."""
    P = 12
    model, cdf = make_model(B, O, P)
    lut = O.lookup_from_cdf(cdf, P)
    rng = np.random.default_rng(5)
    n_streams, stride = 70, 12
    words = rng.integers(1, 2 ** 32, (n_streams, stride), dtype=np.uint64).astype(np.uint32)
    n_words = rng.integers(0, stride + 1, n_streams).astype(np.uint32)
    words[5, n_words[5] - 1 if n_words[5] else 0] = 0
    n_words[5] = max(n_words[5], 1)
    words[5, n_words[5] - 1] = 0
    want, want_status = O.ans_decode_batch(words, n_words, 100, -50, cdf, P, lookup=lut)
    assert want_status[5] == 3
    enc = B.EncodedBatch(dev(words.view(np.int32)), dev(n_words.view(np.int32)), torch.zeros(n_streams, dtype=torch.int32, device="cuda"), (32, 64, 12))
    got, status = B.ans_decode(enc, model, 100)
    torch.cuda.synchronize()
    assert status.cpu().numpy().tolist() == want_status.tolist()
    ok = want_status == 0
    assert np.array_equal(got.cpu().numpy()[ok], want[ok])


def test_categorical_table_model(B, O):
    """A tabulated (non-Gaussian) model through cst_model_create_table, P = 24, bucket decoder."""
    rng = np.random.default_rng(11)
    probs = rng.dirichlet(np.ones(300) * 0.3)
    cdf = O.categorical_fast_cdf(probs, 24)
    model = B.Model.from_cdf(cdf, 0, 24)
    sym = O.synth_symbols(9, 0, 100, 500, 0, cdf, 24)
    want_words, want_n, _ = O.ans_encode_batch(sym, 0, cdf, 24)
    enc = B.ans_encode(dev(sym), model, (32, 64, 24))
    torch.cuda.synchronize()
    words, n_words, status = enc.to_numpy()
    assert (status == 0).all() and n_words.tolist() == want_n.tolist()
    for s in range(100):
        assert words[s, : n_words[s]].tolist() == want_words[s, : want_n[s]].tolist()
    dec, _ = B.ans_decode(enc, model, 500)
    torch.cuda.synchronize()
    assert np.array_equal(dec.cpu().numpy(), sym)


def test_large_alphabet_lut64(B, O):
    """n_symbols > 256 at P = 12 exercises the 64-bit lookup entries."""
    gm = O.GaussianModel(-1000, 1000, 17.0, 300.0, 12, 16)
    cdf = gm.cdf_table()
    model = B.Model.quantized_gaussian(-1000, 1000, 17.0, 300.0, 12)
    assert model.cdf().tolist() == cdf.tolist()
    sym = O.synth_symbols(3, 0, 129, 300, -1000, cdf, 12)
    want_words, want_n, _ = O.ans_encode_batch(sym, -1000, cdf, 12)
    enc = B.ans_encode(dev(sym), model, (32, 64, 12))
    dec, st = B.ans_decode(enc, model, 300)
    torch.cuda.synchronize()
    words, n_words, status = enc.to_numpy()
    assert n_words.tolist() == want_n.tolist()
    for s in range(129):
        assert words[s, : n_words[s]].tolist() == want_words[s, : want_n[s]].tolist()
    assert np.array_equal(dec.cpu().numpy(), sym)


@pytest.mark.parametrize("P", [8, 9, 10, 11, 12])
@pytest.mark.parametrize("kind", ["rare", "common", "mixed"])
def test_hand_scheduled_tiles_extremes(B, O, P, kind):
    """The whole-tile encoder/decoder of the (32,64) preset (P <= 12) at the extremes of the word rate: every symbol
    the rarest one (P bits each: the window and ring schedules run at their worst case, 12 words per 32-symbol tile
    at P = 12), every symbol the most probable one (almost no words), and a mix that switches rate between streams
    and inside a stream.  Full waves take the main-loop statement, the last partial wave the per-tile path."""
    lo, hi = -20, 20
    gm = O.GaussianModel(lo, hi, 0.3, 2.5, P, 32)
    cdf = gm.cdf_table()
    model = B.Model.quantized_gaussian(lo, hi, 0.3, 2.5, P)
    probs = np.diff(cdf.astype(np.int64))
    rare, common = int(np.argmin(probs)) + lo, int(np.argmax(probs)) + lo
    n_streams, n_per = 64 * 3 + 5, 32 * 9 + 7
    rng = np.random.default_rng(P * 31 + len(kind))
    if kind == "rare":
        sym = np.full((n_streams, n_per), rare, dtype=np.int32)
    elif kind == "common":
        sym = np.full((n_streams, n_per), common, dtype=np.int32)
    else:
        sym = O.synth_symbols(77, 0, n_streams, n_per, lo, cdf, P)
        sym[::3] = rare
        sym[1::3, : n_per // 2] = common
        sym[1::3, n_per // 2:] = rare
        sym[:, rng.integers(0, n_per, 40)] = hi
    want_words, want_n, want_status = O.ans_encode_batch(sym, lo, cdf, P)
    assert (want_status == 0).all()
    enc = B.ans_encode(dev(sym), model, (32, 64, P))
    torch.cuda.synchronize()
    words, n_words, status = enc.to_numpy()
    assert (status == 0).all() and n_words.tolist() == want_n.tolist()
    for s in range(n_streams):
        assert words[s, : n_words[s]].tolist() == want_words[s, : want_n[s]].tolist(), f"stream {s}"
    dec, dstatus = B.ans_decode(enc, model, n_per)
    torch.cuda.synchronize()
    assert (dstatus.cpu().numpy() == 0).all() and np.array_equal(dec.cpu().numpy(), sym)
    # same streams from the packed buffer: every 16-byte alignment of a stream start occurs
    packed, offsets = B.compact(enc)
    dec2, _ = B.ans_decode((packed, enc.n_words), model, n_per, offsets=offsets, config=(32, 64, P))
    torch.cuda.synchronize()
    assert np.array_equal(dec2.cpu().numpy(), sym)


@pytest.mark.parametrize("seed", range(int(os.environ.get("CST_STRESS_SEEDS", "12"))))
def test_random_shapes_and_models(B, O, seed):
    """Randomised sweep over what selects a kernel variant: precision (hand-scheduled statements for P <= 12, generic
    steps above), alphabet size (encoder table / decoder lookup footprints), batch shape (full waves, partial last wave,
    fewer than two tiles, ragged tails), slab stride (aligned default vs arbitrary), packed vs slab decoding."""
    rng = np.random.default_rng(1000 + seed)
    P = int(rng.choice([8, 9, 10, 11, 12, 12, 12, 16, 24]))
    n_sym = int(rng.integers(2, min(300, (1 << P) - 1)))
    lo = int(rng.integers(-500, 500))
    probs = rng.dirichlet(np.ones(n_sym) * rng.choice([0.05, 0.3, 3.0]))
    cdf = O.categorical_fast_cdf(probs, P)
    model = B.Model.from_cdf(cdf, lo, P)
    n_streams = int(rng.choice([1, 63, 64, 65, 128, 200, 257, 1000, 2500]))
    n_per = int(rng.choice([0, 1, 31, 32, 33, 63, 64, 65, 100, 256, 500, 701]))
    sym = O.synth_symbols(seed, 0, n_streams, n_per, lo, cdf, P)
    stride = None
    if rng.random() < 0.4 and n_per > 0:
        stride = B.max_words(n_per, (32, 64, P)) + int(rng.integers(1, 7))      # not a multiple of 16 words
    want_words, want_n, want_status = O.ans_encode_batch(sym, lo, cdf, P, stride=stride) if stride else O.ans_encode_batch(sym, lo, cdf, P)
    enc = B.ans_encode(dev(sym), model, (32, 64, P), stride=stride)
    torch.cuda.synchronize()
    words, n_words, status = enc.to_numpy()
    assert status.tolist() == want_status.tolist() and n_words.tolist() == want_n.tolist()
    for s in range(n_streams):
        assert words[s, : n_words[s]].tolist() == want_words[s, : want_n[s]].tolist(), f"stream {s}"
    dec, dstatus = B.ans_decode(enc, model, n_per)
    torch.cuda.synchronize()
    assert (dstatus.cpu().numpy() == 0).all() and np.array_equal(dec.cpu().numpy(), sym)
    if n_per > 0:
        packed, offsets = B.compact(enc)
        dec2, _ = B.ans_decode((packed, enc.n_words), model, n_per, offsets=offsets, config=(32, 64, P))
        torch.cuda.synchronize()
        assert np.array_equal(dec2.cpu().numpy(), sym)


@pytest.mark.parametrize("P", [1, 2, 4, 7])
@pytest.mark.parametrize("cfg_ws", [(32, 64), (16, 32)])
def test_small_precisions(B, O, P, cfg_ws):
    """Precisions below 8 bits stay on the generic steps (the 32-bit-halves forms need P >= 8): tiny alphabets,
    including the 1-bit two-symbol model."""
    W, S = cfg_ws
    n_sym = min(1 << P, 5)
    rng = np.random.default_rng(P)
    cuts = np.sort(rng.choice(np.arange(1, 1 << P), n_sym - 1, replace=False))
    cdf = np.concatenate(([0], cuts, [1 << P])).astype(np.uint32)      # any strictly increasing table is a valid model
    model = B.Model.from_cdf(cdf, -1, P)
    sym = O.synth_symbols(4, 0, 130, 257, -1, cdf, P)
    want_words, want_n, want_status = O.ans_encode_batch(sym, -1, cdf, P, W, S)
    enc = B.ans_encode(dev(sym), model, (W, S, P))
    torch.cuda.synchronize()
    words, n_words, status = enc.to_numpy()
    assert status.tolist() == want_status.tolist() and n_words.tolist() == want_n.tolist()
    mask = (1 << W) - 1
    for s in range(130):
        assert (words[s, : n_words[s]] & mask).tolist() == want_words[s, : want_n[s]].tolist()
    dec, dstatus = B.ans_decode(enc, model, 257)
    torch.cuda.synchronize()
    assert (dstatus.cpu().numpy() == 0).all() and np.array_equal(dec.cpu().numpy(), sym)


@pytest.mark.parametrize("cfg", [(32, 64, 24), (32, 64, 16), (16, 32, 16)], ids=lambda c: "W%dS%dP%d" % c)
def test_alphabet_too_large_for_lds(B, O, cfg):
    """More than ~3800 symbols: the 16-byte encoder entries no longer fit next to the rings and tiles in LDS and stay in
    HBM / L2; the decoder uses its bucket index (P = 24) or the global lookup tables (P = 16)."""
    W, S, P = cfg
    n_sym = 6000
    rng = np.random.default_rng(P)
    cdf = O.categorical_fast_cdf(rng.dirichlet(np.ones(n_sym) * 0.2), P)
    model = B.Model.from_cdf(cdf, -3000, P)
    sym = O.synth_symbols(8, 0, 150, 200, -3000, cdf, P)
    want_words, want_n, want_status = O.ans_encode_batch(sym, -3000, cdf, P, W, S)
    enc = B.ans_encode(dev(sym), model, cfg)
    torch.cuda.synchronize()
    words, n_words, status = enc.to_numpy()
    assert status.tolist() == want_status.tolist() and n_words.tolist() == want_n.tolist()
    mask = (1 << W) - 1
    for s in range(150):
        assert (words[s, : n_words[s]] & mask).tolist() == want_words[s, : want_n[s]].tolist()
    dec, dstatus = B.ans_decode(enc, model, 200)
    torch.cuda.synchronize()
    assert (dstatus.cpu().numpy() == 0).all() and np.array_equal(dec.cpu().numpy(), sym)


def test_full_size_c2_properties(B, O):
    """BASELINE config C2 at full size (65 536 x 4096): round trip is the identity, and a sample of
    streams is bit-identical to the oracle."""
    P, n_streams, n_per = 12, 65536, 4096
    model, cdf = make_model(B, O, P)
    # symbols generated on the host in slices to bound time: 512 streams from the oracle recipe, tiled with a
    # per-block permutation so that streams differ
    base = O.synth_symbols(0xC0FFEE, 0, 512, n_per, -50, cdf, P)
    dsym = dev(base).repeat(n_streams // 512, 1)
    shift = torch.arange(n_streams, device="cuda") // 512
    idx = (torch.arange(n_per, device="cuda")[None, :] + shift[:, None]) % n_per
    dsym = torch.gather(dsym, 1, idx).contiguous()
    enc = B.ans_encode(dsym, model, (32, 64, 12))
    dec, status = B.ans_decode(enc, model, n_per)
    torch.cuda.synchronize()
    assert int(enc.status.abs().sum().item()) == 0 and int(status.abs().sum().item()) == 0
    assert torch.equal(dec, dsym)
    sample = [0, 1, 511, 512, 30000, 65535]
    host = dsym[sample].cpu().numpy()
    want_words, want_n, _ = O.ans_encode_batch(host, -50, cdf, P)
    for k, s in enumerate(sample):
        assert enc.stream(s).tolist() == want_words[k, : want_n[k]].tolist()
    avg = enc.total_words() / n_streams
    assert 650 < avg < 740  # ~692 words per stream


# ------------------------------------------------------------------ compaction (single-pass scan + gather)

@pytest.mark.parametrize("n_streams,n_per", [(0, 8), (1, 40), (255, 33), (256, 64), (257, 100), (5000, 37), (70001, 24)])
def test_compact_offsets_and_words(B, O, n_streams, n_per):
    """cst_compact_words: offsets = exclusive prefix sum of the word counts (look-back over up to 274 workgroups),
    packed = concatenation of the streams' words; asynchronous, total on device."""
    P = 12
    cdf = O.GaussianModel(-20, 20, 1.5, 4.0, P, 32).cdf_table()
    model = B.Model.from_cdf(cdf, -20, P)
    sym = O.synth_symbols(7, 0, n_streams, n_per, -20, cdf, P)
    if n_streams > 3:
        sym[3, :] = -20                              # a very cheap stream: one or two words
    enc = B.ans_encode(dev(sym), model, (32, 64, P)) if n_streams else None
    if n_streams == 0:
        from constriction_amd.batched import EncodedBatch
        enc = EncodedBatch(torch.empty((0, 16), dtype=torch.int32, device="cuda"), torch.empty(0, dtype=torch.int32, device="cuda"),
                           torch.empty(0, dtype=torch.int32, device="cuda"), (32, 64, P))
    packed, offsets = B.compact(enc)
    torch.cuda.synchronize()
    words, n_words, _ = enc.to_numpy()
    off = offsets.cpu().numpy()
    want = np.concatenate([[0], np.cumsum(n_words.astype(np.int64))])
    assert off.tolist() == want.tolist()
    pk = packed.cpu().numpy().view(np.uint32)
    flat = np.concatenate([words[s, : n_words[s]] for s in range(n_streams)] + [np.zeros(0, np.uint32)])
    assert np.array_equal(pk[: off[-1]], flat)
    if n_streams > 10:
        # a packed buffer that is too small, at the C ABI: streams that do not fit are skipped, the total still tells
        # (batched.compact turns that into a ValueError: test_compact_reports_a_packed_buffer_that_is_too_small)
        from constriction_amd import _native as N
        cap = int(off[n_streams // 2]) + 1
        small = torch.full((cap,), -1, dtype=torch.int32, device="cuda")
        off2 = torch.empty(n_streams + 1, dtype=torch.int64, device="cuda")
        scratch = torch.empty(N.lib().cst_compact_scratch_bytes(n_streams), dtype=torch.uint8, device="cuda")
        N.check(N.lib().cst_compact_words(enc.words.data_ptr(), enc.words.shape[1], enc.n_words.data_ptr(), n_streams, off2.data_ptr(),
                                          small.data_ptr(), cap, scratch.data_ptr(), None), "cst_compact_words")
        torch.cuda.synchronize()
        assert off2.cpu().numpy().tolist() == want.tolist() and int(off2[-1]) > cap
        sm = small.cpu().numpy().view(np.uint32)
        assert np.array_equal(sm[: off[n_streams // 2]], flat[: off[n_streams // 2]])


def test_bench_symbols_match_the_oracle(B, O):
    """bench.py's device-side workload generator is the SURVEY 8(d) recipe of oracle.synth_symbols, bit for bit."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", str(__import__("pathlib").Path(__file__).resolve().parent.parent / "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for P, begin, n_streams, n_per in [(12, 0, 300, 257), (12, 65536 * 3 + 5, 64, 4096), (24, 17, 100, 100)]:
        cdf = O.GaussianModel(-50, 50, 3.2, 9.6, P, 32).cdf_table()
        got = bench.synth_symbols_device(0xC0FFEE, begin, n_streams, n_per, -50, torch.from_numpy(cdf.astype(np.int64)).cuda(), P, chunk=128)
        want = O.synth_symbols(0xC0FFEE, begin, n_streams, n_per, -50, cdf, P)
        assert np.array_equal(got.cpu().numpy(), want)


# ------------------------------------------------------------------ more than one wave per SIMD: small-footprint kernels

@pytest.mark.parametrize("P,n_sym", [(12, 101), (8, 40), (12, 256)])
@pytest.mark.parametrize("n_per,stride_extra", [(96, 0), (100, 0), (64, 5)])
def test_more_than_one_wave_per_simd(B, O, P, n_sym, n_per, stride_extra):
    """Batches of more than 256 streams per CU take cst_ans_small.hip (two waves per SIMD): full waves through the
    generated main loops, the partial last wave, the ragged top symbols and unaligned slabs through the compiler-scheduled
    paths; words of every stream against the oracle, decode from slabs and from the packed layout."""
    n_streams = 65536 + 64 * 3 + 17
    lo = -n_sym // 2
    rng = np.random.default_rng(P * 1000 + n_sym + n_per)
    cdf = O.categorical_fast_cdf(rng.dirichlet(np.ones(n_sym) * 0.7), P)
    model = B.Model.from_cdf(cdf, lo, P)
    sym = O.synth_symbols(11, 0, n_streams, n_per, lo, cdf, P)
    sym[65600, 5] = lo + n_sym                                   # one impossible symbol in a full wave of the last workgroups
    sym[n_streams - 3, 0] = lo - 1                               # and one in the partial wave
    stride = B.max_words(n_per, (32, 64, P)) + stride_extra
    want_words, want_n, want_status = O.ans_encode_batch(sym, lo, cdf, P, stride=stride)
    enc = B.ans_encode(dev(sym), model, (32, 64, P), stride=stride)
    torch.cuda.synchronize()
    words, n_words, status = enc.to_numpy()
    assert status.tolist() == want_status.tolist() and status[65600] == 1 and status[n_streams - 3] == 1
    assert n_words.tolist() == want_n.tolist()
    mask = np.arange(stride, dtype=np.uint32)[None, :] < n_words[:, None]
    assert np.array_equal(np.where(mask, words, 0), np.where(mask, want_words, 0))
    dec, dstatus = B.ans_decode(enc, model, n_per)
    packed, offsets = B.compact(enc)
    dec2, _ = B.ans_decode((packed, enc.n_words), model, n_per, offsets=offsets, config=(32, 64, P))
    torch.cuda.synchronize()
    good = status == 0
    assert (dstatus.cpu().numpy()[good] == 0).all()
    assert np.array_equal(dec.cpu().numpy()[good], sym[good]) and np.array_equal(dec2.cpu().numpy()[good], sym[good])


def test_c5_shard_full_size(B, O):
    """Config C5's per-GPU shard at FULL size -- 131 072 streams x 4096 symbols (two waves per SIMD) -- encode, compaction,
    the gather of the packed words through the library's RCCL communicator (one rank: the only world size this box has),
    decode from the gathered buffer; words of sampled streams against the oracle, every decoded symbol against the input."""
    import os
    import torch.distributed as dist
    from constriction_amd import dist as D
    P, n_streams, n_per = 12, 131072, 4096
    cdf = O.GaussianModel(-50, 50, 3.2, 9.6, P, 32).cdf_table()
    model = B.Model.quantized_gaussian(-50, 50, 3.2, 9.6, P)
    base = O.synth_symbols(0xC0FFEE, 0, 512, n_per, -50, cdf, P)
    dsym = dev(base).repeat(n_streams // 512, 1)
    shift = torch.arange(n_streams, device="cuda") // 512
    idx = (torch.arange(n_per, device="cuda")[None, :] + 5 * shift[:, None]) % n_per
    dsym = torch.gather(dsym, 1, idx).contiguous()
    enc = B.ans_encode(dsym, model, (32, 64, P))
    packed, offsets = B.compact(enc)
    torch.cuda.synchronize()
    assert int(enc.status.abs().sum().item()) == 0
    total = int(offsets[-1].item())
    assert total == enc.total_words() and 680 * n_streams < total < 700 * n_streams
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("gloo", rank=0, world_size=1)
        created = True
    try:
        comm = D.RcclComm()
        all_packed, all_off = comm.gather_packed(packed, offsets, dst=0)
        torch.cuda.synchronize()
        assert torch.equal(all_off, offsets) and torch.equal(all_packed, packed[:total])
        comm.close()
    finally:
        if created:
            dist.destroy_process_group()
    dec, status = B.ans_decode((all_packed, enc.n_words), model, n_per, offsets=all_off, config=(32, 64, P))
    torch.cuda.synchronize()
    assert int(status.abs().sum().item()) == 0 and torch.equal(dec, dsym)
    sample = [0, 511, 512, 65535, 65536, 100000, n_streams - 1]
    want_words, want_n, _ = O.ans_encode_batch(dsym[sample].cpu().numpy(), -50, cdf, P)
    off = all_off.cpu().numpy()
    pk = all_packed.cpu().numpy().view(np.uint32)
    for k, s in enumerate(sample):
        assert pk[off[s]: off[s + 1]].tolist() == want_words[k, : want_n[k]].tolist()


def test_rccl_gather_through_the_c_abi_single_rank(B, O):
    """cst_rccl_* / cst_gather_sizes_rccl / cst_gather_rccl with a communicator of ONE rank (the only world size a
    single-GPU box offers): ids, sizes all-gather, the root's own copy and the offset re-basing."""
    import os
    import torch.distributed as dist
    from constriction_amd import dist as D
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("gloo", rank=0, world_size=1)
        created = True
    try:
        P = 12
        cdf = O.GaussianModel(-20, 20, 1.5, 4.0, P, 32).cdf_table()
        model = B.Model.from_cdf(cdf, -20, P)
        sym = O.synth_symbols(3, 0, 700, 50, -20, cdf, P)
        enc = B.ans_encode(dev(sym), model, (32, 64, P))
        packed, offsets = B.compact(enc)
        comm = D.RcclComm()
        all_packed, all_off = comm.gather_packed(packed, offsets, dst=0)
        torch.cuda.synchronize()
        total = int(offsets[-1])
        assert torch.equal(all_off, offsets) and torch.equal(all_packed, packed[:total])
        # the inverse (cst_scatter_rccl): the rank gets its words and offsets back and decodes them
        back_packed, back_off = comm.scatter_packed(all_packed, all_off, src=0)
        torch.cuda.synchronize()
        assert torch.equal(back_off, offsets) and torch.equal(back_packed, packed[:total])
        dec, st = B.ans_decode((back_packed, enc.n_words), model, 50, offsets=back_off, config=(32, 64, P))
        torch.cuda.synchronize()
        assert (st.cpu().numpy() == 0).all() and np.array_equal(dec.cpu().numpy(), sym)
        # sizes as a two-rank job would see them on rank 1 are rejected cleanly (bad rank), not hung
        from constriction_amd import _native as N
        assert N.lib().cst_scatter_rccl(comm._h, 1, 1, 0, None, None, comm.last_sizes.ctypes.data, None, None, None) == N.CST_ERR_INVALID_ARGUMENT
        comm.close()
    finally:
        if created:
            dist.destroy_process_group()


def test_golden_vectors_through_the_batched_abi(B, O, golden):
    """Every ANS golden vector with one i.i.d. table model per step -- in particular R3, the reference's (16,32,12)
    lookup-decoder example (src/stream/model/categorical/lookup_contiguous.rs:278-290), which the single-coder drop-in
    (fixed to the (32,64,24) preset like the reference's Python API) cannot express -- through ans_decode / ans_encode of
    the batched API, replicated over a wave and a half of identical streams."""
    from golden_util import models_for
    n_rep = 96
    done = 0
    for vec in golden["vectors"]:
        if vec["coder"] != "ans" or len(vec["steps"]) != 1:
            continue
        step = vec["steps"][0]
        if step["model"]["kind"] not in ("categorical_fast", "table") and not (step["model"]["kind"] == "gaussian" and "means" not in step["model"]):
            continue
        W, S, P = vec["W"], vec["S"], vec["P"]
        m, n = models_for(step, P, O, 32 if W == 32 else 16)
        cdf = m.cdf_table() if hasattr(m, "cdf_table") else m.cdf
        lo = m.lo
        model = B.Model.from_cdf(np.asarray(cdf, dtype=np.uint32), lo, P)
        if step["op"] == "decode":
            words = np.asarray(vec["init"]["compressed"], dtype=np.uint32)
            enc_words = dev(np.tile(words.view(np.int32), (n_rep, 1)))
            n_words = dev(np.full(n_rep, len(words), dtype=np.int32))
            expect = np.atleast_1d(np.asarray(step["expect"], dtype=np.int32))
            dec, st = B.ans_decode((enc_words, n_words), model, len(expect), config=(W, S, P))
            torch.cuda.synchronize()
            assert (st.cpu().numpy() == 0).all()
            assert (dec.cpu().numpy() == expect[None, :]).all(), vec["id"]
        else:
            sym = np.asarray(step["symbols"], dtype=np.int32)
            enc = B.ans_encode(dev(np.tile(sym, (n_rep, 1))), model, (W, S, P))
            torch.cuda.synchronize()
            for s in (0, 63, 64, n_rep - 1):
                assert enc.stream(s).tolist() == vec["expect_compressed"], vec["id"]
        done += 1
    assert done >= 3


def test_roundtrip_launcher_equals_the_wrappers(B, O):
    """batched.ans_roundtrip_launcher (what bench.py's timed loop calls: both C-ABI entry points with pre-converted
    arguments) produces what ans_encode / ans_decode produce"""
    P = 12
    cdf = O.GaussianModel(-50, 50, 3.2, 9.6, P, 32).cdf_table()
    model = B.Model.from_cdf(cdf, -50, P)
    sym = dev(O.synth_symbols(9, 0, 300, 500, -50, cdf, P))
    want = B.ans_encode(sym, model, (32, 64, P))
    enc = B.ans_encode(torch.zeros_like(sym) - 50, model, (32, 64, P))          # buffers to be overwritten
    dec = torch.empty_like(sym)
    step = B.ans_roundtrip_launcher(sym, model, enc, dec)
    step(); step()
    torch.cuda.synchronize()
    assert torch.equal(enc.n_words, want.n_words) and torch.equal(dec, sym) and int(step.decode_status.abs().sum()) == 0
    for s in (0, 64, 299):
        assert enc.stream(s).tolist() == want.stream(s).tolist()


def test_compact_reports_a_packed_buffer_that_is_too_small(B, O):
    P = 12
    cdf = O.GaussianModel(-20, 20, 1.5, 4.0, P, 32).cdf_table()
    model = B.Model.from_cdf(cdf, -20, P)
    sym = O.synth_symbols(5, 0, 300, 200, -20, cdf, P)
    enc = B.ans_encode(dev(sym), model, (32, 64, P))
    total = enc.total_words()
    packed, offsets = B.compact(enc, capacity=total)               # exactly enough
    assert int(offsets[-1]) == total
    with pytest.raises(ValueError):
        B.compact(enc, capacity=total - 1)
    small = (torch.empty(total // 2, dtype=torch.int32, device="cuda"), torch.empty(301, dtype=torch.int64, device="cuda"))
    with pytest.raises(ValueError):
        B.compact(enc, out=small)


def test_noncontiguous_alphabet(B, O):
    """NonContiguousCategoricalEncoderModel / NonContiguousLookupDecoderModel (the reference's test:
    src/stream/model/categorical/lookup_noncontiguous.rs:703-760): symbols 'a','x','c','y' with probabilities 3, 18, 1, 42 of
    2^6, then a larger alphabet of scattered symbols; words = those of the contiguous coder on the symbols' indices."""
    for P, symbols, probs in [(6, [ord(ch) for ch in "axcy"], [3, 18, 1, 42]),
                              (12, [7, -100000, 2**31 - 1, 0, -5, 12345, -(2**31), 99], [1, 1000, 7, 500, 88, 2000, 1, 499])]:
        cdf = np.concatenate([[0], np.cumsum(probs)]).astype(np.uint32)
        assert int(cdf[-1]) == 1 << P
        model = B.Model.from_cdf_noncontiguous(symbols, cdf, P)
        rng = np.random.default_rng(P)
        idx = rng.choice(len(symbols), size=(130, 77), p=np.array(probs) / float(1 << P)).astype(np.int32)
        sym = np.asarray(symbols, dtype=np.int64)[idx].astype(np.int32)
        if P == 6:
            sym[0, :9] = [ord(ch) for ch in "axcxcyaac"]                 # the reference test's message
            idx[0, :9] = ["axcy".index(ch) for ch in "axcxcyaac"]
        want_words, want_n, _ = O.ans_encode_batch(idx, 0, cdf, P)
        enc = B.ans_encode(dev(sym), model, (32, 64, P))
        torch.cuda.synchronize()
        words, n_words, status = enc.to_numpy()
        assert (status == 0).all() and n_words.tolist() == want_n.tolist()
        for s in range(len(sym)):
            assert words[s, : n_words[s]].tolist() == want_words[s, : want_n[s]].tolist()
        dec, dst = B.ans_decode(enc, model, sym.shape[1])
        torch.cuda.synchronize()
        assert (dst.cpu().numpy() == 0).all() and np.array_equal(dec.cpu().numpy(), sym)
        bad = sym.copy()
        bad[5, 3] = 424242                                               # not in the alphabet
        st = B.ans_encode(dev(bad), model, (32, 64, P)).status.cpu().numpy()
        assert st[5] == 1 and (np.delete(st, 5) == 0).all()
        # the same alphabet through the other table-model entry points: range coder, checkpointed streams
        want_rw, want_rn, _ = O.rc_encode_batch(idx, 0, cdf, P)
        renc = B.range_encode(dev(sym), model, (32, 64, P))
        rdec, rst = B.range_decode(renc, model, sym.shape[1])
        torch.cuda.synchronize()
        rwords, rn, rstatus = renc.to_numpy()
        assert (rstatus == 0).all() and rn.tolist() == want_rn.tolist()
        for s in (0, 1, 64, len(sym) - 1):
            assert rwords[s, : rn[s]].tolist() == want_rw[s, : want_rn[s]].tolist()
        assert (rst.cpu().numpy() == 0).all() and np.array_equal(rdec.cpu().numpy(), sym)
        assert B.range_encode(dev(bad), model, (32, 64, P)).status.cpu().numpy()[5] == 1
        cenc, ck = B.ans_encode_checkpointed(dev(sym[:, :70]), model, 10, (32, 64, P))
        cdec, cst = B.ans_decode_checkpointed(cenc, ck, model, 70)
        torch.cuda.synchronize()
        assert (cst.cpu().numpy() == 0).all() and np.array_equal(cdec.cpu().numpy(), sym[:, :70])
        plain = B.ans_encode(dev(sym[:, :70]), model, (32, 64, P))
        assert torch.equal(cenc.n_words, plain.n_words) and cenc.stream(3).tolist() == plain.stream(3).tolist()
    with pytest.raises(ValueError):
        B.Model.from_cdf_noncontiguous([1, 2, 1], np.array([0, 10, 20, 64], dtype=np.uint32), 6)   # duplicate symbol


@pytest.mark.parametrize("P", [13, 16, 20, 24])
def test_bucket_entry_decoder_walks_the_tails(B, O, P):
    """12 < P <= 24 (cst_ans_b16.hip): one 16-byte bucket entry resolves three symbols, the rest is a walk over the cdf
    table.  256 symbols, 100 + 155 of them with probability 1 / 2^P crowded into the first and last bucket, and data drawn
    UNIFORMLY over the alphabet so that most quantiles lie beyond their bucket's third symbol; full and partial waves,
    slabs and the packed layout, tails of every length mod 4."""
    n = 256
    probs = np.ones(n, dtype=np.int64)
    probs[100] = (1 << P) - (n - 1) - 500
    probs[37] += 300; probs[200] += 200
    cdf = np.concatenate([[0], np.cumsum(probs)]).astype(np.uint32)
    assert int(cdf[-1]) == 1 << P
    model = B.Model.from_cdf(cdf, -7, P)
    rng = np.random.default_rng(P)
    for n_streams, n_per in ((192, 640), (70, 96), (64, 64), (3, 100)):
        sym = (rng.integers(0, n, (n_streams, n_per)) - 7).astype(np.int32)
        sym[:, ::3] = 93                                       # (and the bulk symbol in between)
        want_words, want_n, _ = O.ans_encode_batch(sym, -7, cdf, P)
        enc = B.ans_encode(dev(sym), model, (32, 64, P))
        torch.cuda.synchronize()
        words, n_words, status = enc.to_numpy()
        assert (status == 0).all() and n_words.tolist() == want_n.tolist()
        for s in range(n_streams):
            assert words[s, : n_words[s]].tolist() == want_words[s, : want_n[s]].tolist()
        dec, st = B.ans_decode(enc, model, n_per)
        torch.cuda.synchronize()
        assert (st.cpu().numpy() == 0).all() and np.array_equal(dec.cpu().numpy(), sym)
        packed, offsets = B.compact(enc)
        dec2, st2 = B.ans_decode((packed, enc.n_words), model, n_per, offsets=offsets, config=(32, 64, P))
        torch.cuda.synchronize()
        assert (st2.cpu().numpy() == 0).all() and np.array_equal(dec2.cpu().numpy(), sym)
        # decoding more symbols than were encoded reads past the data exactly like the reference (stack.rs:1090-1096)
        want_more, want_st = O.ans_decode_batch(want_words, want_n, n_per + 36, -7, cdf, P)
        more, st3 = B.ans_decode(enc, model, n_per + 36)
        torch.cuda.synchronize()
        assert st3.cpu().numpy().tolist() == want_st.tolist()
        assert np.array_equal(more.cpu().numpy(), want_more)


@pytest.mark.parametrize("coder", ["ans", "range"])
@pytest.mark.parametrize("P,n", [(13, 300), (16, 700), (22, 1024), (20, 257)])
@pytest.mark.parametrize("layout", ["stream_major", "symbol_major"])
def test_large_alphabets_through_the_bucket_entries(B, O, coder, P, n, layout):
    """Alphabets of 257 ... 1024 symbols at 12 < P <= 22: the bucket entry keeps a 10-bit symbol index above a 22-bit cumulative
    and the hand-scheduled decoders take these shapes too (both coders, both layouts).  A skewed model whose tails are crowded
    with unit probabilities (the walk beyond a bucket's third symbol), data drawn over the whole alphabet; words and decoded
    symbols against the oracle."""
    rng = np.random.default_rng(P * 7 + n)
    probs = np.ones(n, dtype=np.int64)
    heavy = rng.choice(n, 12, replace=False)
    rest = (1 << P) - n
    share = rng.dirichlet(np.ones(12)) * rest
    probs[heavy] += share.astype(np.int64)
    probs[heavy[0]] += (1 << P) - int(probs.sum())
    cdf = np.concatenate([[0], np.cumsum(probs)]).astype(np.uint32)
    assert int(cdf[-1]) == 1 << P and (np.diff(cdf.astype(np.int64)) > 0).all()
    lo = -n // 2
    model = B.Model.from_cdf(cdf, lo, P)
    n_streams, n_per = 192, 200
    sym = (rng.integers(0, n, (n_streams, n_per)) + lo).astype(np.int32)
    sym[:, ::2] = np.asarray(heavy)[rng.integers(0, 12, (n_streams, (n_per + 1) // 2))] + lo
    sym[0, :4] = [lo, lo + n - 1, lo + 256, lo + 255]
    enc_f, dec_f = (B.ans_encode, B.ans_decode) if coder == "ans" else (B.range_encode, B.range_decode)
    want_words, want_n, _ = (O.ans_encode_batch if coder == "ans" else O.rc_encode_batch)(sym, lo, cdf, P)
    d_sym = dev(sym if layout == "stream_major" else sym.T)
    enc = enc_f(d_sym, model, (32, 64, P), layout)
    torch.cuda.synchronize()
    words, n_words, status = enc.to_numpy()
    assert (status == 0).all() and n_words.tolist() == want_n.tolist()
    for s in range(n_streams):
        assert words[s, : n_words[s]].tolist() == want_words[s, : want_n[s]].tolist()
    dec, st = dec_f(enc, model, n_per, layout)
    torch.cuda.synchronize()
    got = dec.cpu().numpy()
    assert (st.cpu().numpy() == 0).all() and np.array_equal(got if layout == "stream_major" else got.T, sym)


@pytest.mark.parametrize("n_per", [128, 129, 131, 4095, 4100, 4099])
@pytest.mark.parametrize("base_shift", [0, 1, 3])
@pytest.mark.parametrize("cfg", [(32, 64, 12), (32, 64, 24), (16, 32, 12)], ids=lambda c: "W%dS%dP%d" % c)
def test_rows_of_any_length_through_the_main_loop(B, O, n_per, base_shift, cfg):
    """The ANS coders' main-loop statements (P <= 12, the bucket-entry / unpacked-entry ones of P = 24, the 16-bit preset) on rows
    that do not start on cache-line boundaries: every lane codes the
    symbols in front of ITS row's next 128-byte boundary outside the loop, so that its tiles are whole cache lines (row_skew,
    cst_ans_kernels.hpp).  Row lengths with every residue that matters, symbol buffers that start 4 and 12 bytes off a
    16-byte boundary, full waves plus a partial one; words against the oracle, decoding 70 symbols past the end too."""
    n_streams = 200
    W, S, P = cfg
    cdf = O.GaussianModel(-50, 50, 3.2, 9.6, P, 32).cdf_table()
    model = B.Model.from_cdf(cdf, -50, P)
    sym = O.synth_symbols(n_per, 0, n_streams, n_per, -50, cdf, P)
    want_words, want_n, _ = O.ans_encode_batch(sym, -50, cdf, P, W, S)
    src = torch.zeros(n_streams * n_per + 8, dtype=torch.int32, device="cuda")
    src[base_shift: base_shift + sym.size] = dev(sym).reshape(-1)
    enc = B.ans_encode(src[base_shift: base_shift + sym.size].view(n_streams, n_per), model, cfg)
    torch.cuda.synchronize()
    assert enc.n_words.cpu().numpy().tolist() == want_n.tolist()
    got_words = enc.to_numpy()[0]
    for i in range(n_streams):
        assert (got_words[i, : want_n[i]] & ((1 << W) - 1)).tolist() == want_words[i, : want_n[i]].tolist(), i
    for extra in (0, 70):
        n_dec = n_per + extra
        buf = torch.full((n_streams * n_dec + 8,), -99, dtype=torch.int32, device="cuda")
        out = buf[base_shift: base_shift + n_streams * n_dec].view(n_streams, n_dec)
        dec, st = B.ans_decode(enc, model, n_dec, out=out)
        torch.cuda.synchronize()
        want, want_st = O.ans_decode_batch(want_words, want_n, n_dec, -50, cdf, P, W, S)
        assert st.cpu().numpy().tolist() == want_st.tolist()
        assert np.array_equal(out.cpu().numpy(), want)
        assert (buf[:base_shift].cpu().numpy() == -99).all() and (buf[base_shift + n_streams * n_dec:].cpu().numpy() == -99).all()


def test_packed_container_through_the_gpu(B, O, tmp_path):
    """encode -> compact -> container file -> load -> decode from the packed form; every stream's slice of the file is the
    array one reference coder would have written with `tofile` (src/pybindings/stream/stack.rs:149-166)"""
    from constriction_amd import container
    P = 24
    model, cdf = make_model(B, O, P)
    sym = O.synth_symbols(0xC0FFEE, 3, 200, 333, -50, cdf, P)
    want_words, want_n, _ = O.ans_encode_batch(sym, -50, cdf, P)
    enc = B.ans_encode(dev(sym), model, (32, 64, P))
    packed, offsets = B.compact(enc)
    torch.cuda.synchronize()
    path = tmp_path / "batch.cst"
    container.save(path, packed, offsets, (32, 64, P))
    words, off, cfg = container.load(path)
    assert cfg == (32, 64, P)
    for s in (0, 57, 199):
        assert words[off[s]: off[s + 1]].tolist() == want_words[s, : want_n[s]].tolist()
    n_words = np.diff(off.astype(np.int64)).astype(np.int32)
    dec, st = B.ans_decode((dev(words.view(np.int32)), dev(n_words)), model, 333, offsets=dev(off.astype(np.int64)), config=cfg)
    torch.cuda.synchronize()
    assert (st.cpu().numpy() == 0).all() and np.array_equal(dec.cpu().numpy(), sym)


def test_symbol_major_main_loops(B, O):
    """symbols[t][stream] through the hand-scheduled main loops (scripts/gen_{encode,decode}_loop.py, SYMBOL_MAJOR): full
    waves beside a partial one, ragged top, an impossible symbol inside a tile (that stream only), decoding past the end."""
    P, cfg = 12, (32, 64, 12)
    model, cdf = make_model(B, O, P)
    n_streams, n_per = 196, 203                      # three full waves + four streams; six tiles + eleven symbols
    sym = O.synth_symbols(4711, 0, n_streams, n_per, -50, cdf, P)
    want_words, want_n, want_st = O.ans_encode_batch(sym, -50, cdf, P)
    enc = B.ans_encode(dev(sym.T), model, cfg, "symbol_major")
    torch.cuda.synchronize()
    words, n_words, status = enc.to_numpy()
    assert status.tolist() == want_st.tolist() and n_words.tolist() == want_n.tolist()
    for s in range(n_streams):
        assert words[s, : n_words[s]].tolist() == want_words[s, : want_n[s]].tolist(), f"stream {s}"
    extra = 40
    want_dec, want_dst = O.ans_decode_batch(want_words, want_n, n_per + extra, -50, cdf, P)
    dec, dst = B.ans_decode(enc, model, n_per + extra, "symbol_major")
    torch.cuda.synchronize()
    assert dst.cpu().numpy().tolist() == want_dst.tolist()
    assert np.array_equal(dec.cpu().numpy().T, want_dec)
    bad = sym.copy()
    bad[70, 100] = 51                                # outside the support, in the middle of a tile of a full wave
    enc = B.ans_encode(dev(bad.T), model, cfg, "symbol_major")
    torch.cuda.synchronize()
    st = enc.status.cpu().numpy()
    assert st[70] == 1 and (np.delete(st, 70) == 0).all()


def _clustered_cdf(n_sym, P, cluster, period, rng):
    """runs of `cluster` symbols of probability 1 / 2^P between symbols that share the rest: every run lies inside one bucket
    of 2^(P - 11) quantiles, far more than the three symbols a bucket entry resolves"""
    probs = np.ones(n_sym, dtype=np.int64)
    big = np.arange(0, n_sym, period)
    rest = (1 << P) - n_sym
    share = rng.multinomial(rest, rng.dirichlet(np.ones(len(big)) * 2.0))
    probs[big] += share
    assert probs.sum() == 1 << P and cluster < period
    return np.concatenate([[0], np.cumsum(probs)]).astype(np.uint32)


@pytest.mark.parametrize("coder", ["ans", "range"])
@pytest.mark.parametrize("P,n_sym,period", [(24, 256, 32), (24, 250, 5), (22, 1024, 8), (16, 700, 7), (14, 200, 4), (13, 256, 16)])
def test_second_level_tables(B, O, coder, P, n_sym, period):
    """Buckets in which more than three symbols begin (DecLut::sub_bits, cst_common.hpp): runs of minimal-probability symbols
    in the MIDDLE of the distribution, a few of them (every such bucket gets a second-level table) or hundreds (the first 32
    get one, the others walk the cdf table); symbols drawn so that the runs are hit far more often than their mass says.
    Main-loop statements (200 full-wave streams of 32 k + 7 symbols) against the oracle's words and symbols."""
    rng = np.random.default_rng(P * 1000 + n_sym)
    cdf = _clustered_cdf(n_sym, P, period - 1, period, rng)
    lo = -(n_sym // 2)
    model = B.Model.from_cdf(cdf, lo, P)
    n_streams, n_per = 200, 32 * 8 + 4
    sym = O.synth_symbols(P, 0, n_streams, n_per, lo, cdf, P)
    tiny = rng.integers(0, n_sym, size=sym.shape).astype(np.int32) + lo               # uniform over symbols: mostly the tiny ones
    sym = np.where(rng.random(sym.shape) < 0.3, tiny, sym).astype(np.int32)
    if coder == "ans":
        want_words, want_n, _ = O.ans_encode_batch(sym, lo, cdf, P)
        enc = B.ans_encode(dev(sym), model, (32, 64, P))
        dec, st = B.ans_decode(enc, model, n_per)
    else:
        want_words, want_n, _ = O.rc_encode_batch(sym, lo, cdf, P)
        enc = B.range_encode(dev(sym), model, (32, 64, P))
        dec, st = B.range_decode(enc, model, n_per)
    torch.cuda.synchronize()
    words, n_words, status = enc.to_numpy()
    assert n_words.tolist() == want_n.tolist()
    for s in range(n_streams):
        assert words[s, : n_words[s]].tolist() == want_words[s, : want_n[s]].tolist(), s
    assert (st.cpu().numpy() == 0).all() and np.array_equal(dec.cpu().numpy(), sym)


@pytest.mark.parametrize("coder", ["ans", "range"])
def test_second_level_slots_shared_by_buckets(B, O, coder):
    """Three crowded buckets that map to the SAME second-level slot (buckets b, b + 32, b + 96 of 2048 at P = 24; the lowest
    owns it), one more on a slot of its own: lanes of the owner take its table, lanes of the higher buckets read an entry
    that passes the check (a lower bucket's: longer walk) and must still decode their own symbols; every cluster is hit hard."""
    P, lo = 24, -40
    shift = P - 11
    clusters = [(5, 12), (37, 9), (101, 7), (700, 11)]                  # (bucket, symbols of probability 1 at its start)
    cdf = [0]
    for b, k in clusters:
        cdf.append(b << shift)                                         # one wide symbol up to the bucket's first quantile ...
        for _ in range(k):
            cdf.append(cdf[-1] + 1)                                    # ... then k symbols of probability 1 inside the bucket
    cdf.append(1 << P)
    cdf = np.array(sorted(set(cdf)), dtype=np.uint32)
    n_sym = len(cdf) - 1
    model = B.Model.from_cdf(cdf, lo, P)
    rng = np.random.default_rng(5)
    n_streams, n_per = 128, 32 * 6 + 3
    sym = (rng.integers(0, n_sym, size=(n_streams, n_per)) + lo).astype(np.int32)      # uniform over SYMBOLS: mostly the tiny ones
    if coder == "ans":
        want_words, want_n, _ = O.ans_encode_batch(sym, lo, cdf, P)
        enc = B.ans_encode(dev(sym), model, (32, 64, P))
        dec, st = B.ans_decode(enc, model, n_per)
    else:
        want_words, want_n, _ = O.rc_encode_batch(sym, lo, cdf, P)
        enc = B.range_encode(dev(sym), model, (32, 64, P))
        dec, st = B.range_decode(enc, model, n_per)
    torch.cuda.synchronize()
    words, n_words, _ = enc.to_numpy()
    assert n_words.tolist() == want_n.tolist()
    for s in range(n_streams):
        assert words[s, : n_words[s]].tolist() == want_words[s, : want_n[s]].tolist(), s
    assert (st.cpu().numpy() == 0).all() and np.array_equal(dec.cpu().numpy(), sym)
    # ... and through the compiler-scheduled paths (a partial wave, 16-bit index images do not apply here)
    few = dev(sym[:37, :50].copy())
    if coder == "ans":
        dec2, _ = B.ans_decode(B.ans_encode(few, model, (32, 64, P)), model, 50)
    else:
        dec2, _ = B.range_decode(B.range_encode(few, model, (32, 64, P)), model, 50)
    assert torch.equal(dec2, few)


def test_tuned_stride(B, O, tmp_path, monkeypatch):
    """batched.tuned_stride: max_words for small batches without measuring; for a batch of 2^26 symbols a stride from the
    candidate list, remembered per shape; and stride="tuned" changes where the slabs lie, never what is in them."""
    P = 12
    model, cdf = make_model(B, O, P)
    small = dev(O.synth_symbols(3, 0, 256, 512, -50, cdf, P))
    assert B.tuned_stride(small, model, (32, 64, P)) == B.max_words(512, (32, 64, P))
    with pytest.raises(ValueError):
        B.ans_encode(small, model, (32, 64, P), stride="fastest")
    n_streams, n_per = 16384, 4096
    head = O.synth_symbols(9, 0, 64, n_per, -50, cdf, P)
    sym = dev(head).repeat(n_streams // 64, 1).contiguous()
    base = B.max_words(n_per, (32, 64, P))
    stride = B.tuned_stride(sym, model, (32, 64, P))
    assert base <= stride <= base + 640 and (stride == base or stride % 32 == 0)
    assert B.tuned_stride(sym, model, (32, 64, P)) == stride
    # CST_STRIDE_CACHE: a later process (here: this one with its memory wiped) takes the file's word for it
    cache = tmp_path / "strides.json"
    monkeypatch.setenv("CST_STRIDE_CACHE", str(cache))
    B._TUNED_STRIDES.clear()
    measured = B.tuned_stride(sym, model, (32, 64, P))
    import json
    (key, value), = json.loads(cache.read_text()).items()
    assert value == measured and key == "16384|4096|32|64|12|stream_major|ans|shared|101"
    cache.write_text(json.dumps({key: base + 96}))
    B._TUNED_STRIDES.clear()
    assert B.tuned_stride(sym, model, (32, 64, P)) == base + 96
    monkeypatch.delenv("CST_STRIDE_CACHE")
    B._TUNED_STRIDES.clear()
    stride = B.tuned_stride(sym, model, (32, 64, P))
    enc = B.ans_encode(sym, model, (32, 64, P), stride="tuned")
    assert enc.words.shape == (n_streams, stride)
    want_words, want_n, _ = O.ans_encode_batch(head, -50, cdf, P)
    torch.cuda.synchronize()
    n_words = enc.n_words.cpu().numpy()
    assert (enc.status.cpu().numpy() == 0).all()
    for block in (0, n_streams // 64 - 1):
        words = enc.words[block * 64: block * 64 + 64].cpu().numpy().view(np.uint32)
        for s in range(64):
            assert n_words[block * 64 + s] == want_n[s]
            assert words[s, : want_n[s]].tolist() == want_words[s, : want_n[s]].tolist()
    dec, dstatus = B.ans_decode(enc, model, n_per)
    torch.cuda.synchronize()
    assert (dstatus.cpu().numpy() == 0).all() and torch.equal(dec, sym)
