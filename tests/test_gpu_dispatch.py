"""Which kernel the dispatcher takes for every BASELINE shape at FULL size (65 536 x 4096 and the 131 072-stream C5 shard), by
name (cst_last_kernel_name): plain, int8, P = 24, packed 16-bit words, cold words, and the jump-point forms the default calls now
choose for themselves.  A change of the dispatch rules that silently moves a BASELINE shape to another kernel fails here; the
round trips are checked on the way (decoded symbols = input)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ALT = any(os.environ.get(k) for k in ("CST_NO_N8", "CST_NO_PC_ENCODER", "CST_SMALL_KERNELS", "CST_PC_COMBINED", "CST_NO_PC_WIDE", "CST_DQ_DECODER",
                                      "CST_LANE_GEO", "CST_AUTO_JUMP", "CST_PT_SUB_WAVES"))
N_STREAMS, N_PER = 65536, 4096


@pytest.fixture(scope="module")
def B():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    if ALT:
        pytest.skip("an alternate kernel path is forced: the names are those of the default dispatch")
    if torch.cuda.get_device_properties(0).multi_processor_count != 256:
        pytest.skip("the expected names are those of a 256-CU MI355X")
    from constriction_amd import batched
    return batched


@pytest.fixture(scope="module")
def bench():
    import bench
    return bench


@pytest.fixture(scope="module")
def c2(B, bench):
    out = {}
    for P in (12, 24):
        model = B.Model.quantized_gaussian(bench.LO, bench.HI, bench.MEAN, bench.STD, P)
        cdf_dev = torch.from_numpy(model.cdf().astype(np.int64)).cuda()
        out[P] = (model, bench.synth_symbols_device(bench.SEED, 0, N_STREAMS, N_PER, bench.LO, cdf_dev, P))
    return out


def _roundtrip(B, encode, decode, want_enc, want_dec, symbols):
    enc = encode()
    got_enc = B.last_kernel()
    dec, st = decode(enc)
    got_dec = B.last_kernel()
    assert (got_enc, got_dec) == (want_enc, want_dec)
    assert int(st.abs().sum()) == 0 and int(enc.status.abs().sum()) == 0 and torch.equal(dec, symbols)
    return enc


def test_c2_headline_pair_and_its_cold_form(B, c2):
    model, sym = c2[12]
    enc = _roundtrip(B, lambda: B.ans_encode(sym, model, (32, 64, 12)), lambda e: B.ans_decode(e, model, N_PER),
                     "ans_encode_pc_kernel", "ans_decode_kernel", sym)
    assert enc.jump is None                                           # the headline pair stays the plain pair
    foreign = B.EncodedBatch(enc.words, enc.n_words, enc.status, enc.config)      # words of unknown provenance: the lane-quad decoder
    dec, st = B.ans_decode(foreign, model, N_PER)
    assert B.last_kernel() == "ans_decode_dq_kernel" and torch.equal(dec, sym)


@pytest.mark.parametrize("dtype,enc_name,dec_name", [(torch.int8, "ans_encode_pc_n8_kernel<ckpt>", "ans_decode_small_n8_kernel"),
                                                      (torch.int16, "ans_encode_pc_n16_kernel<ckpt>", "ans_decode_small_n16_kernel")], ids=["int8", "int16"])
def test_c2_narrow_matrices_take_two_lanes_per_stream(B, c2, dtype, enc_name, dec_name):
    model, sym = c2[12]
    d = sym.to(dtype)
    enc = _roundtrip(B, lambda: B.ans_encode(d, model, (32, 64, 12)), lambda e: B.ans_decode(e, model, N_PER, dtype=dtype), enc_name, dec_name, d)
    assert enc.jump.pos.shape == (N_STREAMS, 2)
    _roundtrip(B, lambda: B.ans_encode(d, model, (32, 64, 12), jump_points=0), lambda e: B.ans_decode(e, model, N_PER, dtype=dtype),
               enc_name.replace("<ckpt>", ""), dec_name.replace("_small", ""), d)


def test_c2_at_24_bits(B, c2):
    model, sym = c2[24]
    enc = _roundtrip(B, lambda: B.ans_encode(sym, model, (32, 64, 24)), lambda e: B.ans_decode(e, model, N_PER),
                     "ans_encode_pc_kernel<wide, ckpt>", "ans_decode_b16_small_kernel", sym)
    assert enc.jump.pos.shape == (N_STREAMS, 2)
    _roundtrip(B, lambda: B.ans_encode(sym, model, (32, 64, 24), jump_points=0), lambda e: B.ans_decode(e, model, N_PER),
               "ans_encode_pc_kernel<wide>", "ans_decode_b16_kernel", sym)
    d = sym.to(torch.int8)
    _roundtrip(B, lambda: B.ans_encode(d, model, (32, 64, 24)), lambda e: B.ans_decode(e, model, N_PER, dtype=torch.int8),
               "ans_encode_pc_n8_kernel<wide, ckpt>", "ans_decode_b16_small_n8_kernel", d)


def test_c2_small_preset_plain_and_packed(B, c2):
    model, sym = c2[12]
    _roundtrip(B, lambda: B.ans_encode(sym, model, (16, 32, 12)), lambda e: B.ans_decode(e, model, N_PER), "ans_encode_w16_kernel", "ans_decode_w16_kernel", sym)
    enc = _roundtrip(B, lambda: B.ans_encode(sym, model, (16, 32, 12), packed16=True), lambda e: B.ans_decode(e, model, N_PER),
                     "ans_encode_w16pk_kernel", "ans_decode_w16pk_kernel", sym)
    assert enc.packed16 and enc.jump is None
    # asked for explicitly (random access): the packed encoder notes the points on its way, the chunks decode as streams of their own
    enc = _roundtrip(B, lambda: B.ans_encode(sym, model, (16, 32, 12), packed16=True, jump_points=2), lambda e: B.ans_decode(e, model, N_PER),
                     "ans_encode_w16pk_kernel<ckpt>", "ans_decode_w16pk_kernel", sym)
    assert enc.jump.pos.shape == (N_STREAMS, 2)


def test_c3_tables_per_stream(B, bench):
    mu, sigma = bench.c3_parameters(bench.SEED, 0, N_STREAMS, N_PER, torch.device("cuda"))
    model = B.Model.quantized_gaussian_per_stream(-127, 127, mu, sigma, 12)
    rows = model.cdfs_device()
    sym = bench.synth_symbols_per_stream(bench.SEED, 0, N_PER, -127, rows, 12)
    enc = _roundtrip(B, lambda: B.ans_encode(sym, model, (32, 64, 12)), lambda e: B.ans_decode(e, model, N_PER),
                     "ans_encode_pt_kernel<ckpt>", "ans_decode_pt_sub_kernel", sym)
    assert enc.jump.pos.shape == (N_STREAMS, 8)
    _roundtrip(B, lambda: B.ans_encode(sym, model, (32, 64, 12), jump_points=0), lambda e: B.ans_decode(e, model, N_PER),
               "ans_encode_pt_kernel", "ans_decode_pt_kernel", sym)
    d = sym.to(torch.int8)                                               # round 6: the int8 matrix inside the same loops
    enc = _roundtrip(B, lambda: B.ans_encode(d, model, (32, 64, 12)), lambda e: B.ans_decode(e, model, N_PER, dtype=torch.int8),
                     "ans_encode_pt_n8_kernel<ckpt>", "ans_decode_pt_sub_n8_kernel", d)
    assert enc.jump.pos.shape == (N_STREAMS, 8)


@pytest.mark.parametrize("P", [12, 24])
def test_c4_range_coder(B, c2, P):
    model, sym = c2[P]
    enc = _roundtrip(B, lambda: B.range_encode(sym, model, (32, 64, P)), lambda e: B.range_decode(e, model, N_PER),
                     "range_encode_ckpt_kernel", "range_decode_sub_kernel", sym)
    assert enc.jump.pos.shape == (N_STREAMS, 2)
    _roundtrip(B, lambda: B.range_encode(sym, model, (32, 64, P), jump_points=0), lambda e: B.range_decode(e, model, N_PER),
               "range_encode_fast_kernel", "range_decode_fast_kernel", sym)
    d = sym.to(torch.int8)                                               # round 6: int8 matrices inside the range coder's loops
    enc = _roundtrip(B, lambda: B.range_encode(d, model, (32, 64, P)), lambda e: B.range_decode(e, model, N_PER, dtype=torch.int8),
                     "range_encode_ckpt_n8_kernel", "range_decode_sub_n8_kernel", d)
    assert enc.jump.pos.shape == (N_STREAMS, 2)
    _roundtrip(B, lambda: B.range_encode(d, model, (32, 64, P), jump_points=0), lambda e: B.range_decode(e, model, N_PER, dtype=torch.int8),
               "range_encode_n8_kernel", "range_decode_n8_kernel", d)


def test_c5_shard_runs_the_small_footprint_pair(B, bench, c2):
    model = c2[12][0]
    cdf_dev = torch.from_numpy(model.cdf().astype(np.int64)).cuda()
    sym = bench.synth_symbols_device(bench.SEED, 0, 2 * N_STREAMS, N_PER, bench.LO, cdf_dev, 12)
    enc = _roundtrip(B, lambda: B.ans_encode(sym, model, (32, 64, 12)), lambda e: B.ans_decode(e, model, N_PER, cold=False),
                     "ans_encode_small_kernel", "ans_decode_small_kernel", sym)
    assert enc.jump is None


def test_f1_per_symbol_gaussians(B):
    n_streams, n_per, lo, hi = N_STREAMS, 512, -100, 100              # (a quarter of the bench's rows: the dispatch looks at the streams)
    g = torch.Generator(device="cuda").manual_seed(5)
    mu = (torch.rand((n_streams, n_per), generator=g, device="cuda", dtype=torch.float64) - 0.5) * 60
    sd = torch.exp(torch.rand((n_streams, n_per), generator=g, device="cuda", dtype=torch.float64) * 4 - 1)
    sym = torch.clamp(torch.round(mu + sd * torch.randn((n_streams, n_per), generator=g, device="cuda", dtype=torch.float64)), lo, hi).to(torch.int32)
    enc = B.ans_encode_gaussian(sym, lo, hi, mu, sd)
    assert B.last_kernel() == "ans_encode_gaussian_fused_kernel<ckpt>" and enc.jump.pos.shape == (n_streams, 2)
    dec, st = B.ans_decode_gaussian(enc, lo, hi, mu, sd)
    assert B.last_kernel() == "ans_decode_gaussian_lane_kernel<small>" and int(st.abs().sum()) == 0 and torch.equal(dec, sym)
    plain = B.ans_encode_gaussian(sym, lo, hi, mu, sd, jump_points=0)
    assert B.last_kernel() == "ans_encode_gaussian_fused_kernel" and plain.jump is None
    dec, st = B.ans_decode_gaussian(plain, lo, hi, mu, sd)
    assert B.last_kernel() == "ans_decode_gaussian_lane_kernel" and torch.equal(dec, sym)
    renc = B.range_encode_gaussian(sym, lo, hi, mu, sd)                  # round 6: the range coder's per-symbol calls take them too
    assert B.last_kernel() == "range_encode_gaussian_fused_kernel<ckpt>" and renc.jump.pos.shape == (n_streams, 2)
    dec, st = B.range_decode_gaussian(renc, lo, hi, mu, sd)
    assert B.last_kernel() == "range_decode_gaussian_lane_kernel<small>" and int(st.abs().sum()) == 0 and torch.equal(dec, sym)
