#!/usr/bin/env python3
"""Randomised parity stress of the DEFAULT calls of round 6 (jump_points="auto"): every family that now notes jump points on its own
-- ANS with int32 / int8 / int16 matrices at 8 <= P <= 24, the range coder with int32 / int8 matrices, one table per stream with
int32 / int8 matrices, per-symbol Gaussians on both coders, ragged batches -- on random tables, shapes (partial waves, fewer and more
streams than the chip has lanes), data (model-distributed, uniform, rarest symbols only) and impossible symbols.  Checked per case:
the default call's words / counts / status = the jump_points=0 call's = the CPU oracle's (a sample of streams), its jump table = the
oracle's Pos at those symbols, decode through the table = decode without = the input.  Not part of the suite: minutes of GPU time.
usage: python tests/stress/stress_auto.py [seconds] [seed] [family: ans | range | per_stream | gaussian | ragged]"""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from constriction_amd import batched as B
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
cus = torch.cuda.get_device_properties(0).multi_processor_count
count = {"ans": 0, "range": 0, "per_stream": 0, "gaussian": 0, "ragged": 0}
with_jump = dict.fromkeys(count, 0)
n_streams_total = 0


def random_table(n, P):
    w = rng.gamma(0.3, 1.0, n) + 1e-9
    p = np.maximum(1, np.floor(w / w.sum() * ((1 << P) - n)).astype(np.int64))
    p[int(np.argmax(p))] += (1 << P) - int(p.sum())
    return p, np.concatenate([[0], np.cumsum(p)]).astype(np.uint32)


def same_batch(a, b, tag):
    assert torch.equal(a.n_words, b.n_words) and torch.equal(a.status, b.status), tag
    used = torch.arange(a.words.shape[1], device="cuda")[None, :] < a.n_words[:, None]
    assert bool(((a.words == b.words) | ~used).all()), tag


def shape():
    kind = rng.random()
    if kind < 0.1:                                       # more streams than lanes of one wave per SIMD
        n_streams = cus * 256 + int(rng.integers(1, 3000))
        n_per = 512 * int(rng.choice([1, 2]))
    elif kind < 0.4:
        n_streams = int(rng.integers(1, 3000))
        n_per = 256 * int(rng.integers(2, 17))
    elif kind < 0.7:
        n_streams = 256 * int(rng.integers(1, 65))
        n_per = 512 * int(rng.integers(1, 9))
    else:                                                # a full chip of streams, short rows
        n_streams = int(rng.choice([cus * 128, cus * 256, cus * 256 - 3]))
        n_per = 512 * int(rng.choice([1, 2]))
    return n_streams, n_per


def table_case(coder):
    P = int(rng.integers(8, 13)) if rng.random() < 0.5 else int(rng.integers(13, 25))
    nb = int(rng.choice([4, 4, 1, 2])) if coder == "ans" else int(rng.choice([4, 1]))
    n = int(rng.choice([2, 3, 17, 101, 128, 255, 256]))
    n = min(n, (1 << P) // 2)
    lo = int(rng.integers(-128, 128 - n + 1))
    p, cdf = random_table(n, P)
    model = B.Model.from_cdf(cdf, lo, P)
    n_streams, n_per = shape()
    kind = rng.random()
    if kind < 0.5:
        idx = rng.choice(n, size=(n_streams, n_per), p=p / float(1 << P))
    elif kind < 0.8:
        idx = rng.integers(0, n, (n_streams, n_per))
    else:
        idx = rng.choice(np.flatnonzero(p == p.min()), size=(n_streams, n_per))
    sym = (idx + lo).astype(np.int32)
    t_t = {4: torch.int32, 2: torch.int16, 1: torch.int8}[nb]
    bad = []
    if rng.random() < 0.25:
        cand = [v for v in (lo - 1, lo + n) if -128 <= v <= 127]
        for _ in range(int(rng.integers(1, 4))):
            if cand:
                r = int(rng.integers(n_streams)); sym[r, rng.integers(n_per)] = int(rng.choice(cand)); bad.append(r)
    d = dev(sym).to(t_t)
    tag = f"{coder} bytes={nb} P={P} n={n} lo={lo} streams={n_streams} n_per={n_per} kind={kind:.2f} bad={len(bad)}"
    enc_fn, dec_fn = (B.ans_encode, B.ans_decode) if coder == "ans" else (B.range_encode, B.range_decode)
    plain = enc_fn(d, model, (32, 64, P), jump_points=0)
    auto = enc_fn(d, model, (32, 64, P))
    assert plain.jump is None, tag
    same_batch(plain, auto, tag)
    check = np.unique(np.concatenate([rng.integers(0, n_streams, 48), [0, n_streams - 1], np.array(bad, dtype=np.int64)]))
    want_words, want_n, want_st = (O.ans_encode_batch if coder == "ans" else O.rc_encode_batch)(sym[check], lo, cdf, P)
    words, n_words, status = auto.to_numpy()
    assert status[check].tolist() == want_st.tolist(), tag
    for i, s in enumerate(check):
        if want_st[i] == 0:
            assert n_words[s] == want_n[i] and np.array_equal(words[s, : n_words[s]], want_words[i, : want_n[i]]), (tag, int(s))
    good = status == 0
    if auto.jump is not None:
        iv = auto.jump.interval
        assert iv >= 256 and n_per % iv == 0 and auto.jump.pos.shape == (n_streams, n_per // iv), tag
        ok_rows = check[want_st == 0][:16]
        if coder == "ans":
            wp, ws = O.ans_jump_table(sym[ok_rows], lo, cdf, P, iv)
            assert np.array_equal(auto.jump.pos.cpu().numpy().view(np.uint32)[ok_rows], wp), tag
            assert np.array_equal(auto.jump.state.cpu().numpy().view(np.uint64)[ok_rows], ws), tag
        else:
            wp, wl, wr = O.range_jump_table(sym[ok_rows], lo, cdf, P, iv)
            assert np.array_equal(auto.jump.pos.cpu().numpy().view(np.uint32)[ok_rows], wp), tag
            assert np.array_equal(auto.jump.lower.cpu().numpy().view(np.uint64)[ok_rows], wl), tag
            assert np.array_equal(auto.jump.range.cpu().numpy().view(np.uint64)[ok_rows], wr), tag
    if good.all():
        dec_a, st_a = dec_fn(auto, model, n_per, dtype=t_t)
        dec_p, st_p = dec_fn(plain, model, n_per, dtype=t_t)
        assert int(st_a.abs().sum()) == 0 and int(st_p.abs().sum()) == 0 and torch.equal(dec_a, d) and torch.equal(dec_p, d), tag
    return n_streams, auto.jump is not None


def per_stream_case():
    P, lo, hi = 12, -127, 127
    n_streams, n_per = shape()
    n_streams = min(n_streams, 20000)
    wide = rng.random() < 0.2                              # tables too big for a workgroup's LDS: the generic kernels, no jump points
    mu = rng.uniform(-40, 40, n_streams) if wide else rng.uniform(-10, 10, n_streams)
    sd = np.exp(rng.uniform(np.log(0.3), np.log(40.0 if wide else 16.0), n_streams))
    model = B.Model.quantized_gaussian_per_stream(lo, hi, dev(mu), dev(sd), P)
    rows = model.cdfs_device().cpu().numpy().astype(np.int64)
    u = rng.integers(0, 1 << P, (n_streams, n_per))
    if rng.random() < 0.3:
        u = rng.integers(0, 2, (n_streams, n_per)) * ((1 << P) - 1)      # the two tails: the rarest symbols
    sym = (torch.searchsorted(torch.from_numpy(rows).cuda(), torch.from_numpy(u).cuda(), right=True) - 1 + lo).cpu().numpy()
    sym = sym.astype(np.int32)
    nb = int(rng.choice([4, 1]))
    t_t = torch.int32 if nb == 4 else torch.int8
    d = dev(sym).to(t_t)
    tag = f"per_stream bytes={nb} streams={n_streams} n_per={n_per}"
    plain = B.ans_encode(d, model, (32, 64, P), jump_points=0)
    auto = B.ans_encode(d, model, (32, 64, P))
    same_batch(plain, auto, tag)
    check = np.unique(np.concatenate([rng.integers(0, n_streams, 32), [0, n_streams - 1]]))
    words, n_words, status = auto.to_numpy()
    assert (status == 0).all(), tag
    for s in check:
        c = O.AnsCoder()
        c.encode_iid_table_reverse(sym[s], rows[s].astype(np.uint32), lo, P)
        assert words[s, : n_words[s]].tolist() == c.get_compressed().tolist(), (tag, int(s))
    if auto.jump is not None:
        iv = auto.jump.interval
        for s in check[:8]:
            wp, ws = O.ans_jump_table(sym[s: s + 1], lo, rows[s].astype(np.uint32), P, iv)
            assert np.array_equal(auto.jump.pos[s].cpu().numpy().view(np.uint32), wp[0]), (tag, int(s))
            assert np.array_equal(auto.jump.state[s].cpu().numpy().view(np.uint64), ws[0]), (tag, int(s))
    dec_a, st_a = B.ans_decode(auto, model, n_per, dtype=t_t)
    dec_p, st_p = B.ans_decode(plain, model, n_per, dtype=t_t)
    assert int(st_a.abs().sum()) == 0 and int(st_p.abs().sum()) == 0 and torch.equal(dec_a, d) and torch.equal(dec_p, d), tag
    return n_streams, auto.jump is not None


def gaussian_case():
    coder = str(rng.choice(["ans", "range"]))
    lo = int(rng.integers(-300, 0)); hi = int(rng.integers(1, 300))
    n_streams = int(rng.choice([16384, 16384 + 17, 32768, 40000, cus * 256, cus * 256 + 256]))
    n_per = 16 * int(rng.choice([32, 48, 64, 96, 128]))
    g = torch.Generator(device="cuda").manual_seed(int(rng.integers(1 << 30)))
    mu = lo + (hi - lo) * torch.rand((n_streams, n_per), generator=g, device="cuda", dtype=torch.float64)
    sd = torch.exp(torch.rand((n_streams, n_per), generator=g, device="cuda", dtype=torch.float64) * 7 - 2)
    sym = torch.clamp(torch.round(mu + sd * torch.randn((n_streams, n_per), generator=g, device="cuda", dtype=torch.float64)), lo, hi).to(torch.int32)
    tag = f"gaussian {coder} lo={lo} hi={hi} streams={n_streams} n_per={n_per}"
    enc_fn, dec_fn = (B.ans_encode_gaussian, B.ans_decode_gaussian) if coder == "ans" else (B.range_encode_gaussian, B.range_decode_gaussian)
    plain = enc_fn(sym, lo, hi, mu, sd, jump_points=0)
    auto = enc_fn(sym, lo, hi, mu, sd)
    same_batch(plain, auto, tag)
    assert int(auto.status.abs().sum()) == 0, tag
    h_sym, h_mu, h_sd = sym.cpu().numpy(), mu.cpu().numpy(), sd.cpu().numpy()
    for s in np.unique(np.concatenate([rng.integers(0, n_streams, 6), [0, n_streams - 1]])):
        if coder == "ans":
            c = O.AnsCoder()
            c.encode_gaussian_reverse(h_sym[s], lo, hi, h_mu[s], h_sd[s], 24, 32)
        else:
            c = O.RangeEncoder()
            c.encode(h_sym[s], [O.GaussianModel(lo, hi, float(m), float(v), 24, 32) for m, v in zip(h_mu[s], h_sd[s])], 24)
        assert auto.stream(int(s)).tolist() == c.get_compressed().tolist(), (tag, int(s))
    dec_a, st_a = dec_fn(auto, lo, hi, mu, sd)
    dec_p, st_p = dec_fn(plain, lo, hi, mu, sd)
    assert int(st_a.abs().sum()) == 0 and int(st_p.abs().sum()) == 0 and torch.equal(dec_a, sym) and torch.equal(dec_p, sym), tag
    return n_streams, auto.jump is not None


def ragged_case():
    P = int(rng.choice([8, 12, 16, 24]))
    n = int(rng.choice([2, 17, 101, 256]))
    n = min(n, (1 << P) // 2)
    lo = int(rng.integers(-1000, 1000))
    p, cdf = random_table(n, P)
    model = B.Model.from_cdf(cdf, lo, P)
    n_docs = int(rng.integers(1, 20000))
    lens = np.minimum(rng.geometric(1.0 / float(rng.choice([20, 300, 2000])), n_docs), 20000).astype(np.int64)
    if rng.random() < 0.3:
        lens[rng.integers(0, n_docs, max(1, n_docs // 10))] = 0
    offs = np.concatenate([[0], np.cumsum(lens)])
    flat = (rng.choice(n, size=int(offs[-1]), p=p / float(1 << P)) + lo).astype(np.int32)
    tag = f"ragged P={P} n={n} docs={n_docs} symbols={offs[-1]}"
    d, o = dev(flat), dev(offs)
    plain = B.ans_encode_ragged(d, o, model, (32, 64, P), jump_every=0)
    auto = B.ans_encode_ragged(d, o, model, (32, 64, P))
    assert torch.equal(plain.n_words, auto.n_words) and torch.equal(plain.word_offsets, auto.word_offsets) and torch.equal(plain.status, auto.status), tag
    nw = auto.n_words.to(torch.int64)
    ids = torch.repeat_interleave(torch.arange(n_docs, device="cuda"), nw)
    at = auto.word_offsets[ids] + torch.arange(int(nw.sum()), device="cuda") - (torch.cumsum(nw, 0) - nw)[ids]
    assert torch.equal(plain.words[at], auto.words[at]), tag
    for s in np.unique(rng.integers(0, n_docs, 24)):
        c = O.AnsCoder()
        c.encode_iid_table_reverse(flat[offs[s]: offs[s + 1]], cdf, lo, P)
        assert auto.stream(int(s)).tolist() == c.get_compressed().tolist(), (tag, int(s))
    dec_a, st_a = B.ans_decode_ragged(auto, model, o)
    dec_p, st_p = B.ans_decode_ragged(plain, model, o)
    assert int(st_a.abs().sum()) == 0 and int(st_p.abs().sum()) == 0 and torch.equal(dec_a, d) and torch.equal(dec_p, d), tag
    return n_docs, getattr(auto, "jump", None) is not None


while time.time() < t_end:
    r = rng.random()
    fam = "ans" if r < 0.35 else "range" if r < 0.6 else "per_stream" if r < 0.75 else "gaussian" if r < 0.85 else "ragged"
    if len(sys.argv) > 3:
        fam = sys.argv[3]
    n, jumped = {"ans": lambda: table_case("ans"), "range": lambda: table_case("range"), "per_stream": per_stream_case,
                 "gaussian": gaussian_case, "ragged": ragged_case}[fam]()
    count[fam] += 1
    with_jump[fam] += int(jumped)
    n_streams_total += n
    B.release_scratch()
print(f"stress_auto: {sum(count.values())} cases, {n_streams_total} streams: the default calls agree with the plain calls and the oracle -- "
      + ", ".join(f"{k} {count[k]} ({with_jump[k]} with jump points)" for k in count))
