#!/usr/bin/env python3
"""Randomised parity stress of the per-symbol quantized-Gaussian kernels against the CPU oracle (not part of the test
suite: minutes of GPU time).  Every dispatch: one wave per stream, cdf rows + row lookup (fewer than 64 streams, supports
up to 255), one lane per stream; both coders, both layouts, presets (32,64,P) and (16,32,P); tame and extreme models;
encode + decode round trips and decoding of RANDOM words (every quantile, the tails included).
usage: python tests/stress/stress_per_symbol.py [seconds] [seed]"""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from constriction_amd import batched as B
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
t_end = time.time() + budget
n_cases = 0
while time.time() < t_end:
    coder = str(rng.choice(["ans", "range"]))
    W, S = (32, 64) if rng.random() < 0.7 else (16, 32)
    P = int(rng.choice([12, 16, 20, 24] if W == 32 else [8, 12, 16]))
    prob_bits = 32 if W == 32 else 16
    lo = int(rng.integers(-300, 300))
    n = int(rng.choice([2, 3, 50, 201, 255, 256, 700]))
    n = min(n, 1 << P)
    hi = lo + n - 1
    n_streams = int(rng.choice([1, 2, 40, 63, 64, 65, 200]))
    n_per = int(rng.choice([1, 31, 33, 70, 300, 1700]))
    layout = str(rng.choice(["stream_major", "symbol_major"]))
    extreme = rng.random() < 0.4
    if extreme:
        mu = rng.uniform(lo - 50.0, hi + 50.0, (n_streams, n_per))
        sd = np.exp(rng.uniform(np.log(1e-6), np.log(1e5), (n_streams, n_per)))
    else:
        mu = rng.uniform(lo, hi, (n_streams, n_per))
        sd = np.exp(rng.uniform(np.log(0.3), np.log(max(0.31, n / 4)), (n_streams, n_per)))
    t = (lambda a: a.T) if layout == "symbol_major" else (lambda a: a)
    cfg = (W, S, P)
    tag = f"{coder} {cfg} [{lo},{hi}] streams={n_streams} n_per={n_per} {layout} extreme={extreme}"
    check = list(range(n_streams)) if n_streams <= 8 else sorted(set(rng.integers(0, n_streams, 6).tolist()) | {0, n_streams - 1})

    # ---- decode random words: GPU vs oracle ----
    if coder == "ans":
        stride = n_per + 8
        words = rng.integers(1, 1 << W, (n_streams, stride), dtype=np.uint64).astype(np.uint32)
        enc = B.EncodedBatch(dev(words.view(np.int32)), dev(np.full(n_streams, stride, np.int32)), dev(np.zeros(n_streams, np.int32)), cfg)
        dec, st = B.ans_decode_gaussian(enc, lo, hi, dev(t(mu)), dev(t(sd)), layout)
        torch.cuda.synchronize()
        dec = t(dec.cpu().numpy()); st = st.cpu().numpy()
        for s in check:
            c = O.AnsCoder(words[s] if W == 32 else words[s].astype(np.uint16), W=W, S=S)
            try:
                want = list(c.decode_gaussian(n_per, lo, hi, mu[s], sd[s], P, prob_bits))
            except Exception:
                assert st[s] != 0, (tag, s, "oracle failed, GPU did not")
                continue
            assert st[s] == 0 and dec[s].tolist() == want, (tag, s)

    # ---- chain coders over the same models: decode random words, put the symbols back (oracle: src/stream/chain.rs) ----
    if rng.random() < 0.5 and n_per * P <= 40 * W * 32:
        stride = n_per + 8
        words = rng.integers(1, 1 << W, (n_streams, stride), dtype=np.uint64).astype(np.uint32)
        oracles, heads, n_pop = [], np.zeros((n_streams, 2), dtype=np.uint64), np.zeros(n_streams, np.uint32)
        for s in range(n_streams):
            c = O.ChainCoder(words[s], W=W, S=S, P=P)
            oracles.append(c); heads[s] = (c.rem_head, c.comp_head); n_pop[s] = len(c.compressed)
        chains = B.ChainBatch(dev(words.view(np.int32)), dev(n_pop.view(np.int32)), dev(heads.view(np.int64)), cfg)
        csym, pushed, n_pushed, cst = B.chain_decode_gaussian(chains, lo, hi, dev(t(mu)), dev(t(sd)), layout)
        back, n_back, est2 = B.chain_encode_gaussian(B.ChainBatch(pushed, n_pushed.clone(), chains.heads, cfg), csym, lo, hi,
                                                     dev(t(mu)), dev(t(sd)), layout)
        torch.cuda.synchronize()
        csym_h, cst, est2 = t(csym.cpu().numpy()), cst.cpu().numpy(), est2.cpu().numpy()
        pushed_h, n_pushed_h, back_h, n_back_h = pushed.cpu().numpy().view(np.uint32), n_pushed.cpu().numpy(), back.cpu().numpy().view(np.uint32), n_back.cpu().numpy()
        for s in check:
            c = oracles[s]
            models = [O.GaussianModel(lo, hi, m, d, P, prob_bits) for m, d in zip(mu[s], sd[s])]
            try:
                want = c.decode(models).tolist()
            except O.ChainCoder.OutOfCompressedData:
                assert cst[s] == 4, (tag, s, "chain: oracle ran out of data, GPU did not say so")
                continue
            assert cst[s] == 0 and csym_h[s].tolist() == want, (tag, s, "chain decode")
            assert pushed_h[s, : n_pushed_h[s]].tolist() == c.remainders, (tag, s, "chain remainders")
            before = len(c.compressed)
            c.encode_reverse(csym_h[s], models)
            assert est2[s] == 0 and back_h[s, : n_back_h[s]].tolist() == c.compressed[before:], (tag, s, "chain encode")
            assert np.array_equal(np.concatenate(c.get_data()), words[s]), (tag, s, "chain restore")

    # ---- encode symbols drawn from the models, compare words, decode back ----
    sym = np.clip(np.rint(mu + sd * rng.standard_normal(mu.shape)), lo, hi).astype(np.int32)
    enc_fn = B.ans_encode_gaussian if coder == "ans" else B.range_encode_gaussian
    dec_fn = B.ans_decode_gaussian if coder == "ans" else B.range_decode_gaussian
    enc = enc_fn(dev(t(sym)), lo, hi, dev(t(mu)), dev(t(sd)), cfg, layout)
    torch.cuda.synchronize()
    est = enc.status.cpu().numpy()
    for s in check:
        try:
            if coder == "ans":
                c = O.AnsCoder(W=W, S=S)
                c.encode_gaussian_reverse(sym[s], lo, hi, mu[s], sd[s], P, prob_bits)
            else:
                c = O.RangeEncoder(W=W, S=S)
                c.encode(sym[s], [O.GaussianModel(lo, hi, m, d, P, prob_bits) for m, d in zip(mu[s], sd[s])], P)
            want = c.get_compressed().tolist()
        except Exception:
            assert est[s] != 0, (tag, s, "oracle refused the symbols, GPU did not")
            continue
        assert est[s] == 0 and enc.stream(s).tolist() == want, (tag, s)
    dec, st = dec_fn(enc, lo, hi, dev(t(mu)), dev(t(sd)), layout)
    torch.cuda.synchronize()
    dec = t(dec.cpu().numpy()); st = st.cpu().numpy()
    good = est == 0
    assert (st[good] == 0).all() and np.array_equal(dec[good], sym[good]), tag
    n_cases += 1
print(f"{n_cases} random per-symbol cases agree with the oracle")
