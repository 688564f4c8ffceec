#!/usr/bin/env python3
"""Randomised parity stress of every batched kernel against the CPU oracle (not part of the test suite: minutes of GPU
time).  Random presets, precisions, alphabets, stream counts / lengths (partial waves, ragged tiles), slab and packed
layouts; words, counts, status and round trips must all agree.  usage: python tests/stress/stress_parity.py [seconds] [seed] [big]"""
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
    coder = rng.choice(["ans", "range"])
    W, S = (32, 64) if coder == "range" or rng.random() < 0.7 else (16, 32)
    P = int(rng.choice([8, 10, 12, 13, 16, 20, 24] if W == 32 else [8, 12, 14, 16]))
    n = int(rng.choice([2, 3, 17, 101, 255, 256, 300]))
    n = min(n, 1 << P)
    # a random table: mostly small probabilities, a few large ones, every symbol at least 1
    w = rng.gamma(0.3, 1.0, n) + 1e-9
    p = np.maximum(1, np.floor(w / w.sum() * ((1 << P) - n)).astype(np.int64))
    p[int(np.argmax(p))] += (1 << P) - int(p.sum())
    assert p.min() >= 1 and int(p.sum()) == 1 << P
    cdf = np.concatenate([[0], np.cumsum(p)]).astype(np.uint32)
    lo = int(rng.integers(-1000, 1000))
    model = B.Model.from_cdf(cdf, lo, P)
    big = len(sys.argv) > 3          # third argument: only shapes that reach the hand-scheduled main loops
    n_streams = int(rng.choice([64, 128, 130, 257, 512] if big else [1, 63, 64, 65, 128, 130, 257, 512]))
    n_per = int(rng.choice([64, 96, 100, 640, 1000, 2052, 4096] if big else [0, 1, 4, 31, 32, 36, 64, 96, 100, 640, 1000, 2052]))
    kind = rng.choice(["model", "uniform"])
    if kind == "model":
        idx = rng.choice(n, size=(n_streams, n_per), p=p / float(1 << P))
    else:
        idx = rng.integers(0, n, (n_streams, n_per))
    sym = (idx + lo).astype(np.int32)
    cfg = (W, S, P)
    if coder == "ans":
        want_words, want_n, want_st = O.ans_encode_batch(sym, lo, cdf, P, W, S)
        enc = B.ans_encode(dev(sym), model, cfg)
    else:
        want_words, want_n, want_st = O.rc_encode_batch(sym, lo, cdf, P, W, S)
        enc = B.range_encode(dev(sym), model, cfg)
    torch.cuda.synchronize()
    words, n_words, status = enc.to_numpy()
    tag = f"{coder} {cfg} n={n} streams={n_streams} n_per={n_per} {kind}"
    assert status.tolist() == want_st.tolist(), tag
    assert n_words.tolist() == want_n.tolist(), tag
    for s in range(n_streams):
        assert words[s, : n_words[s]].tolist() == want_words[s, : want_n[s]].tolist(), (tag, s)
    extra = int(rng.choice([0, 0, 4, 40]))
    if coder == "ans":
        want_dec, want_dst = O.ans_decode_batch(want_words, want_n, n_per + extra, lo, cdf, P, W, S)
        dec, dst = B.ans_decode(enc, model, n_per + extra)
        packed, offsets = B.compact(enc)
        dec2, dst2 = B.ans_decode((packed, enc.n_words), model, n_per + extra, offsets=offsets, config=cfg)
    else:
        want_dec, want_dst = O.rc_decode_batch(want_words, want_n, n_per + extra, lo, cdf, P, W, S)
        dec, dst = B.range_decode(enc, model, n_per + extra)
        packed, offsets = B.compact(enc)
        dec2, dst2 = B.range_decode((packed, enc.n_words), model, n_per + extra, offsets=offsets, config=cfg)
    if coder == "ans":
        # the same batch as symbols[t][stream]: same words, same symbols back
        encT = B.ans_encode(dev(sym.T), model, cfg, "symbol_major")
        decT, dstT = B.ans_decode(encT, model, n_per + extra, "symbol_major")
        torch.cuda.synchronize()
        wT, nT, sT = encT.to_numpy()
        assert sT.tolist() == want_st.tolist() and nT.tolist() == want_n.tolist(), (tag, "symbol-major")
        for s in range(n_streams):
            assert wT[s, : nT[s]].tolist() == want_words[s, : want_n[s]].tolist(), (tag, s, "symbol-major")
        okT = want_dst == 0
        assert dstT.cpu().numpy().tolist() == want_dst.tolist() and np.array_equal(decT.cpu().numpy().T[okT], want_dec[okT]), (tag, "symbol-major decode")
    torch.cuda.synchronize()
    for d, st in ((dec, dst), (dec2, dst2)):
        st = st.cpu().numpy()
        assert st.tolist() == want_dst.tolist(), tag
        ok = want_dst == 0
        assert np.array_equal(d.cpu().numpy()[ok], want_dec[ok]), tag
    n_cases += 1
print(f"{n_cases} random cases agree with the oracle")
