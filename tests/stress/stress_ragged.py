#!/usr/bin/env python3
"""Randomised parity stress of the ragged entry points against the CPU oracle (not part of the test suite).
usage: python tests/stress/stress_ragged.py [seconds] [seed]"""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from constriction_amd import batched as B
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
n_cases = n_streams_total = 0
while time.time() < t_end:
    W, S = (32, 64) if rng.random() < 0.7 else (16, 32)
    P = int(rng.choice([1, 4, 8, 12, 13, 16, 20, 24] if W == 32 else [1, 8, 12, 16]))
    n = min(int(rng.choice([2, 3, 17, 101, 256, 700, 5000])), 1 << P)
    w = rng.gamma(0.3, 1.0, n) + 1e-9
    p = np.maximum(1, np.floor(w / w.sum() * ((1 << P) - n)).astype(np.int64))
    p[int(np.argmax(p))] += (1 << P) - int(p.sum())
    cdf = np.concatenate([[0], np.cumsum(p)]).astype(np.uint32)
    lo = int(rng.integers(-1000, 1000))
    model = B.Model.from_cdf(cdf, lo, P)
    n_docs = int(rng.choice([1, 2, 63, 64, 65, 200, 1000]))
    lengths = rng.integers(0, int(rng.choice([2, 40, 300, 2500])), n_docs)
    docs = [lo + rng.choice(n, size=int(k), p=p / p.sum()).astype(np.int32) for k in lengths]
    eof = None
    if rng.random() < 0.4:                       # documents that end at a terminator: the rarest symbol, appended to each
        eof = lo + int(np.argmin(p))
        docs = [np.concatenate([d[d != eof], [eof]]).astype(np.int32) for d in docs]
    flat, offsets = B.ragged(docs)
    # a third of the cases with a schedule (lane slot i codes stream order[i]): sorted by length, or a random permutation
    r = rng.random()
    order = None if r < 0.67 else ("sorted" if r < 0.84 else torch.from_numpy(rng.permutation(n_docs).astype(np.int32)).cuda())
    enc = B.ans_encode_ragged(flat, offsets, model, (W, S, P), order=order)
    torch.cuda.synchronize()
    assert int(enc.status.abs().sum()) == 0
    n_words = enc.n_words.cpu().numpy()
    for s in rng.choice(n_docs, size=min(n_docs, 12), replace=False):
        words, nw, st = O.ans_encode_batch(docs[s][None, :], lo, cdf, P, W, S)
        assert st[0] == 0 and n_words[s] == nw[0] and enc.stream(int(s)).tolist() == words[0, : nw[0]].tolist(), (W, S, P, n, len(docs[s]))
    dec, status = B.ans_decode_ragged(enc, model, offsets)
    assert int(status.abs().sum()) == 0 and torch.equal(dec, flat)
    if eof is not None:
        dec2, off2, st2 = B.ans_decode_until(enc, model, eof, max_symbols=4000)
        assert int(st2.abs().sum()) == 0 and torch.equal(off2, offsets) and torch.equal(dec2, flat)
    n_cases += 1; n_streams_total += n_docs
print(f"stress_ragged: {n_cases} cases, {n_streams_total} streams agree")
