#!/usr/bin/env python3
"""Randomised parity stress of the producer / consumer ANS encoder (cst_ans_pc.hip) against the CPU oracle (not part of the
test suite: minutes of GPU time).  Only shapes that kernel takes: whole workgroups of 256 streams, rows of whole 32-symbol
tiles, (32,64), 8 <= P <= 12; random tables, slab strides (some too small: CST_STREAM_CAPACITY), impossible symbols.
usage: python tests/stress/stress_pc.py [seconds] [seed]"""
import ctypes as C, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from constriction_amd import batched as B, _native as N
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
n_cases = n_streams_total = 0
lib = N.lib()
while time.time() < t_end:
    P = int(rng.integers(8, 13))
    n = int(rng.choice([2, 3, 17, 101, 255, 256, 300, 1024]))
    n = min(n, 1 << P)
    w = rng.gamma(0.3, 1.0, n) + 1e-9
    p = np.maximum(1, np.floor(w / w.sum() * ((1 << P) - n)).astype(np.int64))
    p[int(np.argmax(p))] += (1 << P) - int(p.sum())
    cdf = np.concatenate([[0], np.cumsum(p)]).astype(np.uint32)
    lo = int(rng.integers(-1000, 1000))
    model = B.Model.from_cdf(cdf, lo, P)
    n_streams = 256 * int(rng.choice([1, 2, 3, 8]))
    n_per = 32 * int(rng.choice([2, 3, 4, 5, 7, 16, 33, 64]))
    idx = rng.choice(n, size=(n_streams, n_per), p=p / float(1 << P)) if rng.random() < 0.5 else rng.integers(0, n, (n_streams, n_per))
    sym = (idx + lo).astype(np.int32)
    for _ in range(int(rng.choice([0, 0, 1, 5]))):          # impossible symbols
        sym[rng.integers(n_streams), rng.integers(n_per)] = int(rng.choice([lo - 1, lo + n, 2 ** 30, -2 ** 31, 2 ** 31 - 1]))
    want_words, want_n, want_st = O.ans_encode_batch(sym, lo, cdf, P)
    full = B.max_words(n_per, (32, 64, P))
    stride = int(rng.choice([full, full + 16, 16 * max(1, int(want_n.max()) // 16), 16, 48]))
    want_st = np.where((want_st == 0) & (want_n > stride), 2, want_st)
    guard = torch.full((n_streams * stride + 1024,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
    d = torch.from_numpy(sym).cuda()
    assert d.data_ptr() % 128 == 0 and guard.data_ptr() % 64 == 0
    n_words = torch.zeros(n_streams, dtype=torch.int32, device="cuda")
    status = torch.zeros(n_streams, dtype=torch.int32, device="cuda")
    N.check(lib.cst_ans_encode_batch(model._h, N.CoderConfig(32, 64, P), C.c_void_p(d.data_ptr()), n_streams, n_per, 0,
                                     C.c_void_p(guard.data_ptr()), stride, C.c_void_p(n_words.data_ptr()), None,
                                     C.c_void_p(status.data_ptr()), 0, None), "cst_ans_encode_batch")
    torch.cuda.synchronize()
    tag = f"P={P} n={n} streams={n_streams} n_per={n_per} stride={stride}"
    got_st, got_n = status.cpu().numpy(), n_words.cpu().numpy()
    assert got_st.tolist() == want_st.tolist(), tag
    assert got_n.tolist() == np.where(want_st == 0, want_n, 0).tolist(), tag
    words = guard.cpu().numpy().view(np.uint32)
    assert (words[n_streams * stride:] == 0x5A5A5A5A).all(), (tag, "words behind the last slab")
    rows = words[: n_streams * stride].reshape(n_streams, stride)
    for s in np.flatnonzero(want_st == 0):
        assert rows[s, : want_n[s]].tolist() == want_words[s, : want_n[s]].tolist(), (tag, int(s))
    n_cases += 1
    n_streams_total += n_streams
print(f"stress_pc: {n_cases} cases, {n_streams_total} streams agree with the oracle")
