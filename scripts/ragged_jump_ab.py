#!/usr/bin/env python3
"""A/B: the ragged calls with and without jump points (round 6) at the bench's shape and around it."""
import sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench
from constriction_amd import batched as B

P, n_sym = 24, 64
w = 0.93 ** np.arange(n_sym)
prob = np.maximum(1, np.floor(w / w.sum() * ((1 << P) - n_sym)).astype(np.int64))
prob[0] += (1 << P) - int(prob.sum())
cdf = np.concatenate([[0], np.cumsum(prob)]).astype(np.uint32)
model = B.Model.from_cdf(cdf, 0, P)
for n_docs in (2_000, 20_000, 100_000, 400_000):
    rng = np.random.default_rng(bench.SEED)
    lengths = np.exp(rng.uniform(np.log(20), np.log(2000), n_docs)).astype(np.int64)
    offsets = np.zeros(n_docs + 1, dtype=np.int64)
    np.cumsum(lengths, out=offsets[1:])
    gen = torch.Generator(device="cuda").manual_seed(1)
    q = torch.randint(0, 1 << P, (int(offsets[-1]),), generator=gen, device="cuda", dtype=torch.int64)
    flat = (torch.searchsorted(torch.from_numpy(cdf.astype(np.int64)).cuda(), q, right=True) - 1).to(torch.int32)
    off_d = torch.from_numpy(offsets).cuda()
    for every in (0, 64, 128, 256, 512):
        enc = B.ans_encode_ragged(flat, off_d, model, (32, 64, P), jump_every=every)
        dec, st = B.ans_decode_ragged(enc, model, off_d)
        assert torch.equal(dec, flat) and int(st.abs().sum()) == 0
        e = bench.event_ms(lambda: B.ans_encode_ragged(flat, off_d, model, (32, 64, P), jump_every=every), 5)
        d = bench.event_ms(lambda: B.ans_decode_ragged(enc, model, off_d, out=dec), 5)
        print(f"{n_docs:7d} documents, {int(offsets[-1]) / 1e6:6.1f} M symbols, a jump point every {every:4d}: encode {e:.3f} ms  decode {d:.3f} ms  [{B.last_kernel()}]")
