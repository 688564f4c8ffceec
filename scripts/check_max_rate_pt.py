#!/usr/bin/env python3
"""Adversarial data for the per-stream-table decoders: every symbol a far-tail symbol of probability 2^-P (12 bits per symbol at
P = 12: a 32-symbol tile consumes exactly 12 words, the maximum the word windows are sized for), plain and through jump points."""
import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from constriction_amd import batched as B

n, k, P = 512, 4096, 12
rng = np.random.default_rng(2)
mu = torch.from_numpy(rng.uniform(-5, 5, n)).cuda(); sd = torch.from_numpy(rng.uniform(0.4, 0.8, n)).cuda()
m = B.Model.quantized_gaussian_per_stream(-127, 127, mu, sd, P)
tails = rng.choice(np.concatenate([np.arange(-127, -40), np.arange(40, 128)]), (n, k)).astype(np.int32)
# ... with a likely symbol here and there, so that the refills fall on every phase of a tile (all-tail data emits a word on the same
# three of every eight steps, tile after tile)
likely = np.rint(mu.cpu().numpy())[:, None].astype(np.int32) + np.zeros((n, k), np.int32)
frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.03
sym = torch.from_numpy(np.where(rng.random((n, k)) < frac, likely, tails)).cuda()
enc = B.ans_encode(sym, m, (32, 64, P))
print("words per stream", float(enc.n_words.float().mean()), "status", int(enc.status.abs().sum()))
dec, st = B.ans_decode(enc, m, k)
print("plain", B.last_kernel(), bool(torch.equal(dec, sym)), int(st.abs().sum()))
for chunks in (4, 8):
    e2, ck = B.ans_encode_checkpointed(sym, m, k // chunks, (32, 64, P))
    d2, s2 = B.ans_decode_checkpointed(e2, ck, m, k)
    print("k =", chunks, B.last_kernel(), bool(torch.equal(d2, sym)), int(s2.abs().sum()), bool(torch.equal(e2.n_words, enc.n_words)))
