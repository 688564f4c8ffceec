#!/usr/bin/env python3
"""BASELINE config C1 (plumbing): ONE stream of 1 000 000 symbols, LeakyQuantizer(-50..=50) x Gaussian(3.2, 9.6),
(W,S,P) = (32,64,24), through the batched C ABI and through the single-coder drop-in, word for word against the oracle."""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from constriction_amd import batched as B
import constriction_amd as constriction
from constriction_amd import stream  # noqa: F401
from oracle import oracle as O

P, lo, hi = 24, -50, 50
gm = O.GaussianModel(lo, hi, 3.2, 9.6, P, 32)
cdf = gm.cdf_table()
sym = O.synth_symbols(0xC0FFEE, 0, 1, 1_000_000, lo, cdf, P)
t0 = time.perf_counter(); want_words, want_n, st = O.ans_encode_batch(sym, lo, cdf, P); t_cpu = time.perf_counter() - t0
model = B.Model.quantized_gaussian(lo, hi, 3.2, 9.6, P)
d = torch.from_numpy(sym).cuda()
enc = B.ans_encode(d, model, (32, 64, P)); torch.cuda.synchronize()
t0 = time.perf_counter(); enc = B.ans_encode(d, model, (32, 64, P)); torch.cuda.synchronize(); t_gpu = time.perf_counter() - t0
words, n_words, status = enc.to_numpy()
assert status[0] == 0 and n_words[0] == want_n[0] and np.array_equal(words[0, : n_words[0]], want_words[0, : want_n[0]])
dec, dst = B.ans_decode(enc, model, 1_000_000); torch.cuda.synchronize()
assert np.array_equal(dec.cpu().numpy(), sym)
coder = constriction.stream.stack.AnsCoder()
coder.encode_reverse(sym[0], constriction.stream.model.QuantizedGaussian(lo, hi, 3.2, 9.6))
assert np.array_equal(coder.get_compressed(), want_words[0, : want_n[0]])
print(f"C1 ok: {want_n[0]} words; one-stream encode: oracle {t_cpu * 1e3:.1f} ms, GPU (one lane!) {t_gpu * 1e3:.1f} ms")
